#!/usr/bin/env python3
"""Where a cold process's statistics phase goes (bench.py's cold_process.host_phases: "statistics"): one 3072 x 3072 fp32
second-moment file from disk to HBM, piece by piece, and get_cov_text_encoder for the four edited layers."""
import os, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from emcid_amd import emcid_main as em, runningstats as rs, synthetic as syn
from emcid_amd.emcid_hparams import EMCIDHyperParams

dev = "cuda:0" if torch.cuda.is_available() else "cpu"
tmp = Path(tempfile.mkdtemp())
hp_d = syn.sd_hparams_dict(layers=(7, 8, 9, 10), mom2_update_weight=4000, edit_weight=0.5)
names = [hp_d["rewrite_module_tmp"].format(l) for l in hp_d["layers"]]
syn.write_stats_cache(tmp / "stats", names, 3072, hp_d["mom2_n_samples"], seed=2, t=6144)
files = [syn.stats_file(tmp / "stats", n, hp_d["mom2_n_samples"]) for n in names]
if dev != "cpu":
    torch.zeros(1, device=dev); torch.cuda.synchronize()


def t(fn, n=3):
    out = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn()
        if dev != "cpu":
            torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return round(min(out), 2), r


ms, dat = t(lambda: rs.read_npz_stored(files[0])); print("read_npz_stored", ms, "ms")
ms, _ = t(lambda: np.fromfile(files[0], dtype=np.uint8)); print("  numpy.fromfile alone", ms, "ms")
ms, z = t(lambda: dict(np.load(files[0]))); print("numpy.load (all members)", ms, "ms")
m = torch.from_numpy(dat["mom2.mom2"]); cnt = int(dat["mom2.count"])
ms, c = t(lambda: (m / cnt).float()); print("moment() on the host", ms, "ms")
ms, _ = t(lambda: c.to(dev)); print("pageable .to(device)", ms, "ms")
if dev != "cpu":
    ms, p = t(lambda: c.pin_memory()); print("pin_memory copy", ms, "ms")
    ms, _ = t(lambda: p.to(dev, non_blocking=True)); print("pinned .to(device)", ms, "ms")
    ms, _ = t(lambda: m.to(dev) / cnt); print("upload raw + divide on the GPU", ms, "ms")
pipe = syn.build_pipe("sd-v1.4", dev, syllables=True)
hp = EMCIDHyperParams(**hp_d)
for i in range(3):
    em.clear_caches()
    t0 = time.perf_counter()
    for n in names:
        em.get_cov_text_encoder(pipe.text_encoder, pipe.tokenizer, n, hp.mom2_dataset, hp.mom2_n_samples, hp.mom2_dtype,
                                stat_dir=str(tmp / "stats"), verbose=False)
    if dev != "cpu":
        torch.cuda.synchronize()
    print("get_cov_text_encoder x 4 layers", round((time.perf_counter() - t0) * 1e3, 1), "ms")

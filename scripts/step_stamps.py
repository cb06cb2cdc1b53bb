#!/usr/bin/env python3
"""Per-workgroup timeline of the fused Cholesky step launches (leaf + trailing tiles + shadow product P = Yt X) of the LAST edited
layer of a 1000-concept edit: start skew after the launch's first workgroup, lifetimes by kind, end of the launch — from
in-kernel s_memrealtime stamps (100 MHz), emcid_debug_step_stamps.  usage: step_stamps.py [N=1000]"""
import ctypes as C, json, os, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from emcid_amd import emcid_main as em, hip
from emcid_amd.emcid_hparams import EMCIDHyperParams

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = "cuda:0"
lib = hip.load()
stamps = torch.zeros(32 * 512 * 4, dtype=torch.int64, device=dev)
lib.emcid_debug_step_stamps(C.c_void_p(stamps.data_ptr()))       # before the first edit: the graphs capture the pointer
work = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
work.mkdir(exist_ok=True)
pipe, reqs, hp_d, cache, stats, names = bench.build_inputs(N, dev, work)
hp = EMCIDHyperParams(**hp_d)
for _ in range(4):
    stamps.zero_()
    em.apply_emcid_to_text_encoder(pipe, reqs, hp, dev, cache_name=cache, stats_dir=stats, verbose=False)
torch.cuda.synchronize()
s = stamps.view(32, 512, 4).cpu()
kinds = {1: "leaf", 2: "trailing", 3: "shadow", 4: "spine", 5: "panel"}
for sl in range(32):
    rows = s[sl][s[sl][:, 3] > 0]
    if rows.numel() == 0:
        continue
    t0 = int(rows[:, 0].min())
    out = {"launch": sl, "unit": "us (100 MHz clock)", "launch_span": (int(rows[:, 2].max()) - t0) / 100}
    for k, name in kinds.items():
        r = rows[rows[:, 3] == k]
        if r.numel():
            st, life = (r[:, 0] - t0).double() / 100, (r[:, 2] - r[:, 0]).double() / 100
            q = lambda v: [round(float(x), 1) for x in torch.quantile(v, torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64))]
            out[name] = {"n": int(r.shape[0]), "start_min_med_max": q(st), "lifetime_min_med_max": q(life),
                         "last_end": round(float(((r[:, 2] - t0).double() / 100).max()), 1)}
    print(json.dumps(out))

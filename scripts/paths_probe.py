#!/usr/bin/env python3
"""Which tensors of a cold 40-concept SD-width edit differ between two runs of the same forward path, and between the native layer
runner and the per-launch path (debugging aid for tests/test_e2e_gpu.py::test_forward_paths_agree)."""
import sys, tempfile
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import clip_forward as cf, edit_engine as ee, emcid_main as em, synthetic as syn
from emcid_amd.emcid_hparams import EMCIDHyperParams
from emcid_amd.nethook import get_parameter

DEV = "cuda:0"
kind, layers, n_req = "sd-v1.4", (7, 8, 9, 10), 40
hidden, inter = syn.ENCODER_DIMS[kind][:2]
tmp = Path(tempfile.mkdtemp())
reqs = syn.make_requests(n_req, ragged=True, names="syllable")
hp_d = syn.sd_hparams_dict(layers=layers, mom2_update_weight=60, mom2_n_samples=100)
names = [hp_d["rewrite_module_tmp"].format(l) for l in layers]
cache = str(tmp / "cache") + "/"
syn.write_vstar_cache(cache, reqs, hidden, seed=1, scale=0.5)
syn.write_stats_cache(tmp / "stats", names, inter, 100, seed=2, t=2 * inter)
runs = []
for mode, native in (("native", True), ("native", True), ("launches", False), ("launches", False), ("native", True)):
    cf.NATIVE_RUNNER = native
    em.clear_caches()
    pipe = syn.build_pipe(kind, DEV, syllables=True)
    hp = EMCIDHyperParams(**hp_d)
    plan = em.prepare_text_encoder_edit(pipe.text_encoder, pipe.tokenizer, reqs, hp, hp.layers, 60, str(tmp / "stats"), cache,
                                        verbose=False)
    edits = ee.run_encoder_edit(plan, trace=True)
    ee.check_info(plan)
    runs.append((mode, edits, plan.factors_from_cache))
for a in range(len(runs)):
    for b in range(a + 1, len(runs)):
        line = [f"{runs[a][0]}#{a} (cached {runs[a][2]}) vs {runs[b][0]}#{b} (cached {runs[b][2]}):"]
        for ea, eb in zip(runs[a][1], runs[b][1]):
            line.append(f"L{ea.layer} K {(ea.K - eb.K).abs().max().item():.1e} Zc {(ea.Zc - eb.Zc).abs().max().item():.1e} "
                        f"dW {(ea.dW - eb.dW).abs().max().item() / ea.dW.abs().max().item():.1e}")
        print(" | ".join(line))

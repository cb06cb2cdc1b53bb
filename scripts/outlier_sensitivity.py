#!/usr/bin/env python3
"""How well-posed is the 1e-4 parity bar on a given synthetic encoder?  The oracle (= the reference, bit for bit on CPU) with the
encoder in fp32 against the same computation with the encoder in fp64: max |dW32 - dW64| / max |dW64| per edited layer.  Plain
Gaussian init: 2-3e-6; `add_trained_like_outliers`: 2-4e-5 (a first, uncompensated version of it: 3-8e-3 — every softmax
saturated; not a parity target for anybody).  usage: python scripts/outlier_sensitivity.py [outl|plain]"""
import copy, sys, tempfile, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from conftest import load_golden
from emcid_amd import synthetic as syn
from oracle import emcid_oracle as orc
torch.set_num_threads(8)
z, meta = load_golden("real_sd_outliers_summary")
outl = (sys.argv[1] != "plain") if len(sys.argv) > 1 else True
reqs = syn.make_requests(meta["n_requests"], names="syllable")
hidden, inter = syn.ENCODER_DIMS[meta["kind"]][:2]
tmp = tempfile.mkdtemp()
cache = tmp + "/cache/"
syn.write_vstar_cache(cache, reqs, hidden, seed=1, scale=0.5)
st = meta["stats"]
syn.write_stats_cache(tmp + "/stats", meta["layer_names"], inter, st["n_samples"], seed=st["seed"], t=st["t"])
res = {}
for dt in (torch.float32, torch.float64):
    pipe = syn.build_pipe(meta["kind"], "cpu", syllables=True, outliers=outl)
    pipe.text_encoder.to(dt)
    w0 = {ln: orc.get_parameter(pipe.text_encoder, ln + ".weight").clone() for ln in meta["layer_names"]}
    hp = copy.deepcopy(meta["hparams"])
    try:
        orc.apply_emcid_to_text_encoder(pipe, reqs, hp, mom2_weight=meta["lam"], edit_weight=meta["ew"], cache_name=cache, stats_dir=tmp + "/stats")
    except Exception as e:
        print("dtype", dt, "failed:", repr(e)[:300]); continue
    res[dt] = [ (orc.get_parameter(pipe.text_encoder, ln + ".weight").double() - w0[ln].double()) for ln in meta["layer_names"]]
if len(res) == 2:
    for li,(a,b) in enumerate(zip(res[torch.float32], res[torch.float64])):
        print("layer", li, "max|dW32 - dW64| / max|dW64| =", ((a-b).abs().max()/b.abs().max()).item(), " max|dW64|", b.abs().max().item())

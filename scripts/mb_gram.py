#!/usr/bin/env python3
"""Gram kernel alone (emcid_gram_accumulate_f32): G += X^T X for X (t, d) fp32, HIP-event timed.
usage: mb_gram.py [d] [t ...]   env EMCID_GRAM_MI=2|4 selects the tile height (128 | 256 rows)."""
import json, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip

d = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
ts = [int(a) for a in sys.argv[2:]] or [4096, 16384, 65536]
dev = "cuda:0"
out = {"d": d, "mi": os.environ.get("EMCID_GRAM_MI", "default")}
for t in ts:
    X = torch.randn(t, d, device=dev)
    G = torch.zeros(d, d, device=dev)
    for _ in range(3):
        hip.gram_accumulate_(G, X)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        hip.gram_accumulate_(G, X)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    ref = (X.double().t() @ X.double()) * (n + 3)
    hip.symmetrize_lower_(G)
    err = ((G.double() - ref).abs().max() / ref.abs().max()).item()
    out[str(t)] = {"us": round(us, 1), "syrk_TF": round(t * d * d / us / 1e6, 1), "rel_err": err}
print(json.dumps(out))

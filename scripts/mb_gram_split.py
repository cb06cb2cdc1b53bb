#!/usr/bin/env python3
"""Gram kernel: sweep of the token split (workgroups = lower tiles x ksplit) at fixed (t, d)."""
import json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
d, t = int(sys.argv[1]), int(sys.argv[2])
X = torch.randn(t, d, device="cuda:0"); G = torch.zeros(d, d, device="cuda:0")
out = {}
for ks in [int(a) for a in sys.argv[3:]]:
    for _ in range(2): hip.gram_accumulate_(G, X, ksplit=ks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): hip.gram_accumulate_(G, X, ksplit=ks)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    out[ks] = [round(us, 1), round(t * d * d / us / 1e6, 1)]
print(json.dumps({"d": d, "t": t, "ksplit: [us, syrk_TF]": out}))

#!/usr/bin/env python3
"""S = I + Yt Yt^T (1024 x 1024 x 3072, lower tiles): the two-phase stream-K form against plain / K-split small tiles."""
import json, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
from scripts.microbench import timeit
dev, d = "cuda:0", 3072
Y = torch.randn(1024, d, dtype=torch.float64, device=dev)
ref = torch.tril(Y @ Y.t())
S = torch.zeros(1024, 1024, dtype=torch.float64, device=dev)
dt = timeit(lambda: hip.dgemm_streamk(0, Y, Y, S, flags=16, wgs=256, diag_add=0.0), iters=20, warmup=3)
print(json.dumps({"variant": "two-phase stream-K (default tiles)", "us": round(dt * 1e6, 1)}))
for cfg in (2, 1):
    S.zero_()
    hip.dgemm_ex(0, 0, Y, Y, S, beta=0.0, flags=16, cfg=cfg)
    err = float((torch.tril(S) - ref).abs().max() / ref.abs().max())
    dt = timeit(lambda: hip.dgemm_ex(0, 0, Y, Y, S, beta=0.0, flags=16, cfg=cfg), iters=20, warmup=3)
    print(json.dumps({"variant": f"plain lower tiles cfg {cfg}", "us": round(dt * 1e6, 1), "err": err}))
    for ks in (2, 4):
        S.zero_()
        hip.dgemm_ex(0, 0, Y, Y, S, beta=1.0, flags=16, cfg=cfg, ksplit=ks)
        err = float((torch.tril(S) - ref).abs().max() / ref.abs().max())
        dt = timeit(lambda: hip.dgemm_ex(0, 0, Y, Y, S, beta=1.0, flags=16, cfg=cfg, ksplit=ks), iters=20, warmup=3)
        print(json.dumps({"variant": f"cfg {cfg} ksplit {ks} (f64 atomics: not reproducible)", "us": round(dt * 1e6, 1), "err": err}))

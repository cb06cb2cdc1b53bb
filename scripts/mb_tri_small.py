#!/usr/bin/env python3
"""The two GEMMs against the explicit inverse factor for SMALL concept counts (M = 128 ... 512 rows): two-phase stream-K over
128 x 128 tiles against mirrored pairs of 32 x 64 tiles (where is the crossover?)."""
import json, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
from scripts.microbench import timeit
dev, d = "cuda:0", 3072
X = torch.tril(torch.randn(d, d, dtype=torch.float64, device=dev))
for M in (128, 256, 384, 512, 768, 1024):
    A = torch.randn(M, d, dtype=torch.float64, device=dev)
    for name, tb, tri in (("Yt=Kt*X^T", 0, 1), ("U=G*X", 1, 2)):
        ref = A @ (X.t() if tb == 0 else X)
        C = torch.zeros(M, d, dtype=torch.float64, device=dev)
        hip.dgemm_streamk(tb, A, X, C, flags=tri, wgs=256)
        e1 = float((C - ref).abs().max() / ref.abs().max())
        t1 = timeit(lambda: hip.dgemm_streamk(tb, A, X, C, flags=tri, wgs=256), iters=20, warmup=3)
        C.zero_()
        hip.dgemm_ex(0, tb, A, X, C, beta=0.0, flags=tri | 32, cfg=2)
        e2 = float((C - ref).abs().max() / ref.abs().max())
        t2 = timeit(lambda: hip.dgemm_ex(0, tb, A, X, C, beta=0.0, flags=tri | 32, cfg=2), iters=20, warmup=3)
        print(json.dumps({"M": M, "shape": name, "streamk_us": round(t1 * 1e6, 1), "pairs32x64_us": round(t2 * 1e6, 1), "err": [e1, e2]}))

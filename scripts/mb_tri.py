#!/usr/bin/env python3
"""Tile-config / K-split sweep for the triangular-operand GEMMs of the explicit-inverse dual solver."""
import json, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
from scripts.microbench import timeit

dev = "cuda:0"
d = 3072
X = torch.tril(torch.randn(d, d, dtype=torch.float64, device=dev))
shapes = [
    ("Yt=Kt*X^T", 0, 0, 1024, X, 1, d * d * 1024),      # B(k,n)=X[n][k] stored [n][k], zero for k>n
    ("U=G*X", 0, 1, 768, X, 2, d * d * 768),              # B stored [k][n], zero for k<n
]
for name, ta, tb, M, B, tri, flops in shapes:
    A = torch.randn(M, d, dtype=torch.float64, device=dev)
    ref = A @ (B.t() if tb == 0 else B)
    for cfg in (0, 1, 2):
        for ks in (0, 100, -16, -32):
            pair = ks == 100
            if pair:
                ks = 0
            C = torch.zeros(M, d, dtype=torch.float64, device=dev)
            beta = 0.0 if ks == 0 else 1.0
            tri_ = tri | (32 if pair else 0)
            hip.dgemm_ex(ta, tb, A, B, C, beta=beta, flags=tri_, cfg=cfg, ksplit=ks)
            err = float((C - ref).abs().max() / ref.abs().max())
            def run():
                if ks != 0:
                    C.zero_()
                hip.dgemm_ex(ta, tb, A, B, C, beta=beta, flags=tri_, cfg=cfg, ksplit=ks)
            dt = timeit(run, iters=20, warmup=3)
            print(json.dumps({"shape": name, "cfg": cfg, "pair": pair, "ksplit": ks, "us": round(dt * 1e6, 1),
                              "tflops_tri": round(flops / dt / 1e12, 1), "err": err}))

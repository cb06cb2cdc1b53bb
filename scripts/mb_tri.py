#!/usr/bin/env python3
"""Tile-config / K-split sweep for the triangular-operand GEMMs of the explicit-inverse dual solver."""
import json, os, sys, torch
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
    variants = [("cfg1", dict(flags=tri, cfg=1)), ("cfg2 pair", dict(flags=tri | 32, cfg=2))]     # (the atomic stream-K cfg 4 is gone)
    for wgs in (256, 512):
        C = torch.zeros(M, d, dtype=torch.float64, device=dev)
        hip.dgemm_streamk(tb, A, B, C, flags=tri, wgs=wgs)
        err = float((C - ref).abs().max() / ref.abs().max())
        dt = timeit(lambda: hip.dgemm_streamk(tb, A, B, C, flags=tri, wgs=wgs), iters=20, warmup=3)
        print(json.dumps({"shape": name, "variant": f"two-phase streamK {wgs} xcd={os.environ.get('EMCID_STREAMK_XCD', '1')}",
                          "us": round(dt * 1e6, 1), "tflops_tri": round(flops / dt / 1e12, 1), "err": err}))
    for label, kw in variants:
        C = torch.zeros(M, d, dtype=torch.float64, device=dev)
        hip.dgemm_ex(ta, tb, A, B, C, beta=0.0, **kw)
        err = float((C - ref).abs().max() / ref.abs().max())
        dt = timeit(lambda: hip.dgemm_ex(ta, tb, A, B, C, beta=0.0, **kw), iters=20, warmup=3)
        print(json.dumps({"shape": name, "variant": label, "us": round(dt * 1e6, 1), "tflops_tri": round(flops / dt / 1e12, 1), "err": err}))

# the N x N system of the dual solver: S = I + Yt Yt^T (lower tiles), 1024 x 1024 x 3072
Y = torch.randn(1024, d, dtype=torch.float64, device=dev)
S = torch.zeros(1024, 1024, dtype=torch.float64, device=dev)
for wgs in (256, 512):
    dt = timeit(lambda: hip.dgemm_streamk(0, Y, Y, S, flags=16, wgs=wgs, diag_add=1.0), iters=20, warmup=3)
    print(json.dumps({"shape": "S=I+Yt*Yt^T", "variant": f"two-phase streamK {wgs}", "us": round(dt * 1e6, 1),
                      "tflops_syrk": round(1024 * 1024 * d / dt / 1e12, 1)}))

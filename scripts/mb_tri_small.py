#!/usr/bin/env python3
"""Round 6: the two products of a 100-concept layer against the cached inverse factor (128 rows against a 3072 x 3072 triangle:
Yt = Kt X^T and P = Yt X, `inv_apply` of the n100 record: 54 us each for 1.2 GF) on the stream-K form at several workgroup
counts, and on the K-split paired-tile form; the MFMA floor is 1.2 GF / 78.6 TF = 15 us, reading the triangle once 38 MB."""
import json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip

dev = "cuda"
d = 3072


def timeit(fn, iters=30, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


X = torch.tril(torch.randn(d, d, dtype=torch.float64, device=dev))
for M in (128, 256):
    for name, tb, tri in (("Yt=Kt*X^T", 0, 1), ("P=Yt*X", 1, 2)):
        A = torch.randn(M, d, dtype=torch.float64, device=dev)
        ref = A @ (X.t() if tb == 0 else X)
        flops = M * d * d
        for wgs in (128, 192, 256, 384, 512, 768, 1024):
            C = torch.zeros(M, d, dtype=torch.float64, device=dev)
            hip.dgemm_streamk(tb, A, X, C, flags=tri, wgs=wgs)
            err = float((C - ref).abs().max() / ref.abs().max())
            dt = timeit(lambda: hip.dgemm_streamk(tb, A, X, C, flags=tri, wgs=wgs))
            print(json.dumps({"M": M, "shape": name, "variant": f"two-phase streamK wgs={wgs}", "us": round(dt * 1e6, 1),
                              "tflops_tri": round(flops / dt / 1e12, 1), "err": err}), flush=True)
        for label, kw in (("cfg1 64x64", dict(flags=tri, cfg=1)), ("cfg2 pair", dict(flags=tri | 32, cfg=2)),
                          ("cfg1 ksplit4 (atomic, beta=1)", dict(flags=tri, cfg=1, ksplit=4, beta=1.0)),
                          ("cfg1 ksplit8 (atomic, beta=1)", dict(flags=tri, cfg=1, ksplit=8, beta=1.0)),
                          ("cfg2 ksplit4 (atomic, beta=1)", dict(flags=tri, cfg=2, ksplit=4, beta=1.0))):
            C = torch.zeros(M, d, dtype=torch.float64, device=dev)
            kw = dict(kw)
            beta = kw.pop("beta", 0.0)
            try:
                hip.dgemm_ex(0, tb, A, X, C, beta=beta, **kw)
            except Exception as e:
                print(json.dumps({"M": M, "shape": name, "variant": label, "error": repr(e)[:200]}), flush=True)
                continue
            err = float((C - ref).abs().max() / ref.abs().max())
            dt = timeit(lambda: hip.dgemm_ex(0, tb, A, X, C, beta=beta, **kw))
            print(json.dumps({"M": M, "shape": name, "variant": label, "us": round(dt * 1e6, 1),
                              "tflops_tri": round(flops / dt / 1e12, 1), "err_first_call": err}), flush=True)

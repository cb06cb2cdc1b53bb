#!/usr/bin/env python3
"""Do other hipBLASLt/rocBLAS solutions beat the default heuristic on the encoder's fp32 projection shapes?"""
import sys, time, os
import torch
import torch.nn.functional as F
dev = "cuda:0"
rows = 6292
shapes = [(768, 2304), (768, 768), (768, 3072), (3072, 768)]


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


xs = {k: torch.randn(rows, k, device=dev) for k in (768, 3072)}
ws = {(k, n): (torch.randn(n, k, device=dev), torch.randn(n, device=dev)) for k, n in shapes}
base = {}
for (k, n) in shapes:
    w, b = ws[(k, n)]
    base[(k, n)] = bench(lambda: F.linear(xs[k], w, b))
if os.environ.get("TUNE", "1") == "1":
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.set_filename("/tmp/emcid_tunable.csv")
    torch.cuda.tunable.set_max_tuning_duration(200)
    torch.cuda.tunable.set_max_tuning_iterations(20)
    t0 = time.perf_counter()
    for (k, n) in shapes:
        w, b = ws[(k, n)]
        F.linear(xs[k], w, b)
    torch.cuda.synchronize()
    print(f"tuning took {time.perf_counter() - t0:.1f} s")
    torch.cuda.tunable.tuning_enable(False)
for (k, n) in shapes:
    w, b = ws[(k, n)]
    t = bench(lambda: F.linear(xs[k], w, b))
    fl = 2.0 * rows * k * n
    print(f"{rows}x{k}->{n}: default {base[(k, n)] * 1e6:7.1f} us ({fl / base[(k, n)] / 1e12:5.1f} TF)   tuned {t * 1e6:7.1f} us ({fl / t / 1e12:5.1f} TF)")

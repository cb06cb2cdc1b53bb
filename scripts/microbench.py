#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (HIP events on the launch stream). Prints one JSON line per case."""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip  # noqa: E402

DEV = "cuda:0"
F64_PEAK = 78.6e12   # [external: AMD MI355X datasheet] fp64 matrix
F32_PEAK = 157.3e12  # MI355X_MICROARCH.md


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_dgemm():
    for (ta, tb, M, N, K) in [(1, 1, 3072, 3072, 1024), (0, 0, 3072, 3072, 128), (0, 0, 1024, 3072, 128),
                              (0, 0, 4096, 4096, 4096), (1, 1, 768, 3072, 1024), (0, 0, 1024, 128, 128),
                              (1, 1, 5120, 5120, 1024)]:
        A = torch.randn((M, K) if ta == 0 else (K, M), dtype=torch.float64, device=DEV)
        B = torch.randn((N, K) if tb == 0 else (K, N), dtype=torch.float64, device=DEV)
        Cm = torch.zeros(M, N, dtype=torch.float64, device=DEV)
        dt = timeit(lambda: hip.dgemm(ta, tb, A, B, Cm))
        fl = 2.0 * M * N * K
        print(json.dumps({"case": "dgemm", "ta": ta, "tb": tb, "M": M, "N": N, "K": K, "ms": dt * 1e3,
                          "tflops": fl / dt / 1e12, "frac_f64_peak": fl / dt / F64_PEAK}))


def bench_edit_layer():
    for (N, d, h) in [(100, 3072, 768), (1000, 3072, 768), (1000, 5120, 1280)]:
        g = torch.Generator().manual_seed(0)
        K = (torch.randn(N, d, generator=g) * 0.3).to(DEV)
        Zc = torch.randn(N, h, generator=g).to(DEV)
        zs_t = torch.randn(N, h, generator=g).to(DEV)
        x = torch.randn(2 * d, d, generator=g).to(DEV)
        Cov = (x.t() @ x) / (2 * d)
        W0 = (torch.randn(h, d, generator=g) * 0.02).to(DEV)
        W = torch.empty_like(W0)
        ws = hip.EditWorkspace(N, d, h, DEV)
        dt = timeit(lambda: hip.edit_layer(K, Zc, zs_t, Cov, 4000.0, 0.5, 4, W0=W0, W=W, ws=ws), iters=5)
        fl = N * d * d + d ** 3 / 3 + 2 * N * d * d + 2 * h * N * d   # SURVEY.md §8d algorithmic flops
        print(json.dumps({"case": "edit_layer", "N": N, "d": d, "h": h, "ms": dt * 1e3, "tflops": fl / dt / 1e12,
                          "frac_f64_peak": fl / dt / F64_PEAK, "info": int(ws.info.item())}))
        # stages
        dp = (d + 127) // 128 * 128
        A = (Cov.double() * 4000 + torch.eye(d, device=DEV, dtype=torch.float64))
        Ap = torch.eye(dp, dtype=torch.float64, device=DEV)
        Ap[:d, :d] = A
        def chol():
            hip.cholesky(Ap.clone())
        t_clone = timeit(lambda: Ap.clone(), iters=5)
        t_chol = timeit(chol, iters=5) - t_clone
        L, inv, info = hip.cholesky(Ap.clone())
        Np = (N + 63) // 64 * 64
        Bt = torch.randn(Np, dp, dtype=torch.float64, device=DEV)
        t_solve = timeit(lambda: hip.cholesky_solve_(L, inv, Bt), iters=5)
        print(json.dumps({"case": "stages", "N": N, "d": d, "chol_ms": t_chol * 1e3,
                          "chol_tflops": d ** 3 / 3 / t_chol / 1e12, "solve_ms": t_solve * 1e3,
                          "solve_tflops": 2 * Np * d * d / t_solve / 1e12}))


def bench_gram():
    for (t, d, ks) in [(3072, 3072, 0), (3072, 3072, 1), (32768, 3072, 0), (32768, 5120, 0)]:
        X = torch.randn(t, d, device=DEV)
        G = torch.zeros(d, d, device=DEV)
        dt = timeit(lambda: hip.gram_accumulate_(G, X, ks))
        fl = float(t) * d * d   # SYRK count
        print(json.dumps({"case": "gram", "t": t, "d": d, "ksplit": ks, "ms": dt * 1e3, "tflops": fl / dt / 1e12,
                          "frac_f32_peak": fl / dt / F32_PEAK}))
        dt2 = timeit(lambda: X.t() @ X)
        print(json.dumps({"case": "gram_torch_mm", "t": t, "d": d, "ms": dt2 * 1e3, "tflops_syrk_equiv": fl / dt2 / 1e12}))


if __name__ == "__main__":
    which = sys.argv[1:] or ["dgemm", "edit", "gram"]
    if "dgemm" in which: bench_dgemm()
    if "edit" in which: bench_edit_layer()
    if "gram" in which: bench_gram()

#!/usr/bin/env python3
"""Calibration point for the split-fp16 projection kernel: torch's (hipBLASLt) bf16 / fp16 matmul on shapes with the SAME matrix-pipe
work as the four SD projections at 6 400 rows (K' = 3 K: three f16 MFMAs per algorithmic multiply-add) — what the vendor
library's tuned 16-bit kernels reach on GEMMs of this size on this chip.  Not on the product path."""
import time, torch
dev = "cuda"


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, M, K, N in (("qkv", 6400, 768, 2304), ("out", 6400, 768, 768), ("fc1", 6400, 768, 3072), ("fc2", 6400, 3072, 768)):
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(M, 3 * K, device=dev, dtype=dt)
        w = torch.randn(N, 3 * K, device=dev, dtype=dt)
        t = timeit(lambda: torch.nn.functional.linear(x, w))
        print(f"{name:4s} {M}x{3 * K}->{N} {str(dt)[6:]:9s} {t:7.1f} us  {2.0 * M * N * 3 * K / t / 1e6:7.1f} TF of MFMA work "
              f"= {2.0 * M * N * K / t / 1e6:6.1f} TF-equivalent", flush=True)

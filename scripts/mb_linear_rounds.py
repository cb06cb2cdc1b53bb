#!/usr/bin/env python3
"""How much of the 128 x 128 four-wave kernel's rate is lost to the partial last round of tiles: the qkv / fc1 projections at row
counts that fill the 512 workgroup slots (2 per compute unit) exactly, and at the trie forward's 6 400 rows."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip

dev = "cuda"
CFG = 64 + 1 + 4 * 1      # 128 x 128, 4 waves, prefetch 2


def timeit(fn, n=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, K, N in (("qkv", 768, 2304), ("fc1", 768, 3072), ("qkv-bigG", 1280, 3840)):
    for M in (3584, 5376, 6400, 7168, 8192, 10752, 14336):
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * 0.05
        b = torch.randn(N, device=dev, generator=g)
        y = torch.empty(M, N, device=dev)
        tiles = (M // 128) * (N // 128)
        t = timeit(lambda: hip.linear(x, w, b, out=y, cfg=CFG))
        print(f"{name:9s} M={M:6d} tiles {tiles:5d} = {tiles / 512:5.2f} rounds of 512 | {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.1f} TF", flush=True)

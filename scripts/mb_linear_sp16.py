#!/usr/bin/env python3
"""Microbenchmark of csrc/gemm_sp16.hip (emcid_linear_sp16_f32: fp32-accurate projections as three f16 MFMAs on split operands)
against the exact-f32 kernel (emcid_linear_f32) on the projection shapes of the trie forward, with the error of both against the
fp64 product (max and rms, relative to the largest |result|).  One line per shape."""
import os, sys, time
from pathlib import Path
import torch
import torch.nn.functional as F
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip

dev = "cuda"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
shapes = [(rows, 768, 2304, "qkv"), (rows, 768, 768, "out"), (rows, 768, 3072, "fc1"), (rows, 3072, 768, "fc2"),
          (3072, 768, 768, "out@query"), (3072, 768, 3072, "fc1@query"), (1000, 3072, 768, "fc2(K)"), (640, 768, 2304, "qkv@n100"),
          (640, 3072, 768, "fc2@n100"), (1000, 5120, 1280, "fc2(K)-bigG"),
          (36335, 768, 2304, "qkv-36k"), (36335, 768, 3072, "fc1-36k"), (2048, 768, 3072, "fc1-2k"), (12000, 768, 2304, "qkv-12k"), (12000, 768, 3072, "fc1-12k"),
          (rows, 1280, 3840, "qkv-bigG"), (rows, 1280, 5120, "fc1-bigG"), (rows, 5120, 1280, "fc2-bigG")]


if os.environ.get("MB_SHAPES"):
    shapes = [sh for sh in shapes if sh[3] in os.environ["MB_SHAPES"].split(",")]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for M, K, N, name in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    b = torch.randn(N, device=dev, generator=g)
    y = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    ref = F.linear(x.double(), w.double(), b.double())
    top = ref.abs().max().item()
    xs, ws = hip.split_rows(x), hip.split_rows(w)
    e32 = (hip.linear(x, w, b).double() - ref).abs()
    e16 = (hip.linear_sp(xs, ws, b).double() - ref).abs()
    line = [f"{name:11s} {M}x{K}->{N}  err f32 max {e32.max().item() / top:.1e} rms {e32.pow(2).mean().sqrt().item() / top:.1e}"
            f" | sp16 max {e16.max().item() / top:.1e} rms {e16.pow(2).mean().sqrt().item() / top:.1e}"]
    t = timeit(lambda: hip.linear(x, w, b, out=y))
    line.append(f"f32 {t:7.1f} us {fl / t / 1e6:6.1f} TF")
    t = timeit(lambda: hip.split_rows(x))
    line.append(f"split(x) {t:6.1f} us")
    names = {0: "128x128", 1: "pp2x128x128", 2: "64x64", 3: "160x128k2"}
    cfgs = [(-1, "auto")] + [(tile + 4 * (pf - 1), f"{names[tile]}/pf{pf}") for tile in (0, 1, 2, 3) for pf in (1, 2)]
    if os.environ.get("MB_DBG", "0") == "1":       # timing-only variants (wrong results)
        cfgs += [(4 + 16, "128x128/noload"), (4 + 32, "128x128/noload-nostore"), (4 + 48, "128x128/mfma-only")]
        cfgs += [(3 + 16, "160x128k2/noload"), (3 + 32, "160x128k2/noload-nostore"), (3 + 48, "160x128k2/mfma-only")]
    cfgs += [(64, "dma256x256"), (128, "dma128x128"), (192, "dma160x128k2")]
    if os.environ.get("MB_DBG", "0") == "1":
        cfgs += [(64 + 16, "dma256x256/noload")]
    yauto = hip.linear_sp(xs, ws, b)
    for cfg, cn in cfgs:
        t = timeit(lambda: hip.linear_sp(xs, ws, b, out=y, cfg=cfg))
        line.append(f"{cn} {t:6.1f} us {fl / t / 1e6:5.1f} TF")
        if cfg in (64, 128, 192):        # the LDS-DMA kernels contract in another order than the register-staged ones: compare, do not equate
            d = (hip.linear_sp(xs, ws, b, cfg=cfg) - yauto).abs().max().item() / top
            line.append(f"(vs auto {d:.1e})")
    print(" | ".join(line), flush=True)

#!/usr/bin/env python3
"""Microbenchmark of csrc/gemm_f32.hip (emcid_linear_f32) against torch's F.linear (hipBLASLt: library default and
TunableOp-tuned) on the projection shapes of the trie forward: rows x K -> N with fused epilogues.  Prints one line per
(shape, variant): microseconds and TFLOP/s (2 M N K)."""
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
          (640, 768, 768, "out@n100"), (640, 768, 3072, "fc1@n100"), (640, 3072, 768, "fc2@n100"), (1000, 5120, 1280, "fc2(K)-bigG"),
          (rows, 1280, 3840, "qkv-bigG"), (rows, 1280, 5120, "fc1-bigG"), (rows, 5120, 1280, "fc2-bigG")]
tune = os.environ.get("MB_TUNE", "1") == "1"


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
    r = torch.randn(M, N, device=dev, generator=g)
    y = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    ref = F.linear(x.double(), w.double(), b.double())
    out = hip.linear(x, w, b)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    line = [f"{name:10s} {M}x{K}->{N}  err {err:.1e}"]
    t = timeit(lambda: F.linear(x, w, b))
    line.append(f"torch {t:7.1f} us {fl / t / 1e6:6.1f} TF")
    if tune:
        tn = torch.cuda.tunable
        tn.enable(True); tn.tuning_enable(True); tn.set_max_tuning_duration(30); tn.set_max_tuning_iterations(10)
        F.linear(x, w, b); torch.cuda.synchronize()
        tn.tuning_enable(False)
        t = timeit(lambda: F.linear(x, w, b))
        tn.enable(False)
        line.append(f"tuned {t:7.1f} us {fl / t / 1e6:6.1f} TF")
    names = {0: "160x128", 1: "128x128", 2: "256x128", 3: "64x64"}
    cfgs = [(-1, "auto")] + [(tile + 4 * (pf - 1), f"{names[tile]}/8w/pf{pf}") for tile in (0, 1, 2, 3) for pf in (1, 2)]
    cfgs += [(64 + tile + 4 * (pf - 1), f"{names[tile]}/4w/pf{pf}") for tile, pf in ((0, 3), (1, 2), (3, 3))]
    t128 = -(-M // 128) * -(-N // 128)
    cfgs += [(hip.linear_split_cfg(parts), f"128x128/splitK{parts}") for parts in (2, 3, 4, 6, 8) if t128 * parts <= 512 and t128 <= 256]
    if os.environ.get("MB_DBG", "0") == "1":       # timing-only variants (wrong results): no loads in the loop / no LDS stores either
        cfgs += [(0 + 4 + 16, "160x128/4w/noload"), (0 + 4 + 32, "160x128/4w/noload-nostore")]
    for cfg, cn in cfgs:
        t = timeit(lambda: hip.linear(x, w, b, out=y, cfg=cfg))
        line.append(f"{cn} {t:6.1f} us {fl / t / 1e6:5.1f} TF")
    t = timeit(lambda: hip.linear(x, w, b, act=hip.ACT_QUICK_GELU, out=y))
    line.append(f"+gelu {t:7.1f}")
    t = timeit(lambda: hip.linear(x, w, b, residual=r, out=y))
    line.append(f"+res {t:7.1f}")
    print(" | ".join(line), flush=True)

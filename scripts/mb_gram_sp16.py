#!/usr/bin/env python3
"""The Stage-0 Gram on the split-fp16 path (emcid_gram_accumulate_sp16_f32) against the exact-f32 SYRK (emcid_gram_accumulate_f32):
microseconds per batch, TF by the SYRK count t d^2, error of both against fp64."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for t, d in ((16384, 3072), (32768, 3072), (106000, 3072), (32768, 5120)):
    x = torch.randn(t, d, device="cuda")
    G = torch.zeros(d, d, device="cuda")
    res = {}
    for name, flag in (("split-fp16", True), ("exact-f32", False)):
        hip.GRAM_SPLIT = flag
        G.zero_()
        hip.gram_accumulate_(G, x, 0)
        if t <= 32768:
            ref = x.double().t() @ x.double()
            low = torch.tril(G.double())
            res[name + " err"] = ((low - torch.tril(ref)).abs().max() / ref.abs().max()).item()
        us = timeit(lambda: hip.gram_accumulate_(G, x, 0))
        res[name] = f"{us:9.1f} us {t * d * d / us / 1e6:6.1f} TF(SYRK count)"
    hip.GRAM_SPLIT = True
    print(t, d, res, flush=True)

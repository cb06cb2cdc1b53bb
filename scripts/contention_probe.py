#!/usr/bin/env python3
"""The factorization chain (side stream) underneath a stream of fp32 GEMMs shaped like the encoder forward."""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
dev = "cuda:0"
d, L = 3072, 4
covs = []
for l in range(L):
    X = torch.randn(2 * d, d, device=dev)
    covs.append((X.t() @ X / (2 * d)).contiguous())
fac = hip.factor_cov(covs, 4000.0, 0.5, inverse=False)
A = torch.randn(6292, 768, device=dev)
W1 = torch.randn(768, 3072, device=dev)
W2 = torch.randn(3072, 768, device=dev)
side = torch.cuda.Stream(priority=int(os.environ.get("PRIO", "0")))
main = torch.cuda.current_stream()


def run(n_gemm, with_chain):
    torch.cuda.synchronize()
    e0, e1, s0, s1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e0.record()
    if with_chain:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            s0.record()
            hip.factor_cov(covs, 4000.0, 0.5, fac, inverse=False)
            s1.record()
    x = A
    for _ in range(n_gemm):
        hdn = x @ W1
        x = hdn @ W2
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), (s0.elapsed_time(s1) if with_chain else 0.0)


for _ in range(2):
    run(12, True)
print("gemms alone    : main %.3f ms" % run(12, False)[0])
print("chain alone    : side %.3f ms" % run(0, True)[1])
m, s = run(12, True)
print("both           : main %.3f ms, side %.3f ms" % (m, s))

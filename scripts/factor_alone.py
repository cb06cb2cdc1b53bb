#!/usr/bin/env python3
"""factor_cov (batched Cholesky of 4 x 3072^2 + explicit inverse factors) on an otherwise idle GPU."""
import sys, time
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
fac = hip.factor_cov(covs, 4000.0, 0.5)
torch.cuda.synchronize()
for _ in range(3):
    fac = hip.factor_cov(covs, 4000.0, 0.5, fac)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    fac = hip.factor_cov(covs, 4000.0, 0.5, fac)
torch.cuda.synchronize()
print(f"factor_cov alone: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms, info {int(fac.info.item())}")

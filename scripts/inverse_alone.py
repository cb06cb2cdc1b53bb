#!/usr/bin/env python3
"""cov_inverse (X = inv(L) for 4 x 3072^2 factors, batched) on an otherwise idle GPU + its accuracy."""
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
    hip.cov_inverse(fac)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    hip.cov_inverse(fac)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
flops = L * sum(512 * (512 * i) ** 2 + 512 * 512 * (512 * i) for i in range(1, d // 512))
err = max(float((torch.tril(fac.X(l)) @ torch.tril(fac.L(l)) - torch.eye(d, dtype=torch.float64, device=dev)).abs().max()) for l in range(L))
print(f"cov_inverse(all) alone: {dt * 1e3:.3f} ms  {flops / dt / 1e12:.1f} TF  |XL-I| {err:.2e}")

#!/usr/bin/env python3
"""emcid_cholesky_f64 alone (graph replay through the C ABI) for the sizes the dual solver factors."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
dev = "cuda:0"
for n in (1024, 3072):
    X = torch.randn(2 * n, n, dtype=torch.float64, device=dev)
    A0 = (X.t() @ X / (2 * n) + torch.eye(n, dtype=torch.float64, device=dev)).contiguous()
    L = torch.zeros_like(A0)
    inv = torch.empty(int(hip.load().emcid_inverse_workspace_doubles(n)), dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    A = A0.clone()

    def run():
        A.copy_(A0)
        hip._check(hip.load().emcid_cholesky_f64(hip._ptr(A), hip._ptr(L), n, n, hip._ptr(inv), hip._ptr(info, torch.int32),
                                                  hip._stream(A)), "chol")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    err = float((torch.tril(L) @ torch.tril(L).t() - A0).abs().max())
    print(f"cholesky n={n}: {dt * 1e6:8.1f} us (incl. a {n}x{n} copy)  err {err:.2e} info {int(info.item())}")

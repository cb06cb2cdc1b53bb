#!/usr/bin/env python3
"""The dominant fp64 kernel of the bench (Yt = Kt X^T against the explicit inverse factor: gemm_f64_streamk_kernel, 128x128
tiles, triangular K range cut into 256 equal runs, 1024 x 3072 x 3072) a few times — target for `rocprofv3 --pmc` passes
(FETCH_SIZE and WRITE_SIZE in separate runs, MI355X_MICROARCH.md §rocprofv3 PMC slots)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
d, N = 3072, 1024
X = torch.tril(torch.randn(d, d, dtype=torch.float64, device="cuda:0"))
Kt = torch.randn(N, d, dtype=torch.float64, device="cuda:0")
Yt = torch.zeros(N, d, dtype=torch.float64, device="cuda:0")
for _ in range(5):
    hip.dgemm_ex(0, 0, Kt, X, Yt, flags=1, cfg=4, ksplit=256)
torch.cuda.synchronize()

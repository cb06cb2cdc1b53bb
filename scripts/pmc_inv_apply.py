#!/usr/bin/env python3
"""The dominant fp64 kernel of the bench (Yt = Kt X^T against the explicit inverse factor: two-phase stream-K, 128x128 tiles,
triangular K range cut into 256 equal runs, 1024 x 3072 x 3072) and its sibling (U = V X, 768 rows) a few times — target for
`rocprofv3 --pmc` passes (scripts/pmc_passes.sh)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
d, N = 3072, 1024
X = torch.tril(torch.randn(d, d, dtype=torch.float64, device="cuda:0"))
Kt = torch.randn(N, d, dtype=torch.float64, device="cuda:0")
Yt = torch.zeros(N, d, dtype=torch.float64, device="cuda:0")
V = torch.randn(768, d, dtype=torch.float64, device="cuda:0")
U = torch.zeros(768, d, dtype=torch.float64, device="cuda:0")
for _ in range(5):
    hip.dgemm_streamk(0, Kt, X, Yt, flags=1, wgs=256)
    hip.dgemm_streamk(1, V, X, U, flags=2, wgs=256)
torch.cuda.synchronize()

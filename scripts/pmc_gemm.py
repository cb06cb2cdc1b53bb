#!/usr/bin/env python3
"""One dgemm shape, a few launches — target for rocprofv3 --pmc passes."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
M = N = K = 4096
A = torch.randn(M, K, dtype=torch.float64, device="cuda:0"); B = torch.randn(N, K, dtype=torch.float64, device="cuda:0")
C = torch.zeros(M, N, dtype=torch.float64, device="cuda:0")
for _ in range(3): hip.dgemm(0, 0, A, B, C)
torch.cuda.synchronize()

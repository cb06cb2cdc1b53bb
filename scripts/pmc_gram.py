#!/usr/bin/env python3
"""A few launches of the Gram kernel at Stage-0 shape (t 110 000 rows, d 3072) — target for `rocprofv3 --pmc` passes."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
t, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 110000), 3072
X = torch.randn(t, d, device="cuda:0"); G = torch.zeros(d, d, device="cuda:0")
for _ in range(4):
    hip.gram_accumulate_(G, X, ksplit=20)
torch.cuda.synchronize()

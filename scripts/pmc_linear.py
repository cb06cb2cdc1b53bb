#!/usr/bin/env python3
"""The forward's projection GEMMs (csrc/gemm_f32.hip, emcid_linear_f32) on the four SD-v1.4 shapes of a 6 400-row trie
(qkv, out, fc1 + quick_gelu, fc2 + residual), a few launches each — target for `rocprofv3 --pmc` passes (scripts/pmc_passes.sh)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
rows = 6400
g = torch.Generator(device="cuda").manual_seed(1)
for K, N, act, res in ((768, 2304, 0, False), (768, 768, 0, True), (768, 3072, 1, False), (3072, 768, 0, True)):
    x = torch.randn(rows, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(rows, N, device="cuda", generator=g) if res else None
    y = torch.empty(rows, N, device="cuda")
    for _ in range(5):
        hip.linear(x, w, b, act=act, residual=r, out=y)
torch.cuda.synchronize()

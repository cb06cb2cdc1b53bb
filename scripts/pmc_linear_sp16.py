#!/usr/bin/env python3
"""The split-fp16 projection kernel (csrc/gemm_sp16.hip, emcid_linear_sp16_f32) on ONE shape of a 6 400-row trie
(argv: qkv | out | fc1 | fc2 [cfg]), a few launches — target for `rocprofv3 --pmc` passes (scripts/pmc_passes.sh)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
rows = 6400
shapes = {"qkv": (768, 2304, 0, False), "out": (768, 768, 0, True), "fc1": (768, 3072, 1, False), "fc2": (3072, 768, 0, True)}
K, N, act, res = shapes[sys.argv[1] if len(sys.argv) > 1 else "qkv"]
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else -1
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows, K, device="cuda", generator=g)
w = torch.randn(N, K, device="cuda", generator=g) * 0.05
b = torch.randn(N, device="cuda", generator=g)
r = torch.randn(rows, N, device="cuda", generator=g) if res else None
y = torch.empty(rows, N, device="cuda")
xs, ws = hip.split_rows(x), hip.split_rows(w)
for _ in range(20):
    hip.linear_sp(xs, ws, b, act=act, residual=r, out=y, cfg=cfg)
torch.cuda.synchronize()

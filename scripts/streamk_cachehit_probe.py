#!/usr/bin/env python3
"""Is the stream-K GEMM against the triangle bound by its operand fetches?  The same launch (1024 x 3072 x 3072, tri = 1) with
both operands given row stride 0 — every row of every tile is the same 128-byte line, so every global load hits L1/L2 — against
the real operands.  (The products are garbage; only the time matters.)"""
import json, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
from scripts.microbench import timeit
dev, d = "cuda:0", 3072
X = torch.tril(torch.randn(d, d, dtype=torch.float64, device=dev))
A = torch.randn(1024, d, dtype=torch.float64, device=dev)
C = torch.zeros(1024, d, dtype=torch.float64, device=dev)
out = {}
out["real_operands_us"] = round(timeit(lambda: hip.dgemm_streamk(0, A, X, C, flags=1, wgs=256), iters=20, warmup=3) * 1e6, 1)
A0, X0 = A[:1].expand(1024, d), X[-1:].expand(d, d)
out["stride0_operands_us"] = round(timeit(lambda: hip.dgemm_streamk(0, A0, X0, C, flags=1, wgs=256), iters=20, warmup=3) * 1e6, 1)
A1 = A[:128].repeat(8, 1)       # 8 row tiles share 128 distinct rows?  no: distinct memory; keep as the L2-resident variant
Xs = X[:128].repeat(24, 1).contiguous()   # B panels all equal to the first 128 rows: 3 MB... still distinct memory
out["note"] = "stride-0: all lines identical (L1 hits)"
print(json.dumps(out))

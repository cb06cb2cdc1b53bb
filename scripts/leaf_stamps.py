#!/usr/bin/env python3
"""Phase timing of the Cholesky leaf (diagnostic build path with s_memtime stamps)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
x = torch.randn(256, 128, dtype=torch.float64, device="cuda:0")
A = (x.t() @ x + torch.eye(128, dtype=torch.float64, device="cuda:0")).contiguous()
L = torch.zeros_like(A); inv = torch.zeros_like(A); info = torch.zeros(1, dtype=torch.int32, device="cuda:0")
st = torch.zeros(32, dtype=torch.int64, device="cuda:0")
lib = hip.load()
for _ in range(3):
    hip._check(lib.emcid_debug_leaf_stamps(hip._ptr(A), hip._ptr(L), hip._ptr(inv), hip._ptr(info), hip._ptr(st), hip._stream(A)), "dbg")
torch.cuda.synchronize()
s = st.cpu().tolist()
names = ["start", "loaded"] + sum([[f"A{p}", f"B{p}", f"C{p}"] for p in range(3)], []) + ["A3", "lvl1", "lvl2", "stored"]
vals = [v for v in s if v][:len(names)]
print("ticks (100 MHz s_memtime? shader clock) deltas:")
for n, a, b in zip(names[1:], vals[:-1], vals[1:]):
    print(f"  {n:8s} {b - a:8d}")
print("total", vals[-1] - vals[0])
err = (torch.tril(L) @ torch.tril(L).t() - A).abs().max().item()
print("factor err", err, "inv err", (inv @ torch.tril(L) - torch.eye(128, dtype=torch.float64, device="cuda:0")).abs().max().item())
import os
mid = os.environ.get("MIDSTAMP") == "1"
a = s[16:16 + (17 if mid else 13)]
if any(a):
    names2 = ["load"] + [f"step{i}" for i in range(4)] + (sum([[f"step{i} valu", f"step{i} mfma"] for i in range(4, 8)], []) if mid else [f"step{i}" for i in range(4, 8)]) + ["(stamp)", "out-lds", "out-stage"]
    print("phase A (panel 1) detail:")
    for n, x, y in zip(names2, a[:-1], a[1:]):
        if y:
            print(f"  {n:10s} {y - x:8d}")

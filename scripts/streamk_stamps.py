#!/usr/bin/env python3
"""Where a two-phase stream-K launch spends its time, per workgroup (in-kernel shader-clock stamps, diagnostic build path):
start skew, end spread, cycles in K loops / partial publishes / last-ticket reductions / epilogues."""
import json, sys, ctypes as C
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip

dev = "cuda:0"
d = 3072
X = torch.tril(torch.randn(d, d, dtype=torch.float64, device=dev))
cases = [("Yt=Kt*X^T", 0, 1024, 1), ("U=G*X", 1, 768, 2)]
lib = hip.load()
for name, tb, M, tri in cases:
    A = torch.randn(M, d, dtype=torch.float64, device=dev)
    Cm = torch.zeros(M, d, dtype=torch.float64, device=dev)
    for _ in range(3):
        hip.dgemm_streamk(tb, A, X, Cm, flags=tri, wgs=256)
    stamps = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
    lib.emcid_debug_streamk_stamps(C.c_void_p(stamps.data_ptr()))
    hip.dgemm_streamk(tb, A, X, Cm, flags=tri, wgs=256)
    torch.cuda.synchronize()
    lib.emcid_debug_streamk_stamps(None)
    s = stamps.view(256, 8).cpu().double()
    t0 = s[:, 0].min()
    q = lambda v: [round(float(x)) for x in torch.quantile(v, torch.tensor([0.0, 0.5, 0.9, 1.0], dtype=torch.float64))]
    print(json.dumps({"shape": name, "unit": "shader cycles (quantiles 0/50/90/100 over 256 workgroups)",
                      "start_after_first": q(s[:, 0] - t0), "end_after_first": q(s[:, 1] - t0), "lifetime": q(s[:, 1] - s[:, 0]),
                      "k_loops": q(s[:, 2]), "publish": q(s[:, 3]), "reduce": q(s[:, 4]), "epilogue": q(s[:, 5]),
                      "segments": q(s[:, 6]), "k_loop_cycles_per_kstep": round(float(s[:, 2].sum() / (19200 if tri == 1 else 14400)), 1)}))

#!/usr/bin/env python3
"""Accuracy of the dual solver (explicit inverse factors) when the statistics are ill-conditioned: C = Q diag(s) Q^T with
log-uniform spectrum over `decades` decades; error of dW against the fp64 LU of the full system (same inputs, computed
here with torch in fp64 on the GPU — a check of the algebra, not the parity oracle)."""
import sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
dev = "cuda:0"
N, d, h, lam, ew = 200, 3072, 768, 4000.0, 0.5
g = torch.Generator(device=dev).manual_seed(0)
Q, _ = torch.linalg.qr(torch.randn(d, d, dtype=torch.float64, device=dev, generator=g))
for decades in (2, 4, 6, 8, 10):
    sp = torch.logspace(0, -decades, d, dtype=torch.float64, device=dev)
    C = ((Q * sp) @ Q.t()).float().contiguous()
    C = ((C + C.t()) * 0.5).contiguous()
    K = torch.randn(N, d, device=dev, generator=g) * 0.3
    Zc = torch.randn(N, h, device=dev, generator=g)
    zs_t = torch.randn(N, h, device=dev, generator=g)
    W0 = torch.randn(h, d, device=dev, generator=g)
    s = (ew / 0.5) ** 0.5
    K64 = K.double() * s
    A = lam * (C * (1 - ew) / 0.5).double() + K64.t() @ K64
    adj = torch.linalg.solve(A, K64.t())
    upd = ((zs_t - Zc).double() * s).t() @ adj.t()
    out = {}
    for use_inv in (True, False):
        fac = hip.factor_cov([C], lam, ew, inverse=use_inv)
        W = torch.empty(h, d, device=dev)
        res = hip.edit_layer_dual_apply(K, Zc, zs_t, fac, 0, ew, 1, W0, W, use_inverse=use_inv)
        ok = int(fac.info.item()) == 0 and int(res["ws"].info.item()) == 0
        out[use_inv] = ((res["dW"].double() - upd).abs().max() / upd.abs().max()).item() if ok else float("nan")
    print(f"cond(C) ~ 1e{decades}: rel dW error  GEMMs against inv(L) {out[True]:.2e}   block substitution {out[False]:.2e}")

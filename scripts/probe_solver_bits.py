#!/usr/bin/env python3
"""Is the apply-only dual solve (emcid_edit_dual_apply_*) bit-reproducible from call to call?  Fixed K / Zc / v* / factors, 40
calls per concept count with other work of varying length put on the stream in front of each; distinct results counted."""
import hashlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
DEV = "cuda:0"
d, h, lam, ew = 3072, 768, 4000.0, 0.5
g = torch.Generator().manual_seed(1)
x = torch.randn(2 * d, d, generator=g) * torch.exp(torch.linspace(0, -3, d))
Cov = ((x.t() @ x) / (2 * d)).to(DEV)
fac = hip.factor_cov([Cov * 1.5 + torch.eye(d, device=DEV) * 1e-3, Cov], lam, ew)
noise = torch.randn(4096, 4096, device=DEV)
for N in (int(v) for v in (sys.argv[1:] or ["40", "64", "100", "128", "200", "1000"])):
    K = (torch.randn(N, d, generator=g) * 0.3).to(DEV)
    Zc = torch.randn(N, h, generator=g).to(DEV)
    zs_t = torch.randn(N, h, generator=g).to(DEV)
    W0 = (torch.randn(h, d, generator=g) * 0.02).to(DEV)
    for use_inverse in (True, False):
        seen = {}
        for it in range(40):
            W = torch.empty(h, d, dtype=torch.float32, device=DEV)
            for _ in range(it % 4):
                noise @ noise
            hip.edit_layer_dual_apply(K, Zc, zs_t, fac, 1, ew, 2, W0, W, use_inverse=use_inverse)
            torch.cuda.synchronize()
            key = hashlib.md5(W.cpu().numpy().tobytes()).hexdigest()
            seen[key] = seen.get(key, 0) + 1
        print(f"N {N:5d} use_inverse {use_inverse}: {len(seen)} distinct results in 40 calls {sorted(seen.values(), reverse=True)}", flush=True)

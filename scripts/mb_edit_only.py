#!/usr/bin/env python3
import json, os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
from scripts.microbench import timeit
N, d, h = 1000, 3072, 768
g = torch.Generator().manual_seed(0)
K = (torch.randn(N, d, generator=g) * 0.3).cuda(); Zc = torch.randn(N, h, generator=g).cuda(); zs = torch.randn(N, h, generator=g).cuda()
x = torch.randn(2 * d, d, generator=g).cuda(); Cov = (x.t() @ x) / (2 * d)
W0 = (torch.randn(h, d, generator=g) * 0.02).cuda(); W = torch.empty_like(W0)
ws = hip.EditWorkspace(N, d, h, "cuda:0")
dt = timeit(lambda: hip.edit_layer(K, Zc, zs, Cov, 4000.0, 0.5, 4, W0=W0, W=W, ws=ws), iters=20, warmup=3)
print(json.dumps({"GRAPH": os.environ.get("EMCID_GRAPH"), "LOOKAHEAD": os.environ.get("EMCID_LOOKAHEAD"), "edit_layer_ms": round(dt * 1e3, 3), "info": int(ws.info.item())}))

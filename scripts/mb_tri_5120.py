import json, sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from emcid_amd import hip
from scripts.microbench import timeit
dev, d = "cuda:0", 5120
X = torch.tril(torch.randn(d, d, dtype=torch.float64, device=dev))
for M, name, tb, tri in ((1024, "Yt=Kt*X^T", 0, 1), (1280, "U=V*X", 1, 2), (384, "Yt(N=300)", 0, 1)):
    A = torch.randn(M, d, dtype=torch.float64, device=dev)
    C = torch.zeros(M, d, dtype=torch.float64, device=dev)
    t1 = timeit(lambda: hip.dgemm_streamk(tb, A, X, C, flags=tri, wgs=256), iters=10, warmup=3)
    t2 = timeit(lambda: hip.dgemm_ex(0, tb, A, X, C, beta=0.0, flags=tri | 32, cfg=2), iters=10, warmup=3)
    print(json.dumps({"d": d, "M": M, "shape": name, "streamk_us": round(t1 * 1e6, 1), "pairs32x64_us": round(t2 * 1e6, 1), "pair_wgs": (M // 32) * 20}))

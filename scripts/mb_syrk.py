import json, sys, torch
sys.path.insert(0, "/root/repo" if __import__("os").path.exists("/root/repo/emcid_amd") else __import__("os").environ["GRAFT_REPO_ROOT"])
from emcid_amd import hip
from scripts.microbench import timeit
dev, d = "cuda:0", 3072
Y = torch.randn(1024, d, dtype=torch.float64, device=dev)
ref = torch.tril(Y @ Y.t())
S = torch.zeros(1024, 1024, dtype=torch.float64, device=dev)
dt = timeit(lambda: hip.dgemm_streamk(0, Y, Y, S, flags=16, wgs=256, diag_add=0.0), iters=20, warmup=3)
print("streamk", round(dt * 1e6, 1))
for cfg in (2, 1, 0):
    for ks in (1, 2, 4):
        S.zero_()
        try:
            hip.dgemm_ex(0, 0, Y, Y, S, beta=0.0 if ks == 1 else 1.0, flags=16, cfg=cfg, ksplit=ks)
            err = float((torch.tril(S) - ref).abs().max() / ref.abs().max())
            f = lambda: hip.dgemm_ex(0, 0, Y, Y, S, beta=0.0 if ks == 1 else 1.0, flags=16, cfg=cfg, ksplit=ks)
            dt = timeit(f, iters=20, warmup=3)
            print("cfg", cfg, "ksplit", ks, round(dt * 1e6, 1), "err", err)
        except Exception as e:
            print("cfg", cfg, ks, "ERR", str(e)[:80])

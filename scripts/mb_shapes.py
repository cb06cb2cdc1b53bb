#!/usr/bin/env python3
import json, os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
from scripts.microbench import timeit
for (ta, tb, M, N, K, beta) in [(0, 0, 1000, 2560, 512, 1.0), (0, 0, 1000, 1536, 512, 1.0), (0, 0, 1000, 512, 512, 1.0),
                                (0, 1, 1000, 2560, 512, 1.0), (0, 0, 2560, 512, 512, 1.0), (0, 0, 1024, 512, 2048, 1.0),
                                (0, 0, 1000, 512, 512, 0.0)]:
    A = torch.randn((M, K) if ta == 0 else (K, M), dtype=torch.float64, device="cuda:0")
    B = torch.randn((N, K) if tb == 0 else (K, N), dtype=torch.float64, device="cuda:0")
    C = torch.zeros(M, N, dtype=torch.float64, device="cuda:0")
    dt = timeit(lambda: hip.dgemm(ta, tb, A, B, C, alpha=-1.0, beta=beta), iters=20, warmup=3)
    print(json.dumps({"cfg": os.environ.get("EMCID_GEMM_CFG"), "ks": os.environ.get("EMCID_GEMM_KSPLIT"), "ta": ta, "tb": tb, "M": M, "N": N, "K": K, "beta": beta,
                      "us": round(dt * 1e6, 1), "tflops": round(2.0 * M * N * K / dt / 1e12, 1)}))

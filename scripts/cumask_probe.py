#!/usr/bin/env python3
"""Does a CU partition (hipExtStreamCreateWithCUMask) let the factorization chain run beside the forward-shaped GEMMs?
side stream = `n_side` CUs, main stream = the others; layouts: 'block' (the first n CUs) or 'stride' (every 256/n-th)."""
import ctypes, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip
dev = "cuda:0"
torch.cuda.init(); torch.zeros(1, device=dev)
rt = ctypes.CDLL("libamdhip64.so")
rt.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
rt.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


d, L = 3072, 4
covs = []
for l in range(L):
    X = torch.randn(2 * d, d, device=dev)
    covs.append((X.t() @ X / (2 * d)).contiguous())
fac = hip.factor_cov(covs, 4000.0, 0.5, inverse=False)
A = torch.randn(6292, 768, device=dev)
W1 = torch.randn(768, 3072, device=dev)
W2 = torch.randn(3072, 768, device=dev)


def run(main, side, n_gemm, with_chain):
    torch.cuda.synchronize()
    e0, e1, s0, s1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.cuda.stream(main):
        e0.record()
    if with_chain:
        with torch.cuda.stream(side):
            s0.record()
            hip.factor_cov(covs, 4000.0, 0.5, fac, inverse=False)
            s1.record()
    with torch.cuda.stream(main):
        x = A
        for _ in range(n_gemm):
            hdn = x @ W1
            x = hdn @ W2
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), (s0.elapsed_time(s1) if with_chain else 0.0)


plain_main, plain_side = torch.cuda.Stream(), torch.cuda.Stream()
configs = [("no mask", plain_main, plain_side)]
for n_side, layout in [(32, "block"), (32, "stride"), (64, "block"), (64, "stride"), (48, "block")]:
    side_bits = list(range(n_side)) if layout == "block" else list(range(0, 256, 256 // n_side))
    main_bits = [b for b in range(256) if b not in set(side_bits)]
    configs.append((f"side {n_side} CUs ({layout})", masked_stream(main_bits), masked_stream(side_bits)))
for name, m, s in configs:
    for _ in range(2):
        run(m, s, 12, True)
    ga = run(m, s, 12, False)[0]
    ca = run(m, s, 0, True)[1]
    bm, bs = run(m, s, 12, True)
    print(f"{name:28s} gemms alone {ga:6.3f}  chain alone {ca:6.3f}  both: main {bm:6.3f} side {bs:6.3f}  end {max(bm, bs):6.3f}")

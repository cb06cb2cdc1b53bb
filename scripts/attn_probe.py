#!/usr/bin/env python3
"""Which SDPA backend is fastest for the edit batch shape (B=3000, H=12, S=9, D=64, fp32, causal mask)?"""
import sys, torch, time
from torch.nn.attention import sdpa_kernel, SDPBackend
import torch.nn.functional as F
dev = "cuda:0"
for (B, H, S, D) in [(3000, 12, 9, 64), (3000, 12, 17, 64), (3000, 12, 77, 64), (3000, 20, 9, 64)]:
    q, k, v = (torch.randn(B, H, S, D, device=dev) for _ in range(3))
    mask = torch.zeros(B, 1, S, S, device=dev).masked_fill(torch.ones(S, S, device=dev).triu(1).bool(), float("-inf"))
    for name, be in [("math", SDPBackend.MATH), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("flash", SDPBackend.FLASH_ATTENTION), ("default", None)]:
        try:
            def run():
                if be is None:
                    return F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
                with sdpa_kernel(be):
                    return F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
            for _ in range(3): run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): run()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            print(f"B{B} H{H} S{S} {name:10s} {dt*1e3:8.3f} ms")
        except Exception as e:
            print(f"B{B} H{H} S{S} {name:10s} ERR {str(e)[:80]}")

#!/usr/bin/env python3
"""Do two independent halves of the trie forward's projections, on two streams, fill each other's partial last rounds?  The four
GEMMs of a layer (qkv, out, fc1 + quick_gelu, fc2 + residual) x 7 layers on 6 400 rows in one stream against 2 x 3 200 rows on two
streams (each stream's launches depend on each other through their operands, as in the forward)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from emcid_amd import hip

dev = "cuda"
h, d, L = 768, 3072, 7
g = torch.Generator(device=dev).manual_seed(1)
Wqkv = torch.randn(3 * h, h, device=dev, generator=g) * 0.03
Wo = torch.randn(h, h, device=dev, generator=g) * 0.03
W1 = torch.randn(d, h, device=dev, generator=g) * 0.03
W2 = torch.randn(h, d, device=dev, generator=g) * 0.03


def chain(x, bufs):
    qkv, o, m, y = bufs
    for _ in range(L):
        hip.linear(x, Wqkv, out=qkv)
        hip.linear(qkv[:, :h], Wo, out=o)
        hip.linear(o, W1, act=hip.ACT_QUICK_GELU, out=m)
        hip.linear(m, W2, residual=x, out=y)
    return y


def bufs(rows):
    return (torch.empty(rows, 3 * h, device=dev), torch.empty(rows, h, device=dev), torch.empty(rows, d, device=dev),
            torch.empty(rows, h, device=dev))


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
x = torch.randn(rows, h, device=dev, generator=g) * 0.1
b_all = bufs(rows)
one = timeit(lambda: chain(x, b_all))
flops = 2.0 * rows * L * (h * 3 * h + h * h + 2 * h * d)
print(f"one stream, {rows} rows: {one:.3f} ms  {flops / one / 1e9:.1f} TF")
for parts in (2, 3):
    per = rows // parts
    xs = [x[i * per:(i + 1) * per] for i in range(parts)]
    bs = [bufs(per) for _ in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]

    def split():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for s, xi, bi in zip(streams, xs, bs):
            with torch.cuda.stream(s):
                chain(xi, bi)
        for s in streams:
            cur.wait_stream(s)

    def serial():
        for xi, bi in zip(xs, bs):
            chain(xi, bi)

    t2 = timeit(split)
    t1 = timeit(serial)
    print(f"{parts} x {per} rows: {parts} streams {t2:.3f} ms {flops / t2 / 1e9:.1f} TF | one after the other {t1:.3f} ms {flops / t1 / 1e9:.1f} TF")

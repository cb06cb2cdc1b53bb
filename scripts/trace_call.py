#!/usr/bin/env python3
"""Ordered launch list of ONE steady-state call from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: calls
are delimited by the embedding + LayerNorm kernel (the first launch of every call).  Prints start offset, duration and the idle
gap before each kernel, then per-kernel totals: what a latency-bound call (100 concepts) spends where.
usage: trace_call.py <dir or csv> [which call, default -3] [marker substring, default embed_layernorm]"""
import csv, glob, os, re, sys
from collections import defaultdict

src = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
marker = sys.argv[3] if len(sys.argv) > 3 else "embed_layernorm"
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
lo = starts[which]
hi = starts[which + 1] if which + 1 < 0 and which + 1 + len(starts) < len(starts) else len(rows)
call = rows[lo:hi]


def short(n):
    n = re.sub(r"^void ", "", n).replace("emcid::", "")
    return re.sub(r"\(.*", "", n)[:70]


t0, prev = call[0][0], call[0][0]
tot, busy = defaultdict(lambda: [0, 0.0]), 0.0
for s, e, n in call:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {short(n)}")
    tot[short(n)][0] += 1
    tot[short(n)][1] += (e - s) / 1e3
    busy += (e - s) / 1e3
    prev = max(prev, e)
print(f"\n{len(call)} kernels, span {(prev - t0) / 1e3:.1f} us, sum of durations {busy:.1f} us")
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{us:9.1f} us  x{c:3d}  {n}")

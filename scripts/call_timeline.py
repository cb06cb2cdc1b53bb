#!/usr/bin/env python3
"""Device timeline of whole apply_* calls from a `rocprofv3 --kernel-trace --output-format csv` run of scripts/host_profile.py:
the trace is cut into calls at idle gaps longer than CUT_MS (between two calls the GPU waits for the host's tokenization), and
for each of the last calls prints span, busy time and every idle gap longer than GAP_US with the kernels on either side.
usage: call_timeline.py <dir or csv> [n_calls=3] [gap_us=25] [cut_ms=1.5]"""
import csv, glob, os, re, sys

src = sys.argv[1]
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
gap_us = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
cut_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 1.5
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    n = re.sub(r"^void ", "", n).replace("emcid::", "").replace("at::native::", "")
    if n.startswith("Cijk"):
        m = re.search(r"MT(\d+x\d+x\d+)", n)
        return "hipblaslt_" + (m.group(1) if m else "gemm")
    return n.split("(")[0][:70]


calls, cur = [], [rows[0]]
end = rows[0][1]
for r in rows[1:]:
    if (r[0] - end) / 1e6 > cut_ms:
        calls.append(cur)
        cur = []
    cur.append(r)
    end = max(end, r[1])
calls.append(cur)
print(f"{len(rows)} launches, {len(calls)} segments")
for seg in calls[-n_calls:]:
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    busy, cs, ce = 0, seg[0][0], seg[0][1]
    gaps = []
    for s, e, n in seg[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append((ce, s))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print(f"--- call: {len(seg)} launches, span {(t1 - t0) / 1e3:.0f} us, busy {busy / 1e3:.0f} us, idle {(t1 - t0 - busy) / 1e3:.0f} us "
          f"({sum(1 for a, b in gaps if (b - a) / 1e3 > gap_us)} gaps > {gap_us:.0f} us holding "
          f"{sum(b - a for a, b in gaps if (b - a) / 1e3 > gap_us) / 1e3:.0f} us)")
    ends = sorted(seg, key=lambda r: r[1])
    for a, b in gaps:
        if (b - a) / 1e3 > gap_us:
            before = max((r for r in seg if r[1] <= a), key=lambda r: r[1])
            after = min((r for r in seg if r[0] >= b), key=lambda r: r[0])
            print(f"   gap {(b - a) / 1e3:7.1f} us at +{(a - t0) / 1e3:8.0f} us   after [{short(before[2])}]  before [{short(after[2])}]")

# per-kernel totals over the last n_calls segments (steady state: the first call's tuning launches are far behind)
from collections import defaultdict
tot = defaultdict(lambda: [0, 0])
nseg = 0
for seg in calls[-n_calls:]:
    nseg += 1
    for s_, e_, n_ in seg:
        k = short(n_)
        tot[k][0] += e_ - s_
        tot[k][1] += 1
print(f"--- kernel totals per segment (mean over {nseg} segments), us: name, launches, total, mean")
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"   {k:72s} {c / nseg:7.1f} {t / nseg / 1e3:9.1f} {t / c / 1e3:8.2f}")

# launch-by-launch listing of the solve of the last edited layer (from its prep_kr_kernel to the end of the call), with durations
# and the gap before each launch: argv[5] = "list"
if len(sys.argv) > 5 and sys.argv[5] == "list":
    seg = calls[-1]
    starts = [i for i, r in enumerate(seg) if "prep_kr_kernel" in r[2]]
    if starts:
        prev_end = seg[starts[-1] - 1][1] if starts[-1] > 0 else seg[starts[-1]][0]
        t0l = seg[starts[-1]][0]
        print("--- last layer's solve, launch by launch: +start us, duration us, gap before us, kernel")
        for s_, e_, n_ in seg[starts[-1]:]:
            print(f"   +{(s_ - t0l) / 1e3:8.1f} {(e_ - s_) / 1e3:8.1f} {(s_ - prev_end) / 1e3:7.1f}  {short(n_)}")
            prev_end = max(prev_end, e_)

# argv[5] = "head:<n>": the first n launches of the last full call (prefix forward), launch by launch
if len(sys.argv) > 5 and sys.argv[5].startswith("head:"):
    n_ = int(sys.argv[5].split(":")[1])
    seg = max(calls[-4:], key=len)
    t0l, prev_end = seg[0][0], seg[0][0]
    print("--- first launches of the call: +start us, duration us, gap before us, kernel")
    for s_, e_, k_ in seg[:n_]:
        print(f"   +{(s_ - t0l) / 1e3:8.1f} {(e_ - s_) / 1e3:8.1f} {(s_ - prev_end) / 1e3:7.1f}  {short(k_)}")
        prev_end = max(prev_end, e_)

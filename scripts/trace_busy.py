#!/usr/bin/env python3
"""Device busy time against device span per call, from a rocprofv3 --kernel-trace CSV of bench.py: calls are separated by the idle
gaps > 0.25 ms the host leaves between them; per call: kernels, span, union of the kernel intervals, what lies between kernels
(count and total of the gaps, histogram), and the same split by which kernel FOLLOWS the gap.  usage: trace_busy.py <dir or csv>"""
import csv, glob, os, statistics, sys
from collections import Counter, defaultdict
src = sys.argv[1]
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
calls, cur = [], [rows[0]]
end = rows[0][1]
for s, e, n in rows[1:]:
    if s - end > 250_000:
        calls.append(cur)
        cur = []
    cur.append((s, e, n))
    end = max(end, e)
calls.append(cur)
sizes = Counter(len(c) for c in calls if len(c) >= 50)      # (the process's set-up shows as many short runs)
main = max(k for k, v in sizes.items() if v >= 5)      # the longest run that repeats: a whole call the host never starved
print("calls", len(calls), "most common kernel count per call", main, "x", sizes[main])
spans, busys, gaps_tot, gap_by = [], [], [], defaultdict(list)
hist = Counter()
for c in calls:
    if len(c) != main:
        continue
    t0, t1 = c[0][0], max(e for _, e, _ in c)
    busy, gtot, end = 0, 0, c[0][0]
    per = defaultdict(int)
    for s, e, n in c:
        if s > end:
            g = s - end
            gtot += g
            hist[min(int(g / 1000), 20)] += 1
            per[n.split("(")[0][-60:]] += g
            busy += e - s
        else:
            busy += max(0, e - end)
        end = max(end, e)
    spans.append((t1 - t0) / 1e6); busys.append(busy / 1e6); gaps_tot.append(gtot / 1e6)
    for k, v in per.items():
        gap_by[k].append(v / 1e6)
print(f"span ms median {statistics.median(spans):.3f}  busy {statistics.median(busys):.3f}  between kernels {statistics.median(gaps_tot):.3f}")
print("gap histogram (us bucket: count over the counted calls):", dict(sorted(hist.items())))
for k, v in sorted(gap_by.items(), key=lambda kv: -statistics.median(kv[1]))[:6]:
    print(f"  before {k:62s} {statistics.median(v):.3f} ms per call")
dur, cnt = defaultdict(float), defaultdict(int)
n_counted = 0
for c in calls:
    if len(c) != main:
        continue
    n_counted += 1
    for s_, e_, n_ in c:
        key = n_.split("(")[0][-70:]
        dur[key] += e_ - s_
        cnt[key] += 1
print("kernel time per call:")
for k, v in sorted(dur.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  {k:72s} {v / n_counted / 1e6:7.3f} ms in {cnt[k] / n_counted:5.1f} launches")

#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints every kernel longer than MIN_MS and every idle gap longer than MIN_MS on the
device timeline (with the kernels on either side), to find stalls.  usage: trace_gaps.py <dir or csv> [min_ms]"""
import csv, sys, glob, os
src = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90], r.get("Queue_Id", "")))
rows.sort()
t0 = rows[0][0]
prev_end, prev_name = rows[0][1], rows[0][2]
print("kernels", len(rows), "span ms", (rows[-1][1] - t0) / 1e6)
for s, e, name, q in rows:
    if (e - s) / 1e6 > min_ms:
        print(f"LONG  {(s - t0) / 1e6:10.2f} ms  dur {(e - s) / 1e6:8.2f} ms  q{q} {name}")
    if (s - prev_end) / 1e6 > min_ms:
        print(f"GAP   {(prev_end - t0) / 1e6:10.2f} ms  gap {(s - prev_end) / 1e6:8.2f} ms  after [{prev_name}]  before q{q} [{name}]")
    if e > prev_end:
        prev_end, prev_name = e, name

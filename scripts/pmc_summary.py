#!/usr/bin/env python3
"""Per-kernel means of every counter found under <dir>/*/**/*counter_collection.csv (rocprofv3 --pmc passes) + durations from
the kernel traces; derived: MFMA pipe busy fraction, wait fractions, effective clock, L2 hit rate, memory-side bytes."""
import csv, glob, json, sys
from collections import defaultdict

root = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> per-dispatch values
for f in glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True):
    per_dispatch = defaultdict(float)
    for r in csv.DictReader(open(f)):
        per_dispatch[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, _, c), v in per_dispatch.items():
        vals[k][c].append(v)
dur = defaultdict(list)
for f in glob.glob(f"{root}/sq1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, cs in vals.items():
    if not any(s in k for s in ("emcid", "Cijk")):
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    rec = {"launches": max(len(v) for v in cs.values()), "counters": {c: round(x, 1) for c, x in sorted(m.items())}}
    if k in dur:
        rec["avg_ns"] = sum(dur[k]) / len(dur[k])
    g = m.get("GRBM_GUI_ACTIVE")
    if g:
        cyc = g / 8.0                                    # summed over the 8 XCDs
        rec["cycles_per_launch"] = cyc
        if k in dur:
            rec["effective_clock_ghz"] = cyc / rec["avg_ns"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            rec["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)       # 256 CUs x 4 SIMDs
    w = m.get("SQ_WAVE_CYCLES")
    if w:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in m:
                rec[c.lower() + "_over_wave_cycles"] = m[c] / w
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        rec["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "FETCH_SIZE" in m:
        rec["fetch_bytes_corrected"] = m["FETCH_SIZE"] * 1024 * 2      # KB; x2: gfx950 tallies 128-B requests at 64 B
    if "WRITE_SIZE" in m:
        rec["write_bytes"] = m["WRITE_SIZE"] * 1024
    if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
        rec["lds_conflict_over_active"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    out[k[:160]] = rec
print(json.dumps(out, indent=1))

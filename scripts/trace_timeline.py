"""Timeline of one bench step from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py.

Steps are delimited by scale_cov_kernel (the first launch of factor_cov, one per step).  Prints per-kernel-name totals,
the union of busy time, and optionally the ordered launch list with gaps — the ground truth for where a step's wall
clock goes when the solver runs as hipGraphs (the per-class HIP-event table in bench.py is taken in eager mode)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "hipblaslt_" + (m.group(1) if m else "gemm")
    name = name.replace("emcid::", "")
    name = re.sub(r"at::native::", "", name)
    m = re.match(r"(gemm_f64_kernel<[^>]*>)", name)
    if m:
        return m.group(1).replace("emcid::", "").replace(" ", "")
    return name.split("(")[0][:60]


def main():
    path = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -2     # which step (index into scale_cov occurrences)
    verbose = len(sys.argv) > 3
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"],
                         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r["Grid_Size_Y"])) *
                         max(1, int(r["Grid_Size_Z"]))))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[2].startswith("scale_cov_kernel")]
    print(f"{len(rows)} launches, {len(marks)} steps (scale_cov marks)")
    lo = marks[which]
    hi = marks[which + 1] if which + 1 < len(marks) and which != -1 else len(rows)
    # the forward of the step starts a few launches before/after the mark; take launches from the end of the previous
    # step's last kernel: extend back while the gap to the previous launch is < 50 us
    while lo > 0 and rows[lo][0] - rows[lo - 1][1] < 30_000 and lo - 1 not in marks and (lo - 1) > (marks[which - 1] if which != 0 else -1) + 50:
        if rows[lo - 1][2].startswith(("apply_u", "gemm_f64")):
            break
        lo -= 1
    seg = rows[lo:hi]
    t0 = seg[0][0]
    t1 = max(r[1] for r in seg)
    print(f"step window: {(t1 - t0) / 1e3:.1f} us, {len(seg)} launches")
    tot = defaultdict(lambda: [0, 0])
    for s, e, n, q, g in seg:
        tot[n][0] += e - s
        tot[n][1] += 1
    # union busy
    busy = 0
    cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, *_ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"busy union {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
    for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print(f"{d / 1e3:10.1f} us {c:5d}x  avg {d / c / 1e3:7.1f}  {n}")
    if verbose:
        prev_e = t0
        for s, e, n, q, g in seg:
            print(f"{(s - t0) / 1e3:10.1f} +{(e - s) / 1e3:7.1f} gap {(s - prev_e) / 1e3:7.1f} q{q} wg{g:6d} {n}")
            prev_e = max(prev_e, e)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condenses the four per-shape counter summaries of the split-fp16 projection kernel (scripts/round4_measure.sh:
profiles/<tag>_pmc_sp16_{qkv,out,fc1,fc2}.json, made by pmc_passes.sh + pmc_summary.py from scripts/pmc_linear_sp16.py) into the
file bench.py reads for `roofline.traffic`: profiles/<round>_pmc_linear_sp16.json (r06_a -> r06_pmc_linear_sp16.json).  usage: pmc_sp16_traffic.py <tag> [rows=6400]"""
import json, sys
from pathlib import Path

repo = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 6400
shapes = {"qkv": (768, 2304, False), "out": (768, 768, True), "fc1": (768, 3072, False), "fc2": (3072, 768, True)}   # K, N, residual
per, total = {}, 0.0
for name, (K, N, res) in shapes.items():
    d = json.load(open(repo / "profiles" / f"{tag}_pmc_sp16_{name}.json"))
    kern = [k for k in d if "linear_sp16" in k]
    assert len(kern) == 1, (name, list(d))
    e = d[kern[0]]
    traffic = e["fetch_bytes_corrected"] + e["write_bytes"]
    algo = 4 * (rows * K + N * K + rows * N + (rows * N if res else 0))
    per[name] = {"kernel": kern[0].split("(")[0], "traffic_bytes": traffic, "algorithmic_bytes": algo, "ratio": traffic / algo,
                 "avg_us": e["avg_ns"] / 1e3, "mfma_pipe_busy": e["mfma_pipe_busy_frac"], "l2_hit_rate": e["l2_hit_rate"],
                 "clock_ghz": e["effective_clock_ghz"]}
    total += traffic
out = {"traffic_bytes_per_launch": total / len(shapes), "per_shape": per,
       "note": f"memory-side (L2-miss) bytes per launch of the split-fp16 projection kernel, mean over the four SD-v1.4 projection shapes of "
               f"a {rows}-row trie (one launch of each per layer; profiles/{tag}_pmc_sp16_{{qkv,out,fc1,fc2}}.json): FETCH_SIZE x2 (gfx950 "
               f"tallies 128-B requests at 64 B) + WRITE_SIZE; Infinity-Cache hits are counted (the planes of the weights stay resident "
               f"there), so this is L2-miss traffic, not HBM traffic; algorithmic = 4 (rows K + N K + rows N [+ rows N residual]) bytes "
               f"(the split planes take 4 bytes per element like the fp32 matrices they stand for)"}
round_tag = tag.split("_")[0] if tag.startswith("r") else "r05"          # r06_a -> r06
(repo / "profiles" / f"{round_tag}_pmc_linear_sp16.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps({k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in per.items()}, indent=1))

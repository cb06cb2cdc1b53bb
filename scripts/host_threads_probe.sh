#!/bin/bash
# Headline step and host phases under different tokenizer / v*-reader thread counts (one bench run each, no extras).
# usage: scripts/host_threads_probe.sh   (writes gpurun_out/host_threads/*.json, prints one line per setting)
out=gpurun_out/host_threads; mkdir -p "$out"
for cfg in "4 4" "8 4" "4 2" "8 2" "2 4" "4 8" "4 4"; do
  set -- $cfg
  EMCID_TOK_THREADS=$1 EMCID_READ_THREADS=$2 python bench.py --steps 30 --warmup 5 --no-variants --no-stage0 --no-cpu-baseline --no-gemm-ab \
      > "$out/t$1_r$2.json" 2> "$out/t$1_r$2.err" || { echo "tok $1 read $2 FAILED"; tail -3 "$out/t$1_r$2.err"; exit 1; }
  python - "$out/t$1_r$2.json" "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
h = d["host_phases_ms_per_call"]
print(f"tok {sys.argv[2]} read {sys.argv[3]}: ms_per_step {d['ms_per_step']:.3f} median {d['ms_per_call_median']:.3f} | tokenize {h['tokenize+lookup']:.3f} trie {h['trie']:.3f} "
      f"prefix {h['prefix launches']:.3f} vstar-check {h['vstar check']:.3f} join {h['vstar join + h2d']:.3f}")
PY
done

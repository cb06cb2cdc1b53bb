#!/bin/bash
# A/B of one environment switch on the bench: scripts/ab_env.sh NAME VAR=VALUE_A VAR=VALUE_B [bench args...]
# Runs bench.py alternately (A, B, A, B) and prints value / ms_per_step / device_ms_per_step / chol classes of each run.
name=$1; a=$2; b=$3; shift 3
out=gpurun_out/$name; mkdir -p "$out"
for i in 1 2; do
  env "$a" python bench.py "$@" > "$out/a$i.json" 2> "$out/a$i.err"
  env "$b" python bench.py "$@" > "$out/b$i.json" 2> "$out/b$i.err"
done
python - "$out" "$a" "$b" <<'PY'
import json, sys
out, a, b = sys.argv[1:4]
for tag, label in (("a1", a), ("b1", b), ("a2", a), ("b2", b)):
    try:
        d = json.loads(open(f"{out}/{tag}.json").read().strip().splitlines()[-1])
        kc = {k: round(v.get("ms_per_step", 0), 3) for k, v in d.get("kernel_classes", {}).items()}
        print(label, "value", d["value"], "ms", d["ms_per_step"], "dev", d.get("device_ms_per_step"), kc)
    except Exception as e:
        print(label, "ERR", e, open(f"{out}/{tag}.err").read()[-400:])
PY

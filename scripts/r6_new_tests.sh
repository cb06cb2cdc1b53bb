# round 6: the new parity tests (realistic request shapes, parallel switch test), then a bench run with the variant records
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r06_b}
timeout -k 10 600 python -m pytest tests -x -q -m gpu --durations=15 -k "realistic or process_switches or stale or sdxl_real_dims or streamk or dgemm" > gpurun_out/${TAG}_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_tests.txt
tail -25 gpurun_out/${TAG}_tests.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-stage0 --no-cpu-full --no-gemm-ab > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"
tail -3 gpurun_out/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("ms", round(d["ms_per_step"],3), "device", round(d["device_ms_per_step"],3), "frac", round(d["roofline"]["frac"],3))
print(json.dumps(d["config"].get("assumes"), indent=1))
for k in ("n100","n125"):
    r=d.get(k,{})
    print(k, {kk: r.get(kk) for kk in ("ms_per_call_median","device_ms_per_step","device_ms_in_bracketed_kernels","launches_per_step","solve_ms_per_step","trie_rows_of_tokens","error")})
    for c,v in sorted((r.get("kernel_classes") or {}).items(), key=lambda kv:-kv[1]["ms_per_step"]): print("   ", c, v)
print("shape_checks", json.dumps(d.get("shape_checks"), indent=1))
PY
echo done

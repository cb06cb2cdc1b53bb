set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06_e
timeout -k 10 600 python -m pytest tests -x -q -m gpu --durations=12 -k "tree_attention or shadow or sdxl_real_dims or sdxl_apply_twice or stage0 or prepare_starts_over or templated or layer_stats or switches" > gpurun_out/${TAG}_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_tests.txt
tail -20 gpurun_out/${TAG}_tests.txt
timeout -k 10 120 python scripts/prelaunch_profile.py > gpurun_out/${TAG}_prelaunch.txt 2>&1; cat gpurun_out/${TAG}_prelaunch.txt | tail -10
timeout -k 10 300 python scripts/bench_stage0.py --captions 100000 > gpurun_out/${TAG}_stage0.json 2> gpurun_out/${TAG}_stage0.err; cat gpurun_out/${TAG}_stage0.json | cut -c1-600
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_s0prof -- python3 scripts/bench_stage0.py --captions 100000 > gpurun_out/${TAG}_stage0_under_rocprof.json 2> gpurun_out/${TAG}_s0prof.err && echo prof ok
f=$(find gpurun_out/${TAG}_s0prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_stage0_kernel_stats.csv; rm -rf gpurun_out/${TAG}_s0prof
head -9 gpurun_out/${TAG}_stage0_kernel_stats.csv | cut -c1-200
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-stage0 --no-cpu-baseline --no-gemm-ab --no-variants > gpurun_out/${TAG}_quick.json 2> gpurun_out/${TAG}_quick.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_quick.json").read().strip().splitlines()[-1])
print("ms", round(d["ms_per_step"],3), "median", round(d["ms_per_call_median"],3), "device", round(d["device_ms_per_step"],3), "linear", round(d["kernel_classes"]["linear"]["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), "solve", round(d["solve"]["ms_per_step"],3))
print(d["host_phases_ms_per_call"])
PY
echo done

set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06_d
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/${TAG}_suite.txt 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_suite.txt
tail -32 gpurun_out/${TAG}_suite.txt
timeout -k 10 120 python scripts/prelaunch_profile.py > gpurun_out/${TAG}_prelaunch.txt 2>&1; cat gpurun_out/${TAG}_prelaunch.txt | tail -14
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-stage0 --no-cpu-baseline --no-gemm-ab --no-variants > gpurun_out/${TAG}_quick.json 2> gpurun_out/${TAG}_quick.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_quick.json").read().strip().splitlines()[-1])
print("ms", round(d["ms_per_step"],3), "median", round(d["ms_per_call_median"],3), "device", round(d["device_ms_per_step"],3), "linear", round(d["kernel_classes"]["linear"]["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), "solve", round(d["solve"]["ms_per_step"],3))
print(d["host_phases_ms_per_call"])
PY
EMCID_FEW_ROWS_KSPLIT=0 timeout -k 10 120 python scripts/soak_n.py 100 100 2>&1 | tail -3
timeout -k 10 120 python scripts/soak_n.py 100 100 2>&1 | tail -3
echo done

# the whole GPU suite with per-test durations, smoke(), and a quick headline run (round 6)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r06_suite}
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=60 > gpurun_out/$TAG.txt 2>&1; echo "pytest rc $?" >> gpurun_out/$TAG.txt
tail -4 gpurun_out/$TAG.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-stage0 --no-cpu-baseline --no-gemm-ab --no-variants > gpurun_out/${TAG}_quick.json 2> gpurun_out/${TAG}_quick.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_quick.json").read().strip().splitlines()[-1])
print("ms", round(d["ms_per_step"],3), "median", round(d["ms_per_call_median"],3), "device", round(d["device_ms_per_step"],3), "linear", round(d["kernel_classes"]["linear"]["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), "solve", round(d["solve"]["ms_per_step"],3), round(d["solve"]["frac_f64_mfma_peak"],3))
PY
echo done

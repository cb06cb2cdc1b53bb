# round 5, fourth GPU call: GPU suite with the corrected guard, microbenchmark of all six LDS-DMA forms on more shapes, the A/B of
# the 16x16x32 default on the headline, the full bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r05_t4.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_t4.txt
tail -4 gpurun_out/r05_t4.txt
MB_DBG=0 timeout -k 10 420 python scripts/mb_linear_sp16_r5.py > gpurun_out/r05_mb_linear_sp16_forms.txt 2>&1; echo "mb rc $?"
out=gpurun_out/mfma16_ab; mkdir -p $out
for i in 1 2; do for v in 0 1; do
  EMCID_SP16_MFMA16=$v timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-stage0 --no-cpu-baseline --no-gemm-ab --no-variants > $out/v${v}_$i.json 2> $out/v${v}_$i.err
done; done
python - > gpurun_out/r05_mfma16_ab.txt <<PY
import json
for i in (1,2):
    for v in (0,1):
        d=json.loads(open(f"gpurun_out/mfma16_ab/v{v}_{i}.json").read().strip().splitlines()[-1])
        pc=sorted(d["ms_per_call"])
        print("MFMA16",v,"run",i,"mean",round(d["ms_per_step"],3),"median",round(d["ms_per_call_median"],3),"p10",round(pc[6],2),"p90",round(pc[53],2),"device",round(d["device_ms_per_step"],3),"linear",round(d["kernel_classes"]["linear"]["ms_per_step"],3),"frac",round(d["roofline"]["frac"],3))
PY
cat gpurun_out/r05_mfma16_ab.txt
echo done

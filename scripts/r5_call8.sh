# round 5, eighth GPU call: the projection with its epilogue vectors prefetched into LDS — kernel parity, stamps, quick bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "sp16" > gpurun_out/r05_t8.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_t8.txt
tail -3 gpurun_out/r05_t8.txt
MB_SHAPES=qkv,fc1,out,fc2 timeout -k 10 300 python scripts/mb_linear_sp16_r5.py > gpurun_out/r05_mb_linear_sp16_side.txt 2>&1
tail -40 gpurun_out/r05_mb_linear_sp16_side.txt
timeout -k 10 300 python bench.py --steps 60 --warmup 5 --no-stage0 --no-cpu-baseline --no-gemm-ab --no-variants > gpurun_out/r05_quick8.json 2> gpurun_out/r05_quick8.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_quick8.json').read().strip().splitlines()[-1])
print('ms/step',d['ms_per_step'],'value',d['value'],'roofline',d['roofline'])
print({k:v for k,v in d.get('breakdown',{}).items()} if 'breakdown' in d else list(d.keys()))
PY
echo done

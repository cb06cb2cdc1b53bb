# round 5: the 150 ms call of profiles/r05_b_bench.json (timed step 16 of 20, in both bench processes of that box) — which phase?
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for k in 1 2 3 4 5; do
flag=
timeout -k 10 300 python bench.py $flag --no-stage0 --no-cpu-baseline --no-gemm-ab --no-variants > gpurun_out/r05_outlier_$k.json 2> gpurun_out/r05_outlier_$k.err
python - $k <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r05_outlier_{sys.argv[1]}.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],3), 'median', round(d['ms_per_call_median'],3), [round(x,1) for x in d['ms_per_call']])
print(d['slowest_call']); print(d['gc_in_timed_region'])
PY
grep -i "warn\|pivot\|stale\|fallback" gpurun_out/r05_outlier_$k.err | head -5
done

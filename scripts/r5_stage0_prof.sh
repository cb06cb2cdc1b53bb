# round 5: the Stage-0 path as shipped under rocprofv3 (kernel table), and on its own (tokens/s)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tag=${1:-r05_b}
timeout -k 10 300 python scripts/bench_stage0.py --captions 100000 > gpurun_out/${tag}_stage0.json 2> gpurun_out/${tag}_stage0.err; cat gpurun_out/${tag}_stage0.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_s0prof -- python3 scripts/bench_stage0.py --captions 100000 > gpurun_out/${tag}_stage0_under_rocprof.json 2> gpurun_out/${tag}_s0prof.err && echo prof ok
f=$(find gpurun_out/${tag}_s0prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_stage0_kernel_stats.csv; rm -rf gpurun_out/${tag}_s0prof
head -14 gpurun_out/${tag}_stage0_kernel_stats.csv | cut -c1-200

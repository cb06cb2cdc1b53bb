set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06_c
timeout -k 10 600 python -m pytest tests -x -q -m gpu --durations=15 -k "realistic or process_switches or stale or n1500 or gram" > gpurun_out/${TAG}_tests.txt 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_tests.txt
tail -22 gpurun_out/${TAG}_tests.txt
timeout -k 10 200 python scripts/mb_tri_small.py > gpurun_out/${TAG}_mb_tri_small.txt 2>&1; echo "mb rc $?"
cat gpurun_out/${TAG}_mb_tri_small.txt | cut -c1-200
timeout -k 10 300 python scripts/bench_stage0.py --captions 100000 > gpurun_out/${TAG}_stage0.json 2> gpurun_out/${TAG}_stage0.err; cat gpurun_out/${TAG}_stage0.json | cut -c1-1500
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_s0prof -- python3 scripts/bench_stage0.py --captions 100000 > gpurun_out/${TAG}_stage0_under_rocprof.json 2> gpurun_out/${TAG}_s0prof.err && echo prof ok
f=$(find gpurun_out/${TAG}_s0prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_stage0_kernel_stats.csv; rm -rf gpurun_out/${TAG}_s0prof
head -9 gpurun_out/${TAG}_stage0_kernel_stats.csv | cut -c1-200
echo done

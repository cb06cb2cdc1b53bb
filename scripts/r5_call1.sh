# round 5, first GPU call: parity of the 16x16x32 forms, their microbenchmark with in-kernel clocks, leaf stamps, Stage-0 kernel table
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "linear_sp16_vs_torch and (256 or 320)" > gpurun_out/r05_t1.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_t1.txt
tail -3 gpurun_out/r05_t1.txt
timeout -k 10 500 python scripts/mb_linear_sp16_r5.py > gpurun_out/r05_mb_linear_sp16_dbg.txt 2>&1; echo "mb rc $?"
timeout -k 10 100 python scripts/leaf_stamps.py > gpurun_out/r05_leaf_stamps.txt 2>&1; echo "leaf rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_s0prof -- python3 scripts/bench_stage0.py --captions 100000 > gpurun_out/r05_stage0_under_rocprof.json 2> gpurun_out/r05_s0prof.err; echo "s0 rc $?"
f=$(find gpurun_out/r05_s0prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_stage0_kernel_stats.csv; rm -rf gpurun_out/r05_s0prof
echo done

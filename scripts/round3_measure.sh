# round-3 measurement set: bench line, rocprofv3 kernel stats of the same command, counter passes of the edit's kernels and of the
# forward GEMM alone.  usage (on the GPU box): bash scripts/round3_measure.sh <tag>
set -o pipefail
tag=${1:-r03_a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && echo bench ok
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --no-cpu-baseline --no-stage0 --no-variants --no-gemm-ab > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.err && echo prof ok
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv; rm -rf gpurun_out/${tag}_prof
bash scripts/pmc_passes.sh gpurun_out/${tag}_pmc_lin scripts/pmc_linear.py > gpurun_out/${tag}_pmc_linear_summary.json 2> gpurun_out/${tag}_pmc_lin.err && echo pmc linear ok
find gpurun_out/${tag}_pmc_lin -name "*.csv" -delete; find gpurun_out/${tag}_pmc_lin -type d -empty -delete
bash scripts/pmc_passes.sh gpurun_out/${tag}_pmc scripts/pmc_edit_steps.py 4 > gpurun_out/${tag}_pmc_edit_summary.json 2> gpurun_out/${tag}_pmc.err && echo pmc edit ok
find gpurun_out/${tag}_pmc -name "*.csv" -delete; find gpurun_out/${tag}_pmc -type d -empty -delete

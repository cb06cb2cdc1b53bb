set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r02n_bench.json 2> gpurun_out/r02n_bench.err && echo bench ok
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02n_prof -- python3 bench.py --no-cpu-baseline --no-stage0 > gpurun_out/r02n_bench_under_rocprof.json 2> gpurun_out/r02n_prof.err && echo prof ok
f=$(find gpurun_out/r02n_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r02n_bench_kernel_stats.csv; rm -rf gpurun_out/r02n_prof
bash scripts/pmc_passes.sh gpurun_out/r02n_pmc scripts/pmc_edit_steps.py 4 > gpurun_out/r02n_pmc_summary.json 2> gpurun_out/r02n_pmc.err && echo pmc ok
find gpurun_out/r02n_pmc -name "*.csv" -delete; find gpurun_out/r02n_pmc -type d -empty -delete

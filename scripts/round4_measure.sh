# round-4 measurement set: bench line, rocprofv3 kernel stats of the same command, counter passes of the edit's kernels and of the
# split-fp16 projection kernel alone, one shape per run (the four SD-v1.4 shapes of a 6 400-row trie).
# usage (on the GPU box): bash scripts/round4_measure.sh <tag> [quick]
set -o pipefail
tag=${1:-r04_a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && echo bench ok
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --no-cpu-baseline --no-stage0 --no-variants --no-gemm-ab > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.err && echo prof ok
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv; rm -rf gpurun_out/${tag}_prof
[ "$2" = quick ] && exit 0
for s in qkv out fc1 fc2; do
  bash scripts/pmc_passes.sh gpurun_out/${tag}_pmc_sp16_$s scripts/pmc_linear_sp16.py $s > gpurun_out/${tag}_pmc_sp16_$s.json 2> gpurun_out/${tag}_pmc_sp16_$s.err && echo pmc $s ok
  find gpurun_out/${tag}_pmc_sp16_$s -name "*.csv" -delete; find gpurun_out/${tag}_pmc_sp16_$s -type d -empty -delete
done
bash scripts/pmc_passes.sh gpurun_out/${tag}_pmc scripts/pmc_edit_steps.py 4 > gpurun_out/${tag}_pmc_edit_summary.json 2> gpurun_out/${tag}_pmc.err && echo pmc edit ok
find gpurun_out/${tag}_pmc -name "*.csv" -delete; find gpurun_out/${tag}_pmc -type d -empty -delete

# round-6 measurement set, in two GPU calls (a call is limited to 20 minutes):
#   bash scripts/round5_measure.sh <tag> bench   the bench line, then rocprofv3 kernel stats of the same command
#   bash scripts/round5_measure.sh <tag> pmc     counter passes: the projection kernel alone, one SD-v1.4 shape of a 6 400-row trie per
#                                                run (-> profiles/r06_pmc_linear_sp16.json, which bench.py reads for roofline.traffic),
#                                                and the edit's kernels over four calls
set -o pipefail
tag=${1:-r06_a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$2" = bench ]; then
  timeout -k 10 700 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && echo bench ok
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python3 bench.py --no-cpu-baseline --no-stage0 --no-variants --no-gemm-ab > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_prof.err && echo prof ok
  f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv; rm -rf gpurun_out/${tag}_prof
  timeout -k 10 200 python scripts/soak.py > gpurun_out/${tag}_soak.txt 2>&1 && echo soak ok
  exit 0
fi
for s in qkv out fc1 fc2; do
  bash scripts/pmc_passes.sh gpurun_out/${tag}_pmc_sp16_$s scripts/pmc_linear_sp16.py $s > gpurun_out/${tag}_pmc_sp16_$s.json 2> gpurun_out/${tag}_pmc_sp16_$s.err && echo pmc $s ok
  find gpurun_out/${tag}_pmc_sp16_$s -name "*.csv" -delete; find gpurun_out/${tag}_pmc_sp16_$s -type d -empty -delete
done
bash scripts/pmc_passes.sh gpurun_out/${tag}_pmc scripts/pmc_edit_steps.py 4 > gpurun_out/${tag}_pmc_edit_summary.json 2> gpurun_out/${tag}_pmc.err && echo pmc edit ok
find gpurun_out/${tag}_pmc -name "*.csv" -delete; find gpurun_out/${tag}_pmc -type d -empty -delete

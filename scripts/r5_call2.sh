# round 5, second GPU call: the LDS-transposed epilogue of the 16x16x32 forms (parity + microbenchmark with per-workgroup
# prologue / loop / epilogue stamps), the stale-cache guard, the sequential SDXL apply, the non-contiguous layer list
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -x -q -k "(linear_sp16_vs_torch and (256 or 320)) or rewritten_through_data or apply_twice or non_contiguous or toy_sd_apply" > gpurun_out/r05_t2.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_t2.txt
tail -5 gpurun_out/r05_t2.txt
MB_SHAPES=qkv,fc1,out,fc2,qkv-36k timeout -k 10 400 python scripts/mb_linear_sp16_r5.py > gpurun_out/r05_mb_linear_sp16_epi.txt 2>&1; echo "mb rc $?"
echo done

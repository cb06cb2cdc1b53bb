#!/bin/bash
# rocprofv3 counter passes over one target script (counters in their own runs, --kernel-trace only; see MI355X_MICROARCH.md
# "rocprofv3 PMC slots").  usage: scripts/pmc_passes.sh <out_dir> <script.py> [args...]   -> <out_dir>/summary.json
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
pass() {  # name counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -- python3 "$TARGET" "${ARGS[@]}" > "$out/$name.log" 2>&1
}
TARGET=$1; shift; ARGS=("$@")
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 "$(dirname "$0")/pmc_summary.py" "$out" > "$out/summary.json"
cat "$out/summary.json"

#!/bin/bash
# One rocprofv3 SQ-counter pass over a conv_lab binary and case; prints per-kernel means of kernels matching PATTERN.
# usage (from the repository root on the GPU box): tools/micro/pmc_lab.sh BINARY CASE PATTERN
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp || exit 1
rm -rf /tmp/pmc_lab
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
    --output-format csv -d /tmp/pmc_lab -o p -- "$ROOT/tools/micro/$1" "$2" > /tmp/pmc_lab.log 2>&1 || { tail -5 /tmp/pmc_lab.log; exit 1; }
python3 "$ROOT/tools/pmc_kernel.py" "$3" "$(find /tmp/pmc_lab -name '*counter_collection.csv' | head -1)"
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC \
    --output-format csv -d /tmp/pmc_lab2 -o p -- "$ROOT/tools/micro/$1" "$2" > /tmp/pmc_lab2.log 2>&1 || { tail -5 /tmp/pmc_lab2.log; exit 1; }
python3 "$ROOT/tools/pmc_kernel.py" "$3" "$(find /tmp/pmc_lab2 -name '*counter_collection.csv' | head -1)"

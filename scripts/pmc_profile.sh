#!/bin/bash
# rocprofv3 counter passes for the tracing kernel (run on the GPU box via gpurun).
# usage: scripts/pmc_profile.sh <case> <outdir-under-gpurun_out> [extra prof_driver args]
set -u
CASE=${1:-step}; OUT=${2:-pmc_$CASE}; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out/$OUT
i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_REQ_sum TCC_READ_sum" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_LEVEL_VMEM SQ_INSTS_FLAT" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64" ; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $ROOT/gpurun_out/$OUT/pass$i -- python3 $ROOT/scripts/prof_driver.py --case $CASE "$@" > $ROOT/gpurun_out/$OUT/pass$i.log 2>&1 || echo "pass $i failed (see log)"
done
python3 $ROOT/scripts/pmc_summary.py $ROOT/gpurun_out/$OUT > $ROOT/gpurun_out/$OUT/summary.txt 2>&1
cat $ROOT/gpurun_out/$OUT/summary.txt

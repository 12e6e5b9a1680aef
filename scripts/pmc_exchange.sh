#!/bin/bash
# rocprofv3 SQ counter passes of the exchange kernel beside the lane kernel (128x128x64, 1e7 photons per launch).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for MODE in 0 1; do
  export MCBRAT_EXCHANGE=$MODE MCBRAT_FLUSH_LANES=24
  OUT=$ROOT/gpurun_out/pmc_x$MODE; mkdir -p $OUT
  i=0
  for SET in \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
    "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" \
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_LEVEL_VMEM SQ_INSTS_FLAT" ; do
    i=$((i+1))
    rocprofv3 --pmc $SET --output-format csv -d $OUT/pass$i -- python3 $ROOT/scripts/prof_driver.py --case landsat --thr 32 > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
  done
  python3 $ROOT/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
  echo "== exchange=$MODE"; grep -E "SQ_INSTS_VALU\"|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_THREAD_CYCLES_VALU|SQ_ACTIVE_INST_VALU|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_ANY|SQ_WAIT_ANY|SQ_WAVES|SQ_LDS_BANK|SQ_ACTIVE_INST_LDS|SQ_WAIT_INST_LDS" $OUT/summary.txt
done

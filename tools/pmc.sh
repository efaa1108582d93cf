#!/bin/bash
# usage: tools_pmc.sh <outdir> [bench args...]   -- collects SQ / TCC counter groups in separate passes
OUT=$1; shift
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/$OUT; cd /tmp
for i in 1 2 3 4; do
  case $i in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES";;
    2) C="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS";;
    3) C="FETCH_SIZE GRBM_GUI_ACTIVE";;
    4) C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum";;
  esac
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-entry "$@" > $R/$OUT/run$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $OUT

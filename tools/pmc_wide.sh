#!/bin/bash
# SQ counter passes over one bench workload, summary lines of the wide join kernel only: tools/pmc_wide.sh <outdir> [bench args...]
OUT=$1; shift
export TMPDIR=/tmp; R=$PWD; mkdir -p $R/$OUT; cd /tmp
for i in 1 2; do
  case $i in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES";;
    2) C="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS";;
  esac
  timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-entry "$@" > $R/$OUT/run$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $OUT | grep -E "fj_count_join" 

#!/bin/bash
# rocprofv3 passes for the non-headline configurations (one MI355X):
# kernel stats + SQ counters of `bench.py --config CFG --no-variants`.
# usage (repo root, through gpurun): bash tools/profile_variants.sh TAG CFG...
set -e
TAG=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
cd $R
for CFG in "$@"; do
  OUT=$R/gpurun_out/prof_${TAG}_$CFG
  mkdir -p $OUT
  B="bench.py --config $CFG --no-variants --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B --steps 10 --warmup 2 > $OUT/stats.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sqa -- python3 $B --steps 3 --warmup 2 > $OUT/sqa.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/sqb -- python3 $B --steps 3 --warmup 2 > $OUT/sqb.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $B --steps 3 --warmup 2 > $OUT/fetch.log 2>&1
  echo "$CFG done"
done

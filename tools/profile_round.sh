#!/bin/bash
# rocprofv3 passes of the default bench command on the GPU box (one MI355X):
# kernel stats, then FETCH_SIZE / WRITE_SIZE / SQ counters in separate --pmc runs.
# usage (from the repo root, through gpurun): bash tools/profile_round.sh TAG
set -e
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
BENCH="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-variants > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-variants > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sqa -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-variants > $OUT/sqa.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/sqb -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-variants > $OUT/sqb.log 2>&1
echo done

#!/bin/bash
# PMC passes over one eager DIS-MF step (run on the GPU box through gpurun).  Each counter group is its own
# rocprofv3 run with --kernel-trace only (no other trace domains), as the pool requires.
#   usage: bash scripts/prof_pmc.sh <outdir under gpurun_out/>
set -u
OUT=/root/repo/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="/root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph"
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq -o sq \
  --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
  -- python3 $ARGS > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch -o fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE \
  -- python3 $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/write -o write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum \
  -- python3 $ARGS > $OUT/write.json 2> $OUT/write.err
ls -la $OUT/*/ | tail -20

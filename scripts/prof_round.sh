#!/bin/bash
# One profiling round on the GPU box: bench line, rocprofv3 kernel-trace stats of the same command, and the PMC
# passes (separate runs, --kernel-trace only) used for roofline.traffic.   usage: bash scripts/prof_round.sh <tag>
set -u
TAG=$1
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --arch single_frame --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $OUT/bench_sf.json 2> $OUT/bench_sf.err
python bench.py --arch single_frame --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $OUT/bench_sf_bf16.json 2> $OUT/bench_sf_bf16.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $OUT/bench_prof.json 2> $OUT/bench_prof.err
ARGS="/root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-extra-legs"
rocprofv3 --kernel-trace --output-format csv -d $OUT/sq -o sq \
  --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
  -- python3 $ARGS > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch -o fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE \
  -- python3 $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/write -o write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum \
  -- python3 $ARGS > $OUT/write.json 2> $OUT/write.err
# keep the merged-back volume small: the per-dispatch kernel trace is summarised by the stats csv
rm -f $OUT/trace/*kernel_trace.csv $OUT/sq/*kernel_trace.csv $OUT/fetch/*kernel_trace.csv $OUT/write/*kernel_trace.csv
ls $OUT $OUT/trace

#!/bin/bash
# DIS-SF (BASELINE config 2) profiling round on the GPU box, both storage modes: rocprofv3 kernel-trace stats of the bench command
# and the FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) over eager steps.  scripts/make_sf_profile_summary.py
# turns the output into profiles/<tag>_sf*.   usage: bash scripts/prof_sf_round.sh <tag>
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp
for MODE in f32 bf16; do
  OUT=/root/repo/gpurun_out/$TAG/sf_$MODE
  mkdir -p $OUT
  A="/root/repo/bench.py --arch single_frame --dtype $MODE --no-cpu-baseline --no-extra-legs"
  python3 $A --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $A --steps 10 --warmup 3 > $OUT/bench_prof.json 2> $OUT/bench_prof.err
  rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch -o fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- python3 $A --steps 2 --warmup 1 --no-graph > $OUT/fetch.json 2> $OUT/fetch.err
  rocprofv3 --kernel-trace --output-format csv -d $OUT/write -o write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 $A --steps 2 --warmup 1 --no-graph > $OUT/write.json 2> $OUT/write.err
  rm -f $OUT/trace/*kernel_trace.csv $OUT/fetch/*kernel_trace.csv $OUT/write/*kernel_trace.csv
done
ls /root/repo/gpurun_out/$TAG/sf_f32 /root/repo/gpurun_out/$TAG/sf_bf16

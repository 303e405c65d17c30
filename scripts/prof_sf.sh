#!/bin/bash
# rocprofv3 kernel-trace stats of the DIS-SF bench (scripts/prof_round.sh profiles DIS-MF).  usage: bash scripts/prof_sf.sh <tag>
set -u
TAG=$1
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sf -o run -- python3 /root/repo/bench.py --arch single_frame --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_sf_prof.json 2> $OUT/bench_sf_prof.err
rm -f $OUT/trace_sf/*kernel_trace.csv
ls $OUT/trace_sf

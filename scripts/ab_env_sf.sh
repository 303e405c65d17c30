#!/bin/bash
# usage: ab_env_sf.sh <tag> "<envA>" "<envB>" rounds    (DIS-SF fp32 and bf16-storage legs, alternating runs of two environments)
TAG=$1; A=$2; B=$3; R=${4:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { for mode in f32 bf16; do
  extra="--dtype f32"; [ $mode = bf16 ] && extra="--dtype bf16"
  env $2 python bench.py --arch single_frame $extra --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/$1_$3_$mode.json 2> $OUT/$1_$3_$mode.err
  python - $OUT/$1_$3_$mode.json "$1 [$2] $mode" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:36s} {d['value']:8.1f} frames/s  {d['ms_per_step']:.3f} ms/step")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done; }
for r in $(seq 1 $R); do run A "$A" $r; run B "$B" $r; done

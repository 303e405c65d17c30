#!/bin/bash
# usage: ab_env.sh <tag> "<envA>" "<envB>" rounds
TAG=$1; A=$2; B=$3; R=${4:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { env $2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/$1_$3.json 2> $OUT/$1_$3.err
  python - $OUT/$1_$3.json "$1 [$2]" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print(f"{sys.argv[2]:28s} {d['value']:8.1f} frames/s  {d['ms_per_step']:.3f} ms/step  dominant {r.get('avg_launch_ms', 0)*1e3:.1f} us x {r.get('launches_per_step')}")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for r in $(seq 1 $R); do run A "$A" $r; run B "$B" $r; done

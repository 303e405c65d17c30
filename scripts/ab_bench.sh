#!/bin/bash
# Same-box A/B of two builds of libdis_hip.so on the headline bench line (alternating runs; cdna_hip_programming.md rule 24).
#   bash scripts/ab_bench.sh <tag> <libA.so|default> <libB.so|default> [rounds] [extra bench args]
TAG=$1; A=$2; B=$3; R=${4:-2}; shift 4 2>/dev/null
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { # name lib round
  if [ "$2" = default ]; then env -u DIS_HIP_LIB python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs "${EXTRA[@]}" > $OUT/$1_$3.json 2> $OUT/$1_$3.err
  else DIS_HIP_LIB=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs "${EXTRA[@]}" > $OUT/$1_$3.json 2> $OUT/$1_$3.err; fi
  python - $OUT/$1_$3.json $1 <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print(f"{sys.argv[2]:8s} {d['value']:8.1f} frames/s  {d['ms_per_step']:.3f} ms/step  dominant {r.get('avg_launch_ms', 0)*1e3:.1f} us/launch")
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
EXTRA=("$@")
for r in $(seq 1 $R); do run A $A $r; run B $B $r; done

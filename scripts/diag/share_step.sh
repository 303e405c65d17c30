#!/bin/bash
# Two independent single-rank eager DIS-MF benches sharing one GPU (time-slicing): prints the final loss terms of both.
# A healthy run ends near [0.0577, 0.0188, 0.0072, ...]; the corrupted one near [0.216, 0.0056, 0.0025, ...].   usage: share_step.sh [ENV=VAL ...]
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
(env "$@" timeout 300 python bench.py --no-graph --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/sh1.log 2>&1 &)
env "$@" timeout 300 python bench.py --no-graph --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/sh2.log 2>&1
sleep 4
for f in sh1 sh2; do tail -1 gpurun_out/$f.log | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print([round(v, 4) for v in d['loss_terms']][:4])
except Exception as e:
    print('ERR', l[-200:])
"; done

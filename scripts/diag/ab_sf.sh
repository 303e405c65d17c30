#!/bin/bash
# same-box A/B of an environment switch on the DIS-SF benches: scripts/diag/ab_sf.sh VAR  (VAR=0 is the old form)
cd /root/repo
V=$1
for r in 1 2 3; do
  for dt in bf16 f32; do
    a=$(python bench.py --arch single_frame --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    b=$(env $V=0 python bench.py --arch single_frame --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    echo "$dt new $a old($V=0) $b"
  done
done

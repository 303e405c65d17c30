#!/bin/bash
# same-box A/B of two builds of the fused backward launch (scripts/diag/fb_variant.sh): the probe alternately with each library,
# separate processes; prints the fused column of every run (the two-launch column is the box's control).
#   scripts/diag/fb_ab.sh head new [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do
  for v in $A $B; do
    echo "== $v"
    DIS_HIP_LIB=$PWD/build_variants/libdis_hip_$v.so timeout 300 python scripts/diag/bwd_fused_probe.py 6 2>&1 | tail -7 | awk '{n=NF; printf "%-28s two %7.1f  fused %7.1f\n", $1, $(n-6), $(n-3)}'
  done
done

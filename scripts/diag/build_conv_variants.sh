#!/bin/bash
# Diagnostic libraries for scripts/diag/conv_bound.py: conv2d.hip + conv_f16x2.hip with one knock-out flag set each
# (see the F2_KO_* / F2_CLK block in csrc/conv_f16x2.hip).  Built HERE (hipcc cross-compiles), they travel to the GPU box
# inside build_variants/ (git-ignored).  Never loaded by the package.
#   bash scripts/diag/build_conv_variants.sh [variant ...]
set -eu
cd "$(dirname "$0")/../../depthinspace_amd/csrc"
OUT=../../build_variants
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops"
[ -f conv2d.o ] || /opt/rocm/bin/hipcc $FLAGS -c conv2d.hip -o conv2d.o 2>/dev/null
declare -A V
V[base]="-DF2_CLK -DF2_PF2=0"
V[ko_store]="-DF2_CLK -DF2_PF2=0 -DF2_KO_STORE"
V[ko_load]="-DF2_CLK -DF2_PF2=0 -DF2_KO_LOAD"
V[ko_mem]="-DF2_CLK -DF2_PF2=0 -DF2_KO_LOAD -DF2_KO_STORE"
V[ko_mfma]="-DF2_CLK -DF2_PF2=0 -DF2_KO_MFMA"
V[ko_split]="-DF2_CLK -DF2_PF2=0 -DF2_KO_SPLIT"
V[ko_epi]="-DF2_CLK -DF2_PF2=0 -DF2_KO_EPI"
V[ko_compute]="-DF2_CLK -DF2_PF2=0 -DF2_KO_MFMA -DF2_KO_SPLIT -DF2_KO_EPI"
V[ko_mem_mfma]="-DF2_CLK -DF2_PF2=0 -DF2_KO_LOAD -DF2_KO_STORE -DF2_KO_MFMA"
V[wait1]="-DF2_CLK -DF2_PF2=0 -DF2_WAIT_MODE=1"
V[wait2]="-DF2_CLK -DF2_PF2=0 -DF2_WAIT_MODE=2"
V[early]="-DF2_CLK -DF2_PF2=0 -DF2_LOAD_SCHED=1"
V[early_wait1]="-DF2_CLK -DF2_PF2=0 -DF2_LOAD_SCHED=1 -DF2_WAIT_MODE=1"
V[pf2]="-DF2_CLK"
V[pf2_early]="-DF2_CLK -DF2_LOAD_SCHED=1"
V[pf2_ko_mfma]="-DF2_CLK -DF2_KO_MFMA"
V[w_ko_mfma]="-DF2_CLK -DF2W_KO_MFMA"
V[w_ko_load]="-DF2_CLK -DF2W_KO_LOAD"
V[w_ko_both]="-DF2_CLK -DF2W_KO_LOAD -DF2W_KO_MFMA"
V[w_ko_split]="-DF2_CLK -DF2W_KO_SPLIT"
V[w_wpc1]="-DF2_CLK -DF2W_WPC=1"
V[snake]="-DF2_CLK -DF2_SNAKE"
V[early_wait2]="-DF2_CLK -DF2_PF2=0 -DF2_LOAD_SCHED=1 -DF2_WAIT_MODE=2"
NAMES="${@:-${!V[@]}}"
for v in $NAMES; do
  (
    /opt/rocm/bin/hipcc $FLAGS ${V[$v]} -c conv_f16x2.hip -o $OUT/f2_$v.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libf2_$v.so conv2d.o adam.o $OUT/f2_$v.o
    : keep $OUT/f2_$v.o for relinking
    echo "built $v"
  ) &
  # at most 4 compilers at once (8 CPUs, ~3 GB each)
  while [ "$(jobs -r | wc -l)" -ge 4 ]; do sleep 1; done
done
wait
ls -la $OUT

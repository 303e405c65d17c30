#!/bin/bash
# build_variants/libdis_hip_<name>.so = the library with conv_bwd_fused.hip compiled with extra flags (A/B builds: DIS_HIP_LIB=...)
# FB_SRC=<absolute path>: another revision of the file (e.g. git show HEAD:depthinspace_amd/csrc/conv_bwd_fused.hip > build_variants/head.hip)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../../depthinspace_amd/csrc"
mkdir -p ../../build_variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -I. -c ${FB_SRC:-conv_bwd_fused.hip} -o ../../build_variants/fb_$NAME.o 2> /dev/null
OBJS=$(ls *.o | grep -v hasan | grep -v conv_bwd_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libdis_hip_$NAME.so $OBJS ../../build_variants/fb_$NAME.o
echo built build_variants/libdis_hip_$NAME.so

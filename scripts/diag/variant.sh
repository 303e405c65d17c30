#!/bin/bash
# build_variants/libdis_hip_<name>.so = the library with ONE .hip file compiled with extra flags (A/B builds: DIS_HIP_LIB=<path>).
# usage: scripts/diag/variant.sh <name> <file.hip> [-DFLAG ...]      (needs the regular build's objects: make -C depthinspace_amd/csrc)
set -e
NAME=$1; FILE=$2; shift 2
cd "$(dirname "$0")/../../depthinspace_amd/csrc"
mkdir -p ../../build_variants
BASE=$(basename $FILE .hip)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -I. -c $FILE -o ../../build_variants/${BASE}_$NAME.o 2> /dev/null
OBJS=$(ls *.o | grep -v hasan | grep -v "^$BASE.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libdis_hip_$NAME.so $OBJS ../../build_variants/${BASE}_$NAME.o
echo built build_variants/libdis_hip_$NAME.so

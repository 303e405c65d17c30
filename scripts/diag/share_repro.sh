#!/bin/bash
# Builds and runs scripts/diag/share_repro.hip: first the victim alone, then the victim next to two load processes.
# Every process is a fresh child of this shell (nothing exec'ed from a process that has touched the GPU).
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 scripts/diag/share_repro.hip -o gpurun_out/share_repro -ldl || exit 1
# DIS_DUMP=1 bash scripts/diag/share_repro.sh: per-thread / per-block sums from a diagnostic build of the two conv files
if [ -n "${DIS_DUMP:-}" ]; then
  if [ ! -f depthinspace_amd/libdis_hip_dump.so ]; then
    (cd depthinspace_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DDIS_STATS_DUMP ${DIS_DUMP_FLAGS:-} \
       -shared conv2d.hip conv_f16x2.hip -o ../libdis_hip_dump.so) || exit 1
  fi
  export DIS_LIB=depthinspace_amd/libdis_hip_dump.so
fi
ITERS=${1:-6000}
echo "=== victim alone"
timeout 300 ./gpurun_out/share_repro victim $ITERS
echo "=== victim next to two synthetic load processes (MFMA + memory loops: few registers, no LDS - they co-reside with the victim)"
timeout 120 ./gpurun_out/share_repro load 45 > gpurun_out/share_load1.log 2>&1 &
L1=$!
timeout 120 ./gpurun_out/share_repro load 45 > gpurun_out/share_load2.log 2>&1 &
L2=$!
sleep 2
timeout 300 ./gpurun_out/share_repro victim $ITERS
kill $L1 $L2 2>/dev/null
wait $L1 $L2 2>/dev/null
cat gpurun_out/share_load1.log gpurun_out/share_load2.log

# the load under which the anomaly was seen in round 2: other processes running the DIS-MF training step itself (127 KB of LDS
# and 512 threads per workgroup: the queues of the processes cannot co-reside on a CU and are time-sliced)
echo "=== victim next to two processes running the DIS-MF training step (python bench.py)"
timeout 200 python bench.py --steps 1500 --warmup 2 --no-cpu-baseline --no-eager-leg > gpurun_out/share_bench1.log 2>&1 &
B1=$!
timeout 200 python bench.py --steps 1500 --warmup 2 --no-cpu-baseline --no-eager-leg > gpurun_out/share_bench2.log 2>&1 &
B2=$!
sleep 45   # (import torch, build the step, capture the graph)
for rep in 1 2 3 4; do
  echo "--- victim process $rep"
  timeout 300 ./gpurun_out/share_repro victim $ITERS
done
kill $B1 $B2 2>/dev/null
wait $B1 $B2 2>/dev/null
tail -c 300 gpurun_out/share_bench1.log

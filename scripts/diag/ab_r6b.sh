#!/bin/bash
# same-box A/B of the round-6 launch-count changes (weight-gradient slices in place + the larger zero arena) and of the Conv3D block cap
cd /root/repo
O=gpurun_out/ab_r6b; mkdir -p $O
python -m pytest tests/test_bwd_fused_gpu.py tests/test_step_gpu.py tests/test_token_contracts_gpu.py tests/test_net_ops_gpu.py -x -q -m gpu > $O/tests.log 2>&1
tail -3 $O/tests.log
for r in 1 2; do
  python scripts/diag/conv3d_bwd_modes.py 2>&1 | grep " det:" > $O/c3_base_$r.txt
  DIS_HIP_LIB=/root/repo/build_variants/libdis_hip_cap1024.so python scripts/diag/conv3d_bwd_modes.py 2>&1 | grep " det:" > $O/c3_cap1024_$r.txt
done
grep . $O/c3_*.txt
for r in 1 2 3; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/bench_new_$r.json 2> /dev/null
  DIS_GW_INPLACE=0 DIS_ARENA_DOUBLES=8388608 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/bench_old_$r.json 2> /dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_r6b/bench_*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value'], 1), round(d['ms_per_step'], 3))
PY

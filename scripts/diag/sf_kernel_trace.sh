#!/bin/bash
# Per-dispatch kernel trace of one eager DIS-SF fp32 step (scripts/sf_profile.py), reduced to the conv_gen.hip kernels:
# name, grid, duration.   usage (GPU box): bash scripts/diag/sf_kernel_trace.sh <outname>
OUT=/root/repo/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o run -- python3 /root/repo/scripts/sf_profile.py 8 $2 > $OUT/calls.txt 2> $OUT/err.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last step = the last third of the dispatches (2 warm-up steps + the recorded one)
n = len(rows) // 3
with open(out + '/dispatches.txt', 'w') as g:
    for r in rows[-n:]:
        name = r['Kernel_Name']
        if not any(k in name for k in ('convh2', 'convg', 'convb', 'conv_f16x2', 'conv_bf16x3', 'conv_wgrad')):
            continue
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        g.write(f"{us:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4} lds {r.get('LDS_Block_Size','?'):>7} vgpr {r.get('VGPR_Count','?'):>4}+{r.get('Accum_VGPR_Count','?'):<4}  {name[:150]}\n")
PY
rm -rf $OUT/trace

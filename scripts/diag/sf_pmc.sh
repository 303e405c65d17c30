#!/bin/bash
# HBM bytes per dispatch of the conv_gen.hip kernels in one eager DIS-SF step (separate --pmc passes, --kernel-trace only).
# usage (GPU box): bash scripts/diag/sf_pmc.sh <outname> [bf16]
OUT=/root/repo/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch -o fetch --pmc FETCH_SIZE -- python3 /root/repo/scripts/sf_profile.py 8 $2 > $OUT/fetch.txt 2> $OUT/fetch.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/write -o write --pmc WRITE_SIZE -- python3 /root/repo/scripts/sf_profile.py 8 $2 > $OUT/write.txt 2> $OUT/write.err
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
def load(kind, counter):
    f = glob.glob(f'{out}/{kind}/**/*counter_collection.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    n = len(rows) // 3
    return rows[-n:]
fe, wr = load('fetch', 'FETCH_SIZE'), load('write', 'WRITE_SIZE')
agg = collections.OrderedDict()
for rows, key in ((fe, 'r'), (wr, 'w')):
    for r in rows:
        k = (r['Kernel_Name'][:110], r['Grid_Size'])
        a = agg.setdefault(k, {'r': 0.0, 'w': 0.0, 'n': 0})
        a[key] += float(r['Counter_Value'])
        if key == 'r': a['n'] += 1
with open(out + '/hbm.txt', 'w') as g:
    for (name, grid), a in sorted(agg.items(), key=lambda kv: -(kv[1]['r'] * 2 + kv[1]['w'])):
        if not any(t in name for t in ('convh2', 'convg', 'convb', 'conv_f16x2', 'conv_wgrad', 'absmax')):
            continue
        g.write(f"{a['n']:3d} x  read {a['r'] * 2 / 1e3 / max(a['n'], 1):9.1f} MB (2 x FETCH_SIZE KB)  write {a['w'] / 1e3 / max(a['n'], 1):9.1f} MB  grid {grid:>8}  {name}\n")
PY
rm -rf $OUT/fetch $OUT/write

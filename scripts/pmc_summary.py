#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv written with --output-format csv).

    python scripts/pmc_summary.py gpurun_out/pmc1/sq/sq_counter_collection.csv [more.csv ...] > profiles/x.csv

One row per kernel: calls, average duration (us) and the per-dispatch average of every counter found.
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads (MI355X_MICROARCH.md, HBM section), so the corrected HBM read volume is 2 * FETCH_SIZE.
"""
import csv
import sys
from collections import defaultdict


def main(paths):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(float)
    seen = set()
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r['Kernel_Name']
                c = r['Counter_Name']
                agg[k][c] += float(r['Counter_Value'])
                calls[k][c] += 1
                key = (path, r['Dispatch_Id'])
                if key not in seen:
                    seen.add(key)
                    dur[(k, path)] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    counters = sorted({c for k in agg for c in agg[k]})
    rows = []
    for k in agg:
        n = max(calls[k].values())
        d = max(v for (kk, p), v in dur.items() if kk == k) / n / 1e3
        rows.append((d * n, k, n, d, [agg[k][c] / calls[k][c] if calls[k][c] else float('nan') for c in counters]))
    rows.sort(reverse=True)
    print(','.join(['"Kernel"', 'Calls', 'AvgUs'] + counters))
    for _, k, n, d, vals in rows:
        print(','.join([f'"{k}"', str(n), f'{d:.1f}'] + [f'{v:.1f}' for v in vals]))


if __name__ == '__main__':
    main(sys.argv[1:])

#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration, share) from a rocprofv3 `*_results.db` (rocpd sqlite),
written as CSV.  Same columns as `rocprofv3 --stats --output-format csv` kernel_stats.

    python scripts/rocpd_stats.py gpurun_out/xyz/run_results.db > profiles/xyz_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, c, s, a, mn, mx in rows:
        print(f'"{n}",{c},{s},{a:.1f},{100.0 * s / tot:.4f},{mn},{mx}')


if __name__ == '__main__':
    main(sys.argv[1])

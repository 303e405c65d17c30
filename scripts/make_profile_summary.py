#!/usr/bin/env python
"""Turn one profiling round (gpurun_out/<tag>/, written by scripts/prof_round.sh on the GPU box) into the tracked
summaries under profiles/:

  <tag>_bench.json / <tag>_bench_sf.json / <tag>_bench_sf_bf16.json   the bench lines (DIS-MF headline, DIS-SF fp32, DIS-SF bf16 storage)
  <tag>_bench_under_rocprof.json           the bench line printed under the profiler
  <tag>_kernel_stats.csv                   rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 3`
  <tag>_pmc_sq.csv                         per-kernel averages of the SQ counters (one eager step)
  <tag>_pmc_mem.csv                        per-kernel FETCH_SIZE / WRITE_SIZE / L2 hit counters (separate passes)
  <tag>_kernels.md                         ranking: time per step, measured HBM bytes and GB/s, MFMA/wait fractions
  roofline_traffic.json                    HBM bytes per launch of the dominant kernel (read by bench.py)

    python scripts/make_profile_summary.py r1v5
"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# every <ACT, ACCUM, STATS, ...> instance of the 32 -> 32 kernel (bench.py's roofline): the two-term fp16 kernel (default since
# round 3), else the three-term bf16 kernel
DOMINANTS = ('conv_bwd_fused_kernel<', 'conv_f16x2_kernel<32, 32,', 'conv_bf16x3_kernel<32, 32,')


def main(tag):
    src = os.path.join(ROOT, 'gpurun_out', tag)
    dst = os.path.join(ROOT, 'profiles')
    for a, b in (('bench.json', f'{tag}_bench.json'), ('bench_sf.json', f'{tag}_bench_sf.json'),
                 ('bench_sf_bf16.json', f'{tag}_bench_sf_bf16.json'),
                 ('bench_prof.json', f'{tag}_bench_under_rocprof.json'),
                 ('trace/run_kernel_stats.csv', f'{tag}_kernel_stats.csv')):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(dst, b))
    summ = os.path.join(ROOT, 'scripts', 'pmc_summary.py')
    with open(os.path.join(dst, f'{tag}_pmc_sq.csv'), 'w') as f:
        subprocess.check_call([sys.executable, summ, os.path.join(src, 'sq', 'sq_counter_collection.csv')], stdout=f)
    with open(os.path.join(dst, f'{tag}_pmc_mem.csv'), 'w') as f:
        subprocess.check_call([sys.executable, summ, os.path.join(src, 'fetch', 'fetch_counter_collection.csv'),
                               os.path.join(src, 'write', 'write_counter_collection.csv')], stdout=f)
    stats = list(csv.DictReader(open(os.path.join(dst, f'{tag}_kernel_stats.csv'))))
    prof = json.loads(open(os.path.join(dst, f'{tag}_bench_under_rocprof.json')).read().strip().splitlines()[-1])
    # steps in the trace = launches of the once-per-step Adam counter kernel (warm-up + capture + timed graph replays + the
    # eager-launch leg + the per-call roofline step); before round 2: steps + warmup + 1
    adv = [r for r in stats if r['Name'].startswith('adam_advance_kernel')]
    nsteps = int(adv[0]['Calls']) if adv else prof['steps'] + prof['warmup'] + 1
    sq = {r['Kernel']: r for r in csv.DictReader(open(os.path.join(dst, f'{tag}_pmc_sq.csv')))}
    mem = {r['Kernel']: r for r in csv.DictReader(open(os.path.join(dst, f'{tag}_pmc_mem.csv')))}
    lines = ['| kernel | launches/step | ms/step | avg us | HBM read MB (2xFETCH_SIZE) | HBM write MB | GB/s | L2 hit | '
             'wait_any | wait_inst | active |', '|---|---|---|---|---|---|---|---|---|---|---|']
    tot = sum(float(r['TotalDurationNs']) for r in stats)
    for r in stats[:40]:
        k = r['Name']
        us = float(r['AverageNs']) / 1e3
        m, q = mem.get(k), sq.get(k)
        if m:
            rd, wr = float(m['FETCH_SIZE']) * 2 / 1024, float(m['WRITE_SIZE']) / 1024
            hit = float(m['TCC_HIT_sum']) / max(float(m['TCC_HIT_sum']) + float(m['TCC_MISS_sum']), 1)
            gbs = (rd + wr) / float(m['AvgUs']) * 1e3
            mtxt = f'{rd:.1f} | {wr:.1f} | {gbs:.0f} | {hit:.2f}'
        else:
            mtxt = ' | | | '
        if q:
            wc = max(float(q['SQ_WAVE_CYCLES']), 1)
            qtxt = f"{float(q['SQ_WAIT_ANY'])/wc:.2f} | {float(q['SQ_WAIT_INST_ANY'])/wc:.2f} | {float(q['SQ_ACTIVE_INST_ANY'])/wc:.2f}"
        else:
            qtxt = ' | | '
        lines.append(f"| `{k[:90]}` | {int(r['Calls'])/nsteps:.1f} | {float(r['TotalDurationNs'])/1e6/nsteps:.3f} | {us:.1f} | "
                     f"{mtxt} | {qtxt} |")
    with open(os.path.join(dst, f'{tag}_kernels.md'), 'w') as f:
        f.write(f'# {tag}: kernel ranking of the DIS-MF bs=4 step (sum of kernel time {tot/1e6/nsteps:.1f} ms/step over '
                f'{nsteps} profiled steps)\n\n'
                'HBM columns: PMC passes of one eager step (`scripts/prof_round.sh`), FETCH_SIZE doubled as '
                'MI355X_MICROARCH.md prescribes for gfx950; wait/active columns are fractions of SQ_WAVE_CYCLES.\n\n')
        f.write('\n'.join(lines) + '\n')
    fam, DOMINANT = [], DOMINANTS[0]
    # the dominant family = the one with the largest summed time in the PMC run (bench.py picks the same way from its live timings)
    best = None
    for D_ in DOMINANTS:
        fam_ = [m for k, m in mem.items() if D_ in k]
        t_ = sum(float(m['AvgUs']) * int(m['Calls']) for m in fam_)
        if fam_ and (best is None or t_ > best[0]):
            best = (t_, D_, fam_)
    DOMINANT, fam = (best[1], best[2]) if best else (DOMINANTS[0], [])
    # whole-step HBM traffic of the PMC passes: sum over ALL kernels of (2 x FETCH_SIZE + WRITE_SIZE) x calls, per step (the
    # once-per-step Adam counter kernel counts the steps of the PMC run)
    adv_m = [m for k, m in mem.items() if k.startswith('adam_advance_kernel')]
    pmc_steps = int(adv_m[0]['Calls']) if adv_m else 4
    step_bytes = sum((float(m['FETCH_SIZE']) * 2 + float(m['WRITE_SIZE'])) * 1024 * int(m['Calls']) for m in mem.values()
                     if m['FETCH_SIZE'] not in ('', 'nan') and m['WRITE_SIZE'] not in ('', 'nan')) / pmc_steps
    step_us = sum(float(m['AvgUs']) * int(m['Calls']) for m in mem.values()) / pmc_steps
    if fam:
        calls = sum(int(m['Calls']) for m in fam)
        rd = sum(float(m['FETCH_SIZE']) * 2 * 1024 * int(m['Calls']) for m in fam) / calls
        wr = sum(float(m['WRITE_SIZE']) * 1024 * int(m['Calls']) for m in fam) / calls
        json.dump({'kernel': DOMINANT + ' ACT, ACCUM, STATS, ...> (all instances, launch-weighted)',
                   'hbm_bytes_per_launch': rd + wr, 'read_bytes': rd, 'write_bytes': wr,
                   'launches_averaged': calls, 'launches_per_step': calls / pmc_steps,
                   'step_hbm_bytes': step_bytes, 'step_kernel_ms_under_pmc': step_us / 1e3, 'pmc_steps': pmc_steps,
                   'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over one eager step; '
                             'FETCH_SIZE (KB) doubled (gfx950 counts 64 B per 128-B request), WRITE_SIZE (KB) as read',
                   'source': f'profiles/{tag}_pmc_mem.csv'}, open(os.path.join(dst, 'roofline_traffic.json'), 'w'), indent=1)
    # family totals of the template kernels (the ranking above lists every instance on its own line)
    famt = {}
    for r in stats:
        base = r['Name'].split('<')[0].replace('void ', '')
        if '<' in r['Name']:
            e = famt.setdefault(base, [0, 0.0])
            e[0] += int(r['Calls'])
            e[1] += float(r['TotalDurationNs'])
    with open(os.path.join(dst, f'{tag}_kernels.md'), 'a') as f:
        f.write('\n## template families (all instances)\n\n| family | launches/step | ms/step | avg us |\n|---|---|---|---|\n')
        for base, (c, t) in sorted(famt.items(), key=lambda kv: -kv[1][1])[:12]:
            f.write(f'| `{base}` | {c/nsteps:.1f} | {t/1e6/nsteps:.3f} | {t/1e3/c:.1f} |\n')
    print('wrote profiles/', tag)


if __name__ == '__main__':
    main(sys.argv[1])

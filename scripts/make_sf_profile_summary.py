#!/usr/bin/env python
"""Turn scripts/prof_sf_round.sh's output (gpurun_out/<tag>/sf_f32, sf_bf16) into the tracked DIS-SF evidence under profiles/:

  <tag>_sf_<mode>_bench.json / _bench_under_rocprof.json   the bench lines
  <tag>_sf_<mode>_kernel_stats.csv                          rocprofv3 --kernel-trace --stats of the bench command
  <tag>_sf_<mode>_pmc_mem.csv                               per-kernel FETCH_SIZE / WRITE_SIZE / L2 counters (eager steps)
  <tag>_sf_<mode>_kernels.md                                ranking with measured HBM bytes and GB/s
  roofline_traffic_sf_<mode>.json                           HBM bytes per step and per conv launch (read by bench.py --arch single_frame)

    python scripts/make_sf_profile_summary.py r5v2
"""
import csv, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV_FAMILIES = ('convh2_kernel', 'convg2_fwd_kernel', 'convg_fwd_kernel', 'convg3_fwd_kernel', 'convb_halo_kernel', 'convb_fwd_kernel',
                 'convb_fwd128_kernel', 'conv_f16x2_kernel', 'conv_bf16x3_kernel')


def main(tag):
    dst = os.path.join(ROOT, 'profiles')
    for mode in ('f32', 'bf16'):
        src = os.path.join(ROOT, 'gpurun_out', tag, 'sf_' + mode)
        if not os.path.isdir(src):
            continue
        pre = f'{tag}_sf_{mode}'
        for a, b in (('bench.json', pre + '_bench.json'), ('bench_prof.json', pre + '_bench_under_rocprof.json'),
                     ('trace/run_kernel_stats.csv', pre + '_kernel_stats.csv')):
            if os.path.exists(os.path.join(src, a)):
                shutil.copy(os.path.join(src, a), os.path.join(dst, b))
        with open(os.path.join(dst, pre + '_pmc_mem.csv'), 'w') as f:
            subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'pmc_summary.py'),
                                   os.path.join(src, 'fetch', 'fetch_counter_collection.csv'),
                                   os.path.join(src, 'write', 'write_counter_collection.csv')], stdout=f)
        stats = list(csv.DictReader(open(os.path.join(dst, pre + '_kernel_stats.csv'))))
        mem = {r['Kernel']: r for r in csv.DictReader(open(os.path.join(dst, pre + '_pmc_mem.csv')))}
        adv = [r for r in stats if r['Name'].startswith('adam_advance_kernel')]
        nsteps = int(adv[0]['Calls']) if adv else 14
        adv_m = [m for k, m in mem.items() if k.startswith('adam_advance_kernel')]
        pmc_steps = int(adv_m[0]['Calls']) if adv_m else 4
        ok = [m for m in mem.values() if m['FETCH_SIZE'] not in ('', 'nan') and m['WRITE_SIZE'] not in ('', 'nan')]
        step_bytes = sum((float(m['FETCH_SIZE']) * 2 + float(m['WRITE_SIZE'])) * 1024 * int(m['Calls']) for m in ok) / pmc_steps
        step_us = sum(float(m['AvgUs']) * int(m['Calls']) for m in ok) / pmc_steps
        conv = [m for k, m in mem.items() if any(fam in k for fam in CONV_FAMILIES) and 'wgrad' not in k and m in ok]
        ccalls = sum(int(m['Calls']) for m in conv)
        cbytes = sum((float(m['FETCH_SIZE']) * 2 + float(m['WRITE_SIZE'])) * 1024 * int(m['Calls']) for m in conv)
        json.dump({'kernel': 'forward / input-gradient conv launches of DispNetS (' + ', '.join(CONV_FAMILIES) + ')',
                   'hbm_bytes_per_launch': cbytes / max(ccalls, 1), 'launches_averaged': ccalls, 'launches_per_step': ccalls / pmc_steps,
                   'step_hbm_bytes': step_bytes, 'step_kernel_ms_under_pmc': step_us / 1e3, 'pmc_steps': pmc_steps,
                   'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over eager steps; FETCH_SIZE (KB) doubled '
                             '(gfx950 counts 64 B per 128-B request), WRITE_SIZE (KB) as read',
                   'source': f'profiles/{pre}_pmc_mem.csv'}, open(os.path.join(dst, f'roofline_traffic_sf_{mode}.json'), 'w'), indent=1)
        tot = sum(float(r['TotalDurationNs']) for r in stats)
        lines = ['| kernel | launches/step | ms/step | avg us | HBM read MB (2xFETCH_SIZE) | HBM write MB | GB/s | L2 hit |', '|---|---|---|---|---|---|---|---|']
        listed = 0.0
        for r in stats[:45]:
            k = r['Name']
            m = mem.get(k)
            if m and m in ok:
                rd, wr = float(m['FETCH_SIZE']) * 2 / 1024, float(m['WRITE_SIZE']) / 1024
                hit = float(m['TCC_HIT_sum']) / max(float(m['TCC_HIT_sum']) + float(m['TCC_MISS_sum']), 1)
                mtxt = f"{rd:.1f} | {wr:.1f} | {(rd + wr) / float(m['AvgUs']) * 1e3:.0f} | {hit:.2f}"
            else:
                mtxt = ' | | | '
            listed += float(r['TotalDurationNs'])
            lines.append(f"| `{k[:100]}` | {int(r['Calls'])/nsteps:.1f} | {float(r['TotalDurationNs'])/1e6/nsteps:.3f} | {float(r['AverageNs'])/1e3:.1f} | {mtxt} |")
        fam = {}
        for r in stats:
            base = r['Name'].split('<')[0].split('(')[0].replace('void ', '')
            e = fam.setdefault(base, [0, 0.0])
            e[0] += int(r['Calls'])
            e[1] += float(r['TotalDurationNs'])
        with open(os.path.join(dst, pre + '_kernels.md'), 'w') as f:
            f.write(f'# {pre}: kernel ranking of the DIS-SF bs=8 step, {"bf16 activation storage" if mode == "bf16" else "fp32"} '
                    f'(sum of kernel time {tot/1e6/nsteps:.2f} ms/step over {nsteps} profiled steps; the 45 kernels listed: {listed/1e6/nsteps:.2f} ms)\n\n'
                    f'Whole step, PMC passes over {pmc_steps} eager steps: {step_bytes/1e9:.2f} GB of HBM traffic per step (2 x FETCH_SIZE + WRITE_SIZE), '
                    f'{step_us/1e3:.2f} ms of kernel time under the counters.\n\n')
            f.write('\n'.join(lines) + '\n\n## families (every kernel of the step: nothing unranked)\n\n| family | launches/step | ms/step | avg us |\n|---|---|---|---|\n')
            for base, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
                f.write(f'| `{base}` | {c/nsteps:.1f} | {t/1e6/nsteps:.3f} | {t/1e3/c:.1f} |\n')
        print('wrote', pre)


if __name__ == '__main__':
    main(sys.argv[1])

"""GPU: bench.py itself, as the driver runs it - the single-GPU line's roofline object and the 2-rank path (`--backend gloo`: both
ranks on the one GPU of the box, the same code path as RCCL minus the transport), so that the first multi-GPU run of the driver
cannot fail on plumbing.  Child processes (fresh interpreters), small step counts."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_roofline_covers_the_whole_kernel_family():
    d = _run(['--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-extra-legs'])
    assert d['n_gpus'] == 1 and d['config']['hip_graph'] is True and d['step_mode'] == 'graph'
    r = d['roofline']
    # the two kernel families of the 3x3 32 -> 32 layers: conv_f16x2_kernel (forward, GroupNorm-on-load forward, the input gradients that
    # keep a launch of their own) and - round 6 - conv_bwd_fused_kernel (input + weight gradient in one launch); the one with the larger
    # share of the step is `roofline`, the other `roofline.other_family`; together they are the round-5 line's 76 launches
    o = r['other_family']
    assert o is not None and ('conv_bwd_fused' in r['kernel']) != ('conv_bwd_fused' in o['kernel'])
    assert r['launches_per_step'] + o['launches_per_step'] == 76, (r['launches_by_entry_point'], o['launches_by_entry_point'])
    assert sum(r['launches_by_entry_point'].values()) == r['launches_per_step']
    fused = r if 'conv_bwd_fused' in r['kernel'] else o
    assert any(k.startswith('dis_conv2d_bwd_fused_f16x2') for k in fused['launches_by_entry_point'])
    assert 0.05 < o['frac_hbm'] < 1.0 and 0.02 < o['frac_mfma'] < 1.0
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] == ('GB/s' if r['bound'] == 'hbm' else 'TFLOP/s')
    assert abs(r['frac'] - max(r['frac_mfma'], r['frac_hbm'])) < 1e-12
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert 0.05 < r['frac_mfma'] < 1.0 and 0.05 < r['frac_hbm'] < 1.0
    assert r['flop_per_algorithmic_byte'] < 2 * 72.0 + 1e-6   # (72 flop/B for one product of a plain conv; the fused launch does two per byte set)
    st = r['step_traffic']
    assert st is None or (st['ratio'] > 1.0 and st['algorithmic_bytes_per_step'] > 8e10)
    # labels say which kernel served an entry point
    assert any('conv_f16x2_kernel' in k or 'conv_bwd_fused_kernel' in k for k in d['kernel_ms_one_eager_step'])


def test_two_ranks_gloo_bench_runs_and_replicas_agree():
    d = _run(['--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == [0, 1]
    assert d['replicas_equal'] is True
    assert d['config']['global_batch'] == 8 and d['config']['parallelism'] == 'dp2'
    assert d['step_mode'].startswith('eager')
    assert d['multi_gpu_measured'].startswith('gloo')
    assert d['value'] > 0 and d['adam_steps_taken'] >= 3

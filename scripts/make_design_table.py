#!/usr/bin/env python
"""DESIGN.md section 3's CURRENT-STATE kernel table, generated from a profiling round's tracked summaries
(profiles/<tag>_kernel_stats.csv: rocprofv3 --kernel-trace --stats of the bench command; profiles/<tag>_pmc_mem.csv: FETCH_SIZE /
WRITE_SIZE / L2 counters of one eager step) - one row per kernel family of the DIS-MF step: launches per step, ms per step, HBM
bytes per step (PMC), rate, L2 hit, plus what the family replaces in the reference, its file and its bound.

    python scripts/make_design_table.py r6v1            # prints the table; --write replaces the block between the markers in DESIGN.md
"""
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# family prefix -> (label, file, what it replaces (reference lines, /root/reference/model/...), bound)
FAMILIES = [
    ('conv_f16x2_kernel', 'two-term fp16 3x3 conv: forward / input gradient, 16 / 32 channels (GroupNorm on load, GroupNorm backward on load, channel sums, deferred block outputs)',
     'csrc/conv_f16x2.hip', 'Conv2d 3x3 + SELU (+ GroupNorm passes next to it) of ResNetBlock / Block2D3D, multi_frame_networks.py:338-345,514-542',
     'HBM (exposed latency in a power-limited clock; 72 flop/B < the 104 flop/B balance of 3 fp16 products)'),
    ('conv_wgrad_f16x2_kernel', 'two-term fp16 weight gradient (3x3 of the 16-channel and 16 <-> 32 layers; 4x4 stride 2), slabs per workgroup', 'csrc/conv_f16x2.hip',
     'the weight / bias gradient of the convs without a fused instance', 'its own instruction stream (transposing LDS reads, split, 2 barriers per tile); HBM 4 TB/s'),
    ('conv_bwd_fused_kernel', 'input + weight gradient of a 3x3 conv 32 -> 32 in one launch (round 6), every on-load / epilogue form of the input gradient', 'csrc/conv_bwd_fused.hip',
     'one Conv2d backward node of ResNetBlock / Block2D3D, multi_frame_networks.py:338-345,514-542',
     'LDS bandwidth of the transposing reads (dW phase) + serial vector-instruction issue at one wave per SIMD; HBM 3.4 TB/s (profiles/r6_bwd_fused.md)'),
    ('conv3d_bwd2_kernel', 'Conv3D backward, class-ordered deterministic form', 'csrc/conv3d_knn.hip', 'Conv3D backward incl. the gather scatter, multi_frame_networks.py:469-512',
     'dependent-instruction latency at 2 waves per SIMD'),
    ('conv3d_fwd_kernel', 'Conv3D forward (top-9-of-36 selection shared by all blocks)', 'csrc/conv3d_knn.hip', 'Conv3D.forward, :469-512', 'latency of the selection -> geometry -> gather chain'),
    ('conv3d_select_kernel', 'top-9-of-36 neighbour selection, ids equal torch.topk', 'csrc/conv3d_knn.hip', 'torch.topk, :498', 'VALU (exact key arithmetic)'),
    ('conv_k4s2', '4x4 stride-2 conv (two-term fp16): forward, one-launch input gradient', 'csrc/conv_k4s2.hip', 'Block2D3D.conv2_1, :338-345', 'HBM / LDS'),
    ('conv_fwd_kernel', 'exact-fp32 MFMA convs: 1x1 multi-frame conv (128 <-> 32) fwd / dgrad, 4 -> 16 stem', 'csrc/conv2d.hip', 'conv_mf, conv1, :406-416,:216-227', 'HBM (128-channel operand: 453 MB per launch)'),
    ('conv_wgrad_kernel', 'exact-fp32 MFMA weight gradients of the same', 'csrc/conv2d.hip', 'their weight gradients', 'HBM'),
    ('wgrad_reduce_kernel', 'slab reduce of the weight gradients (fp64 sums, fixed order)', 'csrc/conv2d.hip', '(part of the weight gradient)', 'launch latency (8 us each)'),
    ('gather_warped_feat', 'flow-guided feature gather (tiled) + CSR backward', 'csrc/layout_ops.hip', '96 warps per sample and their backward, :83-99,347-360', 'HBM / L2'),
    ('csr_', 'CSR index of the feature warps (once per forward)', 'csrc/layout_ops.hip', '(index structure, no reference counterpart)', 'HBM'),
    ('resize_nhwc', 'bilinear resize nhwc (tiled), forward + gather-form backward', 'csrc/layout_ops.hip', 'F.interpolate, :42-51', 'HBM'),
    ('gn_', 'GroupNorm launches of their own (apply, residual sums, coefficients)', 'csrc/norm_act.hip', 'GroupNorm(1, C) where no conv carries the pass', 'HBM (apply) / latency (coefficients, 11 us each)'),
    ('mf_geometry', 'unproject / view change / warped xyz / fb masks, once per forward', 'csrc/layout_ops.hip', ':172-214', 'HBM'),
    ('mask_weight', 'slot weighting of the multi-frame features', 'csrc/layout_ops.hip', ':410', 'HBM'),
    ('census_', 'census 9x9 window loss (multi-estimate), forward / backward', 'csrc/pixel_ops.hip', 'ext_functions.py:156-183 (the CTD torchext seam)', 'VALU (81 taps x rsq)'),
    ('geo_loss', 'flow-consistency terms (12 directions in one launch each way)', 'csrc/pixel_ops.hip', 'networks.py:554-661', 'exact divisions (fwd), float atomics (bwd)'),
    ('lcn_', 'local contrast normalisation', 'csrc/pixel_ops.hip', 'networks.py:336-377', 'VALU'),
    ('head_', 'disparity head 3x3 -> 1 + sigmoid affine', 'csrc/conv2d.hip', 'networks.py SigmoidAffine / predict_disp', 'HBM'),
    ('smooth_', 'edge-aware smoothness loss', 'csrc/pixel_ops.hip', 'networks.py:411-431', 'HBM'),
    ('pattern_warp', 'pattern warp by disparity (grid_sample border)', 'csrc/pixel_ops.hip', 'networks.py:372', 'HBM'),
    ('adam', 'Adam on the flat parameter buffer', 'csrc/adam.hip', 'torch.optim.Adam', 'HBM (2 MB)'),
    ('at::native', 'ATen glue left in the step (zero fills, joins of two-consumer gradients, scalar loss weighting)', '-', '-', 'launch latency'),
]


def main(tag, write=False):
    stats = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.csv'))))
    mem = {r['Kernel']: r for r in csv.DictReader(open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_mem.csv')))}
    adv = [r for r in stats if r['Name'].startswith('adam_advance_kernel')]
    nsteps = int(adv[0]['Calls']) if adv else 1
    fam = {}
    other = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    tot_ns = 0.0
    for r in stats:
        name = re.sub(r'^void ', '', r['Name'])
        calls, ns = float(r['Calls']) / nsteps, float(r['TotalDurationNs']) / nsteps
        tot_ns += ns
        m = mem.get(r['Name'])
        eager_calls = float(m['Calls']) if m and 'Calls' in m else None
        rd = wr = hit = miss = 0.0
        if m:
            per_rd, per_wr = float(m['FETCH_SIZE']) * 2 * 1024, float(m['WRITE_SIZE']) * 1024   # (KB per launch: FETCH_SIZE doubled on gfx950)
            rd, wr = per_rd * calls, per_wr * calls
            hit, miss = float(m['TCC_HIT_sum']) * calls, float(m['TCC_MISS_sum']) * calls
        for pre, *_ in FAMILIES:
            if name.startswith(pre):
                f = fam.setdefault(pre, [0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
                break
        else:
            f = other
        f[0] += calls; f[1] += ns; f[2] += rd; f[3] += wr; f[4] += hit; f[5] += miss
    lines = ['| kernel family (file) | replaces (reference) | launches / step | ms / step | HBM GB / step (PMC: read + write) | TB/s | L2 hit | bound |',
             '|---|---|---|---|---|---|---|---|']
    for pre, label, path, repl, bound in FAMILIES:
        if pre not in fam:
            continue
        c, ns, rd, wr, hit, miss = fam[pre]
        gb = (rd + wr) / 1e9
        lines.append(f'| `{pre}*` - {label} (`{path}`) | {repl} | {c:.0f} | {ns / 1e6:.2f} | {rd / 1e9:.1f} + {wr / 1e9:.1f} | '
                     f'{gb / (ns / 1e9) / 1e3 if ns else 0:.1f} | {hit / max(hit + miss, 1):.2f} | {bound} |')
    c, ns, rd, wr, hit, miss = other
    lines.append(f'| everything else | | {c:.0f} | {ns / 1e6:.2f} | {rd / 1e9:.1f} + {wr / 1e9:.1f} | | | |')
    lines.append(f'| **sum of kernel time** | | | **{tot_ns / 1e6:.2f}** | | | | |')
    text = '\n'.join(lines)
    if write:
        p = os.path.join(ROOT, 'DESIGN.md')
        s = open(p).read()
        a, b = '<!-- kernel-table:begin -->', '<!-- kernel-table:end -->'
        i, j = s.index(a) + len(a), s.index(b)
        s = s[:i] + f'\n(generated by `python scripts/make_design_table.py {tag} --write` from `profiles/{tag}_kernel_stats.csv` / `{tag}_pmc_mem.csv`)\n\n' + text + '\n' + s[j:]
        open(p, 'w').write(s)
    else:
        print(text)


if __name__ == '__main__':
    main(sys.argv[1], '--write' in sys.argv)

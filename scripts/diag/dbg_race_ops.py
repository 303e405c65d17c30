"""Diagnostic: which high-level op of the DIS-MF forward first produces a different result when several processes share the GPU?
The results of the wrapped ops functions are cloned (async copies only, no reductions) and compared with the first pass afterwards."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from depthinspace_amd import synth, ops
from depthinspace_amd.model import multi_frame_networks, multi_frame_worker
from depthinspace_amd.trainer import FlatAdam

WRAP = ['conv2d', 'group_norm', 'gather_warped_feat', 'conv3d_knn', 'conv2d_multi', 'resize_nhwc', 'conv2d_scaled_in',
        'conv3d_select', 'mf_geometry', 'mf_geometry_resize', 'slot_weights', 'planar_to_nhwc', 'nhwc_to_planar', 'resize_planar',
        'pack4_nhwc', 'disp_head', 'lcn', 'gather_csr']


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    only = set(sys.argv[2].split(',')) if len(sys.argv) > 2 and sys.argv[2] != 'all' else None
    small_only = len(sys.argv) > 3 and sys.argv[3] == 'small'
    H = W = 64
    args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic', architecture='multi_frame',
                              epochs=1, warmup_epochs=150, train_batch_size=1, max_disp=128)
    settings = synth.make_settings(H, W)
    torch.manual_seed(0)
    w = multi_frame_worker.Worker(args, settings=settings)
    net = multi_frame_networks.FuseNet((H, W), settings.K, settings.baseline).cuda()
    w.build_losses()
    w.current_epoch = 2
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=1234).items()}
    trace = []

    def wrap(name):
        f = getattr(ops, name)

        def g(*a, **k):
            r = f(*a, **k)
            if only is None or name in only:
                outs = r if isinstance(r, (tuple, list)) else (r,)
                keep = [o for o in outs if isinstance(o, torch.Tensor) and o.dtype != torch.int32]
                if small_only:
                    keep = [o for o in keep if o.numel() <= 64]
                trace.append((name, [o.detach().clone() for o in keep]))
            return r
        return g
    for n in WRAP:
        if hasattr(ops, n):
            setattr(ops, n, wrap(n))
    ref, found = None, 0
    hist = {}
    with torch.no_grad():
        for it in range(N):
            del trace[:]
            w.copy_data(batch, device=w.train_device, requires_grad=False, train=True)
            flow = w.read_optical_flow(True)
            out = w.net_forward(net, flow)
            trace.append(('OUT', [out.detach().clone()]))
            torch.cuda.synchronize()
            if ref is None:
                ref = list(trace)
                continue
            for ci, ((n0, t0), (n1, t1)) in enumerate(zip(ref, trace)):
                bad = [k for k, (a, b) in enumerate(zip(t0, t1)) if not torch.equal(a, b)]
                if bad:
                    found += 1
                    key = (ci, n0)
                    hist[key] = hist.get(key, 0) + 1
                    if found <= 8:
                        a, b = t0[bad[0]], t1[bad[0]]
                        d = (a.double() - b.double()).abs()
                        print(f'pass {it}: first differing op #{ci} {n0} (output {bad[0]} shape {tuple(a.shape)}): '
                              f'{int((d > 0).sum())} of {a.numel()} elements differ, max {float(d.max()):.2e}; '
                              f'previous ops: {[n for n, _ in ref[max(0, ci - 3):ci]]}', flush=True)
                    break
    print(f'{found} deviating passes of {N}; first differing op histogram: {sorted(hist.items(), key=lambda kv: -kv[1])[:6]}')


if __name__ == '__main__':
    main()

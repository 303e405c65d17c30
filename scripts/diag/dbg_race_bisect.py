"""Diagnostic: find the first C-ABI call whose results differ between evaluations of the same DIS-MF forward (+ backward with 'bwd')
when several processes share the GPU.  Every float tensor argument of every lib.call is checksummed on-stream (no host
synchronisation inside the pass); a deviating pass is compared call by call with the first pass."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from depthinspace_amd import synth, lib
from depthinspace_amd.model import multi_frame_networks, multi_frame_worker
from depthinspace_amd.trainer import FlatAdam


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    do_bwd = len(sys.argv) > 2 and sys.argv[2] == 'bwd'
    H = W = 64
    args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic', architecture='multi_frame',
                              epochs=1, warmup_epochs=150, train_batch_size=1, max_disp=128)
    settings = synth.make_settings(H, W)
    torch.manual_seed(0)
    w = multi_frame_worker.Worker(args, settings=settings)
    net = multi_frame_networks.FuseNet((H, W), settings.K, settings.baseline).cuda()
    w.build_losses()
    w.current_epoch = 2
    opt = FlatAdam(net.parameters(), lr=1e-4)
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=1234).items()}
    orig = lib.call
    trace = []

    def traced(name, *a):
        orig(name, *a)
        sums = []
        for t in a:
            if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() > 0:
                if t.dtype in (torch.float32, torch.float64):
                    sums.append(torch.nan_to_num(t.detach().double()).abs().sum())   # (uninitialised workspaces may hold NaN)
                elif t.dtype == torch.uint8:   # (int32 = CSR buffers: they contain scratch areas)
                    sums.append(t.detach().double().sum())
        trace.append((name, torch.stack(sums) if sums else None))
    lib.call = traced
    ref = None
    found = 0
    for it in range(N):
        del trace[:]
        w.copy_data(batch, device=w.train_device, requires_grad=False, train=True)
        opt.zero_grad()
        flow = w.read_optical_flow(True)
        out = w.net_forward(net, flow)
        if do_bwd:
            sum(w.loss_forward(out, True, flow)).backward()
        torch.cuda.synchronize()
        cur = [(n, None if s is None else s.cpu()) for n, s in trace]
        if ref is None:
            ref = cur
            continue
        assert len(cur) == len(ref)
        for ci, ((n0, s0), (n1, s1)) in enumerate(zip(ref, cur)):
            assert n0 == n1
            if s0 is None:
                continue
            if not torch.equal(s0, s1):
                rel = ((s0 - s1).abs() / (s0.abs() + 1e-300)).tolist()
                found += 1
                if found <= 6:
                    prev = ', '.join(n for n, _ in ref[max(0, ci - 3):ci])
                    print(f'pass {it}: first differing call #{ci} {n0}: rel diffs per tensor arg {["%.1e" % r for r in rel]} '
                          f'(previous calls: {prev})', flush=True)
                break
    print(f'{found} deviating passes of {N} ({"fwd+bwd" if do_bwd else "fwd"}; {len(ref)} calls per pass)')


if __name__ == '__main__':
    main()

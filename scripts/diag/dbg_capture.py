"""Diagnostic for a hipGraph capture crash: captures pieces of the DIS-SF step, one piece per subprocess."""
import argparse
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def child(stage, dtype, H, W):
    import faulthandler
    faulthandler.enable()
    import torch
    from depthinspace_amd import synth
    from depthinspace_amd.model import single_frame_worker, networks
    from depthinspace_amd.trainer import FlatAdam, GraphedStep
    args = argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic',
                              architecture='single_frame', epochs=1, warmup_epochs=150, train_batch_size=1, max_disp=128)
    settings = synth.make_settings(H, W)
    w = single_frame_worker.Worker(args, settings=settings)
    w.build_losses()
    w.current_epoch = 2
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=5).items()}
    torch.manual_seed(0)
    net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes,
                               act_dtype=torch.bfloat16 if dtype == 'bf16' else torch.float32).cuda()
    opt = FlatAdam(net.parameters(), lr=1e-4)
    x = torch.randn(4, 2, H, W, device='cuda')

    def fwd():
        return net(x)

    def fwdbwd():
        opt.zero_grad()
        out = net(x)
        sum(o.sum() for o in out).backward()

    if stage in ('fwd', 'fwdbwd'):
        f = fwd if stage == 'fwd' else fwdbwd
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f()
            f()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            f()
        g.replay()
        torch.cuda.synchronize()
    elif stage == 'step':
        gs = GraphedStep(w, net, opt, batch, use_graph=True, warmup=1)
        gs.run()
        gs.run()
        torch.cuda.synchronize()
    elif stage == 'eager_then_step':
        for _ in range(2):
            w.train_step(net, opt, batch)
        gs = GraphedStep(w, net, opt, batch, use_graph=True, warmup=1)
        gs.run()
        torch.cuda.synchronize()
    elif stage.startswith('exact'):
        if 'oracle' in stage:
            from oracle import dis_oracle as O  # noqa
        import numpy as np
        batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=5, scene='bumps').items()}
        for name, dt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
            if 'onlybf' in stage and dt == torch.float32:
                continue
            torch.manual_seed(0)
            net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes, act_dtype=dt).cuda()
            opt = FlatAdam(net.parameters(), lr=1e-4)
            p0 = opt.flat_p.clone()
            ls = []
            for _ in range(3):
                errs, _ = w.train_step(net, opt, batch)
                if 'nosync' not in stage:
                    ls.append([float(e) for e in errs])
            if dt == torch.bfloat16:
                if 'del' in stage:
                    del errs, _
                if 'gc' in stage:
                    import gc
                    gc.collect()
                    torch.cuda.empty_cache()
                g = GraphedStep(w, net, opt, batch, use_graph=True, warmup=(2 if 'w2' in stage else 1))
                g.run()
                g.run()
                torch.cuda.synchronize()
    elif stage.startswith('two_nets'):
        # the flow of tests/test_sf_bf16_gpu.py::test_worker_trains_and_evaluates_in_bf16
        scene = 'bumps' if 'bumps' in stage else 'plane'
        batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=5, scene=scene).items()}
        for dt in (torch.float32, torch.bfloat16):
            torch.manual_seed(0)
            net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes, act_dtype=dt).cuda()
            opt = FlatAdam(net.parameters(), lr=1e-4)
            for _ in range(3):
                w.train_step(net, opt, batch)
            if 'graph_both' in stage or dt == torch.bfloat16:
                gs = GraphedStep(w, net, opt, batch, use_graph=True, warmup=1)
                gs.run()
                torch.cuda.synchronize()
    print('OK', stage, dtype, H, W)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
    else:
        for dtype, H, W in (('bf16', 64, 64),):
            for stage in ('exact', 'exact_onlybf'):
                r = subprocess.run([sys.executable, __file__, 'child', stage, dtype, str(H), str(W)], capture_output=True, text=True)
                tail = (r.stdout.strip().splitlines() or [''])[-1]
                print(f'{dtype} {H}x{W} {stage}: rc={r.returncode} {tail}', flush=True)
                if r.returncode != 0:
                    err = [l for l in r.stderr.splitlines() if 'File "/' in l and 'repo' in l]
                    print('   ', ' | '.join(err[:6]))

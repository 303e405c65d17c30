"""Run-to-run variance of one training step (SURVEY.md section 5 asks for it: the reference's scatter kernels are atomic and its
results move from run to run).  The same DIS-MF / DIS-SF step is evaluated several times from identical parameters and inputs.
The forward side (index selection, disparities) is required to repeat to 1e-5 px (in practice bit for bit: its only
order-dependent sums are fp64); the losses are summed through fp64 atomics and are required to repeat to 1e-6 relative; the
gradients pass through the one float scatter that is left (the geo-loss backward warp; the Conv3D feature gradient is
class-ordered and reproducible since round 3, tests/test_net_ops_gpu.py::test_conv3d_class_ordered_backward) and are required to
repeat to 1e-5 of the gradient's largest entry - two orders below the tolerance of the parity tests that read them."""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(arch, bs):
    return argparse.Namespace(use_pseudo_gt=False, lcn_radius=5, track_length=4, data_type='synthetic', architecture=arch,
                              epochs=1, warmup_epochs=150, train_batch_size=bs, max_disp=128)


@pytest.mark.parametrize('arch', ['multi_frame', 'single_frame'])
def test_step_repeats(arch):
    from depthinspace_amd import synth
    from depthinspace_amd.model import multi_frame_networks, multi_frame_worker, single_frame_worker, networks
    from depthinspace_amd.trainer import FlatAdam
    H = W = 64
    settings = synth.make_settings(H, W)
    torch.manual_seed(0)
    if arch == 'multi_frame':
        w = multi_frame_worker.Worker(_args(arch, 1), settings=settings)
        net = multi_frame_networks.FuseNet((H, W), settings.K, settings.baseline).cuda()
    else:
        w = single_frame_worker.Worker(_args(arch, 1), settings=settings)
        net = networks.DispDecoder(channels_in=2, max_disp=128, imsizes=w.imsizes).cuda()
    w.build_losses()
    w.current_epoch = 2
    opt = FlatAdam(net.parameters(), lr=1e-4)
    batch = {k: torch.from_numpy(v) for k, v in synth.make_batch(settings, 1, 4, seed=4321, scene='bumps').items()}
    runs = []
    for _ in range(4):
        w.copy_data(batch, device=w.train_device, requires_grad=False, train=True)
        opt.zero_grad()
        flow = w.read_optical_flow(True)
        out = w.net_forward(net, flow)
        losses = w.loss_forward(out, True, flow)
        sum(losses).backward()
        torch.cuda.synchronize()
        disp = out[0] if isinstance(out, (list, tuple)) else out   # finest scale
        runs.append((disp.detach().clone(), np.array([float(l) for l in losses]), opt.flat_g.clone()))
    d0, l0, g0 = runs[0]
    gmax = float(g0.abs().max())
    worst_g = 0.0
    for d, l, g in runs[1:]:
        # (GroupNorm statistics are fp64 atomic sums: a different arrival order can move a float32 scale by one ulp, so
        # bit equality is the usual outcome but not a guarantee)
        assert float((d - d0).abs().max()) < 1e-5, 'forward pass does not repeat'
        np.testing.assert_allclose(l, l0, rtol=1e-6, atol=0)
        worst_g = max(worst_g, float((g - g0).abs().max()) / gmax)
    print(arch, 'run-to-run: max gradient difference / max |gradient| =', worst_g)
    assert worst_g < 1e-5

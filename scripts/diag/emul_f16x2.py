"""CPU emulation: what would a 2-term fp16 operand split (x = h1 + h2, products a1b1 + a1b2 + a2b1, power-of-two scale per
tensor) in the 3x3 stride-1 16/32-channel convs (forward, dgrad, wgrad) do to the DIS-MF step, end to end?  Compares the
oracle step with emulated convs against the plain fp32 oracle step on the same neighbour sets.
    python scripts/diag/emul_f16x2.py [size] [mode]      mode: f16x2 | bf16x3 | none"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import dis_oracle as O
from depthinspace_amd import synth

MODE = sys.argv[2] if len(sys.argv) > 2 else 'f16x2'


def split_round(v):
    """value the matrix unit effectively sees: sum of the split terms (exact in fp64)"""
    z = torch.zeros_like(v, dtype=torch.float64)
    if MODE == 'none':
        return v.double(), (v.double(), z)
    if MODE == 'bf16x3':
        a = v.bfloat16().float(); r = v - a
        b = r.bfloat16().float(); r2 = r - b
        c = r2.bfloat16().float()
        return a.double() + b.double() + c.double(), (a.double(), z)
    m = float(v.abs().max())
    if m == 0:
        return v.double(), (v.double(), z)
    s = 2.0 ** (14 - int(np.floor(np.log2(m))))   # max -> [2^14, 2^15)
    x = v * s
    h1 = x.half().float()
    h2 = (x - h1).half().float()
    return (h1.double() + h2.double()) / s, (h1.double() / s, h2.double() / s)


class EmuConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        xr, (x1, x2) = split_round(x)
        wr, (w1, w2) = split_round(w)
        y = F.conv2d(xr, wr, None) - F.conv2d(x2, w2, None)   # drop the a2*b2 term
        return (y + b.double().view(1, -1, 1, 1)).float()

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gr, (g1, g2) = split_round(gy)
        wr, (w1, w2) = split_round(w)
        xr, (x1, x2) = split_round(x)
        gx = torch.nn.grad.conv2d_input(x.shape, wr, gr) - torch.nn.grad.conv2d_input(x.shape, w2, g2)
        gw = torch.nn.grad.conv2d_weight(xr, w.shape, gr) - torch.nn.grad.conv2d_weight(x2, w.shape, g2)
        return gx.float(), gw.float(), gy.sum((0, 2, 3))


orig = O._pconv


def emu_pconv(p, name, x, stride=1):
    w = p[name + '.weight']
    if stride == 1 and w.shape[-1] == 3 and w.shape[0] in (16, 32) and w.shape[1] in (16, 32, 48, 96):
        return EmuConv.apply(F.pad(x, (1,) * 4), w, p[name + '.bias'])
    return orig(p, name, x, stride)


def run(emul, size, seed_b=4321, seed_p=14):
    H = W = size
    settings = synth.make_settings(H, W)
    batch = synth.make_batch(settings, 1, 4, seed=seed_b, scene='bumps', motion=1.5)
    params = O.init_params(O.mf_param_shapes(), seed=seed_p)
    ctx = O.StepContext(settings)
    O._pconv = emu_pconv if emul else orig
    try:
        return O.train_step(ctx, 'multi_frame', params, {k: torch.from_numpy(v) for k, v in batch.items()}, epoch=2)
    finally:
        O._pconv = orig


if __name__ == '__main__':
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    torch.set_num_threads(8)
    O.CONV3D_TAP = []
    ref = run(False, size)
    tap, O.CONV3D_TAP = O.CONV3D_TAP, None
    sets = {}
    for lname, tag in (('conv3d_1', 'core'), ('conv3d_2', 'quarter')):
        sets[tag] = torch.stack([c['idx'] for c in tap if c['name'] == f'blocks.0.{lname}'], 0)
    O.CONV3D_FORCE = sets
    emu = run(True, size)
    O.CONV3D_FORCE = None
    d = (emu['out'] - ref['out']).abs()
    print(MODE, 'disp L1 %.3e max %.3e' % (float(d.mean()), float(d.max())))
    print('loss terms max rel', max(abs(float(a) - float(b)) / (abs(float(b)) + 1e-12) for a, b in zip(emu['vals'], ref['vals'])))
    rows = []
    for k, g in ref['grads'].items():
        if g is None:
            continue
        rows.append((float((emu['grads'][k] - g).abs().max()) / (float(g.abs().max()) + 1e-30), k))
    rows.sort(reverse=True)
    print('worst gradient rel errors:', rows[:5])

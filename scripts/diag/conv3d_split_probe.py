"""Conv3D backward (class-ordered form): does splitting the 16 (target, batch) samples into groups whose class chains run on
DIFFERENT streams shorten the call?  Samples are independent (disjoint rows of the feature gradient), only the classes of one
sample are ordered.  Times: one call on all targets; the same work as G calls on tl / G targets each, on one stream and on G streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from depthinspace_amd import ops


def run(h, w, stride, reps=20):
    tl, bs, C = 4, 4, 32
    g = torch.Generator(device='cuda').manual_seed(1)
    yy, xx = torch.meshgrid(torch.arange(h, device='cuda', dtype=torch.float32), torch.arange(w, device='cuda', dtype=torch.float32), indexing='ij')
    geom = torch.empty(tl, bs, h, w, tl, 4, device='cuda')
    z = 1.0 + 0.3 * torch.rand(tl, bs, h, w, tl, device='cuda', generator=g)
    geom[..., 0] = (xx[None, None, :, :, None] / w - 0.5) * z
    geom[..., 1] = (yy[None, None, :, :, None] / w - 0.5) * z
    geom[..., 2] = z
    geom[..., 3] = (torch.rand(tl, bs, h, w, tl, device='cuda', generator=g) > 0.1).float()
    wf = torch.randn(tl, bs, h, w, tl, C, device='cuda', generator=g)
    ps = [torch.randn(s, device='cuda', generator=g) * 0.3 for s in ((16, 3), (16,), (32, 16), (32,), (32, 32))]
    idx = ops.conv3d_select(geom, stride)
    ho, wo = idx.shape[2:4]
    y, agg = torch.empty((tl, bs, ho, wo, C), device='cuda'), torch.empty((tl, bs, ho, wo, C), device='cuda')
    ops.lib.call('dis_conv3d_knn_fwd_agg', geom, wf, *ps, idx, y, agg, tl, bs, h, w, stride)
    gy = torch.randn(y.shape, device='cuda', generator=g)
    gw = torch.zeros_like(wf)
    P = lambda t: t.data_ptr()
    f = ops.lib.fn('dis_conv3d_knn_bwd_det')
    streams = [torch.cuda.Stream() for _ in range(4)]

    # (the kernels see tl * bs independent samples and a fixed slot count: a contiguous chunk of the flattened sample dimension is
    #  passed as tl = 4 targets of bs / G batch entries)
    flat = lambda t: t.reshape((tl * bs,) + tuple(t.shape[2:]))
    gF, wF, iF, yF, aF, gyF, gwF = (flat(t) for t in (geom, wf, idx, y, agg, gy, gw))

    def call(k, G, gp, ws, stream):
        n0 = k * (tl * bs // G)
        rc = f(P(gF[n0]), P(wF[n0]), *[P(p) for p in ps], P(iF[n0]), P(yF[n0]), P(aF[n0]), P(gyF[n0]), P(gwF[n0]), P(gp), P(ws), tl, bs // G,
               h, w, stride, stream.cuda_stream)
        assert rc == 0, rc

    for G in (1, 2, 4):
        nt = tl * bs // G
        gps = [torch.empty(1632, device='cuda') for _ in range(G)]
        wss = [torch.empty(ops.lib.fn('dis_conv3d_knn_bwd_det_workspace')(tl, bs // G, h, w, stride), device='cuda') for _ in range(G)]
        for multi in ((False,) if G == 1 else (False, True)):
            main = torch.cuda.current_stream()

            def once():
                if multi:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    for k in range(G):
                        streams[k].wait_event(ev)
                        call(k, G, gps[k], wss[k], streams[k])
                    for k in range(G):
                        e2 = torch.cuda.Event()
                        e2.record(streams[k])
                        main.wait_event(e2)
                else:
                    for k in range(G):
                        call(k, G, gps[k], wss[k], main)
            for _ in range(3):
                once()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                once()
            e1.record()
            torch.cuda.synchronize()
            print(f'h={h} w={w} stride={stride}: {G} group(s) of {nt} sample(s), {"one stream per group" if multi else "one stream"}: '
                  f'{e0.elapsed_time(e1) / reps:.3f} ms', flush=True)


run(256, 216, 2)
run(128, 108, 1)

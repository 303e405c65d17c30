"""Diagnostic: is the FIRST launch of conv3d_bwd in a fresh process different from the following ones?  Spawns pairs of fresh
processes that run concurrently on the GPU; each calls dis_conv3d_knn_fwd/bwd several times on the same inputs."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def child(tag):
    import torch
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(11)
    out = []
    for (h, w, stride) in ((32, 32, 2), (16, 16, 1)):
        tl, bs, C = 4, 1, 32
        geom = torch.randn(tl, bs, h, w, tl, 4, generator=g)
        geom[..., 2] = geom[..., 2].abs() + 1.0
        geom[..., 3] = (torch.rand(tl, bs, h, w, tl, generator=g) > 0.2).float()
        geom = geom.cuda()
        wf = torch.randn(tl, bs, h, w, tl, C, generator=g).cuda()
        ps = [torch.randn(s, generator=g).cuda() * 0.3 for s in ((16, 3), (16,), (32, 16), (32,), (32, 32))]
        idx = ops.conv3d_select(geom, stride)
        ho, wo = idx.shape[2:4]
        y = torch.empty((tl, bs, ho, wo, C), device='cuda')
        args = (geom, wf, ps[0], ps[1], ps[2], ps[3], ps[4], idx)
        ops.lib.call('dis_conv3d_knn_fwd', *args, y, tl, bs, h, w, stride)
        gy = torch.randn(y.shape, generator=g).cuda()
        res = []
        for it in range(4):
            gwf = torch.zeros_like(wf)
            gp = torch.empty(1632, device='cuda')
            acc = torch.empty(ops.lib.fn('dis_conv3d_knn_bwd_workspace')(), device='cuda')
            ops.lib.call('dis_conv3d_knn_bwd', *args, y, gy, gwf, gp, acc, tl, bs, h, w, stride)
            torch.cuda.synchronize()
            res.append((gp.clone(), gwf.clone()))
        ref = res[-1]
        for it in range(3):
            dw = float((res[it][0][:1024] - ref[0][:1024]).abs().max()) / float(ref[0][:1024].abs().max())
            dr = float((res[it][0][1024:] - ref[0][1024:]).abs().max()) / float(ref[0][1024:].abs().max())
            df = float((res[it][1] - ref[1]).abs().max()) / float(ref[1].abs().max())
            out.append(f's{stride} it{it}: w {dw:.2e} mlp {dr:.2e} feat {df:.2e}')
    print(tag, ' | '.join(out), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(sys.argv[2])
    else:
        for rnd in range(5):
            ps = [subprocess.Popen([sys.executable, __file__, 'child', f'r{rnd}p{k}'], stdout=subprocess.PIPE,
                                   stderr=subprocess.DEVNULL, text=True) for k in range(3)]
            for p in ps:
                print(p.communicate()[0].strip())

"""errors of the Conv3D backward modes against the golden gradients of tests/golden/ops.npz (DIS_CONV3D_BWD=det|agg|atomic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import dis_oracle as O
from depthinspace_amd import ops


def relerr(a, b):
    a = a.detach().cpu().double(); b = b.double()
    return float((a - b).abs().max() / b.abs().max())


G = np.load(os.path.join(ROOT, 'tests', 'golden', 'ops.npz'))
for stride in (1, 2):
    xyz, feat, mask = [torch.from_numpy(G[k]) for k in ('c3_xyz', 'c3_feat', 'c3_mask')]
    tl, bs, C, h, w = feat.shape
    p = O.init_params({k: v for k, v in O.mf_param_shapes().items() if k.startswith('blocks.0.conv3d_1')}, seed=5)
    pd = {k[len('blocks.0.conv3d_1.'):]: v.detach().cuda().requires_grad_(True) for k, v in p.items()}
    geom1 = torch.cat([xyz, mask], dim=2).permute(1, 3, 4, 0, 2)
    geom = geom1.unsqueeze(0).expand(tl, -1, -1, -1, -1, -1).contiguous().cuda()
    wf = feat.permute(1, 3, 4, 0, 2).unsqueeze(0).expand(tl, -1, -1, -1, -1, -1).contiguous().cuda().requires_grad_(True)
    idx = ops.conv3d_select(geom, stride)
    y = ops.conv3d_knn(geom, wf, pd['dense1.0.weight'], pd['dense1.0.bias'], pd['dense2.0.weight'], pd['dense2.0.bias'], pd['w'],
                       idx, stride)
    ho, wo = y.shape[2:4]
    out = ops.group_norm(y.view(tl * bs, ho, wo, C), pd['bn.weight'], pd['bn.bias']).view(tl, bs, ho, wo, C)
    gfull = torch.zeros(tl, bs, ho, wo, C)
    gfull[1] = torch.from_numpy(G[f'c3_s{stride}_go']).permute(0, 2, 3, 1)
    out.backward(gfull.cuda())
    print(ops.CONV3D_BWD, 'stride', stride, 'gfeat', '%.2e' % relerr(wf.grad[1].permute(3, 0, 4, 1, 2), torch.from_numpy(G[f'c3_s{stride}_gfeat'])),
          ' '.join('%s %.2e' % (k_, relerr(pd[k_].grad, torch.from_numpy(G[f'c3_s{stride}_g:{k_}'])))
                   for k_ in ('w', 'dense1.0.weight', 'dense1.0.bias', 'dense2.0.weight', 'dense2.0.bias')))

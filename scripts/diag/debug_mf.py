"""Stage-by-stage comparison of the HIP FuseNet forward against the CPU oracle (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
from oracle import dis_oracle as O
from depthinspace_amd import synth, ops
from depthinspace_amd.model import multi_frame_networks as M

def rel(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

H = W = 64; bs = 1; tl = 4
st = synth.make_settings(H, W)
batch = synth.make_batch(st, bs, 4, seed=1234)
p = O.init_params(O.mf_param_shapes(), seed=11)
ctx = O.StepContext(st)
data = O.copy_data(ctx, {k: torch.from_numpy(v) for k, v in batch.items()})
flow = O.read_optical_flow(data, 4)
ir, amb, d = data['im0'], data['ambient0'], data['primary_disp']
depth = O.disp_to_depth(d, ctx.focal, ctx.baseline)
R, t = data['R'], data['t']
h, w = H // 2, W // 2
with torch.no_grad():
    x = torch.cat((ir, amb), 2).reshape(tl * bs, 3, H, W)
    x = torch.cat([x, d.reshape(tl * bs, 1, H, W)], dim=1)
    c1 = F.selu(O._pconv(p, 'conv1.1', x, stride=2))
    c2 = F.selu(O._pconv(p, 'conv2.1', c1))
    c3 = F.selu(O._pconv(p, 'conv3.1', c2))
    c4 = F.selu(O._pconv(p, 'conv4.1', c3))
    r1 = O._resblock(p, 'res1', c4)
    r3 = O._resblock(p, 'res3', O._resblock(p, 'res2', r1))
    feat0 = r3.view(tl, bs, *r3.shape[1:])
    depth_core = O.resize_ac(depth, (h, w)); flow_core = O.resize_flow(flow, (h, w))
    wxyz, wmask = O.mf_geometry(depth_core, O.mf_core_rays(st.K, H, W), R, t, flow_core)
    f = feat0
    blocks = []
    for b in range(4):
        f = O.mf_block(p, f'blocks.{b}.', f, wxyz, wmask, flow_core, tl)
        blocks.append(f)
    out_ref = O.mf_forward(p, st.K, ir, amb, d, depth, R, t, flow)

net = M.FuseNet((H, W), st.K, st.baseline).cuda()
net.load_state_dict({k: v.detach() for k, v in p.items()})
c = lambda t_: t_.contiguous().cuda()
with torch.no_grad():
    irc, ambc, dc, depc = c(ir), c(amb), c(d), c(depth)
    N, HW = tl * bs, H * W
    x4 = ops.pack4_nhwc([(irc, 2 * HW), (irc.view(-1)[HW:], 2 * HW), (ambc, HW), (dc, HW)], N, H, W)
    print('x4', rel(x4.permute(0, 3, 1, 2), x))
    cv = lambda x_, s, stride, pad: ops.conv2d(x_, s[1].weight, s[1].bias, stride, pad, 1, need_dgrad=False)[0]
    g1 = cv(x4, net.conv1, 2, 1); print('conv1', rel(g1.permute(0, 3, 1, 2), c1))
    g2 = cv(g1, net.conv2, 1, 1); print('conv2', rel(g2.permute(0, 3, 1, 2), c2))
    g3 = cv(g2, net.conv3, 1, 1); print('conv3', rel(g3.permute(0, 3, 1, 2), c3))
    g4 = cv(g3, net.conv4, 1, 1); print('conv4', rel(g4.permute(0, 3, 1, 2), c4))
    q1 = net.res1(g4); print('res1', rel(q1.permute(0, 3, 1, 2), r1))
    q3 = net.res3(net.res2(q1)); print('res3', rel(q3.permute(0, 3, 1, 2), r3))
    feat = q3.view(tl, bs, h, w, 32)
    flw = M.FlowDict({k: c(v) for k, v in flow.items()})
    ff = M.stack_flows(flw, tl)
    fc_p = ops.resize_planar(ff, (h, w), True, flow_scale=(w / W, h / H))
    for i in range(4):
        for j in range(4):
            if i != j:
                e = rel(fc_p[i * 4 + j], flow_core[f'flow_{i}{j}'])
                if e > 1e-6: print('flow core', i, j, e)
    flows = ops.planar_to_nhwc(fc_p.view(16 * bs, 2, h, w)).view(16, bs, h, w, 2)
    dcore = ops.resize_planar(depc.view(tl, bs, H, W), (h, w), True)
    print('depth_core', rel(dcore, depth_core[:, :, 0]))
    geom = ops.mf_geometry(dcore, c(R), c(t), flows, net._Ki_host, 2, 2)
    print('geom xyz', rel(geom[..., :3].permute(0, 4, 1, 5, 2, 3), wxyz), 'mask mismatch',
          float((geom[..., 3].permute(0, 4, 1, 2, 3).unsqueeze(3).cpu() != wmask).float().mean()))
    hq, wq = 16, 16
    fq_p = ops.resize_planar(fc_p, (hq, wq), True, flow_scale=(wq / w, hq / h))
    flows_q = ops.planar_to_nhwc(fq_p.view(16 * bs, 2, hq, wq)).view(16, bs, hq, wq, 2)
    geom_q = ops.mf_geometry_resize(geom, (hq, wq))
    fcur = feat
    for b in range(4):
        blk = net.blocks[b]
        # sub-stages of block b
        fref = feat0 if b == 0 else blocks[b - 1]
        B = f'blocks.{b}.'
        wfeat = torch.stack([O._gather_warped_feat(fref, flow_core, ti, tl) for ti in range(tl)], 0)
        wf = ops.gather_warped_feat(fcur, flows)
        print(b, 'wf', rel(wf.permute(0, 4, 1, 5, 2, 3), wfeat))
        o3d1_ref = torch.stack([O.conv3d_knn(p, B + 'conv3d_1', wxyz[ti], wfeat[ti], wmask[ti], 2, tl) for ti in range(tl)], 0)
        o3d1 = blk.conv3d_1(geom, wf)
        print(b, 'o3d1', rel(o3d1.permute(0, 1, 4, 2, 3), o3d1_ref))
        fcur = blk(fcur, geom, geom_q, flows, flows_q)
        print(b, 'block out', rel(fcur.permute(0, 1, 4, 2, 3), blocks[b]))
    out = net(irc, ambc, dc, depc, c(R), c(t), flw)
    print('out', rel(out, out_ref), float((out.cpu() - out_ref).abs().mean()))

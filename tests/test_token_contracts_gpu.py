"""The inter-node contracts of the GroupNorm-backward tokens (ops._GN_LAZY) fail LOUDLY and in the step that broke them.

A token stands for the gradient wrt a conv output whose only consumer is a GroupNorm (reference: GroupNorm(1, C) behind a conv,
model/multi_frame_networks.py:307-345,514-542).  It is a view of a NaN-filled pool: if autograd ever sums it with the gradient of a
second consumer, the conv's backward raises in that backward pass; a token nobody redeemed is reported by FlatAdam.step() BEFORE
the update."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _chain(second_consumer):
    from depthinspace_amd import ops
    g = torch.Generator().manual_seed(3)
    n, h, w, c = 2, 32, 24, 32
    x = torch.randn(n, h, w, c, generator=g).cuda().requires_grad_(True)
    w1 = (torch.randn(c, c, 3, 3, generator=g) * 0.05).cuda().requires_grad_(True)
    b1 = torch.zeros(c, device='cuda', requires_grad=True)
    gamma = torch.ones(c, device='cuda', requires_grad=True)
    beta = torch.zeros(c, device='cuda', requires_grad=True)
    y, st = ops.conv2d(x, w1, b1, 1, 1, ops.ACT_NONE, want_stats=True)
    assert getattr(y, '_gn_lazy_ok', 0), 'the 3x3 32->32 conv output is answered with tokens'
    o = ops.group_norm(y, gamma, beta, stats=st)
    # (a consumer in front of the GroupNorm's output that leaves channel sums, so that the GroupNorm's backward takes the token route)
    w2 = (torch.randn(c, c, 3, 3, generator=g) * 0.05).cuda().requires_grad_(True)
    z, _ = ops.conv2d(o, w2, None, 1, 1, ops.ACT_NONE)
    loss = z.sum()
    if second_consumer:
        loss = loss + (y * 0.5).sum()   # y has a second consumer: autograd would add the token to a real gradient
    return loss, (x, w1, b1, gamma, beta, w2)


def test_token_route_is_taken_and_clean():
    from depthinspace_amd import ops
    if ops.lib.fn('dis_get_conv_split')() != 1 or not ops.GN_LAZY:
        pytest.skip('tokens: two-term fp16 kernels')
    loss, leaves = _chain(False)
    loss.backward()
    torch.cuda.synchronize()
    ops.check_backward_complete()
    for t in leaves:
        assert t.grad is not None and bool(torch.isfinite(t.grad).all())


def test_token_with_second_consumer_raises_in_that_backward():
    from depthinspace_amd import ops
    if ops.lib.fn('dis_get_conv_split')() != 1 or not ops.GN_LAZY:
        pytest.skip('tokens: two-term fp16 kernels')
    loss, _ = _chain(True)
    try:
        with pytest.raises(RuntimeError, match='second consumer'):
            loss.backward()
    finally:
        ops._GN_LAZY.clear()
        ops._GN_TOKEN_FOR.clear()
        ops._GN_PRE.clear()


def test_tokens_are_nan_and_unique():
    from depthinspace_amd import ops
    dev = torch.device('cuda', torch.cuda.current_device())
    a, b = ops._new_token(dev), ops._new_token(dev)
    assert a.data_ptr() != b.data_ptr()
    assert bool(torch.isnan(a).all()) and bool(torch.isnan(b.expand(2, 3)).all())


def test_stale_token_raises_at_optimizer_step_before_the_update():
    from depthinspace_amd import ops
    from depthinspace_amd.trainer import FlatAdam
    p = torch.nn.Parameter(torch.ones(8, device='cuda'))
    opt = FlatAdam([p], lr=1e-2)
    opt.zero_grad()
    p.grad.add_(1.0)
    before = p.detach().clone()
    dev = torch.device('cuda', torch.cuda.current_device())
    tok = ops._new_token(dev)
    ops._GN_LAZY[tok.data_ptr()] = (tok, None, None, None, 0)   # a token nobody redeemed
    with pytest.raises(RuntimeError, match='not redeemed'):
        opt.step()
    assert torch.equal(p.detach(), before), 'the check runs before the update'
    assert not ops._GN_LAZY
    opt.step()
    assert not torch.equal(p.detach(), before)


def test_gn_bwd_coef_parameter_gradients_are_bit_stable():
    """dis_gn_bwd_coef's last-arriving block adds the samples' dgamma / dbeta parts (agent-scope acq_rel arrival): over many launches
    with n >= 16 blocks the result never moves and equals dis_gn_bwd_from_sums'."""
    from depthinspace_amd import ops
    L = ops.lib
    g_ = torch.Generator().manual_seed(5)
    n, h, w, c = 24, 40, 36, 32
    q = torch.randn(n, h, w, c, generator=g_).cuda()
    gq = torch.randn(n, h, w, c, generator=g_).cuda()
    gamma = (torch.rand(c, generator=g_) + 0.5).cuda()
    st = torch.stack([q.double().sum(dim=(1, 2, 3)), (q.double() ** 2).sum(dim=(1, 2, 3))], 1).reshape(-1).contiguous()
    slots = L.fn('dis_conv2d_gnsums_slots')()
    ab0 = torch.zeros(n, slots, 2, c, dtype=torch.float64, device='cuda')
    ab0[:, 0, 0] = gq.double().sum(dim=(1, 2))
    ab0[:, 0, 1] = (gq.double() * q.double()).sum(dim=(1, 2))
    ncoef = n * (c + 2) + 4 * n * c + 2
    gpre_ref = torch.empty_like(gq)
    gg_ref, gb_ref = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    coef_ref = torch.empty(ncoef, dtype=torch.float32, device='cuda')
    L.call('dis_gn_bwd_from_sums', gq, q, st, gamma, ab0, slots, gpre_ref, gg_ref, gb_ref, coef_ref, n, h * w, c, 1e-5, 0)
    counter = torch.zeros(2, dtype=torch.int32, device='cuda')
    for rep in range(200):
        coef = torch.empty(ncoef, dtype=torch.float32, device='cuda')
        gg, gb = torch.full((c,), float('nan'), device='cuda'), torch.full((c,), float('nan'), device='cuda')
        L.call('dis_gn_bwd_coef', st, gamma, ab0, slots, coef, gg, gb, counter, n, h * w, c, 1e-5)
        assert torch.equal(gg, gg_ref) and torch.equal(gb, gb_ref), rep
        assert torch.equal(coef[:n * (c + 2)], coef_ref[:n * (c + 2)]), rep
    assert int(counter[0]) == 0

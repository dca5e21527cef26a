"""GPU parity of the set-abstraction first layer on the neighbour lists (sug_sa_first_fwd / bwd) against the
operator chain of the reference (model/pointnet2_utils.py:107-135, 193-198): index_points(xyz, idx) - new_xyz,
index_points(points, idx), cat, 1x1 conv (+bias), train-mode BatchNorm2d, ReLU -- in plain fp32 torch on the
grouped [B,S,ns,3+D] tensor.  Tolerance 1e-4 forward (north star), 1e-3 relative on the gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(xyz, points, new_xyz, idx, W, b, bn, groups):
    B, S, ns = idx.shape
    bi = torch.arange(B, device=xyz.device).view(B, 1, 1)
    g = xyz[bi, idx.long()] - new_xyz.view(B, S, 1, 3)
    if points is not None:
        g = torch.cat((g, points[bi, idx.long()]), dim=-1)
    outs = []
    for gg in g.chunk(groups, dim=0):
        y = gg @ W.t() + b
        outs.append(torch.relu(bn(y.reshape(-1, y.shape[-1])).view(y.shape)))
    return torch.cat(outs)


@pytest.mark.parametrize('geo', [False, True])
@pytest.mark.parametrize('B,N,S,ns,D,C,groups,train', [
    (4, 256, 64, 32, 0, 64, 1, True),        # sa1: xyz only
    (4, 128, 32, 64, 128, 128, 2, True),     # sa2: xyz + 128 features, paired domains
    (2, 200, 50, 20, 5, 64, 1, True),        # ragged sizes, ns not a multiple of the row batch
    (4, 128, 32, 64, 128, 128, 1, False),    # eval mode (running statistics)
])
def test_sa_first_layer_vs_torch(B, N, S, ns, D, C, groups, train, geo):
    """geo: the round-4 form y = Pf[j] + b + Wx.(x_j - c_s) (sug_sa_first_geo_*); else P[j] - Q[s] (sug_sa_first_*)."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(B + N + C)
    xyz = torch.rand(B, N, 3, generator=g).cuda()
    points = torch.randn(B, N, D, generator=g).cuda() if D else None
    new_xyz = xyz[:, :S].clone()
    idx = torch.randint(0, N, (B, S, ns), generator=g, dtype=torch.int32).cuda()
    idx[:, :, ns // 2:] = idx[:, :, :1]                     # ball query pads short lists with the first index
    W = (torch.randn(C, 3 + D, generator=g) / (3 + D) ** 0.5).cuda()
    b = (torch.randn(C, generator=g) * 0.1).cuda()
    probe = torch.randn(B, S, ns, C, generator=g).cuda()

    def make_bn():
        bn = torch.nn.BatchNorm1d(C).cuda().train(train)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(-1.0, 1.5, C))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
            bn.running_mean.copy_(torch.linspace(-0.2, 0.2, C))
            bn.running_var.copy_(torch.linspace(0.5, 1.5, C))
        return bn

    bn_r, bn_k = make_bn(), make_bn()
    Wr, br = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    pr = points.clone().requires_grad_(True) if D else None
    ref = _reference(xyz, pr, new_xyz, idx, Wr, br, bn_r, groups)
    (ref * probe).sum().backward()

    Wk, bk = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    pk = points.clone().requires_grad_(True) if D else None
    with ops.bn_groups(groups):
        if geo:
            wx = Wk[:, :3].contiguous()
            Px = ops.linear_rows(xyz, wx)
            Pf = None if pk is None else ops.linear_rows(pk, Wk[:, 3:].contiguous())
            Q = ops.sub_row_bias(ops.linear_rows(new_xyz, wx), bk)
            out = ops.sa_first_layer_geo(Pf, Px, Q, idx, xyz, new_xyz, wx, bk, bn_k)
        else:
            P = ops.linear_rows(xyz if pk is None else torch.cat((xyz, pk), dim=-1), Wk)
            Q = ops.linear_rows(new_xyz, Wk[:, :3]) - bk
            out = ops.sa_first_layer(P, Q, idx, bn_k)
    (out * probe).sum().backward()

    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)

    def rel(a, b_):
        return float((a - b_).norm() / b_.norm().clamp_min(1e-12))

    assert rel(Wk.grad, Wr.grad) < 1e-3, 'dW %.3e' % rel(Wk.grad, Wr.grad)
    assert rel(bn_k.weight.grad, bn_r.weight.grad) < 1e-3
    assert rel(bn_k.bias.grad, bn_r.bias.grad) < 1e-3
    if D:
        assert rel(pk.grad, pr.grad) < 1e-3, 'dpoints %.3e' % rel(pk.grad, pr.grad)
    if train:       # the bias in front of a train-mode BatchNorm has zero gradient; both sides hold rounding noise
        lim = 1e-3 * float(probe.abs().sum() / C) + 1e-4
        assert float(bk.grad.abs().max()) <= lim and float(br.grad.abs().max()) <= lim
        torch.testing.assert_close(bn_k.running_mean, bn_r.running_mean, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(bn_k.running_var, bn_r.running_var, rtol=1e-4, atol=1e-5)
        assert int(bn_k.num_batches_tracked) == groups
    else:
        assert rel(bk.grad, br.grad) < 1e-3


def test_sa_first_layer_config3_shape_is_finite():
    """Config-3 sa2 shape (128 clouds x 128 centroids x 64 samples = 1M rows, 131 -> 128 channels): finite,
    forward bit-reproducible, statistics identities of the output (every channel's BN input has the batch
    mean / variance that the coefficients say)."""
    from sug_amd import ops
    B, N, S, ns, D, C = 128, 512, 128, 64, 128, 128
    g = torch.Generator().manual_seed(9)
    xyz = torch.rand(B, N, 3, generator=g).cuda()
    points = torch.randn(B, N, D, generator=g).cuda().requires_grad_(True)
    idx = torch.randint(0, N, (B, S, ns), generator=g, dtype=torch.int32).cuda()
    W = (torch.randn(C, 3 + D, generator=g) / 11).cuda().requires_grad_(True)
    b = torch.zeros(C).cuda().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    outs = []
    for _ in range(2):
        with ops.bn_groups(2):
            P = ops.linear_rows(torch.cat((xyz, points), dim=-1), W)
            Q = ops.linear_rows(xyz[:, :S].contiguous(), W[:, :3]) - b
            out = ops.sa_first_layer(P, Q, idx, bn)
        outs.append(out.detach())
    assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())
    out.square().sum().backward()
    assert bool(torch.isfinite(points.grad).all()) and bool(torch.isfinite(W.grad).all())
    # relu(BN(y)) with gamma = 1, beta = 0: per group and channel, mean of max(xhat, 0) of a unit-variance
    # variable lies in (0.2, 0.6) whatever its shape
    m = outs[0].view(2, -1, C).mean(dim=1)
    assert float(m.min()) > 0.2 and float(m.max()) < 0.6


def test_sa_first_forms_are_equally_accurate_against_fp64():
    """Does the cancellation in P[j] - Q[s] (two O(|W| |x|) numbers for a value of O(|W| r)) cost accuracy against the
    reference's grouped form W.(x_j - c_s)?  Measured on unit-scale clouds with 0.2 balls: no -- against fp64 the
    pre-activations of P - Q, of the geometric form (sug_sa_first_geo_*) and of the reference composition in fp32 agree to
    the same 7e-7 (6.6e-7 / 7.3e-7 / 7.6e-7).  Asserted: both kernel forms within 2x the reference composition's own fp32
    error."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(7)
    B, N, S, ns, C = 2, 2048, 256, 32, 64
    xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).cuda()
    new_xyz = xyz[:, :S].clone()
    off = (torch.rand(B, S, ns, 3, generator=g) * 0.4 - 0.2).cuda()          # neighbours inside a small ball
    idx = torch.randint(0, N, (B, S, ns), generator=g, dtype=torch.int32).cuda()
    bi = torch.arange(B, device='cuda').view(B, 1, 1)
    xyz = xyz.clone()
    xyz[bi.expand(B, S, ns).reshape(-1), idx.long().reshape(-1)] = (new_xyz.view(B, S, 1, 3) + off).reshape(-1, 3)
    W = torch.randn(C, 3, generator=g).cuda()
    b = (torch.randn(C, generator=g) * 0.1).cuda()
    bn = torch.nn.BatchNorm1d(C).cuda().eval()              # eval mode with unit statistics: z = relu(y) shows y itself
    with torch.no_grad():
        bn.running_var.fill_(1.0 - bn.eps)
        d64 = (xyz.double()[bi, idx.long()] - new_xyz.double().view(B, S, 1, 3)) @ W.double().t() + b.double()
        ref32 = (xyz[bi, idx.long()] - new_xyz.view(B, S, 1, 3)) @ W.t() + b
        P = ops.linear_rows(xyz, W)
        Q = ops.linear_rows(new_xyz, W) - b
        y_pq = ops.sa_first_layer(P, Q, idx, bn)
        y_geo = ops.sa_first_layer_geo(None, P, Q, idx, xyz, new_xyz, W, b, bn)
    pos = d64 > 0.05                                        # compare where the ReLU passes the value
    e = lambda y: float(((y.double() - d64)[pos]).abs().max())
    e_pq, e_geo, e_ref = e(y_pq), e(y_geo), e(ref32)
    print('max |y - fp64| on the passing entries: P - Q %.2e, geometric %.2e, reference composition in fp32 %.2e' % (e_pq, e_geo, e_ref))
    assert e_geo <= 2.0 * e_ref + 1e-7 and e_pq <= 2.0 * e_ref + 1e-7


@pytest.mark.parametrize('cfg', ['sa1_adapt', 'sa1_adapt_node_pass', 'sa2', 'sa2_eval'])
def test_middle_layer_folded_into_the_fused_last_layer_matches_separate_passes(cfg):
    """Round 5 (ops.bn_act_pointmlp_max, sug_pointmlp_max_layer_fwd_xf): the middle SA-MLP layer's BatchNorm + ReLU applied
    inside the fused last-layer kernel == the separate BatchNorm pass + fused last layer (SUG_SA_MID_FUSED=0): outputs, the
    SA-node features (adapt), every gradient and every BatchNorm buffer; paired domain groups; the node-pass form (last layer
    without autograd) and eval mode."""
    from sug_amd import ops
    from sug_amd.model.pointnet2_utils import PointNetSetAbstraction
    sa1 = cfg.startswith('sa1')
    B, N = 4, 1024 if sa1 else 512
    gen = torch.Generator().manual_seed(len(cfg))
    xyz = torch.rand(B, N, 3, generator=gen).cuda()
    pts = None if sa1 else (torch.randn(B, N, 128, generator=gen) * 0.5).cuda()
    adapt = sa1
    tail_grad = cfg != 'sa1_adapt_node_pass'
    train = cfg != 'sa2_eval'

    def run(fused):
        torch.manual_seed(1)
        sa = (PointNetSetAbstraction(256, 0.2, 32, 3, [64, 64, 128], False) if sa1
              else PointNetSetAbstraction(64, 0.4, 64, 128 + 3, [128, 128, 256], False)).cuda().train(train)
        with torch.no_grad():
            for bn in sa.mlp_bns:
                C = bn.num_features
                bn.weight.copy_(torch.linspace(-1.0, 1.5, C))
                bn.bias.copy_(torch.linspace(-0.3, 0.3, C))
                bn.running_mean.copy_(torch.linspace(-0.2, 0.2, C))
                bn.running_var.copy_(torch.linspace(0.5, 1.5, C))
        p = None if pts is None else pts.clone().requires_grad_(True)
        keep, ops.SA_MID_FUSED = ops.SA_MID_FUSED, fused
        try:
            torch.manual_seed(5)                       # FPS start draw
            with ops.bn_groups(2):
                r = sa.rows(xyz, p, adapt=adapt, tail_grad=tail_grad) if adapt else sa.rows(xyz, p)
        finally:
            ops.SA_MID_FUSED = keep
        outs = list(r[1:])
        gg = torch.Generator(device='cuda').manual_seed(2)
        loss = 0
        for o in outs:
            if o.requires_grad:
                loss = loss + (o * torch.randn(o.shape, device='cuda', generator=gg)).sum()
        grads = {}
        if torch.is_tensor(loss):
            loss.backward()
            grads = {k: v.grad.clone() for k, v in sa.named_parameters() if v.grad is not None}
            if p is not None and p.grad is not None:
                grads['points'] = p.grad.clone()
        bufs = {k: v.clone() for k, v in sa.named_buffers()}
        return [o.detach() for o in outs], grads, bufs

    o1, g1, b1 = run(True)
    o0, g0, b0 = run(False)
    assert len(o1) == len(o0) == (2 if adapt else 1)
    for a, b in zip(o1, o0):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), cfg
    assert set(g1) == set(g0), (sorted(g1), sorted(g0))
    if train and cfg != 'sa2_eval':
        assert g1, 'no gradients?'
    if cfg == 'sa1_adapt_node_pass':
        assert not any(k.startswith(('mlp_convs.2', 'mlp_bns.2')) for k in g1)      # the last layer ran without autograd
    gmax = max([float(v.norm()) for v in g0.values()] + [1e-30])
    for k in g0:
        d = float((g1[k] - g0[k]).norm()) / max(float(g0[k].norm()), 1e-3 * gmax)
        assert d <= 2e-4, (cfg, k, d)
    for k in b0:
        assert float((b1[k].double() - b0[k].double()).abs().max()) <= 1e-5 * max(1.0, float(b0[k].double().abs().max())), (cfg, k)

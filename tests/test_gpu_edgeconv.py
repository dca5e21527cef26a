"""GPU parity of the fused EdgeConv layer (GEMM split + gather/BN-stat/max kernel + exact
BN backward) against the oracle's k-expanded formulation.  fp32 tolerance 1e-4 (north star)."""
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _layer(C, Co, seed):
    from sug_amd.model.model_utils import conv_2d
    m = conv_2d(2 * C, Co, 1, activation='leakyrelu', bias=False)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = O.fill_params(shapes, seed)
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize('C,Co,N,k,train', [(3, 64, 256, 20, True), (64, 64, 256, 20, True),
                                            (64, 128, 200, 20, True), (128, 256, 128, 20, True),
                                            (64, 128, 256, 16, False), (8, 12, 100, 5, True)])
def test_edgeconv_forward_backward(C, Co, N, k, train):
    B = 3
    g = torch.Generator().manual_seed(C + Co + N)
    x = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, N, k), generator=g)
    probe = torch.randn(B, Co, N, generator=g)
    m, sd = _layer(C, Co, 5)
    m = m.cuda().train(train)

    # oracle (k-expanded: get_graph_feature -> conv -> BN -> LeakyReLU -> max)
    p = O.as_params({'c.' + kk: v for kk, v in sd.items()})
    xo = x.clone().requires_grad_(True)
    yo = O.conv_bn_act(p, 'c.', O.graph_feature(xo, k, idx), 'leakyrelu', train).max(dim=-1)[0]
    (yo * probe).sum().backward()

    xg = x.transpose(1, 2).contiguous().cuda().requires_grad_(True)
    yg = m.edge_rows(xg, idx.to(torch.int32).cuda())
    (yg * probe.transpose(1, 2).cuda()).sum().backward()

    torch.testing.assert_close(yg.detach().cpu().transpose(1, 2), yo.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xg.grad.cpu().transpose(1, 2), xo.grad, rtol=2e-4, atol=2e-4)
    gw = m.conv[0].weight.grad.cpu()
    torch.testing.assert_close(gw, p['c.conv.0.weight'].grad, rtol=2e-4, atol=2e-4 * float(p['c.conv.0.weight'].grad.abs().max()))
    torch.testing.assert_close(m.conv[1].weight.grad.cpu(), p['c.conv.1.weight'].grad, rtol=2e-4, atol=2e-3)
    torch.testing.assert_close(m.conv[1].bias.grad.cpu(), p['c.conv.1.bias'].grad, rtol=2e-4, atol=2e-3)
    if train:   # running statistics follow nn.BatchNorm2d (momentum 0.1, unbiased running var)
        torch.testing.assert_close(m.conv[1].running_mean.cpu(), p['c.conv.1.running_mean'], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(m.conv[1].running_var.cpu(), p['c.conv.1.running_var'], rtol=1e-5, atol=1e-6)


def test_edgeconv_is_deterministic():
    B, C, Co, N, k = 2, 64, 128, 512, 20
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, N, C, generator=g).cuda()
    idx = torch.randint(0, N, (B, N, k), generator=g, dtype=torch.int32).cuda()
    m, _ = _layer(C, Co, 6)
    m = m.cuda().train()
    outs = []
    for _ in range(2):
        xi = x.clone().requires_grad_(True)
        y = m.edge_rows(xi, idx)
        y.square().sum().backward()
        outs.append((y.detach().clone(), xi.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('C,rows_shape,slope,train', [(64, (3, 500), 0.0, True), (512, (2, 256), 0.01, True),
                                                      (12, (1000,), 1.0, True), (128, (4, 64), 0.0, False)])
def test_bn_act_rows_vs_torch(C, rows_shape, slope, train):
    """Own BN(+LeakyReLU/ReLU) on rows, forward/backward/running stats, vs torch's F.batch_norm (fp32 reference)."""
    import torch.nn.functional as F
    from sug_amd import ops
    g = torch.Generator().manual_seed(C)
    y = (torch.randn(*rows_shape, C, generator=g) * 2 + 0.5).cuda()
    probe = torch.randn(*rows_shape, C, generator=g).cuda()
    bn_a, bn_b = torch.nn.BatchNorm1d(C).cuda().train(train), torch.nn.BatchNorm1d(C).cuda().train(train)
    with torch.no_grad():
        w = 1 + 0.3 * torch.randn(C, generator=g)
        w[::5] = -w[::5]
        for bn in (bn_a, bn_b):
            bn.weight.copy_(w.cuda()); bn.bias.copy_(torch.randn(C, generator=g).cuda() if bn is bn_a else bn_a.bias)
            bn.running_mean.fill_(0.1); bn.running_var.fill_(1.3)
    ya, yb = y.clone().requires_grad_(True), y.clone().requires_grad_(True)
    out_a = ops.bn_act_rows(ya, bn_a, slope)
    ref = F.batch_norm(yb.reshape(-1, C), bn_b.running_mean, bn_b.running_var, bn_b.weight, bn_b.bias, train, 0.1, 1e-5)
    ref = F.leaky_relu(ref, slope).view_as(yb) if slope != 1.0 else ref.view_as(yb)
    (out_a * probe).sum().backward()
    (ref * probe).sum().backward()
    torch.testing.assert_close(out_a, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ya.grad, yb.grad, rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(bn_a.weight.grad, bn_b.weight.grad, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(bn_a.bias.grad, bn_b.bias.grad, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn_a.running_var, bn_b.running_var, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('B,N,C,train', [(3, 200, 64, True), (2, 1024, 512, True), (2, 100, 70, False)])
def test_bn_act_pool_vs_torch(B, N, C, train):
    """Fused bn5 -> leaky_relu(0.2) -> max|mean pooling vs the torch fp32 composition."""
    import torch.nn.functional as F
    from sug_amd import ops
    g = torch.Generator().manual_seed(N + C)
    y = (torch.randn(B, N, C, generator=g) + 0.3).cuda()
    pm, pa = torch.randn(B, C, generator=g).cuda(), torch.randn(B, C, generator=g).cuda()
    bn_a, bn_b = torch.nn.BatchNorm1d(C).cuda().train(train), torch.nn.BatchNorm1d(C).cuda().train(train)
    with torch.no_grad():
        w = 1 + 0.3 * torch.randn(C, generator=g)
        w[::4] = -w[::4]
        for bn in (bn_a, bn_b):
            bn.weight.copy_(w.cuda())
            bn.running_var.fill_(0.8)
    ya, yb = y.clone().requires_grad_(True), y.clone().requires_grad_(True)
    mx, mean = ops.bn_act_pool(ya, bn_a, 0.2)
    r = F.leaky_relu(F.batch_norm(yb.reshape(-1, C), bn_b.running_mean, bn_b.running_var, bn_b.weight, bn_b.bias,
                                  train, 0.1, 1e-5), 0.2).view(B, N, C)
    rmx, rmean = r.max(dim=1)[0], r.mean(dim=1)
    ((mx * pm).sum() + (mean * pa).sum()).backward()
    ((rmx * pm).sum() + (rmean * pa).sum()).backward()
    torch.testing.assert_close(mx, rmx, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(mean, rmean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ya.grad, yb.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(bn_a.weight.grad, bn_b.weight.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(bn_a.bias.grad, bn_b.bias.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(bn_a.running_var, bn_b.running_var, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('R,M,N', [(32768, 64, 64), (32768, 128, 3), (32768, 3, 64), (4096, 512, 128), (1000, 70, 33),
                                   (131072, 64, 131), (8192, 512, 512), (3000, 130, 259), (65536, 256, 128)])
def test_linear_dw_vs_torch(R, M, N):
    """Split-K MFMA weight-gradient kernel vs a torch fp64 reference; bit-reproducible."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(R + M + N)
    x = torch.randn(R, N, generator=g).cuda().requires_grad_(True)
    W = (torch.randn(M, N, generator=g) * 0.1).cuda().requires_grad_(True)
    b = torch.randn(M, generator=g).cuda().requires_grad_(True)
    probe = torch.randn(R, M, generator=g).cuda()
    (ops.linear_rows(x, W, b) * probe).sum().backward()
    ref = (probe.double().t() @ x.detach().double())
    torch.testing.assert_close(W.grad.double(), ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    torch.testing.assert_close(x.grad, probe @ W.detach(), rtol=1e-4, atol=1e-5)
    # bias gradient: column sums of g from the same launch (fp64 combine of per-chunk partials)
    refb = probe.double().sum(0)
    torch.testing.assert_close(b.grad.double(), refb, rtol=1e-5, atol=2e-6 * float(probe.abs().sum(0).max()))
    g1, b1 = W.grad.clone(), b.grad.clone()
    W.grad, b.grad = None, None
    (ops.linear_rows(x, W, b) * probe).sum().backward()
    assert torch.equal(g1, W.grad) and torch.equal(b1, b.grad)


class _BN:
    """Minimal stand-in for an nn.BatchNorm module (parameters + running buffers)."""

    def __init__(self, C, seed):
        g = torch.Generator().manual_seed(seed)
        self.weight = (torch.randn(C, generator=g)).cuda().requires_grad_(True)
        self.bias = (torch.randn(C, generator=g) * 0.1).cuda().requires_grad_(True)
        self.running_mean = torch.zeros(C).cuda()
        self.running_var = torch.ones(C).cuda()
        self.num_batches_tracked = torch.zeros((), dtype=torch.long).cuda()
        self.training, self.eps, self.momentum = True, 1e-5, 0.1


def _grouped_vs_separate(run, x, C, seed):
    """run(x_part, bn) -> tuple of outputs.  Grouped (bn_groups(2)) must equal two calls bit for bit."""
    from sug_amd import ops
    res = []
    for grouped in (False, True):
        bn = _BN(C, seed)
        xs = x.clone().requires_grad_(True)
        if grouped:
            with ops.bn_groups(2):
                outs = run(xs, bn)
        else:
            h = xs.shape[0] // 2
            o1, o2 = run(xs[:h], bn), run(xs[h:], bn)
            outs = tuple(torch.cat((a, b), 0) for a, b in zip(o1, o2))
        probe = [torch.randn(o.shape, generator=torch.Generator().manual_seed(5 + i)).cuda() for i, o in enumerate(outs)]
        sum((o * p).sum() for o, p in zip(outs, probe)).backward()
        res.append(([o.detach() for o in outs], xs.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var,
                    bn.num_batches_tracked))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][1], res[1][1]), float((res[0][1] - res[1][1]).abs().max())
    for i in (2, 3):       # dgamma / dbeta: the two group sums are added in fp64 instead of fp32
        torch.testing.assert_close(res[0][i], res[1][i], rtol=1e-5, atol=1e-5)
    for i in (4, 5, 6):
        assert torch.equal(res[0][i], res[1][i])


@pytest.mark.parametrize('shape', [(8, 1, 128, 1024), (8, 256, 64), (4, 32, 16, 259)])
def test_bn_act_rows_groups(shape):
    from sug_amd import ops
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1)).cuda()
    _grouped_vs_separate(lambda t, bn: (ops.bn_act_rows(t, bn, 0.0),), x, shape[-1], 3)


def test_bn_act_pool_groups():
    from sug_amd import ops
    x = torch.randn(8, 200, 512, generator=torch.Generator().manual_seed(2)).cuda()
    _grouped_vs_separate(lambda t, bn: ops.bn_act_pool(t, bn, 0.2), x, 512, 4)


def test_edgeconv_groups():
    from sug_amd import ops
    B, N, k, Co = 6, 256, 20, 64
    g = torch.Generator().manual_seed(3)
    pq = torch.randn(B, N, 2 * Co, generator=g).cuda()
    idx = torch.randint(0, N, (B, N, k), generator=g).int().cuda()

    # separate calls see views of one clone: tell the halves apart by address
    class _Run:
        def __call__(self, t, bn):
            off = (t.data_ptr() - t._base.data_ptr()) // (t.element_size() * N * 2 * Co) if t._base is not None else 0
            out, _ = ops.edgeconv_bn_act_max(t, idx[off:off + t.shape[0]], bn.weight, bn.bias, bn.running_mean,
                                             bn.running_var, True, 0.01)
            return (out,)

    _grouped_vs_separate(_Run(), pq, Co, 5)


@pytest.mark.parametrize('rows,C,groups', [(32768, 64, 2), (4096, 512, 1), (1000, 7, 1)])
def test_bn_rows_large_offset(rows, C, groups):
    """Batch statistics of data whose mean dwarfs its spread (y = 100 + 0.01*randn: E[x^2] - mean^2 in fp32 partial
    sums would return noise): the partial sums are taken about a pivot row and combined in fp64, so the normalised
    output matches an fp64 BatchNorm."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    y = (100.0 + 0.01 * torch.randn(rows, C, generator=g)).cuda()
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with ops.bn_groups(groups):
        out = ops.bn_act_rows(y, bn, 1.0)                       # slope 1: no activation
    refs = []
    for yg in y.double().chunk(groups, dim=0):
        m, v = yg.mean(0), yg.var(0, unbiased=False)
        refs.append((yg - m) / torch.sqrt(v + bn.eps))
    ref = torch.cat(refs).float()
    # x - mean carries the fp32 rounding of x itself (ulp(100) = 7.6e-6 against a spread of 0.01): 1e-3 of a unit-variance output
    assert float((out - ref).abs().max()) < 5e-3, float((out - ref).abs().max())
    assert abs(float(out.var(0, unbiased=False).mean()) - 1e-4 / (1e-4 + bn.eps)) < 2e-3      # var / (var + eps)
    rv = torch.ones(C, dtype=torch.float64)
    for yg in y.double().chunk(groups, dim=0):
        rv = 0.9 * rv + 0.1 * yg.var(0, unbiased=True).cpu()
    torch.testing.assert_close(bn.running_var.cpu().double(), rv, rtol=1e-3, atol=1e-7)


def test_pointmlp_and_sa_first_large_offset():
    """The same large-offset check for the fused layers that form their BatchNorm sums inside the kernel:
    the per-point MLP + max (bias 100, spread ~1) and the set-abstraction first layer (P - Q about 100).
    (EdgeConv keeps plain sums: its y = P[idx] + Q comes from bias-free convolutions of normalised activations.)"""
    from sug_amd import ops
    g = torch.Generator().manual_seed(12)
    R, K, C2, seg = 8192, 64, 128, 32
    x = torch.randn(R, K, generator=g).cuda()
    W = (torch.randn(C2, K, generator=g) / 8).cuda()
    b = torch.full((C2,), 100.0).cuda()
    bn = torch.nn.BatchNorm1d(C2).cuda().train()
    out = ops.pointmlp_max(x, W, b, bn, 0.0, seg)
    yd = x.double() @ W.double().t() + 100.0
    u = (yd - yd.mean(0)) / torch.sqrt(yd.var(0, unbiased=False) + bn.eps)
    ref = torch.relu(u).view(-1, seg, C2).max(dim=1)[0].float()
    assert float((out - ref).abs().max()) < 2e-4, float((out - ref).abs().max())

    B, N, S, ns, C = 4, 256, 64, 32, 64
    P = (100.0 + 0.01 * torch.randn(B, N, C, generator=g)).cuda()
    Q = (0.01 * torch.randn(B, S, C, generator=g)).cuda()
    idx = torch.randint(0, N, (B, S, ns), generator=g, dtype=torch.int32).cuda()
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with ops.bn_groups(2):
        z = ops.sa_first_layer(P, Q, idx, bn)
    bi = torch.arange(B, device='cuda').view(B, 1, 1)
    y = (P[bi, idx.long()] - Q.unsqueeze(2)).double()
    refs = []
    for yg in y.chunk(2, dim=0):
        m, v = yg.mean((0, 1, 2)), yg.var((0, 1, 2), unbiased=False)
        refs.append(torch.relu((yg - m) / torch.sqrt(v + bn.eps)))
    ref = torch.cat(refs).float()
    assert float((z - ref).abs().max()) < 1e-2, float((z - ref).abs().max())           # input rounding: ulp(100) / 0.014


@pytest.mark.parametrize('C,Co,N,B,G', [(3, 64, 1024, 8, 2), (64, 128, 256, 4, 2), (128, 256, 160, 2, 1), (64, 64, 1000, 3, 1)])
def test_edgeconv_fused_matches_gemm_path(C, Co, N, B, G):
    """The layer with the GEMM inside the gather kernel (sug_edgeconv_fused_layer_fwd, the default) against the
    library-GEMM + gather path (SUG_EDGECONV_FUSED=0) on the same inputs: activations, BatchNorm coefficients /
    running buffers per domain group, and every gradient (x, conv weight, conv bias, gamma, beta) -- with a conv bias,
    mixed-sign gains, a destination slice of a wider buffer and a cloud size that is not a multiple of the tile."""
    from sug_amd import ops
    from sug_amd.model.model_utils import conv_2d
    k = 20
    g = torch.Generator().manual_seed(C * 7 + Co + N)
    x = (torch.randn(B, N, C, generator=g) * 0.8).cuda()
    idx = torch.randint(0, N, (B, N, k), generator=g, dtype=torch.int32).cuda()
    probe = torch.randn(B, N, Co, generator=g).cuda()
    res = []
    for fused in (False, True):
        m = conv_2d(2 * C, Co, 1, activation='leakyrelu', bias=True)
        m.load_state_dict(O.fill_params({kk: tuple(v.shape) for kk, v in m.state_dict().items()}, 9))
        m = m.cuda().train()
        keep, keepc = ops.EDGECONV_FUSED, ops.EDGECONV_FUSED_MAXC
        ops.EDGECONV_FUSED, ops.EDGECONV_FUSED_MAXC = fused, 128          # (the default fuses Cin <= 64 only)
        try:
            assert bool(ops.edgeconv_fused_supported(N, k, C, Co)) == fused
            xi = x.clone().requires_grad_(True)
            wide = torch.zeros(B, N, Co + 64, device='cuda')
            with ops.bn_groups(G):
                y, coef = m.edge_rows(xi, idx, return_stats=True, out=wide[:, :, 32:32 + Co])
            assert y.data_ptr() == wide[:, :, 32:32 + Co].data_ptr()
            (y * probe).sum().backward()
        finally:
            ops.EDGECONV_FUSED, ops.EDGECONV_FUSED_MAXC = keep, keepc
        assert float(wide[:, :, :32].abs().max()) == 0 and float(wide[:, :, 32 + Co:].abs().max()) == 0
        res.append(dict(y=y.detach().clone(), coef=coef.clone(), gx=xi.grad.clone(), gw=m.conv[0].weight.grad.clone(),
                        gb=m.conv[0].bias.grad.clone(), gg=m.conv[1].weight.grad.clone(), gbeta=m.conv[1].bias.grad.clone(),
                        rm=m.conv[1].running_mean.clone(), rv=m.conv[1].running_var.clone()))
    a, b = res
    torch.testing.assert_close(b['y'], a['y'], rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(b['coef'], a['coef'], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(b['rm'], a['rm'], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(b['rv'], a['rv'], rtol=1e-5, atol=1e-6)
    for key in ('gx', 'gw', 'gg', 'gbeta'):
        rel = float((a[key] - b[key]).norm() / (a[key].norm() + 1e-12))
        assert rel < 1e-4, (key, rel)
    # a conv bias in front of train-mode BatchNorm has zero gradient: both paths return rounding noise only
    assert float(b['gb'].abs().max()) <= 1e-3 * float(b['gw'].abs().max()) + 1e-5


@pytest.mark.parametrize('C,Co', [(64, 64), (128, 256)])
def test_edgeconv_large_offset(C, Co):
    """EdgeConv BatchNorm statistics when the mean of y = W.[x_j - x_i ; x_i] dwarfs its spread (y ~ 100 +- 0.05): the
    sums inside the gather kernels (the fused layer for C = 64, the library-GEMM path for C = 128) are taken about a
    pivot, so the normalised activations match an fp64 evaluation of the layer."""
    from sug_amd import ops
    from sug_amd.model.model_utils import conv_2d
    B, N, k = 4, 256, 20
    g = torch.Generator().manual_seed(C + Co)
    x = (5.0 + 0.01 * torch.randn(B, N, C, generator=g)).cuda()
    idx = torch.randint(0, N, (B, N, k), generator=g, dtype=torch.int32).cuda()
    m = conv_2d(2 * C, Co, 1, activation='leakyrelu', bias=False).cuda().train()
    with torch.no_grad():
        w = torch.randn(Co, 2 * C, generator=g) * 0.3
        w[:, C:] += 20.0 / C                                   # the x_i half: W2 . x_i ~ 100
        m.conv[0].weight.copy_(w.view(Co, 2 * C, 1, 1).cuda())
        gam = 1 + 0.3 * torch.randn(Co, generator=g)
        gam[::3] = -gam[::3]
        m.conv[1].weight.copy_(gam.cuda())
    with ops.bn_groups(2), torch.no_grad():
        out = m.edge_rows(x, idx)
    xd, wd = x.double(), m.conv[0].weight.view(Co, 2 * C).double()
    bi = torch.arange(B, device='cuda').view(B, 1, 1)
    nbr = xd[bi, idx.long()]                                                          # [B,N,k,C]
    y = torch.cat((nbr - xd.unsqueeze(2), xd.unsqueeze(2).expand_as(nbr)), -1) @ wd.t()   # [B,N,k,Co]
    assert float(y.mean().abs()) > 50 and float(y.std(dim=(0, 1, 2)).mean()) < 1.0
    refs = []
    for yg in y.chunk(2, dim=0):
        mu, var = yg.mean((0, 1, 2)), yg.var((0, 1, 2), unbiased=False)
        u = (yg - mu) / torch.sqrt(var + m.conv[1].eps) * m.conv[1].weight.double() + m.conv[1].bias.double()
        refs.append(torch.nn.functional.leaky_relu(u, 0.01).max(dim=2)[0])
    ref = torch.cat(refs).float()
    err = float((out - ref).abs().max())
    print('max deviation from the fp64 layer: %.3e' % err)
    assert err < 5e-3, err                  # input rounding: ulp(100) = 7.6e-6 against a spread of ~0.05

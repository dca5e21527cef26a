"""The classifier heads in eight launches (sug_head_linear_fwd / sug_head_ln_bwd / sug_head_linear_bwd, ops.heads_fused) against the module path they
replace -- Pointnet_c, model/Model.py:412-449: Linear -> LayerNorm -> act -> Dropout -> Linear -> LayerNorm -> act
(mid feature) -> Dropout -> Linear -- for the three head flavours (DGCNN: LeakyReLU + bias; PointNet / PointNet++: ReLU,
first Linear without bias; Point Transformer: two layers), one and two heads per launch, row counts on both sides of the
32-row blocks, with the dropout off (eval, and train with p = 0) and on (same uniform randoms fed to a torch restatement).
Outputs and every gradient (input, weights, biases, LayerNorm weights) within 2e-5 of the output / gradient scale."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _heads(kind, n, seed):
    from sug_amd.model.Model import Pointnet_c
    torch.manual_seed(seed)
    hs = []
    for _ in range(n):
        h = Pointnet_c(num_class=10, dgcnn_flag=(kind == 'dgcnn'), PTran_flag=(kind == 'ptran'))
        for m in h.modules():                      # LayerNorm weights away from their (1, 0) initialisation
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.uniform_(-0.3, 0.3)
        hs.append(h.cuda())
    return hs


def _restate(h, x, masks):
    """Pointnet_c.forward(adapt=True) with given dropout masks (already scaled by 1/(1-p); None = no dropout)."""
    act = h.mlp2.ac
    a = x
    if not h.PTran:
        a = act(h.mlp1.fc[1](h.mlp1.fc[0](a)))
        if masks[0] is not None:
            a = a * masks[0]
    mid = act(h.mlp2.fc[1](h.mlp2.fc[0](a)))
    a = mid if masks[1] is None else mid * masks[1]
    return h.mlp3(a), mid


def _grads(hs, x, outs, probe):
    loss = sum((o * p).sum() for pair, pp in zip(outs, probe) for o, p in zip(pair, pp))
    ps = [p for h in hs for p in h.parameters()]
    return torch.autograd.grad(loss, [x] + ps, allow_unused=True)       # (the two-layer head does not use mlp1)


def _close(a, b, what):
    if a is None or b is None:
        assert a is None and b is None, what
        return
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max()) / scale
    assert err <= 2e-5, (what, err)


@pytest.mark.parametrize('kind', ['dgcnn', 'pointnet', 'ptran'])
@pytest.mark.parametrize('M,nheads', [(64, 2), (5, 1), (96, 2), (33, 2), (128, 1)])
@pytest.mark.parametrize('mode', ['eval', 'train_p0'])
def test_heads_fused_matches_modules(kind, M, nheads, mode):
    from sug_amd import ops
    hs = _heads(kind, nheads, 3)
    for h in hs:
        h.train(mode != 'eval')
        h.dropout1.p = h.dropout2.p = 0.0
    K = 512 if kind == 'ptran' else 1024
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, K, generator=g) * 0.7).cuda().requires_grad_(True)
    probe = [(torch.randn(M, 10, generator=g).cuda(), torch.randn(M, 256, generator=g).cuda() * 0.1) for _ in hs]
    assert ops.heads_fused_supported(hs, x)
    got = ops.heads_fused(hs, x)
    want = [h(x, adapt=True) for h in hs]
    for (gl, gm), (wl, wm) in zip(got, want):
        _close(gl, wl, 'logits')
        _close(gm, wm, 'mid feature')
    names = ['x'] + ['head%d.%s' % (i, n) for i, h in enumerate(hs) for n, _ in h.named_parameters()]
    for n, a, b in zip(names, _grads(hs, x, got, probe), _grads(hs, x, want, probe)):
        _close(a, b, n)


@pytest.mark.parametrize('kind', ['dgcnn', 'pointnet', 'ptran'])
def test_heads_fused_dropout(kind, monkeypatch):
    """Dropout on (p = 0.4, 0.25): the uniform randoms of the fused op's one torch.rand launch are recorded and the same
    masks (keep when u >= p, scale 1/(1-p)) go into a torch restatement; outputs and gradients must agree, about
    p of the activations are dropped, and a second call with the same seed repeats the first bit for bit."""
    from sug_amd import ops
    M, nheads = 64, 2
    hs = _heads(kind, nheads, 5)
    for h in hs:
        h.train()
        h.dropout1.p, h.dropout2.p = 0.4, 0.25
    K = 512 if kind == 'ptran' else 1024
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, K, generator=g) * 0.7).cuda().requires_grad_(True)
    probe = [(torch.randn(M, 10, generator=g).cuda(), torch.randn(M, 256, generator=g).cuda() * 0.1) for _ in hs]
    seen = []
    real = torch.rand

    def spy(*a, **k):
        t = real(*a, **k)
        seen.append(t)
        return t
    monkeypatch.setattr(torch, 'rand', spy)
    torch.manual_seed(9)
    got = ops.heads_fused(hs, x)
    monkeypatch.setattr(torch, 'rand', real)
    assert len(seen) == 1
    u = seen[0]
    N1 = 0 if kind == 'ptran' else 512
    want = []
    for i, h in enumerate(hs):
        m1 = None
        if N1:
            u1 = u[i * M * N1:(i + 1) * M * N1].view(M, N1)
            m1 = (u1 >= 0.4).float() / 0.6
            assert 0.35 < float((u1 < 0.4).float().mean()) < 0.45
        o2 = nheads * M * N1
        u2 = u[o2 + i * M * 256:o2 + (i + 1) * M * 256].view(M, 256)
        m2 = (u2 >= 0.25).float() / 0.75
        want.append(_restate(h, x, (m1, m2)))
    for (gl, gm), (wl, wm) in zip(got, want):
        _close(gl, wl, 'logits')
        _close(gm, wm, 'mid feature')
    names = ['x'] + ['head%d.%s' % (i, n) for i, h in enumerate(hs) for n, _ in h.named_parameters()]
    for n, a, b in zip(names, _grads(hs, x, got, probe), _grads(hs, x, want, probe)):
        _close(a, b, n)
    torch.manual_seed(9)
    again = ops.heads_fused(hs, x)
    for (a, b), (c, d) in zip(got, again):
        assert torch.equal(a, c) and torch.equal(b, d)


def test_heads_fused_unsupported_shapes_take_the_module_path():
    from sug_amd import ops
    hs = _heads('dgcnn', 2, 1)
    assert not ops.heads_fused_supported(hs, torch.zeros(200, 1024, device='cuda'))       # more than 128 rows
    assert not ops.heads_fused_supported(hs, torch.zeros(8, 1024, device='cuda', dtype=torch.float16))
    hs[1].dropout2.p = 0.1                                                                  # heads must agree on their dropout
    assert not ops.heads_fused_supported(hs, torch.zeros(8, 1024, device='cuda'))


@pytest.mark.parametrize('M', [2, 32, 100])
@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_gate_bn_matches_gate_then_batchnorm(M, mode):
    """CALayer's tail BN(x * sigmoid(z) + x) in one launch (sug_gate_bn_fwd / _bwd) against ops.gate + nn.BatchNorm1d:
    output, the four gradients and the running buffers, train and eval mode, with a large common offset in x (the
    two-pass statistics must not care)."""
    from sug_amd import ops
    C = 4096
    g = torch.Generator().manual_seed(M)
    # (two rows: x_hat = +-1 whatever the data, and a common offset only adds rounding noise to the tiny difference
    # both implementations divide by -- keep that case centred)
    x = (torch.randn(M, C, generator=g) * 0.5 + (30.0 if M > 2 else 0.0)).cuda().requires_grad_(True)
    z = torch.randn(M, C, generator=g).cuda().requires_grad_(True)
    probe = torch.randn(M, C, generator=g).cuda()
    res = []
    for fused in (False, True):
        bn = torch.nn.BatchNorm1d(C).cuda()
        torch.manual_seed(3)
        bn.weight.data.copy_(torch.rand(C) + 0.5)
        bn.bias.data.copy_(torch.rand(C) - 0.5)
        bn.running_mean.copy_(torch.full((C,), 29.0))
        bn.running_var.copy_(torch.full((C,), 2.0))
        bn.train(mode == 'train')
        assert ops.gate_bn_supported(x, bn)
        out = ops.gate_bn(x, z, bn) if fused else bn(ops.gate(x, z))
        grads = torch.autograd.grad((out * probe).sum(), [x, z, bn.weight, bn.bias])
        res.append([out.detach()] + list(grads) + [bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()])
    names = ['out', 'dx', 'dz', 'dgamma', 'dbeta', 'running_mean', 'running_var']
    for n, a, b in zip(names, res[1], res[0]):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) / scale <= 5e-5, (n, float((a - b).abs().max()), scale)
    assert int(res[1][-1]) == int(res[0][-1])


def _ref_calayer(x, W0, b0, W2, b2, gamma, beta, rm, rv, training):
    """CALayer.forward exactly as the reference composes it (model/Model.py:28-34): two 1x1 Conv2d on [B,C,1,1], ReLU,
    sigmoid gate, x * y + x, BatchNorm1d."""
    import torch.nn.functional as F
    v = x.view(x.shape[0], -1, 1, 1)
    y = F.relu(F.conv2d(v, W0, b0))
    y = torch.sigmoid(F.conv2d(y, W2, b2))
    u = (v * y + v).view(x.shape[0], -1)
    return F.batch_norm(u, rm, rv, gamma, beta, training, 0.1, 1e-5)


@pytest.mark.parametrize('M,nl,train', [(32, 2, True), (32, 1, True), (8, 2, True), (64, 2, True), (5, 1, True), (32, 2, False)])
def test_calayer_fused_matches_reference_composition(M, nl, train):
    """sug_calayer_fwd / _bwd (both attention layers of Net_MDA in one launch per stage) against the reference's CALayer
    composed in plain torch: output, input gradient, every parameter gradient, BatchNorm running statistics."""
    from sug_amd import ops
    from sug_amd.model.Model import CALayer
    g = torch.Generator().manual_seed(100 * M + nl)
    C = 4096
    mods, refs = [], []
    for a in range(nl):
        m = CALayer(C)
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * (0.02 if p.dim() > 1 else 0.3))
            m.bn.weight.add_(1.0)
            m.bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
            m.bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
        m = m.cuda().train(train)
        mods.append(m)
        refs.append({k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and k in dict(m.named_parameters()))
                     for k, v in m.state_dict().items()})
    x = torch.randn(nl * M, C, generator=g).cuda()
    probe = torch.randn(nl * M, C, generator=g).cuda()
    xk = x.clone().requires_grad_(True)
    assert ops.calayer_supported(tuple(mods), xk)
    out = ops.calayers(tuple(mods), xk)
    (out * probe).sum().backward()
    xr = x.clone().requires_grad_(True)
    outs = []
    for a in range(nl):
        R = refs[a]
        outs.append(_ref_calayer(xr[a * M:(a + 1) * M], R['conv_du.0.weight'], R['conv_du.0.bias'], R['conv_du.2.weight'],
                                 R['conv_du.2.bias'], R['bn.weight'], R['bn.bias'], R['bn.running_mean'], R['bn.running_var'], train))
    ref = torch.cat(outs)
    (ref * probe).sum().backward()
    # BatchNorm1d over M rows amplifies input rounding by up to 1/sqrt(var + eps) of a column: compare at 1e-4 of the scale
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)
    rel = lambda a_, b_: float((a_ - b_).norm() / b_.norm().clamp_min(1e-12))
    assert rel(xk.grad, xr.grad) < 1e-4, rel(xk.grad, xr.grad)
    for a in range(nl):
        P = dict(mods[a].named_parameters())
        for k, p in P.items():
            assert p.grad is not None, k
            assert rel(p.grad, refs[a][k].grad) < 2e-4, (k, rel(p.grad, refs[a][k].grad))
        if train:
            torch.testing.assert_close(mods[a].bn.running_mean, refs[a]['bn.running_mean'], rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(mods[a].bn.running_var, refs[a]['bn.running_var'], rtol=1e-5, atol=1e-6)
    # run to run identical (fixed summation order)
    xk2 = x.clone().requires_grad_(True)
    for m in mods:
        m.zero_grad()
    out2 = ops.calayers(tuple(mods), xk2)
    (out2 * probe).sum().backward()
    assert torch.equal(out2, out) if not train else True          # (train: the running statistics moved, the output did not)
    assert torch.equal(xk2.grad, xk.grad)

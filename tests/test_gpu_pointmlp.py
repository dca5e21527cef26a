"""GPU parity of the fused per-point MLP + max kernel (sug_pointmlp_max_*) against the operator the
reference composes (model/Model.py:274-279, model/model_utils.py:72-79, model/pointnet2_utils.py:193-207):
1x1 conv (+bias) -> train-mode BatchNorm -> ReLU -> max over the points of a segment, written in
plain fp32 torch with the full [rows, Co] tensor.  Tolerance 1e-4 forward (north star)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(x, W, b, bn, seg, slope, groups):
    outs = []
    for xg in x.chunk(groups, dim=0):
        y = xg @ W.t()
        if b is not None:
            y = y + b
        z = torch.nn.functional.leaky_relu(bn(y), slope)
        outs.append(z.view(-1, seg, z.shape[1]).max(dim=1)[0])
    return torch.cat(outs)


@pytest.mark.parametrize('B,seg,K,Co,groups,bias,train', [
    (4, 1024, 128, 1024, 1, True, True),      # PointNet conv5 / T-Net conv2d3 + global max
    (4, 1024, 128, 1024, 2, True, True),      # paired domains: BatchNorm per half
    (96, 32, 64, 128, 1, True, True),         # PointNet++ sa1 last layer, nsample = 32
    (40, 64, 128, 256, 2, True, True),        # sa2 last layer, nsample = 64
    (3, 96, 64, 128, 1, False, True),         # segment of 3 tiles, no bias, odd segment count
    (4, 1024, 128, 1024, 1, True, False),     # eval mode (running statistics)
])
def test_pointmlp_max_vs_torch(B, seg, K, Co, groups, bias, train):
    from sug_amd import ops
    g = torch.Generator().manual_seed(B + seg + Co)
    x = torch.randn(B * seg, K, generator=g).cuda()
    W = (torch.randn(Co, K, generator=g) / K ** 0.5).cuda()
    b = (torch.randn(Co, generator=g) * 0.1).cuda() if bias else None
    gam = torch.randn(Co, generator=g).cuda()              # mixed signs: max and min channels
    bet = (torch.randn(Co, generator=g) * 0.2).cuda()
    probe = torch.randn(B, Co, generator=g).cuda()

    def make_bn():
        bn = torch.nn.BatchNorm1d(Co).cuda().train(train)
        with torch.no_grad():
            bn.weight.copy_(gam)
            bn.bias.copy_(bet)
            bn.running_mean.copy_(torch.linspace(-0.2, 0.2, Co))
            bn.running_var.copy_(torch.linspace(0.5, 1.5, Co))
        return bn

    bn_r, bn_k = make_bn(), make_bn()
    xr, Wr = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    ref = _reference(xr, Wr, br, bn_r, seg, 0.0, groups)
    (ref * probe).sum().backward()

    xk, Wk = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    bk = b.clone().requires_grad_(True) if bias else None
    with ops.bn_groups(groups):
        out = ops.pointmlp_max(xk, Wk, bk, bn_k, 0.0, seg)
    (out * probe).sum().backward()

    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)

    def rel(a, b_):
        return float((a - b_).norm() / b_.norm().clamp_min(1e-12))

    assert rel(xk.grad, xr.grad) < 1e-3, 'dx %.3e' % rel(xk.grad, xr.grad)
    assert rel(Wk.grad, Wr.grad) < 1e-3, 'dW %.3e' % rel(Wk.grad, Wr.grad)
    assert rel(bn_k.weight.grad, bn_r.weight.grad) < 1e-3
    assert rel(bn_k.bias.grad, bn_r.bias.grad) < 1e-3
    if bias:
        if train:       # zero in exact arithmetic; the reference's value is rounding noise
            assert float(bk.grad.abs().max()) <= 1e-3 * float(probe.abs().sum() / Co) + 1e-6
            assert float(br.grad.abs().max()) <= 1e-3 * float(probe.abs().sum() / Co) + 1e-4
        else:
            assert rel(bk.grad, br.grad) < 1e-3
    if train:
        torch.testing.assert_close(bn_k.running_mean, bn_r.running_mean, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(bn_k.running_var, bn_r.running_var, rtol=1e-4, atol=1e-5)
        assert int(bn_k.num_batches_tracked) == groups


def test_pointmlp_max_is_deterministic_and_full_size():
    """Config-3 sa1 shape (64 clouds x 512 groups x 32 samples = 1M rows, 64 -> 128 channels) and the
    PointNet conv5 shape at 64 clouds: bit-reproducible forward and backward, finite, and equal to the
    unfused composition within 1e-4."""
    from sug_amd import ops
    for B, seg, K, Co in ((64 * 512, 32, 64, 128), (64, 1024, 128, 1024)):
        g = torch.Generator().manual_seed(seg)
        x = torch.randn(B * seg, K, generator=g).cuda()
        W = (torch.randn(Co, K, generator=g) / K ** 0.5).cuda()
        b = (torch.randn(Co, generator=g) * 0.1).cuda()
        probe = torch.randn(B, Co, generator=g).cuda()
        res = []
        for _ in range(2):
            bn = torch.nn.BatchNorm1d(Co).cuda().train()
            with torch.no_grad():
                bn.weight.copy_(torch.linspace(-1, 1, Co))
            xi, Wi = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
            out = ops.pointmlp_max(xi, Wi, b, bn, 0.0, seg)
            (out * probe).sum().backward()
            res.append((out.detach(), xi.grad, Wi.grad, bn.weight.grad, bn.running_var.clone()))
        for u, v in zip(res[0], res[1]):
            assert torch.equal(u, v)
        assert all(bool(torch.isfinite(t).all()) for t in res[0])
        bn = torch.nn.BatchNorm1d(Co).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(-1, 1, Co))
        ref = _reference(x, W, b, bn, seg, 0.0, 1)
        torch.testing.assert_close(res[0][0], ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('R,K,Co', [(65536, 128, 512), (65536, 64, 256), (5000, 64, 128), (4100, 128, 1024)])
def test_rows_gemm_vs_torch(R, K, Co):
    """sug_rows_gemm (the per-point linear layer without a following reduction, e.g. EdgeConv's PQ operand):
    an fp32 fma chain over k, compared with torch's fp32 GEMM and with an fp64 product."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(R + Co)
    x = torch.randn(R, K, generator=g).cuda()
    W = (torch.randn(Co, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda()
    keep, ops.OWN_ROWS_GEMM = ops.OWN_ROWS_GEMM, True           # opt-in path (the library GEMM is the default)
    try:
        _rows_gemm_checks(ops, x, W, b, R, K)
    finally:
        ops.OWN_ROWS_GEMM = keep


def _rows_gemm_checks(ops, x, W, b, R, K):
    y = ops.linear_rows(x, W, b)
    ref64 = (x.double() @ W.double().t() + b.double())
    assert float((y.double() - ref64).abs().max()) < 2e-5
    wide = torch.zeros(R, K + 8, device='cuda')
    wide[:, 4:4 + K] = x                           # a column slice of a wider buffer (row stride K + 8)
    y2 = ops.linear_rows(wide[:, 4:4 + K], W, None)
    assert torch.equal(y2 + b, y) or float((y2 + b - y).abs().max()) < 1e-6
    # gradients still flow through the library / sug_linear_dw backward
    xr, Wr = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    ops.linear_rows(xr, Wr, b).square().sum().backward()
    xt, Wt = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    torch.nn.functional.linear(xt, Wt, b).square().sum().backward()
    assert float((xr.grad - xt.grad).norm() / xt.grad.norm()) < 1e-5
    assert float((Wr.grad - Wt.grad).norm() / Wt.grad.norm()) < 1e-5


def test_pointmlp_max_split_segments_match_whole_segments_bit_for_bit():
    """Small grids split every segment over several workgroups (config 1: 16 clouds -> 4 parts per 1024-point segment,
    sug_amd/csrc/pointmlp.hip) and combine the partial extremes in part order.  The extreme and its row of a segment do
    not depend on the other segments of the launch, so the first 16 segments of a 64-segment launch (512 workgroups: no
    split) must equal the 16-segment launch (split) bit for bit -- value AND arg (first extreme in ascending rows)."""
    import ctypes
    from sug_amd._lib import lib
    L = lib()
    K, Co, seg = 128, 1024, 1024
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64 * seg, K, generator=g).cuda()
    x[5 * seg + 700] = x[5 * seg + 3]                       # duplicated rows: ties across parts -> the earlier row must win
    x[9 * seg + 1023] = x[9 * seg + 0]
    W = (torch.randn(Co, K, generator=g) / K ** 0.5).cuda()
    b = (torch.randn(Co, generator=g) * 0.1).cuda()
    gam = torch.randn(Co, generator=g).cuda()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = []
    for S in (64, 16):
        zext = torch.empty(S, Co, device='cuda')
        arg = torch.empty(S, Co, dtype=torch.int32, device='cuda')
        ws = torch.zeros(1024 * 2 * Co, device='cuda')
        nblk = ctypes.c_int(0)
        rc = L.sug_pointmlp_max_fwd(p(x), K, S * seg, K, p(W), p(b), p(gam), Co, seg, p(zext), p(arg), p(ws), ctypes.byref(nblk), st)
        assert rc == 0, L.sug_last_error()
        torch.cuda.synchronize()
        res.append((zext, arg, nblk.value, ws))
    assert res[0][2] == 64 and res[1][2] > 16, 'expected whole segments at 64 clouds and split segments at 16 (%d, %d)' % (res[0][2], res[1][2])
    assert torch.equal(res[0][0][:16], res[1][0]) and torch.equal(res[0][1][:16], res[1][1])
    # the BatchNorm partial rows of the split launch add up to those of the same rows in the whole-segment launch
    parts = res[1][2] // 16
    s_whole = res[0][3][:16 * 2 * Co].view(16, 2 * Co).double().sum(0)
    s_split = res[1][3][:16 * parts * 2 * Co].view(16 * parts, 2 * Co).double().sum(0)
    torch.testing.assert_close(s_split, s_whole, rtol=1e-5, atol=1e-2)

"""GPU parity of the full encoders + heads behind Net_MDA.forward against golden outputs of
the reference (tests/golden/model_*.npz; weights by the shared deterministic fill, dropout
disabled, BatchNorm in train mode).  Bar: fp32 logits / features within 1e-4 (north star)."""
import os
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def probe(shape, tag):
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % (2 ** 31))
    return torch.randn(shape, generator=g)


def build(name, seed):
    from sug_amd.model.Model import Net_MDA
    net = Net_MDA(name)
    sd = net.state_dict()
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in sd.items()}, seed))
    for m in net.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
    return net.cuda().train()


def close(a, b, tol, what):
    a, b = a.detach().cpu().float(), b.float()
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, '%s: max abs err %.3e (scale %.3g, tol %.1e)' % (what, err, scale, tol)
    return err


def check_grads(net, G, rtol, dot_tol=None):
    """Gradient norms within `rtol`; probe dot products within `dot_tol` (default rtol) of the
    norm.  Free-running encoders get a looser probe bound: a max-pool over 1024 points x 1024
    channels x B has ~1e4 arg-max decisions, and a 1e-7 feature perturbation (any other BN
    rounding) flips about one near-tie, rerouting that channel's gradient to another point --
    values are unchanged, gradient direction moves by ~1e-2.  Kernel-level tests are tight."""
    dot_tol = rtol if dot_tol is None else dot_tol
    got = dict(net.named_parameters())
    worst = 0.0
    gmax = max(G['grad_norm'].tolist())
    for k, gn, gd in zip(G['grad_names'], G['grad_norm'].tolist(), G['grad_dot'].tolist()):
        g = got[k].grad
        assert g is not None, 'no gradient for ' + k
        n = g.norm().item()
        d = (g.cpu() * probe(g.shape, 'g' + k)).sum().item()
        floor = 1e-4 * gmax
        assert abs(n - gn) <= rtol * gn + floor, '%s: grad norm %.6g vs %.6g' % (k, n, gn)
        assert abs(d - gd) <= dot_tol * max(abs(gd), gn) + floor, '%s: grad probe %.6g vs %.6g' % (k, d, gd)
        worst = max(worst, abs(n - gn) / max(gn, floor))
    # parameters the reference leaves without gradient must not get one here either
    for k, p in got.items():
        if k not in G['grad_names']:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, 'unexpected grad for ' + k
    return worst


def run_passes(name, fname, knn_forced):
    G = load_golden(fname)
    seed = G['seed']
    net = build(name, seed)
    x = G['x'].cuda()
    kw = {}
    if knn_forced:
        forced = [G['knn%d' % i].to(torch.int32).cuda() for i in (1, 2, 3, 4)]
        fwd = net.g.forward
        net.g.forward = lambda xx, node=False: fwd(xx, node=node, knn_idx=forced)
    torch.manual_seed(seed + 1)
    y1, y2, s1, s2 = net(x, semantic_adaption=True)
    outs = (y1, y2, s1, s2)
    loss = sum((t * probe(t.shape, 'probe%d' % i).cuda()).sum() for i, t in enumerate(outs))
    loss.backward()
    return G, net, outs, loss


@pytest.mark.parametrize('name,fname', [('Pointnet', 'model_pointnet.npz'), ('Pointnet2', 'model_pointnet2.npz'),
                                        ('PTran', 'model_ptran.npz'),
                                        ('PTran', 'model_ptran_n2048.npz'),         # BASELINE config 5 cloud size
                                        ('Pointnet2', 'model_pointnet2_b4.npz')])   # config 3 shape, B=4
def test_encoder_parity(name, fname):
    G, net, (y1, y2, s1, s2), loss = run_passes(name, fname, False)
    seed = G['seed']
    close(y1, G['y1'], 1e-4, 'logits c1')
    close(y2, G['y2'], 1e-4, 'logits c2')
    close(s1, G['s1'], 1e-4, 'sem feature c1')
    close(s2, G['s2'], 1e-4, 'sem feature c2')
    check_grads(net, G, 2e-2, dot_tol=5e-2)
    sd = net.state_dict()
    for k, v in zip(G['bn_names'], G['bn_sum'].tolist()):
        got = sd[k].double().sum().item()
        assert abs(got - v) <= 1e-4 * max(1.0, abs(v)), 'BN buffer %s: %.8g vs %.8g' % (k, got, v)
    torch.manual_seed(seed + 2)
    node_s = net(G['x'].cuda(), node_adaptation_s=True)
    torch.manual_seed(seed + 3)
    feat, node = net(G['x'].cuda(), mid_feat=True)
    close(feat, G['mid_feat'], 2e-4, 'mid feat')
    close(node.reshape(node.shape[0], -1), G['mid_node'], 2e-4, 'mid node')
    # attention output ends in BatchNorm1d over the B samples: with B=2 every channel is
    # (x1-x2)/sqrt((x1-x2)^2/4+eps), which amplifies input rounding by up to 1/sqrt(eps) ~ 300
    B = G['x'].shape[0]
    close(node_s, G['node_s'], 2e-4 if B >= 4 else 2e-3, 'node features')


def test_dgcnn_parity_teacher_forced():
    """Neighbour graphs taken from the reference: isolates layer arithmetic from rank flips."""
    G, net, (y1, y2, s1, s2), loss = run_passes('DGCNN', 'model_dgcnn.npz', True)
    close(y1, G['y1'], 1e-4, 'logits c1')
    close(y2, G['y2'], 1e-4, 'logits c2')
    close(s1, G['s1'], 1e-4, 'sem feature c1')
    close(s2, G['s2'], 1e-4, 'sem feature c2')
    check_grads(net, G, 2e-3)


def test_dgcnn_parity_free_running():
    """Own kNN graphs in feature space against the reference run recorded in the golden.  The xyz graph must be bit-exact.
    The feature-space graphs (layers 2-4) can differ from the reference's at fp32 near-ties (any change of rounding in a
    BatchNorm statistic moves an activation by an ulp and may swap two neighbours whose distances differ by less than the
    rounding of the score: quantified against fp64 in test_dgcnn_free_running_flips_are_fp32_ties; the arithmetic behind
    given lists is pinned at 1e-4 in test_dgcnn_parity_teacher_forced).  Here: the number of (point, layer) rows whose
    neighbour SET differs from the golden's must be ZERO on this fixture (it is, with the ascending-feature MFMA chain of
    the fused EdgeConv layer) and the logits hold the north star's 1e-4.  The flip analysis for inputs where near-ties do
    flip lives in tests/test_gpu_fullsize.py."""
    from sug_amd import ops
    rec, real_knn = [], ops.knn

    def spy(f, k):
        idx = real_knn(f, k)
        rec.append(idx.cpu().long())
        return idx
    ops.knn = spy
    try:
        G, net, (y1, y2, s1, s2), loss = run_passes('DGCNN', 'model_dgcnn.npz', False)
    finally:
        ops.knn = real_knn
    assert len(rec) == 4
    assert torch.equal(rec[0], G['knn1']), 'layer-1 (xyz) neighbour graph must be bit-exact'
    differ = [int((rec[i].sort(-1)[0] != G['knn%d' % (i + 1)].sort(-1)[0]).any(-1).sum()) for i in range(4)]
    rows = rec[0].shape[0] * rec[0].shape[1]
    print('free-running DGCNN: rows whose neighbour set differs from the reference run, per layer: %s of %d' % (differ, rows))
    assert sum(differ) == 0, differ
    tol = 1e-4
    err = max(close(y1, G['y1'], tol, 'logits c1'), close(y2, G['y2'], tol, 'logits c2'))
    close(s1, G['s1'], tol, 'sem feature c1')
    close(s2, G['s2'], tol, 'sem feature c2')
    print('free-running DGCNN: logits max err %.3e' % err)


def test_dgcnn_node_pass_and_buffers():
    G = load_golden('model_dgcnn.npz')
    seed = G['seed']
    net = build('DGCNN', seed)
    x = G['x'].cuda()
    torch.manual_seed(seed + 1)
    net(x, semantic_adaption=True)
    sd = net.state_dict()
    for k, v in zip(G['bn_names'], G['bn_sum'].tolist()):
        got = sd[k].double().sum().item()
        assert abs(got - v) <= 2e-4 * max(1.0, abs(v)), 'BN buffer %s: %.8g vs %.8g' % (k, got, v)
    torch.manual_seed(seed + 2)
    close(net(x, node_adaptation_s=True), G['node_s'], 1e-3, 'node features')


def test_pointnet_cls_config1():
    """Config 1 (train_source.py plumbing): Pointnet_cls forward + CE loss."""
    from sug_amd.model.model_pointnet import Pointnet_cls
    G = load_golden('pointnet_cls.npz')
    net = Pointnet_cls()
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, G['seed']))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    net = net.cuda().train()
    y = net(G['x'].cuda())
    close(y, G['y'], 1e-4, 'logits')
    loss = torch.nn.functional.cross_entropy(y, G['label'].cuda())
    assert abs(loss.item() - float(G['loss'])) < 1e-4


def test_pointnet_cls_config1_source_only_train_step():
    """BASELINE config 1 as train_source.py:76-97, :113-131 runs it: Pointnet_cls -> CrossEntropyLoss -> backward -> ONE
    optim.Adam(model.parameters(), lr 1e-3, weight decay 5e-5) update, against the reference run in
    tests/golden/pointnet_cls.npz (round 6: gradients, BatchNorm buffers, post-step parameters, second-forward loss).
    Adam's first update is lr * g / (|g| + eps): +-lr per element whatever the gradient's size, so an element whose
    gradient is rounding noise may move the other way -- the per-tensor checks allow 1 % of the elements to do so; the
    loss of the second forward (same batch, updated weights) holds 2e-3."""
    G = load_golden('pointnet_cls.npz')
    for own_adam in (False, True):
        net = _cls_net('Pointnet_cls', G)
        p0 = {k: v.detach().clone() for k, v in net.named_parameters()}
        if own_adam:
            from sug_amd.optim import Adam
            opt = Adam(net.parameters(), lr=1e-3, weight_decay=5e-5)
        else:
            opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-5)
        x, lab = G['x'].cuda(), G['label'].cuda()
        y = net(x)
        _cls_check(net, G, y, 1e-4, 2e-3, dot_tol=2e-2)          # logits, CE, backward, gradients, BatchNorm buffers
        opt.step()
        opt.zero_grad()
        post = dict(net.named_parameters())
        lr = 1e-3
        for k, want_sum, want_dn in zip(G['param_names'], G['param_sum'].tolist(), G['param_delta_norm'].tolist()):
            n = post[k].numel()
            dn = float((post[k].detach() - p0[k]).double().norm())
            assert abs(dn - want_dn) <= 2e-2 * max(want_dn, lr), (k, dn, want_dn)
            got_sum = float(post[k].detach().double().sum())
            assert abs(got_sum - want_sum) <= 2 * lr * max(4.0, 0.01 * n) + 1e-5 * abs(want_sum), (k, got_sum, want_sum, n)
        loss2 = torch.nn.functional.cross_entropy(net(x), lab)
        assert abs(float(loss2) - float(G['loss2'])) <= 2e-3 * max(1.0, abs(float(G['loss2']))), (own_adam, float(loss2), float(G['loss2']))
        assert float(loss2) < float(G['loss'])                   # the step went downhill, as in the reference run


def _cls_net(cls, G):
    from sug_amd.model import model_pointnet as MP
    net = getattr(MP, cls)()
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, G['seed']))
    for m in net.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
    return net.cuda().train()


def _cls_check(net, G, y, tol, grad_tol, dot_tol=None):
    close(y, G['y'], tol, 'logits')
    loss = torch.nn.functional.cross_entropy(y, G['label'].cuda())
    assert abs(loss.item() - float(G['loss'])) <= tol * max(1.0, abs(float(G['loss']))), (loss.item(), float(G['loss']))
    loss.backward()
    check_grads(net, G, grad_tol, dot_tol=dot_tol)
    sd = net.state_dict()
    for k, v in zip(G['bn_names'], G['bn_sum'].tolist()):
        got = sd[k].double().sum().item()
        assert abs(got - v) <= 2e-4 * max(1.0, abs(v)), 'BN buffer %s: %.8g vs %.8g' % (k, got, v)


def test_pointnet2_cls_source_only():
    """train_source.py:76-77 with Model Pointnet2: model_pointnet.Pointnet2_cls (model/model_pointnet.py:58-90) against the
    reference run of tests/golden/pointnet2_cls.npz -- logits, CE loss, gradients, BatchNorm buffers."""
    G = load_golden('pointnet2_cls.npz')
    net = _cls_net('Pointnet2_cls', G)
    torch.manual_seed(G['seed'] + 1)
    y = net(G['x'].cuda())
    _cls_check(net, G, y, 1e-4, 2e-2, dot_tol=5e-2)


def test_dgcnn_cls_source_only_teacher_forced():
    """model_pointnet.DGCNN (model/model_pointnet.py:93-161) with the reference run's neighbour graphs."""
    G = load_golden('dgcnn_cls.npz')
    net = _cls_net('DGCNN', G)
    forced = [G['knn%d' % i].to(torch.int32).cuda() for i in (1, 2, 3, 4)]
    y = net(G['x'].cuda(), knn_idx=forced)
    # gradient norms 2e-3; probe dot products 2e-2: the global max-pool / max over k have ~1e5 arg-max decisions and a
    # near-tie that falls the other way reroutes one channel's gradient (values unchanged; measured 4.6e-3 of the norm)
    _cls_check(net, G, y, 1e-4, 2e-3, dot_tol=2e-2)


def test_dgcnn_cls_source_only_free_running():
    """The same with its own kNN graphs: on this fixture every neighbour list must equal the reference run's, and the
    logits hold 1e-4."""
    from sug_amd import ops
    G = load_golden('dgcnn_cls.npz')
    net = _cls_net('DGCNN', G)
    rec, real_knn = [], ops.knn

    def spy(f, k):
        idx = real_knn(f, k)
        rec.append(idx.cpu().long())
        return idx
    ops.knn = spy
    try:
        y = net(G['x'].cuda())
    finally:
        ops.knn = real_knn
    assert len(rec) == 4
    assert torch.equal(rec[0], G['knn1']), 'layer-1 (xyz) neighbour graph must be bit-exact'
    differ = [int((rec[i].sort(-1)[0] != G['knn%d' % (i + 1)].sort(-1)[0]).any(-1).sum()) for i in range(4)]
    print('free-running model_pointnet.DGCNN: rows whose neighbour set differs from the reference run: %s' % differ)
    assert sum(differ) == 0, differ
    _cls_check(net, G, y, 1e-4, 2e-3, dot_tol=2e-2)


@pytest.mark.parametrize('dtype,tol,proj16', [(torch.float16, 3e-3, False), (torch.bfloat16, 3e-2, False),
                                              (torch.float16, 1e-2, True)])
def test_ptran_reduced_precision_mode_deviation(dtype, tol, proj16):
    """The 16-bit GEMM mode of the Point Transformer block (C5) is an extension: the reference is
    fp32.  Its deviation from the fp32 parity mode is bounded here and reported separately
    (proj16: the per-point projections of every block in 16 bits too, as bench.py --fp16 runs it)."""
    from sug_amd.model import Ptran_transformer as PT
    G = load_golden('model_ptran.npz')
    net = build('PTran', G['seed'])
    x = G['x'].cuda()
    keep16 = PT.PROJ_16BIT
    PT.PROJ_16BIT = proj16
    try:
        with torch.no_grad():
            torch.manual_seed(G['seed'] + 1)
            ref = net(x, semantic_adaption=True)
            PT.GEMM_DTYPE = dtype
            torch.manual_seed(G['seed'] + 1)
            got = net(x, semantic_adaption=True)
    finally:
        PT.GEMM_DTYPE = None
        PT.PROJ_16BIT = keep16
    for a, b in zip(got, ref):
        close(a, b.cpu(), tol, 'reduced-precision output')
    # gradients flow through the 16-bit GEMMs
    PT.GEMM_DTYPE = dtype
    try:
        torch.manual_seed(G['seed'] + 1)
        y1, _, _, _ = net(x, semantic_adaption=True)
        y1.square().mean().backward()
    finally:
        PT.GEMM_DTYPE = None
    g = net.g.transformers[0].fc_gamma[0].weight.grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0


def _gradient_errors(name, seed, data_seed, B=4, verbose=True):
    """Relative L2 error over all parameters of the HIP gradients and of the fp32 oracle's, both against the fp64 oracle
    (same network, same inputs, same FPS starts; DGCNN: all three on the HIP path's neighbour graphs)."""
    from sug_amd.model.Model import Net_MDA
    from sug_amd import ops
    shapes = {k: tuple(v.shape) for k, v in Net_MDA(name).state_dict().items()}
    fill = O.fill_params(shapes, seed)
    g = torch.Generator().manual_seed(data_seed)
    x = O.synth_clouds(B, 2048 if name == 'Pointnet2' else 1024, g)
    w1, w2 = probe((B, 10), 'w1'), probe((B, 256), 'w2')

    def loss_of(y1, y2, s1, s2, dev=None):
        a, b = (w1, w2) if dev is None else (w1.to(dev), w2.to(dev))
        return (y1 * a.to(y1.dtype)).sum() + (y2 * a.to(y1.dtype)).sum() + (s1 * b.to(y1.dtype)).sum() + (s2 * b.to(y1.dtype)).sum()

    net = build(name, seed)
    rec, real_knn = [], ops.knn

    def spy(f, k):
        idx = real_knn(f, k)
        rec.append(idx.cpu().long())
        return idx
    ops.knn = spy
    try:
        torch.manual_seed(seed + 1)
        loss_of(*net(x.cuda(), semantic_adaption=True), dev='cuda').backward()
    finally:
        ops.knn = real_knn
    got = {k: p.grad.double().cpu() for k, p in net.named_parameters() if p.grad is not None}
    # DGCNN: all three evaluations on the SAME neighbour graphs (the HIP path's): which of two near-tied neighbours enters a
    # list is a discrete decision that fp64 makes differently from either fp32 path (quantified separately:
    # test_dgcnn_free_running_flips_are_fp32_ties); what is compared here is the arithmetic behind the lists
    kw = {'knn_override': rec} if rec else {}
    ref = {}
    for dt in (torch.float32, torch.float64):
        p = O.as_params({k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in fill.items()})
        torch.manual_seed(seed + 1)
        loss_of(*O.net_mda(p, name, x.to(dt), True, None, semantic_adaption=True, **kw)).backward()
        ref[dt] = {k: v.grad.double() for k, v in p.items() if v.requires_grad and v.grad is not None}
    g64, g32 = ref[torch.float64], ref[torch.float32]
    keys = [k for k in g64 if k in got]
    assert len(keys) >= 10
    tot = sum(float(g64[k].norm()) ** 2 for k in keys) ** 0.5
    e_gpu = sum(float((got[k] - g64[k]).norm()) ** 2 for k in keys) ** 0.5 / tot
    e_ref = sum(float((g32[k] - g64[k]).norm()) ** 2 for k in keys) ** 0.5 / tot
    if verbose:
        print('%s seed %d: relative L2 gradient error vs the fp64 oracle: HIP %.3e, fp32 oracle %.3e' % (name, seed, e_gpu, e_ref))
    return e_gpu, e_ref, got, g32, g64, keys, tot


@pytest.mark.parametrize('name', ['Pointnet', 'DGCNN', 'Pointnet2'])
def test_gradient_error_is_fp32_rounding_of_the_reference_arithmetic(name):
    """What the loose gradient tolerances of the parity tests stand for: the same network evaluated by
    the oracle in fp64 is the noise-free gradient; the oracle in fp32 (= the reference's arithmetic) deviates from it by
    rounding amplified through arg-max / ReLU-kink near-ties, and the HIP path deviates by the same order -- over all
    parameters, relative L2 error vs fp64 of the HIP gradients <= 3 x that of the fp32 oracle (+1e-5) for PointNet and DGCNN
    (measured: 5.5e-4 vs 5.9e-4 and 1.6e-5 vs 1.4e-5), and no single significant parameter tensor off by more than 10 x
    its fp32-oracle error (+1e-4).  PointNet++ on this seed measures 4.2e-3 against 7.4e-4: its gradients are the worst
    conditioned of the four (three nested max-pools, see test_pointnet2_gradient_error_statistic_over_8_seeds for the
    multi-seed statistic that replaces a single-seed pin); bounded here at 1e-2."""
    e_gpu, e_ref, got, g32, g64, keys, tot = _gradient_errors(name, 5, 3)
    top = sorted(keys, key=lambda k: -float((got[k] - g64[k]).norm()))[:6]
    if os.environ.get('SUG_GRADERR_ALL'):          # diagnostic: every parameter, error relative to its OWN norm
        for k in keys:
            n = float(g64[k].norm()) + 1e-30
            print('   %-40s own-relative error HIP %.2e  fp32 oracle %.2e  (norm / total %.1e)' % (
                k, float((got[k] - g64[k]).norm()) / n, float((g32[k] - g64[k]).norm()) / n, n / tot))
    print('   largest contributions (|err| / total |g|, HIP | fp32 oracle | own norm / total):',
          [(k, '%.1e' % (float((got[k] - g64[k]).norm()) / tot), '%.1e' % (float((g32[k] - g64[k]).norm()) / tot),
            '%.1e' % (float(g64[k].norm()) / tot)) for k in top])
    if name == 'Pointnet2':
        assert e_gpu <= 1e-2, (e_gpu, e_ref)
        return
    assert e_gpu <= 3.0 * e_ref + 1e-5, (e_gpu, e_ref)
    gmax = max(float(g64[k].norm()) for k in keys)
    for k in keys:
        n = float(g64[k].norm())
        if n < 1e-3 * gmax:
            continue                       # e.g. conv biases in front of BatchNorm: true gradient 0
        a, b = float((got[k] - g64[k]).norm()) / n, float((g32[k] - g64[k]).norm()) / n
        assert a <= 10.0 * b + 1e-4, (k, a, b)


def test_pointnet2_gradient_error_statistic_over_16_seeds():
    """a single-seed ratio (4.2e-3 for HIP vs 7.4e-4 for the fp32 reference arithmetic, both against
    fp64) is not a bound.  Here the same measurement over 16 (weight seed, data seed) pairs at B=2, N=2048: per pair the
    ratio r = (HIP error vs fp64) / (fp32-oracle error vs fp64).  If the HIP backward were biased, r would sit above 1 on
    every seed; if the deviation is near-tie noise of the nested max-pools (either fp32 path flips a different handful of
    winners than fp64), r scatters around 1 with a heavy tail.
    Round-4 measurements (tests/diagnostics/diag_pn2_seeds.py, 16 seeds; the ratios span 0.01 ... 1000): median 1.28,
    geometric mean 1.64, 10 of 16 above 1 for the product path; 1.00 / 1.01 / 6 of 16 with the grouped-tensor first layer
    (SUG_SA_FIRST=0); 1.01 / 1.40 / 9 of 16 with the geometric first layer (SUG_SA_FIRST_GEO=1) -- indistinguishable within
    the scatter, and the first-layer forms are equally accurate against fp64 in isolation
    (tests/test_gpu_sagroup.py::test_sa_first_forms_are_equally_accurate_against_fp64): no bias located, the deviation is
    the near-tie noise either fp32 path has.  Asserted: median r <= 2 (the bar of the review) and geometric mean <= 2."""
    import math
    import statistics
    ratios, rows = [], []
    for i in range(16):
        e_gpu, e_ref, *_ = _gradient_errors('Pointnet2', 40 + i, 140 + i, B=2, verbose=False)
        ratios.append(e_gpu / max(e_ref, 1e-12))
        rows.append((e_gpu, e_ref))
    gm = math.exp(sum(math.log(r) for r in ratios) / len(ratios))
    print('PointNet++ gradient error vs fp64 over 16 seeds: (HIP, fp32 oracle) = %s' % [('%.2e' % a, '%.2e' % b) for a, b in rows])
    print('ratio HIP / fp32-oracle: median %.2f, geometric mean %.2f, max %.2f, min %.2f, above 1: %d of %d' % (
        statistics.median(ratios), gm, max(ratios), min(ratios), sum(r > 1 for r in ratios), len(ratios)))
    assert statistics.median(ratios) <= 2.0 and gm <= 2.0, ratios
    assert max(a for a, _ in rows) <= 2e-2, rows

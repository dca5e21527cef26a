"""GPU parity of the MMD kernel and the mmd.py mirror against goldens from the reference."""
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('tag,lsc', [('sem', 5.0), ('geo', 50.0), ('sem32', 5.0)])
def test_mmd_values_and_grads(mmd_golden, tag, lsc):
    from sug_amd.model import mmd
    G = mmd_golden
    X = G[tag + '_X'].cuda().requires_grad_(True)
    Y = G[tag + '_Y'].cuda().requires_grad_(True)
    ls, lt, w = G[tag + '_ls'].cuda(), G[tag + '_lt'].cuda(), G[tag + '_w'].cuda()
    torch.testing.assert_close(mmd.mix_rbf_mmd2(X, Y, mmd.sigma_list).cpu(), G[tag + '_plain'], **TOL)
    torch.testing.assert_close(mmd.mix_rbf_mmd2(X, Y, mmd.sigma_list, sample_weights=w).cpu(), G[tag + '_weighted'], **TOL)
    v = mmd.soft_mmd(ls, X, lt, Y, lsc, sample_weights=w)
    torch.testing.assert_close(v.detach().cpu(), G[tag + '_soft'], **TOL)
    v.backward()
    # The reference's fp32 autograd gradient is itself noisy: it back-propagates through the
    # diagonal K_ii = sum_s exp(-gamma_s * (n_i - 2 g_ii + n_i)), whose three terms (gamma up to
    # 5000) cancel only approximately in fp32.  Bound of that noise (MMD backward):
    m = X.shape[0]
    zmax = max(float(X.abs().max()), float(Y.abs().max()), lsc)
    noise = 5050.5 * (4.0 / (m * m)) * zmax * 4 * torch.finfo(torch.float32).eps
    torch.testing.assert_close(X.grad.cpu(), G[tag + '_soft_gx'], rtol=1e-3, atol=noise)
    torch.testing.assert_close(Y.grad.cpu(), G[tag + '_soft_gy'], rtol=1e-3, atol=noise)
    # tight check against the fp64 oracle (the noise-free value of the same expression)
    Xd, Yd = G[tag + '_X'].double().requires_grad_(True), G[tag + '_Y'].double().requires_grad_(True)
    O.soft_mmd(G[tag + '_ls'], Xd, G[tag + '_lt'], Yd, lsc, G[tag + '_w'].double()).backward()
    s = float(Xd.grad.abs().max())
    torch.testing.assert_close(X.grad.cpu().double(), Xd.grad, rtol=1e-3, atol=2e-4 * s)
    torch.testing.assert_close(Y.grad.cpu().double(), Yd.grad, rtol=1e-3, atol=2e-4 * s)
    torch.testing.assert_close(mmd.hard_mmd(ls, X, ls.clone(), Y).cpu(), G[tag + '_hard'], **TOL)
    torch.testing.assert_close(mmd.max_hard_mmd(ls, X, lt, Y).cpu(), G[tag + '_maxhard'], **TOL)


@pytest.mark.parametrize('tag', ['sem', 'sem32'])
def test_sda_weights_and_mmd_cal(mmd_golden, tag):
    from sug_amd.model import mmd
    G = mmd_golden
    ps, pt = G[tag + '_ps'].cuda(), G[tag + '_pt'].cuda()
    ls, lt = G[tag + '_ls'].cuda(), G[tag + '_lt'].cuda()
    for meth in ('mean2one', 'none'):
        w = mmd.prob_weights_soft(ps, pt, ls, lt, 0.5, meth)
        torch.testing.assert_close(w.cpu(), G[tag + '_pw_' + meth], rtol=1e-4, atol=1e-7)
    cfg = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'none', 'LABEL_WEIGHT': 0.5}
    v = mmd.mmd_cal(ls, G[tag + '_X'].cuda(), lt, G[tag + '_Y'].cuda(), cfg, data_s=ps, data_t=pt)
    torch.testing.assert_close(v.cpu(), G[tag + '_cal_none'], **TOL)
    with pytest.raises(RuntimeError):
        mmd.mmd_cal(ls, ps, lt, pt, {'NAME': 'nope'})


@pytest.mark.parametrize('m,D', [(256, 4106), (5, 7), (33, 266)])
def test_mmd_vs_oracle_sizes(m, D):
    """Full-size (global batch 256) and ragged sizes against the oracle, with gradients."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(m + D)
    X = torch.randn(m, D, generator=g) * (0.05 if D > 1000 else 1.0)
    Y = torch.randn(m, D, generator=g) * (0.05 if D > 1000 else 1.0) + 0.02
    w = torch.rand(m, generator=g) + 0.5
    # fp64 oracle: the fp32 reference gradient carries cancellation noise (see above)
    Xo, Yo = X.double().requires_grad_(True), Y.double().requires_grad_(True)
    vo = O.mix_rbf_mmd2(Xo, Yo, sample_weights=w.double())
    vo.backward()
    v32 = O.mix_rbf_mmd2(X, Y, sample_weights=w)
    Z = torch.cat((X, Y), 0).cuda().requires_grad_(True)
    v = ops.mix_rbf_mmd2_rows(Z, m, w.cuda())
    v.backward()
    torch.testing.assert_close(v.detach().cpu(), v32, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(v.detach().cpu().double(), vo.detach(), rtol=1e-4, atol=1e-5)
    go = torch.cat((Xo.grad, Yo.grad), 0)
    torch.testing.assert_close(Z.grad.cpu().double(), go, rtol=1e-3, atol=2e-4 * float(go.abs().max()))
    # MMD of a sample with itself is exactly 0 in the biased estimator (size-independent property)
    Zs = torch.cat((X, X), 0).cuda()
    assert abs(float(ops.mix_rbf_mmd2_rows(Zs, m))) < 1e-6


@pytest.mark.parametrize('m', [8, 32, 300])
@pytest.mark.parametrize('meth', ['none', 'naive_inverse', 'exp_inverse', 'mean2one'])
def test_sda_prob_weights_kernel_vs_oracle(m, meth):
    """sug_sda_prob_weights against the oracle's prob_weights_soft on random logits / labels."""
    from sug_amd.model import mmd
    g = torch.Generator().manual_seed(m)
    ps, pt = torch.randn(m, 10, generator=g) * 2, torch.randn(m, 10, generator=g) * 2
    ls, lt = torch.randint(0, 10, (m,), generator=g), torch.randint(0, 10, (m,), generator=g)
    want = O.prob_weights_soft(ps, pt, ls, lt, 0.5, meth)
    got = mmd.prob_weights_soft(ps.cuda(), pt.cuda(), ls.cuda(), lt.cuda(), 0.5, meth)
    assert got.shape == want.shape
    torch.testing.assert_close(got.cpu(), want, rtol=2e-4, atol=1e-7)


def test_mmd_cal_with_entropy_weights_matches_reference():
    """mmd_cal(..., args with ENTROPY_WEIGHTS) on the device == the reference's value (tests/golden/entropy.npz):
    cal_sample_weights prefers ENTROPY_WEIGHTS over SEM_WEIGHTS (model/mmd.py:44-53)."""
    from conftest import load_golden
    from sug_amd.model import mmd
    G = load_golden('entropy.npz')
    c = lambda k: G[k].cuda()
    for w in ('none', 'mean2one'):
        args = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'mean2one', 'ENTROPY_WEIGHTS': w, 'LABEL_WEIGHT': 0.5}
        v = mmd.mmd_cal(c('ls'), c('fs'), c('lt'), c('ft'), args, data_s=c('ps'), data_t=c('pt'))
        assert abs(float(v) - float(G['mmd_' + w])) <= 1e-4 * max(1.0, abs(float(G['mmd_' + w]))), (w, float(v), float(G['mmd_' + w]))
        torch.testing.assert_close(mmd.entropy_weights(c('ps'), c('pt'), w).cpu(), G['w_' + w], rtol=1e-4, atol=2e-7)      # (x log(x/y) - x + y of nearby entropies cancels: 6e-8 absolute between scipy and torch)


@pytest.mark.parametrize('tag', ['sem', 'geo', 'sem32'])
def test_unbiased_estimator_vs_reference_golden(tag):
    """mix_rbf_mmd2(biased=False) = _mmd2(biased=False), model/mmd.py:304-308 (round 6; raised NotImplementedError before):
    value (plain and with sample weights) and gradient against the reference run in tests/golden/mmd.npz."""
    from conftest import load_golden
    from sug_amd.model import mmd
    G = load_golden('mmd.npz')
    X, Y = G[tag + '_X'].cuda().requires_grad_(True), G[tag + '_Y'].cuda().requires_grad_(True)
    w = G[tag + '_w'].cuda()
    v = mmd.mix_rbf_mmd2(X, Y, mmd.sigma_list, biased=False)
    vw = mmd.mix_rbf_mmd2(X, Y, mmd.sigma_list, biased=False, sample_weights=w)
    assert abs(float(v) - float(G[tag + '_unbiased'])) <= 1e-4 * max(1.0, abs(float(G[tag + '_unbiased'])))
    assert abs(float(vw) - float(G[tag + '_unbiased_weighted'])) <= 1e-4 * max(1.0, abs(float(G[tag + '_unbiased_weighted'])))
    vw.backward()
    # the fp32 reference gradient is itself noise-limited by the exp(-5000 d^2) cancellation (see the biased test above):
    # compared against the fp64 oracle, with the reference's own deviation from it as the yardstick
    Xo, Yo = G[tag + '_X'].double().requires_grad_(True), G[tag + '_Y'].double().requires_grad_(True)
    O.mix_rbf_mmd2(Xo, Yo, sample_weights=G[tag + '_w'].double(), biased=False).backward()
    go = torch.cat((Xo.grad, Yo.grad), 0)
    got = torch.cat((X.grad, Y.grad), 0).cpu().double()
    ref = torch.cat((G[tag + '_unbiased_gx'], G[tag + '_unbiased_gy']), 0).double()
    scale = float(go.abs().max())
    err, ref_err = float((got - go).abs().max()), float((ref - go).abs().max())
    assert err <= max(2e-4 * scale, 2.0 * ref_err), (err, ref_err, scale)
    # and the biased / unbiased values differ by the diagonal terms exactly: 2 ns / m - (m-1)-pair renormalisation
    vb = mmd.mix_rbf_mmd2(X.detach(), Y.detach(), mmd.sigma_list)
    assert abs(float(vb) - float(G[tag + '_plain'])) <= 1e-4 * max(1.0, abs(float(G[tag + '_plain'])))


def test_entropy_weights_of_a_saturated_row_follow_scipy_kl_div():
    """A one-hot probability row has an fp32 entropy of exactly -0.0: scipy's kl_div (the reference's, dataset_splitter.py:244)
    then gives +inf for that pair ('none'), and 'mean2one' turns it into NaN in that slot and 0 elsewhere -- recorded in
    tests/golden/entropy.npz (ps1 / w1_*).  The plain x log(x/y) - x + y gave NaN for both."""
    from conftest import load_golden
    from sug_amd.model import mmd
    G = load_golden('entropy.npz')
    for w in ('none', 'mean2one'):
        want = G['w1_' + w].reshape(-1)
        for got in (mmd.entropy_weights(G['ps1'].cuda(), G['pt'].cuda(), w).cpu().reshape(-1), O.entropy_weights(G['ps1'], G['pt'], w).reshape(-1)):
            assert torch.equal(torch.isnan(got), torch.isnan(want)) and torch.equal(torch.isinf(got), torch.isinf(want)), (w, got, want)
            fin = torch.isfinite(want)
            torch.testing.assert_close(got[fin], want[fin], rtol=1e-4, atol=2e-7)


@pytest.mark.parametrize('m', [4, 32, 80])
def test_soft_mmd_multi_equals_the_terms_one_by_one(m):
    """mmd.soft_mmd_multi (one launch per stage for the step's three terms, sug_soft_mmd_multi_*) against mmd_cal per term:
    values and gradients bit for bit; wide and narrow operands (both kernel forms in one launch), with and without SDA
    weights, a term whose value is not used, a non-contiguous feature block."""
    from sug_amd.model import mmd
    g = torch.Generator().manual_seed(5 + m)
    dev = 'cuda'
    ls, lt = torch.randint(0, 10, (m,), generator=g).to(dev), torch.randint(0, 10, (m,), generator=g).to(dev)
    geo = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 50.0, 'GEO_WEIGHTS': 'mean2one'}
    sem = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5.0, 'SEM_WEIGHTS': 'mean2one', 'LABEL_WEIGHT': 0.5}
    plain = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 2.0}
    pcs, pct = torch.rand(m, 3, 128, generator=g).to(dev), torch.rand(m, 3, 128, generator=g).to(dev)
    logit = lambda: torch.randn(m, 10, generator=g).to(dev)
    specs = [(4096, geo, pcs, pct), (256, sem, logit(), logit()), (100, sem, logit(), logit()), (37, plain, None, None)]

    def leaves():
        gg = torch.Generator().manual_seed(77)
        out = []
        for D, _, _, _ in specs:
            fs = (0.3 * torch.randn(m, D, generator=gg)).to(dev).requires_grad_(True)
            wide = (0.3 * torch.randn(m, 2 * D, generator=gg)).to(dev).requires_grad_(True)
            out.append((fs, wide))
        return out

    coef = [1.0, 0.5, 0.0, 2.0]                      # term 2: value computed, no gradient flows into it
    res = []
    for multi in (False, True):
        L = leaves()
        feats = [(fs, wide[:, ::2]) for fs, wide in L]          # target blocks with a column stride of 2
        if multi:
            vals = mmd.soft_mmd_multi(ls, lt, [(fs, ft, cfg, ds, dt) for (fs, ft), (_, cfg, ds, dt) in zip(feats, specs)])
        else:
            vals = [mmd.mmd_cal(ls, fs, lt, ft, cfg, data_s=ds, data_t=dt) for (fs, ft), (_, cfg, ds, dt) in zip(feats, specs)]
        total = sum(c * v for c, v in zip(coef, vals) if c != 0.0)
        total.backward()
        res.append(([v.detach().clone() for v in vals], [(fs.grad, wide.grad) for fs, wide in L]))
    (va, ga), (vb, gb) = res
    for a, b in zip(va, vb):
        assert torch.equal(a, b), (a, b)
    for i, ((a1, a2), (b1, b2)) in enumerate(zip(ga, gb)):
        if coef[i] == 0.0:
            assert b1 is None and b2 is None
            continue
        assert torch.equal(a1, b1) and torch.equal(a2, b2), i


@pytest.mark.parametrize('B', [1, 7, 64, 300])
def test_fused_weight_launches_match_their_step_by_step_forms(B):
    """sug_chamfer_weights (distance2weights inside the Chamfer fold launch) against distance2weights(chamfer(...)) and
    sug_sda_prob_weights_multi (several heads, one launch) against the single-head kernel."""
    from sug_amd import ops
    from sug_amd.model import mmd
    g = torch.Generator().manual_seed(B)
    a, b = torch.rand(B, 200, 3, generator=g).cuda(), torch.rand(B, 173, 3, generator=g).cuda()
    dist = ops.chamfer(a, b)
    for method in ('naive_inverse', 'exp_inverse', 'mean2one'):
        w = ops.chamfer_weights(a, b, method)
        ref = mmd.distance2weights(dist, method).reshape(-1)
        if method == 'mean2one':
            assert torch.equal(w, ref)            # an integer scale: exact unless 1/mean sits on an integer
        else:
            torch.testing.assert_close(w, ref, rtol=1e-5, atol=1e-9)
    # geometric_weights routes through it for channel-first clouds as well
    w = mmd.geometric_weights(a.transpose(1, 2).contiguous(), b[:, :173].transpose(1, 2).contiguous(), weighting='mean2one')
    assert torch.equal(w.reshape(-1), mmd.distance2weights(dist, 'mean2one').reshape(-1))
    ls, lt = torch.randint(0, 10, (B,), generator=g).cuda(), torch.randint(0, 10, (B,), generator=g).cuda()
    preds = [(torch.randn(B, 10, generator=g).cuda(), torch.randn(B, 20, generator=g).cuda()[:, ::2]) for _ in range(5)]
    for method in ('none', 'naive_inverse', 'exp_inverse', 'mean2one'):
        many = ops.sda_prob_weights_multi(preds, ls, lt, 0.5, method)
        for (ps, pt), w in zip(preds, many):
            assert torch.equal(w, ops.sda_prob_weights(ps, pt, ls, lt, 0.5, method))

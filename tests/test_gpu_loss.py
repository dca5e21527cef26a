"""GPU parity of the scalar tail of a step (sug_amd/csrc/loss.hip) against plain torch: the cross entropy of both heads on
the source rows of the paired logits (train_dg_single_gpu.py:269-292 with nn.CrossEntropyLoss()), the weighted loss sum
(:314-324) and the copy-free split of a paired tensor."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('M,Mtot,C', [(32, 64, 10), (4, 8, 10), (8, 8, 10), (64, 128, 10), (5, 10, 7)])
def test_ce_pair_matches_torch_cross_entropy(M, Mtot, C):
    from sug_amd import ops
    g = torch.Generator().manual_seed(M * 100 + C)
    y1 = (torch.randn(Mtot, C, generator=g) * 3).cuda().requires_grad_(True)
    y2 = (torch.randn(Mtot, C, generator=g) * 3).cuda().requires_grad_(True)
    lab = torch.randint(0, C, (M,), generator=g).cuda()
    w = 0.5 * 0.7
    r1, r2 = y1.detach().clone().requires_grad_(True), y2.detach().clone().requires_grad_(True)
    crit = torch.nn.CrossEntropyLoss()
    ref = w * (crit(r1[:M], lab) + crit(r2[:M], lab))
    (ref * 1.7).backward()
    got = ops.ce_pair(y1, y2, lab, w)
    (got * 1.7).backward()
    torch.testing.assert_close(got, ref, rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(y1.grad, r1.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(y2.grad, r2.grad, rtol=1e-5, atol=1e-7)
    assert float(y1.grad[M:].abs().sum()) == 0.0 and float(y2.grad[M:].abs().sum()) == 0.0     # target rows: exact zeros


@pytest.mark.parametrize('ignore_index', [-100, 3])
def test_ce_pair_label_semantics_are_cross_entropy_loss_ones(ignore_index):
    """nn.CrossEntropyLoss label semantics: rows labelled ignore_index are skipped and left out of the mean;
    any other out-of-range label -- where torch raises -- poisons loss and gradient with NaN instead of being scored."""
    from sug_amd import ops
    M, Mtot, C = 16, 32, 10
    g = torch.Generator().manual_seed(5)
    y1 = (torch.randn(Mtot, C, generator=g) * 3).cuda().requires_grad_(True)
    y2 = (torch.randn(Mtot, C, generator=g) * 3).cuda().requires_grad_(True)
    lab = torch.randint(0, C, (M,), generator=g)
    lab[lab == ignore_index] = (ignore_index + 1) % C
    lab[[1, 7, 8]] = ignore_index
    lab = lab.cuda()
    r1, r2 = y1.detach().clone().requires_grad_(True), y2.detach().clone().requires_grad_(True)
    crit = torch.nn.CrossEntropyLoss(ignore_index=ignore_index)
    ref = 0.5 * (crit(r1[:M], lab) + crit(r2[:M], lab))
    ref.backward()
    got = ops.ce_pair(y1, y2, lab, 0.5, ignore_index)
    got.backward()
    torch.testing.assert_close(got, ref, rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(y1.grad, r1.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(y2.grad, r2.grad, rtol=1e-5, atol=1e-7)
    assert float(y1.grad[[1, 7, 8]].abs().sum()) == 0.0
    bad = lab.clone()
    bad[2] = C                      # out of range and not the ignore label
    if ignore_index == C:
        return
    z1 = y1.detach().clone().requires_grad_(True)
    v = ops.ce_pair(z1, y2.detach(), bad, 0.5, ignore_index)
    assert torch.isnan(v)
    v.backward()
    assert torch.isnan(z1.grad[2]).all()


def test_loss_combine_matches_the_step_formula():
    from sug_amd import ops
    vals = [torch.tensor(v, device='cuda', requires_grad=True) for v in (2.31, 0.284, 0.571, -0.113)]
    ref_vals = [v.detach().clone().requires_grad_(True) for v in vals]
    wg, ws = 1.0 * 1, 0.5 * 1.0 * 1
    tot, geo, sem = ops.loss_combine(*vals, wg, ws)
    lc, v0, v1, v2 = ref_vals
    rgeo = wg * v0
    rsem = ws * (v1 + v2)
    rtot = lc + rgeo + rsem
    torch.testing.assert_close(tot, rtot, rtol=1e-6, atol=0)
    torch.testing.assert_close(geo, rgeo.detach(), rtol=1e-6, atol=0)
    torch.testing.assert_close(sem, rsem.detach(), rtol=1e-6, atol=0)
    assert not geo.requires_grad and not sem.requires_grad
    tot.backward()
    rtot.backward()
    for a, b in zip(vals, ref_vals):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-6, atol=0)
    # terms may be absent (SEM_SCALE = 0)
    t2, g2, s2 = ops.loss_combine(vals[0].detach(), vals[1].detach(), None, None, 2.0, 0.5)
    assert abs(float(t2) - (2.31 + 2.0 * 0.284)) < 1e-6 and float(s2) == 0.0


def test_split_halves_backward_is_copy_free_for_adjacent_gradients():
    from sug_amd import ops
    t = torch.randn(8, 16, device='cuda', requires_grad=True)
    a, b = ops.split_halves(t)
    assert a.data_ptr() == t.data_ptr() and torch.equal(b, t[4:])
    # adjacent row blocks of one buffer (what mmd_assemble's backward hands back): the pair's gradient IS that buffer
    buf = torch.randn(8, 26, device='cuda')
    (gt,) = torch.autograd.grad((a, b), t, (buf[:4, :16], buf[4:, :16]))
    assert gt.data_ptr() == buf.data_ptr() and torch.equal(gt, buf[:, :16])
    # unrelated gradients, or a missing one: concatenated / zero-filled
    a, b = ops.split_halves(t)
    g1, g2 = torch.randn(4, 16, device='cuda'), torch.randn(4, 16, device='cuda')
    (gt,) = torch.autograd.grad((a, b), t, (g1, g2))
    assert torch.equal(gt, torch.cat((g1, g2)))
    a, b = ops.split_halves(t)
    (gt,) = torch.autograd.grad(a.sum(), t)
    assert torch.equal(gt[:4], torch.ones(4, 16, device='cuda')) and float(gt[4:].abs().sum()) == 0.0


def test_step_with_fused_loss_tail_equals_the_unfused_step(monkeypatch):
    """SUGStep's paired step with the fused tail (ops.ce_pair, ops.loss_combine, ops.split_halves, ops.cloud_rows) against
    the same step with SUG_FUSED_LOSS=0 (torch's CrossEntropyLoss, scalar ops, unbind): same losses and gradients."""
    from bench import BENCH_METHODS, synth
    from oracle import ref_cpu as O
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    res = []
    for fused in ('1', '0'):
        monkeypatch.setenv('SUG_FUSED_LOSS', fused)
        net = Net_MDA('DGCNN')
        net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, 9))
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        net = net.cuda().train()
        tr = SUGStep(net, lr=0.0, weight_decay=0.0, methods=BENCH_METHODS)
        data = synth(4, 1024, 77, 'cuda')
        torch.manual_seed(5)
        fused_before = tr.fused_heads
        tr.fused_heads = True
        from sug_amd import ops
        keep, ops.FUSED_HEADS = ops.FUSED_HEADS, True
        try:
            lc, lg, ls = tr.losses(*data, combine=(fused == '1'))
            tot = tr._total if tr._total is not None else lc + lg + ls
            tr._total = None
            tot.backward()
        finally:
            ops.FUSED_HEADS = keep
            tr.fused_heads = fused_before
        res.append(([float(lc), float(lg), float(ls), float(tot)],
                    {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
    for a, b in zip(res[0][0], res[1][0]):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys()
    gmax = max(float(g.abs().max()) for g in res[1][1].values())
    for k in res[1][1]:
        torch.testing.assert_close(res[0][1][k], res[1][1][k], rtol=2e-4, atol=2e-6 * gmax)

"""The benched dropout (p = 0.4 in both Pointnet_c heads, model/Model.py:423-431) held against the oracle on the SAME
masks: the HIP heads draw ONE tensor of uniform randoms per call from the GPU generator (keep when u >= p) -- a stream the
CPU reference cannot share -- so the test records that tensor, turns it into the keep-masks of (head, layer, domain) and
evaluates oracle.sug_losses with those masks (oracle._drop: F.dropout's arithmetic on a given mask, pinned on CPU by
tests/test_oracle_golden.py).  Losses of one step under bench.BENCH_METHODS within 1e-4; closes VERDICT r4 "weak" 1(b)
(model-level parity ran with p = 0 only)."""
import pytest
import torch

import bench
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _batch(B, N, seed=19):
    g = torch.Generator().manual_seed(seed)
    data, data_t = O.synth_clouds(B, N, g), O.synth_clouds(B, N, g)
    lab, lab_t = torch.randint(0, 10, (B,), generator=g), torch.randint(0, 10, (B,), generator=g)
    return data, lab, data_t, lab_t


@pytest.mark.parametrize('model_name,single_pass', [('Pointnet', False), ('DGCNN', False), ('DGCNN', True), ('PTran', False)])
def test_step_losses_with_dropout_match_oracle_on_the_same_masks(model_name, single_pass):
    from sug_amd import ops
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    B, N, P = 4, 1024, 0.4
    batch = _batch(B, N)
    net = Net_MDA(model_name)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(O.fill_params(shapes, 7))
    assert net.c1.dropout1.p == P and net.c2.dropout2.p == P            # the reference's rate, untouched
    net = net.cuda().train()
    tr = SUGStep(net, lr=0.0, methods=bench.BENCH_METHODS, single_pass=single_pass)
    data, lab, data_t, lab_t = [t.cuda() for t in batch]
    assert ops.heads_fused_supported((net.c1, net.c2), torch.zeros(2 * B, 512 if model_name == 'PTran' else 1024, device='cuda'))
    drawn, lists, starts = [], [], []
    real_rand, real_knn = torch.rand, ops.knn

    def rand_spy(*a, **kw):
        t = real_rand(*a, **kw)
        if kw.get('device') is not None and torch.device(kw['device']).type == 'cuda':
            drawn.append(t)
        return t

    def knn_spy(f, k):
        idx = real_knn(f, k)
        lists.append(idx.cpu().long())
        return idx

    def start_spy(Bn, n):
        t = torch.randint(0, n, (Bn,), dtype=torch.long)
        starts.append(t)
        return t
    torch.rand, ops.knn, ops.START_PROVIDER = rand_spy, knn_spy, start_spy
    try:
        torch.manual_seed(3)
        got = [float(v.detach()) for v in tr.losses(data, lab, data_t, lab_t)]
    finally:
        torch.rand, ops.knn, ops.START_PROVIDER = real_rand, real_knn, None
    assert len(drawn) == 1, 'one launch of uniform randoms per heads call (%d seen)' % len(drawn)
    u = drawn[0].cpu()
    M = 2 * B
    three = model_name != 'PTran'
    N1, N2 = (512 if three else 0), 256
    assert u.numel() == 2 * M * (N1 + N2)
    keep = {}
    for h in range(2):
        k1 = (u[h * M * N1:(h + 1) * M * N1].view(M, N1) >= P) if three else None
        o2 = 2 * M * N1
        k2 = u[o2 + h * M * N2:o2 + (h + 1) * M * N2].view(M, N2) >= P
        keep[h] = (k1, k2)
    half = lambda t, d: None if t is None else t[d * B:(d + 1) * B]
    drop_keep = tuple(tuple((half(keep[h][0], d), half(keep[h][1], d)) for h in range(2)) for d in range(2))
    frac = float(keep[0][1].float().mean())
    assert 0.5 < frac < 0.7, frac
    # the FPS starts the step drew: per sampling stage one [2B] draw (paired pass) or per pass (two-pass step)
    kw = {}
    if model_name == 'DGCNN':
        n_sem = 4
        kw['knn_override'] = ([l[:B] for l in lists[:n_sem]], [l[B:] for l in lists[:n_sem]])
    per_pass = {'Pointnet': 1, 'DGCNN': 1, 'PTran': 4}[model_name]
    # provider calls: paired passes request either [2B] per stage (one FPS call per forward) or [B] per (domain, stage)
    if starts[0].numel() == 2 * B:
        st_of = lambda pass_i, dom: [starts[pass_i * per_pass + s][dom * B:(dom + 1) * B] for s in range(per_pass)]
    else:
        st_of = lambda pass_i, dom: [starts[(pass_i * 2 + dom) * per_pass + s] for s in range(per_pass)]
    as_spec = lambda l: l if model_name in ('Pointnet', 'DGCNN') else tuple(l)
    node_pass = 0 if single_pass else 1
    st4 = [as_spec(st_of(0, 0)), as_spec(st_of(0, 1)), as_spec(st_of(node_pass, 0)), as_spec(st_of(node_pass, 1))]
    p = O.as_params(O.fill_params(shapes, 7))
    with torch.no_grad():
        if model_name == 'DGCNN' and not single_pass:
            # the node pass of the two-pass step runs its own kNN for conv3 / conv4 (other FPS start -> other features):
            # teacher-force the semantic passes only and let the oracle's node passes run free
            want = _dgcnn_two_pass(p, batch, st4, kw['knn_override'], lists, B, drop_keep, P)
        else:
            want = [float(v) for v in O.sug_losses(p, model_name, batch[0], batch[1], batch[2], batch[3],
                                                   dict(bench.BENCH_METHODS['GEO_MMD'][0]), dict(bench.BENCH_METHODS['SEM_MMD'][0]),
                                                   drop_p=P, starts=st4, drop_keep=drop_keep, **kw)]
    print(model_name, 'single_pass' if single_pass else 'two-pass', 'p = 0.4: HIP', got, 'oracle on the same masks', want,
          'kept fraction %.3f' % frac)
    for a, b in zip(got, want):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (got, want)


def _dgcnn_two_pass(p, batch, st4, ko_sem, lists, B, drop_keep, P):
    """oracle.sug_losses for DGCNN's two-pass step with the HIP path's neighbour graphs in BOTH passes: the semantic pass
    made four kNN calls (paired [2B,N,k] lists 0..3), the node pass -- prefix shared -- two more (conv3, conv4: lists 4, 5)."""
    import torch.nn.functional as F
    assert len(lists) == 6, len(lists)
    node_ko = lambda d: [lists[0][d * B:(d + 1) * B], lists[1][d * B:(d + 1) * B], lists[4][d * B:(d + 1) * B], lists[5][d * B:(d + 1) * B]]
    data, lab, data_t, lab_t = batch
    geo, sem = dict(bench.BENCH_METHODS['GEO_MMD'][0]), dict(bench.BENCH_METHODS['SEM_MMD'][0])
    ps1, ps2, fs1, fs2 = O.net_mda(p, 'DGCNN', data, True, st4[0], P, semantic_adaption=True, knn_override=ko_sem[0], drop_keep=drop_keep[0])
    pt1, pt2, ft1, ft2 = O.net_mda(p, 'DGCNN', data_t, True, st4[1], P, semantic_adaption=True, knn_override=ko_sem[1], drop_keep=drop_keep[1])
    loss_cls = 0.5 * F.cross_entropy(ps1, lab) + 0.5 * F.cross_entropy(ps2, lab)
    node_s = O.net_mda(p, 'DGCNN', data, True, st4[2], P, node_adaptation_s=True, knn_override=node_ko(0))
    node_t = O.net_mda(p, 'DGCNN', data_t, True, st4[3], P, node_adaptation_t=True, knn_override=node_ko(1))
    loss_geo = geo['GEO_SCALE'] * O.mmd_cal(lab, node_s, lab_t, node_t, geo, data, data_t)
    l1 = sem['SEM_SCALE'] * O.mmd_cal(lab, fs1, lab_t, ft1, sem, ps1, pt1)
    l2 = sem['SEM_SCALE'] * O.mmd_cal(lab, fs2, lab_t, ft2, sem, ps2, pt2)
    return [float(loss_cls), float(loss_geo), float(0.5 * l1 + 0.5 * l2)]

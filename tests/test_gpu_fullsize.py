"""GPU tests at BASELINE.json's full sizes (config 2: DGCNN 64 clouds x 1024 points per paired
pass, k=20; config 3: PointNet++ 64 clouds x 2048 points, sa1 r=0.2 / nsample=32).

The CPU oracle does not finish these sizes in seconds, so the checks are (a) the same operator
written in plain fp32 torch ON THE DEVICE (the k-expanded / N x S-materialised formulation the
reference uses), (b) size-independent properties (run-to-run bit-reproducibility, BatchNorm
statistics identities, index validity / ordering), and (c) equality between the step's exact
restructurings.  B = 64 is also the shape that takes the XCD-aware workgroup mapping
(`(B & 7) == 0`) of the kNN and EdgeConv kernels."""
import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def _edge_layer(C, Co, seed):
    from sug_amd.model.model_utils import conv_2d
    m = conv_2d(2 * C, Co, 1, activation='leakyrelu', bias=False)
    m.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed))
    return m.cuda().train()


def _edge_reference(m, x, idx):
    """get_graph_feature -> 1x1 conv -> train-mode BN -> LeakyReLU(0.01) -> max over k, k-expanded,
    plain fp32 torch on the device (model_utils.py:188-210, :8-32; Model.py:88-94)."""
    B, N, C = x.shape
    k = idx.shape[2]
    W = m.conv[0].weight.view(m.conv[0].weight.shape[0], -1)
    bn = m.conv[1]
    flat = (idx.long() + torch.arange(B, device=x.device).view(B, 1, 1) * N).reshape(-1)
    nbr = x.reshape(B * N, C)[flat].view(B, N, k, C)
    ctr = x.view(B, N, 1, C).expand(B, N, k, C)
    e = torch.cat((nbr - ctr, ctr), dim=3)                     # [B,N,k,2C]
    y = e.reshape(-1, 2 * C) @ W.t()                            # [B*N*k, Co]
    mean = y.mean(0)
    var = y.var(0, unbiased=False)
    z = (y - mean) * torch.rsqrt(var + bn.eps) * bn.weight + bn.bias
    z = torch.nn.functional.leaky_relu(z, 0.01)
    return z.view(B, N, k, -1).max(dim=2)[0], mean, y.var(0, unbiased=True)


@pytest.mark.parametrize('C,Co', [(3, 64), (64, 64), (64, 128), (128, 256)])
def test_edgeconv_layer_full_size(C, Co):
    """The four EdgeConv layers of DGCNN at 64 x 1024 x k=20: forward and input gradient against
    the k-expanded torch formulation, BatchNorm running statistics, bit-reproducibility."""
    from sug_amd import ops
    B, N, k = 64, 1024, 20
    g = torch.Generator().manual_seed(100 + Co)
    x = (torch.randn(B, N, C, generator=g) * 0.7).cuda()
    idx = ops.knn(x, k)                                          # a real neighbour graph (self included)
    probe = torch.randn(B, N, Co, generator=g).cuda()
    m = _edge_layer(C, Co, 21)
    rm0, rv0 = m.conv[1].running_mean.clone(), m.conv[1].running_var.clone()
    outs = []
    for _ in range(2):
        m.conv[1].running_mean.copy_(rm0)
        m.conv[1].running_var.copy_(rv0)
        m.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = m.edge_rows(xi, idx)
        (y * probe).sum().backward()
        outs.append((y.detach().clone(), xi.grad.clone(), m.conv[0].weight.grad.clone(), m.conv[1].weight.grad.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b), 'EdgeConv layer is not bit-reproducible run to run'
    y, gx, gw, gg = outs[0]
    rm1, rv1 = m.conv[1].running_mean.clone(), m.conv[1].running_var.clone()

    xr = x.clone().requires_grad_(True)
    m.zero_grad()
    yr, mean, var_unb = _edge_reference(m, xr, idx)
    (yr * probe).sum().backward()
    torch.testing.assert_close(y, yr.detach(), rtol=1e-4, atol=1e-4)
    # gradients: an arg-max near-tie may route to another neighbour (value-neutral); bound the norm
    # of the difference instead of every element
    rel = float((gx - xr.grad).norm() / xr.grad.norm())
    assert rel < 2e-3, 'input gradient differs by %.3e (relative L2)' % rel
    relw = float((gw - m.conv[0].weight.grad).norm() / m.conv[0].weight.grad.norm())
    assert relw < 2e-3, 'weight gradient differs by %.3e (relative L2)' % relw
    # BatchNorm identities: running <- 0.9 * running + 0.1 * (batch mean, unbiased batch variance)
    torch.testing.assert_close(rm1, 0.9 * rm0 + 0.1 * mean.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv1, 0.9 * rv0 + 0.1 * var_unb.detach(), rtol=1e-4, atol=1e-5)


def test_bn_act_pool_full_size():
    """bn5 -> LeakyReLU(0.2) -> max | mean over the points at [64, 1024, 512] (Model.py:112-116)."""
    from sug_amd import ops
    B, N, C = 64, 1024, 512
    g = torch.Generator().manual_seed(5)
    y = torch.randn(B, N, C, generator=g).cuda()
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g).cuda())
        bn.bias.copy_(torch.randn(C, generator=g).cuda() * 0.1)
    ref = torch.nn.BatchNorm1d(C).cuda().train()
    ref.load_state_dict(bn.state_dict())
    pm, pa = torch.randn(B, C, generator=g).cuda(), torch.randn(B, C, generator=g).cuda()
    yi = y.clone().requires_grad_(True)
    with ops.bn_groups(2):
        omax, omean = ops.bn_act_pool(yi, bn, 0.2)
    ((omax * pm).sum() + (omean * pa).sum()).backward()
    yr = y.clone().requires_grad_(True)
    halves = []
    for part in yr.chunk(2, dim=0):                              # two forward calls of the reference
        z = torch.nn.functional.leaky_relu(ref(part.transpose(1, 2)), 0.2)
        halves.append((z.max(dim=2)[0], z.mean(dim=2)))
    rmax, rmean = torch.cat([h[0] for h in halves]), torch.cat([h[1] for h in halves])
    ((rmax * pm).sum() + (rmean * pa).sum()).backward()
    torch.testing.assert_close(omax, rmax, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(omean, rmean, rtol=1e-4, atol=1e-5)
    rel = float((yi.grad - yr.grad).norm() / yr.grad.norm())
    assert rel < 1e-3, rel
    torch.testing.assert_close(bn.running_mean, ref.running_mean, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(bn.running_var, ref.running_var, rtol=1e-4, atol=1e-6)


def test_sa1_grouping_full_size():
    """Config 3's sa1 at 64 x 2048: FPS(512) + ball query (r=0.2, nsample=32) properties --
    centroids distinct and start-anchored, the FPS greedy invariant, neighbour rows ascending,
    inside the ball, padded with the first hit, and complete (no closer-indexed hit skipped)."""
    from sug_amd import ops
    B, N, S, ns, r = 64, 2048, 512, 32, 0.2
    g = torch.Generator().manual_seed(9)
    xyz = O.synth_clouds(B, N, g).squeeze(-1).transpose(1, 2).contiguous().cuda()      # [B,N,3]
    start = torch.randint(0, N, (B,), generator=g)
    fidx = ops.fps(xyz, S, start).long()
    assert torch.equal(fidx[:, 0].cpu(), start)
    assert int(fidx.min()) >= 0 and int(fidx.max()) < N
    srt = fidx.sort(dim=1)[0]
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), 'FPS returned a repeated centroid on a duplicate-free cloud'
    cen = torch.gather(xyz, 1, fidx.unsqueeze(-1).expand(B, S, 3))
    # greedy invariant: centroid t is (one of) the farthest point(s) from centroids 0..t-1
    d = ((xyz.unsqueeze(1) - cen[:, :64].unsqueeze(2)) ** 2).sum(-1)          # [B,64,N] direct form, as FPS uses
    run = torch.cummin(d, dim=1)[0]                                            # min over the first t+1 centroids
    for t in (1, 2, 17, 63):
        far = run[:, t - 1].max(dim=1)[0]
        got = torch.gather(run[:, t - 1], 1, fidx[:, t:t + 1]).squeeze(1)
        assert bool((got >= far * (1 - 1e-6)).all()), 'FPS step %d did not pick the farthest point' % t
    idx = ops.ball_query(xyz, cen, r, ns).long()                               # [B,S,ns]
    assert int(idx.min()) >= 0 and int(idx.max()) < N, 'every centroid contains itself: no empty rows'
    dist = -2 * cen @ xyz.transpose(1, 2)
    dist = dist + (cen ** 2).sum(-1, keepdim=True) + (xyz ** 2).sum(-1).unsqueeze(1)   # expanded form, [B,S,N]
    r2 = float(torch.tensor(r ** 2, dtype=torch.float32))
    picked = torch.gather(dist, 2, idx)
    assert bool((picked <= r2 + 1e-6).all()), 'a neighbour lies outside the ball'
    inc = idx[:, :, 1:] > idx[:, :, :-1]
    pad = idx[:, :, 1:] == idx[:, :, :1]
    assert bool((inc | pad).all()), 'rows must ascend, then repeat the first hit'
    # completeness away from the rounding band: the number of distinct hits equals min(ns, #inside)
    inside_lo = (dist <= r2 - 1e-5).sum(-1)
    inside_hi = (dist <= r2 + 1e-5).sum(-1)
    nuniq = (inc.sum(-1) + 1)
    assert bool((nuniq >= inside_lo.clamp(max=ns)).all()) and bool((nuniq <= inside_hi.clamp(max=ns)).all())


@pytest.mark.parametrize('model_name,B,N', [('DGCNN', 32, 1024), ('Pointnet2', 16, 2048)])
def test_full_size_step_finite_reproducible_and_restructurings_agree(model_name, B, N):
    """One SUGStep.step at the benchmark shape: finite losses; the default (paired domains, shared
    prefix) and the unchanged-caller form (four separate model(...) calls) give the same losses;
    two runs of the same configuration agree."""
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    g = torch.Generator().manual_seed(3)
    data, data_t = O.synth_clouds(B, N, g).cuda(), O.synth_clouds(B, N, g).cuda()
    lab, lab_t = torch.randint(0, 10, (B,), generator=g).cuda(), torch.randint(0, 10, (B,), generator=g).cuda()
    shapes = {k: tuple(v.shape) for k, v in Net_MDA(model_name).state_dict().items()}
    res = []
    for pair, share in ((True, True), (True, True), (False, False)):
        net = Net_MDA(model_name)
        net.load_state_dict(O.fill_params(shapes, 8))
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        net = net.cuda().train()
        tr = SUGStep(net, pair_domains=pair, share_prefix=share)
        torch.manual_seed(77)
        out = []
        for _ in range(2):
            out.append([float(v) for v in tr.step(data, lab, data_t, lab_t)])
        res.append(out)
    for run in res:
        for step in run:
            assert all(v == v and abs(v) < 1e4 for v in step), res
    for a, b in zip(res[0][0], res[1][0]):             # same configuration twice: float atomics only
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), res   # in the small SA-node scatter kernels
    for a, b in zip(res[0][0], res[2][0]):             # paired + shared vs four separate passes
        assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), res


def test_dgcnn_free_running_flips_are_fp32_ties():
    """Free-running DGCNN (own neighbour graphs in feature space) against the north star's 1e-4:
      (1) the oracle FED THE HIP NEIGHBOUR LISTS reproduces the HIP logits / features within 1e-4,
          i.e. everything except the rank decisions meets the bar;
      (2) every HIP neighbour that is not in the fp64 top-k of the features the kernel was given
          misses it by less than the fp32 rounding bound of the reference's expanded-form score
          -2<x_i,x_j> + |x_i|^2 + |x_j|^2 (model_utils.py:179-181): the disagreements with the
          reference are exactly the pairs fp32 cannot rank, and the CPU reference's own picks on the
          same features are not closer to the fp64 truth."""
    from conftest import load_golden
    from sug_amd import ops
    from sug_amd.model.Model import Net_MDA
    G = load_golden('model_dgcnn.npz')
    seed = G['seed']
    shapes = {k: tuple(v.shape) for k, v in Net_MDA('DGCNN').state_dict().items()}
    fill = O.fill_params(shapes, seed)
    net = Net_MDA('DGCNN')
    net.load_state_dict(fill)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    net = net.cuda().train()
    rec = []
    real_knn = ops.knn

    def spy(f, k):
        idx = real_knn(f, k)
        rec.append((f.detach().clone(), idx.clone()))
        return idx

    ops.knn = spy
    try:
        torch.manual_seed(seed + 1)
        with torch.no_grad():
            got = net(G['x'].cuda(), semantic_adaption=True)
    finally:
        ops.knn = real_knn
    assert len(rec) == 4
    # (1)
    p = {k: v.clone() for k, v in fill.items()}
    torch.manual_seed(seed + 1)
    with torch.no_grad():
        want = O.net_mda(p, 'DGCNN', G['x'], True, None, semantic_adaption=True,
                         knn_override=[i.cpu().long() for _, i in rec])
    for a, b, nm in zip(got, want, ('y1', 'y2', 's1', 's2')):
        err = float((a.cpu() - b).abs().max())
        assert err <= 1e-4 * max(1.0, float(b.abs().max())), '%s differs from the oracle on the HIP graphs by %.3e' % (nm, err)
    # (2)
    eps = float(torch.finfo(torch.float32).eps)
    report = []
    for f, idx in rec:
        C = f.shape[2]
        f64 = f.double()
        sq = (f64 ** 2).sum(-1)                                                # [B,N]
        score = -(sq.unsqueeze(2) - 2.0 * f64 @ f64.transpose(1, 2) + sq.unsqueeze(1))
        kth = score.topk(20, dim=-1)[0][..., -1:]                              # fp64 20th best per query
        bound = 2.0 * (2 * C + 6) * eps * (sq.unsqueeze(2) + sq.max(dim=1, keepdim=True)[0].unsqueeze(1))
        hip_def = kth - torch.gather(score, 2, idx.long())                     # > 0: outside the fp64 top-20
        cpu_idx = O.knn_idx(f.cpu().transpose(1, 2).contiguous(), 20).cuda()   # the reference's fp32 picks, same features
        cpu_def = kth - torch.gather(score, 2, cpu_idx)
        assert bool((hip_def <= bound).all()), 'a HIP neighbour misses the fp64 top-k by more than fp32 rounding (C=%d)' % C
        n_hip, n_cpu = int((hip_def > 0).sum()), int((cpu_def > 0).sum())
        differs = int((idx.long().sort(-1)[0] != cpu_idx.sort(-1)[0]).any(-1).sum())
        report.append((C, n_hip, n_cpu, differs, float((hip_def / bound).max())))
        # the kernel's k-ordered fma chain is at least as faithful to fp64 as the CPU sgemm
        assert n_hip <= max(2 * n_cpu, n_cpu + 8), report
    print('kNN picks outside the fp64 top-20 (C, HIP, CPU reference, rows whose sets differ, worst deficit / bound):', report)
    assert report[0][3] == 0, 'xyz graph must equal the CPU reference exactly'


@pytest.mark.parametrize('D,mlp,S,ns,radius', [(0, [64, 64, 128], 128, 32, 0.25), (128, [128, 128, 256], 64, 64, 0.45)])
def test_set_abstraction_backward_is_fp64_arithmetic_on_its_own_inputs(D, mlp, S, ns, radius):
    """A set-abstraction layer as the product runs it (PointNetSetAbstraction.rows) -- first MLP layer on the ball-query
    lists (P[j] - Q[s], sug_sa_first_*), middle layer through the BatchNorm rows kernels, last layer + max over the group
    fused with its rank-K BatchNorm backward (sug_pointmlp_max_*) -- against an fp64 restatement of
    model/pointnet2_utils.py:107-135,193-207 on the SAME index sets: every intermediate, the output and every gradient
    (input features, conv weights, BatchNorm weights) within 2e-5 relative L2 (measured: 2e-7 .. 7e-7).
    Discrete decisions are taken out of the comparison, because one of them moves a gradient by far more than rounding
    (one ReLU mask entry of 2 M flipped by a 1e-7 difference: 4e-3 on the first two layers' gradients,
    tools/diag_sa_stack.py module): group maxima whose runner-up is within 1e-4 get no loss on either side, and the ReLU
    masks of the two paths must agree (the seeds used here do; a flip skips the case with a message)."""
    from sug_amd import ops
    from sug_amd.model.pointnet2_utils import sample_and_group_idx
    import torch.nn.functional as F
    rel = lambda a, b: float((a.detach().double().cpu().reshape(b.shape) - b.detach()).norm() / (float(b.detach().norm()) + 1e-300))
    torch.manual_seed(0)
    B, N = 4, 512
    xyz = (torch.rand(B, N, 3) - 0.5).cuda()
    pts = torch.relu(torch.randn(B, N, D) * 0.7 + 0.3).cuda().requires_grad_(True) if D else None
    Ws = [(torch.randn(mlp[0], 3 + D) / (3 + D) ** 0.5).cuda().requires_grad_(True),
          (torch.randn(mlp[1], mlp[0]) / mlp[0] ** 0.5).cuda().requires_grad_(True),
          (torch.randn(mlp[2], mlp[1]) / mlp[1] ** 0.5).cuda().requires_grad_(True)]
    bs = [(torch.randn(c) * 0.1).cuda().requires_grad_(True) for c in mlp]
    bns = [torch.nn.BatchNorm2d(c).cuda().train() for c in mlp]
    for bn in bns:
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.uniform_(-0.2, 0.2)
    torch.manual_seed(11)
    new_xyz, idx = sample_and_group_idx(S, radius, ns, xyz)
    # ---- the product path, op by op as PointNetSetAbstraction.rows chains them
    assert ops.sa_first_layer_supported(mlp[0]) and ops.pointmlp_max_supported(mlp[1], mlp[2], ns)
    P = ops.linear_rows(xyz if pts is None else torch.cat((xyz, pts), -1), Ws[0])
    Q = ops.sub_row_bias(ops.linear_rows(new_xyz, Ws[0][:, :3]), bs[0])
    g0 = ops.sa_first_layer(P, Q, idx, bns[0])
    g1 = ops.bn_act_rows(ops.linear_rows(g0, Ws[1], bs[1]), bns[1], 0.0)
    out = ops.pointmlp_max(g1, Ws[2], bs[2], bns[2], 0.0, ns).view(B, S, -1)
    for t in (g0, g1):
        t.retain_grad()
    # ---- fp64 on the same groups
    xd, cd = xyz.double().cpu(), new_xyz.double().cpu()
    pd = pts.detach().double().cpu().requires_grad_(True) if D else None
    Wd = [w.detach().double().cpu().requires_grad_(True) for w in Ws]
    bd = [b.detach().double().cpu() for b in bs]
    gam = [bn.weight.detach().double().cpu().requires_grad_(True) for bn in bns]
    bet = [bn.bias.detach().double().cpu().requires_grad_(True) for bn in bns]
    bi = torch.arange(B).view(B, 1, 1)
    il = idx.long().cpu()
    grouped = xd[bi, il] - cd.unsqueeze(2)
    if D:
        grouped = torch.cat((grouped, pd[bi, il]), -1)
    layer = lambda a, i: torch.relu(F.batch_norm(a.reshape(-1, a.shape[-1]) @ Wd[i].t() + bd[i], None, None, gam[i], bet[i],
                                                 True, 0.1, 1e-5)).view(B, S, ns, -1)
    g0d = layer(grouped, 0)
    g1d = layer(g0d, 1)
    A = layer(g1d, 2)
    g0d.retain_grad()
    g1d.retain_grad()
    flips = [int(((h.detach().cpu() > 0) != (d.detach() > 0)).sum()) for h, d in ((g0, g0d), (g1, g1d))]
    top = A.topk(2, dim=2)[0]
    clear = ((top[:, :, 0] - top[:, :, 1]) > 1e-4).double()
    print('ReLU mask entries that differ: %s; group maxima without a clear winner: %d of %d' % (flips, int((1 - clear).sum()), clear.numel()))
    if any(flips):
        pytest.skip('a ReLU mask entry differs between the fp32 and the fp64 forward (%s): a discrete decision, not arithmetic' % flips)
    probe = torch.randn(B, S, mlp[2], dtype=torch.float64, generator=torch.Generator().manual_seed(5)) * clear
    (top[:, :, 0] * probe).sum().backward()
    (out * probe.float().cuda()).sum().backward()
    errs = {'g0': rel(g0, g0d), 'g1': rel(g1, g1d),
            'out': float(((out.detach().double().cpu() - top[:, :, 0].detach()) * clear).norm() / top[:, :, 0].detach().norm()),
            'd g1': rel(g1.grad, g1d.grad), 'd g0': rel(g0.grad, g0d.grad)}
    if D:
        errs['d points'] = rel(pts.grad, pd.grad)
    for i in range(3):
        errs['dW%d' % i] = rel(Ws[i].grad, Wd[i].grad)
        errs['dgamma%d' % i] = rel(bns[i].weight.grad, gam[i].grad)
        errs['dbeta%d' % i] = rel(bns[i].bias.grad, bet[i].grad)
    print({k: '%.1e' % v for k, v in errs.items()})
    for k, e in errs.items():
        assert e <= 2e-5, (k, e)

"""GPU parity of the index / gather kernels (through the C ABI) against the golden
fixtures produced by the reference and against the CPU oracle on fresh seeded inputs.

Bar: indices bit-exact.  Where the CPU reference itself has exact score ties (padded
clouds, duplicates) its topk/sort order is arbitrary; there the test demands that the
HIP choice carries the *same score* position by position ("ties").
"""
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def rows(x):            # [B,C,N] -> [B,N,C] on the GPU
    return x.transpose(1, 2).contiguous().cuda()


def assert_same_or_tied(idx_hip, idx_ref, score, ulps=0):
    """idx_* [B,N,k]; score [B,N,M] = the oracle's ranking matrix.  Exact match, or the
    scores at the two choices agree to `ulps` units in the last place at every slot."""
    idx_hip, idx_ref = idx_hip.cpu().long(), idx_ref.cpu().long()
    if torch.equal(idx_hip, idx_ref):
        return 1.0
    sh = torch.gather(score, 2, idx_hip)
    sr = torch.gather(score, 2, idx_ref)
    tol = ulps * torch.finfo(torch.float32).eps * score.abs().max()
    bad = (sh - sr).abs() > tol
    assert not bad.any(), '%d slots differ beyond ties (max score gap %g)' % (int(bad.sum()), float((sh - sr).abs().max()))
    return float((idx_hip == idx_ref).float().mean())


def test_knn_xyz_grid_exact(ops_golden):
    from sug_amd import ops
    x = ops_golden['knn_grid_x']
    idx = ops.knn(rows(x), 20)
    assert torch.equal(idx.cpu().long(), ops_golden['knn_grid_idx'])


def test_knn_xyz_random_exact(ops_golden):
    from sug_amd import ops
    x = ops_golden['knn_rand_x']
    idx = ops.knn(rows(x), 20)
    frac = assert_same_or_tied(idx, ops_golden['knn_rand_idx'], O.knn_neg_dist(x), ulps=0)
    assert frac == 1.0, 'xyz kNN must be bit-exact on tie-free clouds (got %.6f)' % frac


def test_knn_padded_cloud_ties(ops_golden):
    from sug_amd import ops
    x = ops_golden['knn_pad_x']
    idx = ops.knn(rows(x), 20)
    assert_same_or_tied(idx, ops_golden['knn_pad_idx'], O.knn_neg_dist(x), ulps=0)
    # the HIP rule among exact ties is "lowest index first": deterministic
    idx2 = ops.knn(rows(x), 20)
    assert torch.equal(idx, idx2)


def test_knn_feature_space_golden(ops_golden):
    """C=64: length-64 fp32 dot products round differently in the CPU sgemm and in the kernel's
    ascending fma chain, so rank flips are allowed only between scores within a few ulps."""
    from sug_amd import ops
    x = ops_golden['knn_feat_x']
    idx = ops.knn(rows(x), 20)
    frac = assert_same_or_tied(idx, ops_golden['knn_feat_idx'], O.knn_neg_dist(x), ulps=8)
    assert frac > 0.999


@pytest.mark.parametrize('C,k,N', [(64, 20, 1024), (128, 20, 1024), (3, 16, 2048), (3, 20, 1000), (5, 7, 300), (64, 32, 256)])
def test_knn_random_features_vs_oracle(C, k, N):
    from sug_amd import ops
    g = torch.Generator().manual_seed(C * 1000 + k)
    x = torch.randn(2, C, N, generator=g)
    idx = ops.knn(rows(x), k)
    ref = O.knn_idx(x, k)
    # fp32 dot products of length C: CPU sgemm and the kernel's fma chain may round differently
    frac = assert_same_or_tied(idx, ref, O.knn_neg_dist(x), ulps=8 if C > 3 else 0)
    assert frac > 0.999


def test_knn_column_slice_input():
    """ld > C: neighbours of a 64-column slice of a wider row buffer."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(3)
    buf = torch.randn(2, 512, 192, generator=g).cuda()
    a = ops.knn(buf[:, :, 64:128], 20)
    b = ops.knn(buf[:, :, 64:128].contiguous(), 20)
    assert torch.equal(a, b)


def test_knn_reverse_lists():
    from sug_amd import ops
    g = torch.Generator().manual_seed(4)
    idx = torch.randint(0, 300, (3, 300, 20), generator=g, dtype=torch.int32).cuda()
    off, ent = ops.knn_reverse(idx)
    off, ent, idxc = off.cpu(), ent.cpu(), idx.cpu()
    for b in range(3):
        flat = idxc[b].reshape(-1)
        assert off[b, -1] == flat.numel()
        for m in (0, 1, 17, 299):
            lst = ent[b, off[b, m]:off[b, m + 1]]
            want = torch.nonzero(flat == m).reshape(-1).to(torch.int32)
            assert torch.equal(lst, want)


@pytest.mark.parametrize('B,N', [(64, 1024), (5, 1000), (40, 512), (2, 96)])
def test_knn_reverse_lists_full(B, N):
    """Every destination of every cloud: the reverse lists are the stable sort of the flat neighbour array
    (entries in ascending order), including hub points with lists far longer than k (wave rank sort) and
    destinations nobody lists."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(B + N)
    k = 20
    idx = torch.randint(0, N, (B, N, k), generator=g, dtype=torch.int32)
    idx[:, :, 0] = torch.randint(0, 3, (B, N), generator=g, dtype=torch.int32)      # three hubs with ~N/3 entries
    idx[:, : N // 2, 1] = N - 1                                                     # one with N/2
    off, ent = ops.knn_reverse(idx.cuda())
    off, ent = off.cpu().long(), ent.cpu().long()
    flat = idx.reshape(B, -1).long()
    want_ent = torch.argsort(flat, dim=1, stable=True)
    assert torch.equal(ent, want_ent)
    counts = torch.zeros(B, N, dtype=torch.long).scatter_add_(1, flat, torch.ones_like(flat))
    want_off = torch.cat((torch.zeros(B, 1, dtype=torch.long), counts.cumsum(1)), 1)
    assert torch.equal(off, want_off)


def test_fps_golden(ops_golden):
    from sug_amd import ops
    out = ops.fps(rows(ops_golden['fps_cf_xyz']), 64, ops_golden['fps_cf_start'])
    assert torch.equal(out.cpu().long(), ops_golden['fps_cf_idx'])
    out = ops.fps(ops_golden['fps_cl_xyz'].cuda(), 512, ops_golden['fps_cl_start'])
    assert torch.equal(out.cpu().long(), ops_golden['fps_cl_idx'])
    out = ops.fps(rows(ops_golden['fps_dup_xyz']), 32, ops_golden['fps_dup_start'])
    assert torch.equal(out.cpu().long(), ops_golden['fps_dup_idx'])


def test_fps_draws_start_like_reference(ops_golden):
    """The mirror function must consume the CPU generator exactly like the reference."""
    from sug_amd.model import point_utils, pointnet2_utils
    torch.manual_seed(777)
    out = point_utils.farthest_point_sample(ops_golden['fps_cf_xyz'].cuda(), 64)
    assert out.dtype == torch.int64 and torch.equal(out.cpu(), ops_golden['fps_cf_idx'])
    torch.manual_seed(778)
    out = pointnet2_utils.farthest_point_sample(ops_golden['fps_cl_xyz'].cuda(), 512)
    assert torch.equal(out.cpu(), ops_golden['fps_cl_idx'])


@pytest.mark.parametrize('N,npoint', [(100, 100), (1024, 1), (4096, 64), (777, 33)])
def test_fps_vs_oracle_shapes(N, npoint):
    from sug_amd import ops
    g = torch.Generator().manual_seed(N + npoint)
    xyz = O.synth_clouds(3, N, g).squeeze(-1)
    start = torch.randint(0, N, (3,), generator=g)
    out = ops.fps(rows(xyz), npoint, start)
    assert torch.equal(out.cpu().long(), O.fps_cf(xyz, npoint, start))


def test_ball_query_golden(ops_golden):
    from sug_amd import ops
    xyz, new = ops_golden['fps_cf_xyz'], ops_golden['bq_cf_new']
    out = ops.ball_query(rows(xyz), rows(new), 0.3, 64)
    assert torch.equal(out.cpu().long(), ops_golden['bq_cf_r03'])
    out = ops.ball_query(rows(xyz), rows(new), 0.05, 64)
    assert torch.equal(out.cpu().long(), ops_golden['bq_cf_small_r'])
    x1 = ops_golden['fps_cl_xyz'][:1]
    nx1 = O.gather_cl(x1, ops_golden['fps_cl_idx'][:1])
    out = ops.ball_query(x1.cuda(), nx1.cuda(), 0.2, 32)
    assert torch.equal(out.cpu().long(), ops_golden['bq_cl_r02'])


def test_ball_query_no_hit_rows_yield_N():
    from sug_amd import ops
    xyz = torch.rand(1, 50, 3)
    q = torch.full((1, 2, 3), 10.0)
    out = ops.ball_query(xyz.cuda(), q.cuda(), 0.1, 8).cpu().long()
    ref = O.ball_query_cl(0.1, 8, xyz, q)
    assert torch.equal(out, ref) and int(out.min()) == 50


def test_knn_query_golden(ops_golden):
    from sug_amd import ops
    xyz, moved = ops_golden['fps_cf_xyz'], ops_golden['bq_cf_moved']
    idx, dist = ops.knn_query(rows(xyz), rows(moved), 64, want_dist=True)
    score = O.sqdist_cf(moved, xyz)
    assert_same_or_tied(idx, ops_golden['bq_cf_knn'], score, ulps=0)
    assert torch.equal(dist.cpu(), torch.gather(score, 2, idx.cpu().long()))


@pytest.mark.parametrize('N,S,k', [(2048, 5, 16), (300, 7, 64), (64, 3, 64)])
def test_knn_query_vs_oracle(N, S, k):
    from sug_amd import ops
    g = torch.Generator().manual_seed(N + S)
    xyz = O.synth_clouds(2, N, g).squeeze(-1)
    q = torch.rand(2, 3, S, generator=g) - 0.5
    idx = ops.knn_query(rows(xyz), rows(q), k)
    assert_same_or_tied(idx, O.ball_query_cf(None, k, xyz, q), O.sqdist_cf(q, xyz), ulps=0)


def test_three_nn_and_upsample_golden(ops_golden):
    from sug_amd import ops
    from sug_amd.model import point_utils
    xyz, moved = ops_golden['fps_cf_xyz'], ops_golden['bq_cf_moved']
    idx3, d3 = ops.three_nn_raw(rows(xyz), rows(moved))
    d, i = O.sqdist_cf(xyz, moved).sort(dim=-1)
    assert torch.equal(idx3.cpu().long(), i[:, :, :3])
    assert torch.equal(d3.cpu(), d[:, :, :3])
    out = point_utils.upsample_inter(xyz.cuda(), moved.cuda(), ops_golden['up_p1'].cuda(), ops_golden['up_p2'].cuda(), 3)
    torch.testing.assert_close(out.cpu(), ops_golden['up_out'], rtol=1e-5, atol=1e-6)


def test_gather_group_and_grads(ops_golden):
    from sug_amd import ops
    g = torch.Generator().manual_seed(8)
    feat = torch.randn(2, 200, 24, generator=g)
    idx = torch.randint(0, 200, (2, 50, 6), generator=g)
    f_gpu = feat.cuda().requires_grad_(True)
    f_cpu = feat.clone().requires_grad_(True)
    out = ops.gather_rows(f_gpu, idx.cuda())
    ref = O.gather_cl(f_cpu, idx)
    assert torch.equal(out.detach().cpu(), ref.detach())
    w = torch.randn(ref.shape, generator=g)
    (out * w.cuda()).sum().backward()
    (ref * w).sum().backward()
    torch.testing.assert_close(f_gpu.grad.cpu(), f_cpu.grad, rtol=1e-5, atol=1e-5)
    # grouped max + its gradient
    f_gpu.grad = None
    f_cpu.grad = None
    out = ops.group_max(f_gpu, idx.cuda())
    ref = O.gather_cl(f_cpu, idx).max(dim=2)[0]
    assert torch.equal(out.detach().cpu(), ref.detach())
    w = torch.randn(ref.shape, generator=g)
    (out * w.cuda()).sum().backward()
    (ref * w).sum().backward()
    torch.testing.assert_close(f_gpu.grad.cpu(), f_cpu.grad, rtol=1e-5, atol=1e-5)


def test_index_points_and_graph_feature_mirrors(ops_golden):
    from sug_amd.model import model_utils
    out = model_utils.get_graph_feature(ops_golden['gf_x'].cuda(), k=4, idx=ops_golden['gf_idx'].cuda())
    assert torch.equal(out.cpu(), ops_golden['gf_out'])
    x = ops_golden['knn_grid_x']
    assert torch.equal(model_utils.knn(x.cuda(), 20).cpu(), ops_golden['knn_grid_idx'])


def test_sample_and_group_golden(ops_golden):
    from sug_amd.model import pointnet2_utils as p2
    xs = ops_golden['fps_cl_xyz'][:, :512].contiguous()
    pts = ops_golden['sag_pts']
    torch.manual_seed(779)
    nxyz, npts = p2.sample_and_group(128, 0.4, 64, xs.cuda(), pts.cuda())
    assert torch.equal(nxyz.cpu(), ops_golden['sag_new_xyz'])
    assert torch.equal(npts[:, :, 0].cpu(), ops_golden['sag_new_points_row0'])
    torch.testing.assert_close(npts.sum(dim=2).cpu(), ops_golden['sag_new_points_sum'], rtol=1e-5, atol=1e-5)


def test_chamfer_vs_oracle():
    from sug_amd import ops
    g = torch.Generator().manual_seed(9)
    a, b = O.synth_clouds(3, 1024, g), O.synth_clouds(3, 1024, g)
    d = ops.chamfer(a.squeeze(-1).transpose(1, 2).cuda(), b.squeeze(-1).transpose(1, 2).cuda())
    pa, pb = a.squeeze(-1).transpose(1, 2), b.squeeze(-1).transpose(1, 2)
    dm = ((pa[:, :, None] - pb[:, None]) ** 2).sum(-1)
    ref = dm.min(2)[0].mean(1) + dm.min(1)[0].mean(1)
    torch.testing.assert_close(d.cpu(), ref, rtol=1e-5, atol=1e-7)


def test_adapt_layer_fused_glue_vs_oracle():
    """adapt_layer_off.rows (fused node-offset + interp3/concat kernels, forward and backward)
    against the oracle's adapt_layer_off on identical weights and FPS start."""
    from sug_amd.model.model_utils import adapt_layer_off
    g = torch.Generator().manual_seed(21)
    B, N = 3, 512
    loc = O.synth_clouds(B, N, g).squeeze(-1)                       # [B,3,N]
    fea = torch.randn(B, 64, N, 1, generator=g)
    m = adapt_layer_off()
    sd = O.fill_params({k: tuple(v.shape) for k, v in m.state_dict().items()}, 3)
    m.load_state_dict(sd)
    m = m.cuda().train()
    start = torch.randint(0, N, (B,), generator=g)
    p = O.as_params({'a.' + k: v for k, v in sd.items()})
    fo = fea.clone().requires_grad_(True)
    out_o, node_o, off_o = O.adapt_layer_off(p, 'a.', fo, loc, True, start)
    probe = torch.randn(out_o.shape, generator=g)
    pn = torch.randn(node_o.shape, generator=g)
    ((out_o * probe).sum() + (node_o * pn).sum() + off_o.sum()).backward()
    from sug_amd import ops
    ops.START_PROVIDER = lambda b, n: start
    try:
        fg = fea.cuda().requires_grad_(True)
        out, node, off = m(fg, loc.cuda())
    finally:
        ops.START_PROVIDER = None
    ((out * probe.cuda()).sum() + (node * pn.cuda()).sum() + off.sum()).backward()
    torch.testing.assert_close(off.detach().cpu(), off_o.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(node.detach().cpu(), node_o.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out.detach().cpu(), out_o.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(fg.grad.cpu(), fo.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(m.pred_offset[0].weight.grad.cpu(), p['a.pred_offset.0.weight'].grad, rtol=2e-3, atol=2e-3)


def test_adam_multi_tensor_vs_torch():
    """sug_amd.optim.Adam (sug_adam_step) against torch.optim.Adam: same update, same state layout."""
    from sug_amd.optim import Adam
    g = torch.Generator().manual_seed(11)
    shapes = [(7,), (64, 3, 1, 1), (4097,), (512, 512), (1000, 33), (1,), (4096,), (3, 5, 7)]
    mine = [torch.randn(*s, generator=g).cuda().requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]
    groups = lambda ps: [{'params': ps[:3]}, {'params': ps[3:6], 'lr': 3e-3}, {'params': ps[6:]}]
    om = Adam(groups(mine), lr=1e-3, weight_decay=5e-5)
    ot = torch.optim.Adam(groups(ref), lr=1e-3, weight_decay=5e-5)
    for it in range(4):
        for i, (a, b) in enumerate(zip(mine, ref)):
            if it == 2 and i == 1:          # a tensor without gradient in one step: skipped, its step count lags
                a.grad = b.grad = None
                continue
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** (i % 3 - 1))
            a.grad, b.grad = gr.clone(), gr.clone()
        om.step()
        ot.step()
    for a, b in zip(mine, ref):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7)
    sm, st = om.state_dict(), ot.state_dict()
    assert sm['state'].keys() == st['state'].keys()
    for k in sm['state']:
        assert float(sm['state'][k]['step']) == float(st['state'][k]['step'])
        for name in ('exp_avg', 'exp_avg_sq'):          # fp32 rounding (torch contracts to FMA, this build does not)
            want = st['state'][k][name]
            torch.testing.assert_close(sm['state'][k][name], want, rtol=2e-6, atol=1e-6 * float(want.abs().max()))
    ot.load_state_dict(sm)                  # torch's Adam accepts the state ...
    om.load_state_dict(st)                  # ... and the other way round
    for a, b in zip(mine, ref):
        gr = torch.randn(a.shape, generator=g).cuda()
        a.grad, b.grad = gr.clone(), gr.clone()
    om.step()
    ot.step()
    for a, b in zip(mine, ref):
        torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-7)


@pytest.mark.parametrize('N,S,k', [(1024, 1024, 16), (256, 64, 16), (16, 4, 16), (300, 7, 5)])
def test_knn_query_direct_form_vs_oracle(N, S, k):
    """Point Transformer neighbours: argsort of sum((q - p)^2) (Ptran_transformer.py:32-33), bit-exact."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(N + S + k)
    xyz = torch.rand(2, N, 3, generator=g) * 2 - 1
    qry = xyz[:, :S].contiguous() if S <= N else torch.rand(2, S, 3, generator=g)
    want = O.sqdist_direct(qry, xyz).argsort()[:, :, :k]
    got = ops.knn_query(xyz.cuda(), qry.cuda(), k, direct=True)
    assert torch.equal(got.cpu().long(), want)


def test_knn_full_size_properties():
    """BASELINE size (64 clouds x 1024 points, C=64, k=20): size-independent properties of a kNN
    list -- in range, no duplicates, the point itself first, scores non-increasing along k."""
    from sug_amd import ops
    x = torch.randn(64, 1024, 64, generator=torch.Generator().manual_seed(5)).cuda()
    idx = ops.knn(x, 20).long()
    assert int(idx.min()) >= 0 and int(idx.max()) < 1024
    assert torch.equal(idx[:, :, 0], torch.arange(1024, device='cuda').expand(64, -1))
    srt = idx.sort(dim=2)[0]
    assert bool((srt[:, :, 1:] != srt[:, :, :-1]).all()), 'duplicate neighbour'
    nbr = torch.gather(x.unsqueeze(1).expand(-1, 1024, -1, -1), 2, idx.unsqueeze(-1).expand(-1, -1, -1, 64))
    d = ((nbr.double() - x.double().unsqueeze(2)) ** 2).sum(-1)
    assert bool((d[:, :, 1:] >= d[:, :, :-1] - 1e-4).all()), 'neighbours not ordered by distance'
    # idempotence: a second launch gives the same lists
    assert torch.equal(ops.knn(x, 20).long(), idx)


def test_knn_nan_features_stay_in_range():
    """Clouds with NaN rows (a diverged training step): every slot still holds a valid index, so the
    gathers downstream cannot fault; clean clouds of the same batch are unaffected."""
    from sug_amd import ops
    x = torch.randn(4, 256, 64, generator=torch.Generator().manual_seed(9))
    clean = ops.knn(x.cuda(), 20)
    x[1, 17] = float('nan')
    x[2] = float('nan')
    idx = ops.knn(x.cuda(), 20)
    assert int(idx.min()) >= 0 and int(idx.max()) < 256
    assert torch.equal(idx[0], clean[0]) and torch.equal(idx[3], clean[3])


def test_bad_arguments_raise():
    from sug_amd import ops
    x = torch.randn(2, 64, 3).cuda()
    with pytest.raises(RuntimeError):
        ops.knn_query(x, x, 65)                       # k > 64
    with pytest.raises(RuntimeError):
        ops.knn(torch.randn(2, 64, 3), 4)             # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        ops.three_nn_raw(x, x[:, :2].contiguous())    # fewer than 3 candidates
    with pytest.raises(RuntimeError):
        with ops.bn_groups(2):
            ops.bn_act_rows(torch.randn(3, 8).cuda(), torch.nn.BatchNorm1d(8).cuda(), 0.0)   # 3 rows, 2 groups


@pytest.mark.parametrize('force', ['0', '1', '2'])
def test_knn_packed_key_paths_match_exact_lists(force):
    """The packed-key consumer of sug_knn (knn_pc.hip) against the neighbour lists of the exact (score, index)
    list consumer it replaced (tests/golden/knn_pc_hashes.json, SHA-256 per case; those lists are the ones the
    oracle tests above pin).  force = 0: as shipped (keys in distinct buckets are final, the others are re-ranked
    exactly); 1: every query through the exact re-rank; 2: every query through the exact rescan of its cloud.
    Random and clustered features, exact duplicates, padded clouds, lattices (ties everywhere), N not a multiple
    of the tile, k = 16 and 20."""
    import importlib.util, json, os
    from sug_amd import ops
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    spec = importlib.util.spec_from_file_location('make_knn_hashes', os.path.join(here, 'make_knn_hashes.py'))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    want = json.load(open(os.path.join(here, 'knn_pc_hashes.json')))
    keep = os.environ.get('SUG_KNN_FORCE')
    os.environ['SUG_KNN_FORCE'] = force
    bad = []
    try:
        for name, x in mk.cases().items():
            if force == '2' and x.shape[1] * x.shape[2] > 1024 * 64:
                continue                      # the scalar rescan of every query: small cases only
            for k in (20, 16):
                if mk.digest(ops.knn(x.cuda(), k)) != want['%s_k%d' % (name, k)]:
                    bad.append('%s_k%d' % (name, k))
    finally:
        if keep is None:
            del os.environ['SUG_KNN_FORCE']
        else:
            os.environ['SUG_KNN_FORCE'] = keep
    assert not bad, bad


@pytest.mark.parametrize('rows,C,slope', [(64, 512, 0.2), (64, 256, 0.2), (7, 1024, 0.0), (130, 100, 0.2)])
def test_ln_act_vs_torch(rows, C, slope):
    """LayerNorm + (Leaky)ReLU of the FC heads (fc_layer, model_utils.py:35-57) in one launch: forward within 1e-5 of
    nn.LayerNorm + F.leaky_relu, gradients (input, weight, bias) within 1e-4 relative; bit-reproducible."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).cuda()
    ln_r, ln_k = torch.nn.LayerNorm(C).cuda(), torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        for ln in (ln_r, ln_k):
            ln.weight.copy_(torch.linspace(-1.0, 1.5, C))
            ln.bias.copy_(torch.linspace(-0.3, 0.3, C))
    probe = torch.randn(rows, C, generator=g).cuda()
    xr, xk = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ref = torch.nn.functional.leaky_relu(ln_r(xr), slope)
    (ref * probe).sum().backward()
    assert ops.ln_act_supported(xk, ln_k)
    out = ops.ln_act(xk, ln_k, slope)
    (out * probe).sum().backward()
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)

    def rel(a, b):
        return float((a - b).norm() / b.norm().clamp_min(1e-12))

    assert rel(xk.grad, xr.grad) < 1e-4 and rel(ln_k.weight.grad, ln_r.weight.grad) < 1e-4
    assert rel(ln_k.bias.grad, ln_r.bias.grad) < 1e-4
    g1 = (xk.grad.clone(), ln_k.weight.grad.clone())
    xk.grad, ln_k.weight.grad, ln_k.bias.grad = None, None, None
    (ops.ln_act(xk, ln_k, slope) * probe).sum().backward()
    assert torch.equal(g1[0], xk.grad) and torch.equal(g1[1], ln_k.weight.grad)


@pytest.mark.parametrize('N', [20, 21, 33, 64, 100])
@pytest.mark.parametrize('C', [3, 64])
def test_knn_tiny_clouds(N, C):
    """Clouds smaller than a candidate tile / barely larger than k: the packed-key list never fills its spare slots,
    the first tile carries rows past N; neighbour lists equal the oracle's."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(3, C, N, generator=g)
    idx = ops.knn(x.transpose(1, 2).contiguous().cuda(), 20).cpu().long()
    ref = O.knn_idx(x, 20)
    assert_same_or_tied(idx, ref, O.knn_neg_dist(x), ulps=8 if C > 3 else 0)


@pytest.mark.gpu
def test_small_assembly_kernels_match_their_torch_formulations():
    """sug_mmd_assemble, sug_edge_weight_split, sug_gate_*: single-launch forms of short torch op chains of the
    reference (model/mmd.py:56-66, model_utils.py:188-210 folded into the weights, Model.py:28-34) -- bit for bit
    forward, and the same gradients."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(11)
    dev = 'cuda'
    # soft-MMD operand
    m, D = 7, 301
    fs = torch.randn(m, D, generator=g).to(dev).requires_grad_(True)
    ft = torch.randn(m + 0, 2 * D, generator=g).to(dev)[:, ::2][:, :D].contiguous().requires_grad_(True)
    ls, lt = torch.randint(0, 10, (m,), generator=g).to(dev), torch.randint(0, 10, (m,), generator=g).to(dev)
    Z = ops.mmd_assemble(fs, ft, ls, lt, 2.5)
    oh = torch.zeros(2 * m, 10, device=dev).scatter_(1, torch.cat((ls, lt)).view(-1, 1), 1.0) * 2.5
    Zr = torch.cat((torch.cat((fs, ft), 0), oh), dim=1)
    assert torch.equal(Z, Zr)
    probe = torch.randn(2 * m, D + 10, generator=g).to(dev)
    ga = torch.autograd.grad((Z * probe).sum(), (fs, ft))
    gb = torch.autograd.grad((Zr * probe).sum(), (fs, ft))
    assert all(torch.equal(a, b) for a, b in zip(ga, gb))
    # EdgeConv weight split
    W = torch.randn(24, 2 * 13, generator=g).to(dev).requires_grad_(True)
    S = ops.edge_weight_split(W)
    Sr = torch.cat((W[:, :13], W[:, 13:] - W[:, :13]), dim=0)
    assert torch.equal(S, Sr)
    probe = torch.randn(48, 13, generator=g).to(dev)
    assert torch.equal(torch.autograd.grad((S * probe).sum(), W)[0], torch.autograd.grad((Sr * probe).sum(), W)[0])
    # ... of several layers in one launch each way (the DGCNN encoder's four); a split that is not used hands no gradient
    shapes = [(64, 3), (64, 64), (128, 64), (256, 128), (5, 7)]
    Ws = [torch.randn(co, 2 * c, generator=g).to(dev).requires_grad_(True) for co, c in shapes]
    Ss = ops.edge_weight_split_multi(Ws)
    probes = [torch.randn(2 * co, c, generator=g).to(dev) for co, c in shapes]
    used = [0, 1, 3, 4]
    gm = torch.autograd.grad(sum((Ss[i] * probes[i]).sum() for i in used), Ws, allow_unused=True)
    for i, (Wi, Si) in enumerate(zip(Ws, Ss)):
        one = ops.edge_weight_split(Wi)
        assert torch.equal(Si, one)
        if i in used:
            assert torch.equal(gm[i], torch.autograd.grad((one * probes[i]).sum(), Wi)[0])
        else:
            assert gm[i] is None
    # CALayer gate
    x = torch.randn(5, 4096, generator=g).to(dev).requires_grad_(True)
    z = (3 * torch.randn(5, 4096, generator=g)).to(dev).requires_grad_(True)
    o = ops.gate(x, z)
    orf = x * torch.sigmoid(z) + x
    torch.testing.assert_close(o, orf, rtol=2e-7, atol=0)
    probe = torch.randn(5, 4096, generator=g).to(dev)
    ga = torch.autograd.grad((o * probe).sum(), (x, z))
    gb = torch.autograd.grad((orf * probe).sum(), (x, z))
    for a, b in zip(ga, gb):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize('direct', [False, True])
def test_knn_query_ties_beyond_candidate_buffer(direct):
    """More than 128 candidates at the k-th distance (duplicated points): the select path hands over to the
    round-by-round path; equal distances come out in index order (full-sort semantics, PTran_utils.py:99-136 /
    point_utils.py:107-108)."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(3)
    N, k = 1024, 64
    xyz = torch.rand(2, N, 3, generator=g)
    dup = torch.randperm(N, generator=g)[:300].sort().values
    xyz[0, dup] = xyz[0, dup[0]].clone()
    xyz[1, dup[:140]] = xyz[1, dup[5]].clone()
    qry = torch.stack((xyz[0, dup[0]], xyz[1, dup[5]])).view(2, 1, 3)
    idx = ops.knn_query(xyz.cuda(), qry.cuda(), k, direct=direct).cpu()
    assert idx[0, 0].tolist() == dup[:k].tolist()
    assert idx[1, 0].tolist() == dup[:k].tolist()
    # and a spread of k / N against a stable sort of the same distances (well separated random points)
    for N2, k2 in ((100, 64), (257, 16), (2048, 16), (640, 32), (4096, 64), (64, 64)):
        x = torch.rand(3, N2, 3, generator=g).cuda()
        q = torch.rand(3, 5, 3, generator=g).cuda()
        got, dist = ops.knn_query(x, q, k2, want_dist=True, direct=direct)
        d = ((q[:, :, None, :] - x[:, None, :, :]) ** 2).sum(-1)
        ref = d.sort(dim=-1, stable=True).indices[..., :k2]
        same = (got.long() == ref)
        # near-ties may swap between the two distance formulas: the distances at disagreeing ranks must agree to rounding
        dg = torch.gather(d, 2, got.long())
        dr = torch.gather(d, 2, ref)
        assert torch.all(same | ((dg - dr).abs() <= 1e-6)), (N2, k2)
        assert torch.all(dist[..., 1:] >= dist[..., :-1] - 0.0)
        for b in range(3):
            for s in range(5):
                assert len(set(got[b, s].tolist())) == k2


@pytest.mark.gpu
@pytest.mark.parametrize('N,S,ns,r', [(4096, 37, 32, 0.12), (5000, 9, 16, 0.1), (130, 3, 64, 0.4), (2048, 600, 32, 0.2)])
def test_ball_query_lds_and_global_paths_vs_oracle(N, S, ns, r):
    """Clouds up to 4096 points are staged in LDS (several queries per wave), larger ones take the global-memory
    kernel: both must give the oracle's ascending-index lists (point_utils.py:86-109), padded with the first hit."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(N + S)
    xyz = torch.rand(3, N, 3, generator=g) * 2 - 1
    q = xyz[:, torch.randperm(N, generator=g)[:S]] + 0.01 * torch.randn(3, S, 3, generator=g)
    out = ops.ball_query(xyz.cuda(), q.cuda(), r, ns).cpu().long()
    ref = O.ball_query_cl(r, ns, xyz, q)
    # a candidate within an ulp of the radius may differ between the two evaluations of the same formula: none expected
    assert torch.equal(out, ref)


@pytest.mark.gpu
@pytest.mark.parametrize('R,C,dt', [(65536, 64, torch.float32), (16384, 1024, torch.float32), (8192, 512, torch.float16),
                                     (77, 3, torch.float32), (1, 130, torch.float16)])
def test_colsum_matches_torch(R, C, dt):
    """sug_colsum (bias gradients without the memset of torch's tall reductions) against a float64 sum."""
    from sug_amd import ops
    g = torch.Generator().manual_seed(R + C)
    x = torch.randn(R, C, generator=g).to(dt).cuda()
    ref = x.double().sum(dim=0)
    out = ops.colsum(x)
    assert out.dtype == torch.float32
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-4 * (R ** 0.5))
    torch.testing.assert_close(ops.colsum(x, -1.0).double(), -ref, rtol=1e-5, atol=1e-4 * (R ** 0.5))
    # strided rows (a column slice of a wider buffer)
    wide = torch.randn(R, C + 5, generator=g).to(dt).cuda()
    torch.testing.assert_close(ops.colsum(wide[:, 2:2 + C]).double(), wide[:, 2:2 + C].double().sum(dim=0), rtol=1e-5,
                               atol=1e-4 * (R ** 0.5))

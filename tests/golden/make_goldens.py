#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

Usage (from the repo root, CPU only, ~2 min):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

* puts /root/reference on sys.path, registers empty stub modules for the
  absent third-party imports of model/mmd.py (tkinter/turtle, chamfer_distance,
  h5py, easydict, tensorboardX) and redirects the reference's hard-coded
  'cuda' device strings to 'cpu' (a harness-side monkey-patch; no reference
  file is modified or copied);
* runs reference functions on seeded inputs and stores inputs + outputs as
  small fixtures;
* asserts along the way that oracle/ref_cpu.py (the CPU restatement that
  travels to the GPU box) reproduces the reference on every case.

The fixtures are data only.  The reference itself never leaves this container.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('SUG_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

for name in ('tkinter', 'turtle', 'chamfer_distance', 'h5py', 'easydict', 'tensorboardX'):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules['turtle'].distance = None
sys.modules['chamfer_distance'].ChamferDistance = None
sys.modules['easydict'].EasyDict = dict
sys.modules['tensorboardX'].SummaryWriter = object

# 'cuda' -> 'cpu' redirect for the reference's hard-coded device strings
_orig_to = torch.Tensor.to
_orig_arange = torch.arange


def _cpu_dev(d):
    if isinstance(d, str) and d.startswith('cuda'):
        return 'cpu'
    return d


def _to(self, *a, **kw):
    a = tuple(_cpu_dev(x) for x in a)
    if 'device' in kw:
        kw['device'] = _cpu_dev(kw['device'])
    return _orig_to(self, *a, **kw)


def _arange(*a, **kw):
    if 'device' in kw:
        kw['device'] = _cpu_dev(kw['device'])
    return _orig_arange(*a, **kw)


torch.Tensor.to = _to
torch.arange = _arange

import model.model_utils as r_mu          # noqa: E402  (reference)
import model.point_utils as r_pu          # noqa: E402
import model.pointnet2_utils as r_p2      # noqa: E402
import model.Model as r_M                 # noqa: E402
import model.model_pointnet as r_mp       # noqa: E402
import model.mmd as r_mmd                 # noqa: E402

from oracle import ref_cpu as O           # noqa: E402

torch.set_num_threads(8)


def same(a, b, what, tol=0.0):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if a.dtype.is_floating_point:
        err = (a - b).abs().max().item() if a.numel() else 0.0
        scale = max(1.0, b.abs().max().item()) if b.numel() else 1.0
        assert err <= tol * scale, '%s: restatement differs from reference by %g' % (what, err)
    else:
        assert torch.equal(a, b), '%s: restatement differs from reference' % what


def grid_cloud(B, N, gen, bits=10):
    """Points on the 2^-bits grid in [-1,1): every product / sum in the expanded
    distance is exact in fp32, so kNN ranks are rounding-independent."""
    out = []
    for _ in range(B):
        while True:
            q = torch.randint(-2 ** bits, 2 ** bits, (N * 2, 3), generator=gen)
            q = torch.unique(q, dim=0)
            if q.shape[0] >= N:
                q = q[torch.randperm(q.shape[0], generator=gen)[:N]]
                break
        out.append(q.float() / 2 ** bits)
    return torch.stack(out).permute(0, 2, 1).contiguous()       # [B,3,N]


def save(name, **arrs):
    path = os.path.join(HERE, name)
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if v.dtype == np.int64 and v.size and v.max() < 2 ** 15 and v.min() >= 0 and v.ndim > 0:
            v = v.astype(np.int16)              # indices: keep fixtures small
        conv[k] = v
    np.savez_compressed(path, **conv)
    print('wrote %-28s %7.1f KB' % (name, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ operators
def gen_ops():
    g = torch.Generator().manual_seed(20260301)
    out = {}

    # kNN on xyz: grid cloud (rank is exact), random normalised cloud, padded cloud (ties)
    xg = grid_cloud(2, 256, g)
    out['knn_grid_x'] = xg
    out['knn_grid_idx'] = r_mu.knn(xg, 20)
    same(O.knn_idx(xg, 20), out['knn_grid_idx'], 'knn grid')
    xr = O.synth_clouds(2, 1024, g).squeeze(-1)
    out['knn_rand_x'] = xr
    out['knn_rand_idx'] = r_mu.knn(xr, 20)
    same(O.knn_idx(xr, 20), out['knn_rand_idx'], 'knn rand')
    xp = O.synth_clouds(1, 512, g).squeeze(-1)
    xp[:, :, 460:] = 0.0                       # dataloader-style zero padding (data/dataloader.py:316-321)
    out['knn_pad_x'] = xp
    out['knn_pad_idx'] = r_mu.knn(xp, 20)
    # feature-space kNN (C=64), inputs on a coarse grid so the rank is exact
    xf = torch.randint(-8, 9, (2, 64, 256), generator=g).float() / 8
    xf = xf + torch.arange(256).view(1, 1, 256).float() / 4096      # break ties exactly
    out['knn_feat_x'] = xf
    out['knn_feat_idx'] = r_mu.knn(xf, 20)
    same(O.knn_idx(xf, 20), out['knn_feat_idx'], 'knn feat')
    # graph feature with a given idx
    gf_x = torch.randn(2, 5, 64, generator=g)
    gf_idx = torch.randint(0, 64, (2, 64, 4), generator=g)
    out['gf_x'], out['gf_idx'] = gf_x, gf_idx
    out['gf_out'] = r_mu.get_graph_feature(gf_x, k=4, idx=gf_idx)
    same(O.graph_feature(gf_x, 4, gf_idx), out['gf_out'], 'graph_feature')

    # FPS ([B,C,N] and [B,N,C]) -- explicit seed so the CPU-generator draw is pinned too
    xyz = O.synth_clouds(2, 1024, g).squeeze(-1)
    out['fps_cf_xyz'] = xyz
    torch.manual_seed(777)
    out['fps_cf_idx'] = r_pu.farthest_point_sample(xyz, 64)
    torch.manual_seed(777)
    out['fps_cf_start'] = torch.randint(0, 1024, (2,), dtype=torch.long)
    same(O.fps_cf(xyz, 64, out['fps_cf_start']), out['fps_cf_idx'], 'fps_cf')
    xyz2 = O.synth_clouds(2, 2048, g).squeeze(-1).permute(0, 2, 1).contiguous()
    out['fps_cl_xyz'] = xyz2
    torch.manual_seed(778)
    out['fps_cl_idx'] = r_p2.farthest_point_sample(xyz2, 512)
    torch.manual_seed(778)
    out['fps_cl_start'] = torch.randint(0, 2048, (2,), dtype=torch.long)
    same(O.fps_cl(xyz2, 512, out['fps_cl_start']), out['fps_cl_idx'], 'fps_cl')
    # FPS with duplicated points (arg-max ties -> lowest index)
    xd = xyz[:, :, :256].clone()
    xd[:, :, 128:] = xd[:, :, :128]
    out['fps_dup_xyz'] = xd
    out['fps_dup_start'] = torch.tensor([5, 200])
    _ri = torch.randint
    torch.randint = lambda *a, **kw: out['fps_dup_start'].clone()      # pin the reference's start draw
    try:
        out['fps_dup_idx'] = r_pu.farthest_point_sample(xd, 32)
    finally:
        torch.randint = _ri
    same(O.fps_cf(xd, 32, out['fps_dup_start']), out['fps_dup_idx'], 'fps dup')

    # ball query, [B,C,N]: radius and kNN-by-sort modes (adapt layer sizes)
    new_xyz = r_pu.index_points(xyz, out['fps_cf_idx'])
    out['bq_cf_new'] = new_xyz
    out['bq_cf_r03'] = r_pu.query_ball_point(0.3, 64, xyz, new_xyz)
    same(O.ball_query_cf(0.3, 64, xyz, new_xyz), out['bq_cf_r03'], 'ball_query_cf r')
    moved = new_xyz + 0.01 * torch.randn(new_xyz.shape, generator=g)
    out['bq_cf_moved'] = moved
    out['bq_cf_knn'] = r_pu.query_ball_point(None, 64, xyz, moved)
    same(O.ball_query_cf(None, 64, xyz, moved), out['bq_cf_knn'], 'ball_query_cf knn')
    out['bq_cf_small_r'] = r_pu.query_ball_point(0.05, 64, xyz, new_xyz)      # many short rows -> padding
    # ball query, [B,N,C] (PointNet++ sa1 sizes, one cloud)
    x1 = xyz2[:1]
    nx1 = r_p2.index_points(x1, out['fps_cl_idx'][:1])
    out['bq_cl_r02'] = r_p2.query_ball_point(0.2, 32, x1, nx1)
    same(O.ball_query_cl(0.2, 32, x1, nx1), out['bq_cl_r02'], 'ball_query_cl')
    # sample_and_group (sa2 sizes)
    pts = torch.randn(2, 512, 16, generator=g)
    xs = xyz2[:, :512].contiguous()
    torch.manual_seed(779)
    nxyz, npts = r_p2.sample_and_group(128, 0.4, 64, xs, pts)
    torch.manual_seed(779)
    st = torch.randint(0, 512, (2,), dtype=torch.long)
    o_nxyz, o_npts = O.sample_and_group(128, 0.4, 64, xs, pts, st)
    same(o_nxyz, nxyz, 'sag xyz')
    same(o_npts, npts, 'sag pts')
    out['sag_pts'], out['sag_start'] = pts, st
    out['sag_new_xyz'] = nxyz
    out['sag_new_points_sum'] = npts.sum(dim=2)                  # [B,128,19] checksum over nsample
    out['sag_new_points_row0'] = npts[:, :, 0]

    # upsample_inter + square_distance
    p1 = torch.randn(2, 8, 1024, generator=g)
    p2 = torch.randn(2, 8, 64, generator=g)
    out['up_p1'], out['up_p2'] = p1, p2
    out['up_out'] = r_pu.upsample_inter(xyz, moved, p1, p2, k=3)
    same(O.upsample_inter(xyz, moved, p1, p2, 3), out['up_out'], 'upsample_inter')
    out['sqd_cf'] = r_pu.square_distance(new_xyz, xyz)[:, :4]
    same(O.sqdist_cf(new_xyz, xyz)[:, :4], out['sqd_cf'], 'sqdist')
    save('ops.npz', **out)


# ------------------------------------------------------------------ models
def _load(mod, seed):
    sd = mod.state_dict()
    filled = O.fill_params({k: tuple(v.shape) for k, v in sd.items()}, seed)
    mod.load_state_dict(filled)
    return {k: v.clone() for k, v in filled.items()}


def _no_dropout(mod):
    for m in mod.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0


def _probe(shape, tag):
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % (2 ** 31))
    return torch.randn(shape, generator=g)


def gen_model(name, B, N, seed, fname):
    g = torch.Generator().manual_seed(seed)
    x = O.synth_clouds(B, N, g)
    net = r_M.Net_MDA(name)
    p0 = _load(net, seed)
    _no_dropout(net)
    net.train()
    fps_ranges = {'Pointnet2': [N, 512], 'PTran': [N, 256, 64, 16]}.get(name, [N])   # randint bounds of the FPS draws
    n_fps = len(fps_ranges)
    multi = n_fps > 1
    out = {'x': x, 'seed': seed}

    # pass 1: semantic heads, with gradients
    torch.manual_seed(seed + 1)
    y1, y2, s1, s2 = net(x, semantic_adaption=True)
    loss = sum((t * _probe(t.shape, 'probe%d' % i)).sum() for i, t in enumerate((y1, y2, s1, s2)))
    net.zero_grad()
    loss.backward()
    out.update(y1=y1, y2=y2, s1=s1, s2=s2, loss=loss)
    gnames, gnorm, gdot = [], [], []
    for k, v in net.named_parameters():
        if v.grad is None:
            continue
        gnames.append(k)
        gnorm.append(v.grad.norm().item())
        gdot.append((v.grad * _probe(v.shape, 'g' + k)).sum().item())
    out['grad_names'] = np.array(gnames)
    out['grad_norm'] = np.array(gnorm, dtype=np.float64)
    out['grad_dot'] = np.array(gdot, dtype=np.float64)
    sd1 = net.state_dict()
    bn_names = [k for k in sd1 if k.endswith('running_mean') or k.endswith('running_var')]
    out['bn_names'] = np.array(bn_names)
    out['bn_sum'] = np.array([sd1[k].double().sum().item() for k in bn_names])

    # restatement check (fresh params, same FPS draws)
    p = O.as_params(p0)
    torch.manual_seed(seed + 1)
    starts = [None] * n_fps
    o = O.net_mda(p, name, x, True, starts if multi else None, semantic_adaption=True)
    for a, b, nm in zip(o, (y1, y2, s1, s2), ('y1', 'y2', 's1', 's2')):
        same(a, b, '%s %s' % (name, nm), 2e-6)
    oloss = sum((t * _probe(t.shape, 'probe%d' % i)).sum() for i, t in enumerate(o))
    oloss.backward()
    for k, gn in zip(gnames, gnorm):
        og = p[k].grad
        assert og is not None, k
        assert abs(og.norm().item() - gn) <= 2e-4 * max(1.0, gn), (name, k, og.norm().item(), gn)

    # pass 2: node features through attention_s (continues BN running stats, new FPS draw)
    torch.manual_seed(seed + 2)
    node_s = net(x, node_adaptation_s=True)
    out['node_s'] = node_s
    torch.manual_seed(seed + 2)
    pd = {k: v.detach() for k, v in p.items()}
    o_node = O.net_mda(pd, name, x, True, starts if multi else None, node_adaptation_s=True)
    same(o_node, node_s, name + ' node_s', 2e-5)
    torch.manual_seed(seed + 3)
    feat, node = net(x, mid_feat=True)
    out['mid_feat'], out['mid_node'] = feat, node.reshape(B, -1)

    # the FPS start draws of the three passes, in order
    for i, s in enumerate((seed + 1, seed + 2, seed + 3)):
        torch.manual_seed(s)
        if multi:
            out['start%d' % i] = torch.stack([torch.randint(0, r, (B,)) for r in fps_ranges])
        else:
            out['start%d' % i] = torch.randint(0, N, (B,))

    if name == 'DGCNN':
        # the four neighbour lists of pass 1 (teacher forcing for layer-level parity)
        net2 = r_M.Net_MDA(name)
        net2.load_state_dict(p0)
        net2.train()
        torch.manual_seed(seed + 1)
        with torch.no_grad():
            pp = {k: v.clone() for k, v in p0.items()}
            _, _, (x1, x2, x3, x4) = O.dgcnn_g(pp, 'g.', x, True, None)
            out['knn1'] = r_mu.knn(x.squeeze(-1), 20)
            out['knn2'] = r_mu.knn(x1, 20)
            out['knn3'] = r_mu.knn(x2, 20)
            out['knn4'] = r_mu.knn(x3, 20)
    save(fname, **out)


def gen_pointnet_cls():
    """Config 1 as train_source.py runs it (:76-97 model + criterion + optim.Adam(model.parameters()), :113-131 the step):
    Pointnet_cls forward + CE loss + backward + ONE Adam update (lr 1e-3, weight decay 5e-5), then the loss of a second
    forward on the same batch.  Round 6: gradients, BatchNorm buffers and the post-step parameters are part of the fixture."""
    seed = 31
    g = torch.Generator().manual_seed(seed)
    x = O.synth_clouds(8, 1024, g)
    lab = torch.randint(0, 10, (8,), generator=g)
    net = r_mp.Pointnet_cls()
    p0 = _load(net, seed)
    _no_dropout(net)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=5e-5)
    y = net(x)
    o = O.pointnet_cls({k: v.clone() for k, v in p0.items()}, x, True)
    same(o, y, 'pointnet_cls', 2e-6)
    rec = _cls_record(net, y, lab)              # CE, zero_grad, backward; gradient norms / probe dots / BN buffer sums
    p = O.as_params(p0)
    torch.nn.functional.cross_entropy(O.pointnet_cls(p, x, True), lab).backward()
    for k, gn in zip(rec['grad_names'], rec['grad_norm']):
        assert abs(p[k].grad.norm().item() - gn) <= 2e-4 * max(1.0, gn), (k, p[k].grad.norm().item(), gn)
    opt.step()
    opt.zero_grad()
    pnames = [k for k, _ in net.named_parameters()]
    post = dict(net.named_parameters())
    rec['param_names'] = np.array(pnames)
    rec['param_sum'] = np.array([post[k].detach().double().sum().item() for k in pnames])
    rec['param_dot'] = np.array([(post[k].detach() * _probe(post[k].shape, 'p' + k)).double().sum().item() for k in pnames])
    rec['param_delta_norm'] = np.array([(post[k].detach() - p0[k]).double().norm().item() for k in pnames])
    y2 = net(x)
    rec['loss2'] = torch.nn.functional.cross_entropy(y2, lab).detach()
    save('pointnet_cls.npz', x=x, label=lab, y=y, seed=seed, **rec)


def _cls_record(net, y, lab):
    """CE loss + backward of a source-only classifier; gradient norms / probe dots / BatchNorm buffer sums."""
    loss = torch.nn.functional.cross_entropy(y, lab)
    net.zero_grad()
    loss.backward()
    gnames, gnorm, gdot = [], [], []
    for k, v in net.named_parameters():
        if v.grad is None:
            continue
        gnames.append(k)
        gnorm.append(v.grad.norm().item())
        gdot.append((v.grad * _probe(v.shape, 'g' + k)).sum().item())
    sd = net.state_dict()
    bn_names = [k for k in sd if k.endswith('running_mean') or k.endswith('running_var')]
    return dict(loss=loss, grad_names=np.array(gnames), grad_norm=np.array(gnorm, dtype=np.float64),
                grad_dot=np.array(gdot, dtype=np.float64), bn_names=np.array(bn_names),
                bn_sum=np.array([sd[k].double().sum().item() for k in bn_names]))


def gen_pointnet2_cls():
    """train_source.py:76-77 with Model Pointnet2: model_pointnet.Pointnet2_cls forward + CE loss + backward
    (B=4 so that the BatchNorm1d heads see more than a sign pattern; N=2048 as in BASELINE config 3)."""
    seed, B, N = 33, 4, 2048
    g = torch.Generator().manual_seed(seed)
    x = O.synth_clouds(B, N, g)
    lab = torch.randint(0, 10, (B,), generator=g)
    net = r_mp.Pointnet2_cls()
    p0 = _load(net, seed)
    _no_dropout(net)
    net.train()
    torch.manual_seed(seed + 1)
    y = net(x)
    rec = _cls_record(net, y, lab)
    torch.manual_seed(seed + 1)
    starts = torch.stack([torch.randint(0, r, (B,)) for r in (N, 512)])
    p = O.as_params(p0)
    o = O.pointnet2_cls(p, x, True, (starts[0], starts[1]))
    same(o, y, 'pointnet2_cls', 2e-6)
    torch.nn.functional.cross_entropy(o, lab).backward()
    for k, gn in zip(rec['grad_names'], rec['grad_norm']):
        assert abs(p[k].grad.norm().item() - gn) <= 2e-4 * max(1.0, gn), (k, p[k].grad.norm().item(), gn)
    save('pointnet2_cls.npz', x=x, label=lab, y=y, seed=seed, start0=starts, **rec)


def gen_dgcnn_cls():
    """train_source.py:76-77 with Model DGCNN: model_pointnet.DGCNN forward + CE loss + backward, plus the four
    neighbour lists of the run (teacher forcing / free-running comparison)."""
    seed, B, N = 34, 2, 1024
    g = torch.Generator().manual_seed(seed)
    x = O.synth_clouds(B, N, g)
    lab = torch.randint(0, 10, (B,), generator=g)
    net = r_mp.DGCNN()
    p0 = _load(net, seed)
    _no_dropout(net)
    net.train()
    y = net(x)
    rec = _cls_record(net, y, lab)
    p = O.as_params(p0)
    o, (x1, x2, x3, x4) = O.dgcnn_cls(p, x, True)
    same(o, y, 'dgcnn_cls', 2e-6)
    torch.nn.functional.cross_entropy(o, lab).backward()
    for k, gn in zip(rec['grad_names'], rec['grad_norm']):
        assert abs(p[k].grad.norm().item() - gn) <= 2e-4 * max(1.0, gn), (k, p[k].grad.norm().item(), gn)
    with torch.no_grad():
        knn = [r_mu.knn(t.detach(), 20) for t in (x.squeeze(-1), x1, x2, x3)]
    save('dgcnn_cls.npz', x=x, label=lab, y=y, seed=seed, knn1=knn[0], knn2=knn[1], knn3=knn[2], knn4=knn[3], **rec)


def gen_focal():
    """focal_loss (model/model_utils.py:131-176, the CLS_LOSS: FocalLoss criterion of
    train_dg_single_gpu.py:167,176): first call of a fresh module, uniform and per-class alpha."""
    g = torch.Generator().manual_seed(41)
    out = {}
    for tag, B, alpha, gamma, avg in (('uni', 16, None, 2, True), ('cls', 32, [0.05 * (i + 1) for i in range(10)], 1.5, True),
                                      ('sum', 8, [0.1] * 10, 2, False)):
        pr = (torch.randn(B, 10, generator=g) * 2).requires_grad_(True)
        lab = torch.randint(0, 10, (B,), generator=g)
        crit = r_mu.focal_loss(alpha=alpha, gamma=gamma, num_classes=10, size_average=avg)
        a0 = crit.alpha.clone()
        v = crit(pr, lab)
        (gp,) = torch.autograd.grad(v, pr)
        out[tag + '_pred'], out[tag + '_label'], out[tag + '_alpha'] = pr, lab, a0
        out[tag + '_gamma'] = np.float64(gamma)
        out[tag + '_loss'], out[tag + '_grad'] = v, gp
        same(O.focal_loss(pr, lab, a0, gamma, avg), v, 'focal ' + tag, 1e-6)
    save('focal.npz', **out)


# ------------------------------------------------------------------ MMD
def gen_mmd():
    g = torch.Generator().manual_seed(99)
    out = {}
    for tag, m, D, scale in (('sem', 8, 256, 1.0), ('geo', 8, 4096, 0.05), ('sem32', 32, 256, 1.0)):
        X = (torch.randn(m, D, generator=g) * scale).requires_grad_(True)
        Y = (torch.randn(m, D, generator=g) * scale + 0.1).requires_grad_(True)
        ls = torch.randint(0, 10, (m,), generator=g)
        lt = torch.randint(0, 10, (m,), generator=g)
        w = torch.rand(1, m, generator=g) * 2
        out[tag + '_X'], out[tag + '_Y'], out[tag + '_ls'], out[tag + '_lt'], out[tag + '_w'] = X, Y, ls, lt, w
        v_plain = r_mmd.mix_rbf_mmd2(X, Y, r_mmd.sigma_list)
        v_w = r_mmd.mix_rbf_mmd2(X, Y, r_mmd.sigma_list, sample_weights=w)
        lsc = 50.0 if tag == 'geo' else 5.0
        v_soft = r_mmd.soft_mmd(ls, X, lt, Y, lsc, sample_weights=w)
        v_hard = r_mmd.hard_mmd(ls, X, ls.clone(), Y)
        v_max = r_mmd.max_hard_mmd(ls, X, lt, Y)
        gx, gy = torch.autograd.grad(v_soft, (X, Y))
        out[tag + '_plain'], out[tag + '_weighted'], out[tag + '_soft'] = v_plain, v_w, v_soft
        out[tag + '_hard'], out[tag + '_maxhard'] = v_hard, v_max
        out[tag + '_soft_gx'], out[tag + '_soft_gy'] = gx, gy
        same(O.mix_rbf_mmd2(X, Y), v_plain, 'mmd plain', 1e-6)
        # the unbiased estimator (_mmd2(biased=False), model/mmd.py:304-308; round 6): value and gradient
        v_unb = r_mmd.mix_rbf_mmd2(X, Y, r_mmd.sigma_list, biased=False)
        v_unb_w = r_mmd.mix_rbf_mmd2(X, Y, r_mmd.sigma_list, biased=False, sample_weights=w)
        ugx, ugy = torch.autograd.grad(v_unb_w, (X, Y))
        out[tag + '_unbiased'], out[tag + '_unbiased_weighted'] = v_unb, v_unb_w
        out[tag + '_unbiased_gx'], out[tag + '_unbiased_gy'] = ugx, ugy
        same(O.mix_rbf_mmd2(X, Y, biased=False), v_unb, 'mmd unbiased', 1e-6)
        same(O.mix_rbf_mmd2(X, Y, sample_weights=w, biased=False), v_unb_w, 'mmd unbiased weighted', 1e-6)
        same(O.mix_rbf_mmd2(X, Y, sample_weights=w), v_w, 'mmd weighted', 1e-6)
        same(O.soft_mmd(ls, X, lt, Y, lsc, w), v_soft, 'soft mmd', 1e-6)
        same(O.mmd_cal(ls, X, ls.clone(), Y, {'NAME': 'HARD_MMD'}), v_hard, 'hard mmd', 1e-6)
        same(O.mmd_cal(ls, X, lt, Y, {'NAME': 'MAX_HARD_MMD'}), v_max, 'max hard mmd', 1e-6)
        # SDA weights from head logits (prob_weights_soft, 'mean2one' and 'none')
        ps, pt = torch.randn(m, 10, generator=g), torch.randn(m, 10, generator=g)
        out[tag + '_ps'], out[tag + '_pt'] = ps, pt
        for meth in ('mean2one', 'none'):
            wr = r_mmd.prob_weights_soft(ps, pt, ls, lt, 0.5, meth)
            out[tag + '_pw_' + meth] = wr
            same(O.prob_weights_soft(ps, pt, ls, lt, 0.5, meth), wr, 'prob_weights ' + meth, 1e-6)
        cfg = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'none', 'LABEL_WEIGHT': 0.5}
        v_cal = r_mmd.mmd_cal(ls, X, lt, Y, cfg, data_s=ps, data_t=pt)
        out[tag + '_cal_none'] = v_cal
        same(O.mmd_cal(ls, X, lt, Y, cfg, ps, pt), v_cal, 'mmd_cal', 1e-6)
    save('mmd.npz', **out)


# ------------------------------------------------------------------ training step
def gen_step():
    """Two SUG steps (DGCNN, B=4): losses and post-step parameter checksums, following
    train_dg_single_gpu.py:191-203 (3 Adam optimisers) and :246-335 (the step) with
    TARGET_LOSS 0, ADV_WEIGHT 0, PURE_CLS_EPOCH 0, CE loss, SEM 'none' weights, no GEO weights."""
    seed, B, N = 666, 4, 1024
    g = torch.Generator().manual_seed(seed)
    data, data_t = O.synth_clouds(B, N, g), O.synth_clouds(B, N, g)
    lab, lab_t = torch.randint(0, 10, (B,), generator=g), torch.randint(0, 10, (B,), generator=g)
    net = r_M.Net_MDA('DGCNN')
    p0 = _load(net, seed)
    _no_dropout(net)
    net.train()
    LR, WD = 1e-3, 5e-5
    params = [{'params': v} for k, v in net.g.named_parameters() if 'pred_offset' not in k]
    opt_g = torch.optim.Adam(params, lr=LR, weight_decay=WD)
    opt_c = torch.optim.Adam([{'params': net.c1.parameters()}, {'params': net.c2.parameters()}], lr=LR, weight_decay=WD)
    opt_d = torch.optim.Adam([{'params': net.g.parameters()}, {'params': net.attention_s.parameters()},
                              {'params': net.attention_t.parameters()}], lr=LR, weight_decay=WD)
    geo = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 50, 'GEO_SCALE': 1}
    sem = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'none', 'LABEL_WEIGHT': 0.5, 'SEM_SCALE': 1}
    crit = torch.nn.CrossEntropyLoss()
    losses = []
    torch.manual_seed(seed)
    for step in range(2):
        ps1, ps2, fs1, fs2 = net(data, semantic_adaption=True)
        pt1, pt2, ft1, ft2 = net(data_t, semantic_adaption=True)
        loss_cls = 0.5 * crit(ps1, lab) + 0.5 * crit(ps2, lab)
        node_s = net(data, node_adaptation_s=True)
        node_t = net(data_t, node_adaptation_t=True)
        l_geo = r_mmd.mmd_cal(lab, node_s, lab_t, node_t, geo, data_s=data, data_t=data_t)
        l1 = r_mmd.mmd_cal(lab, fs1, lab_t, ft1, sem, data_s=ps1, data_t=pt1)
        l2 = r_mmd.mmd_cal(lab, fs2, lab_t, ft2, sem, data_s=ps2, data_t=pt2)
        l_sem = 0.5 * l1 + 0.5 * l2
        loss = loss_cls + l_geo + l_sem
        loss.backward()
        opt_d.step(); opt_g.step(); opt_c.step()
        opt_g.zero_grad(); opt_c.zero_grad(); opt_d.zero_grad()
        losses.append([loss_cls.item(), l_geo.item(), l_sem.item()])
        if step == 0:
            # restatement of the same first step
            p = O.as_params(p0)
            torch.manual_seed(seed)
            oc, og, osem = O.sug_losses(p, 'DGCNN', data, lab, data_t, lab_t, geo, sem)
            assert abs(oc.item() - losses[0][0]) < 1e-5 and abs(og.item() - losses[0][1]) < 1e-5 \
                and abs(osem.item() - losses[0][2]) < 1e-5, (oc.item(), og.item(), osem.item(), losses[0])
            # leave the default generator where the reference left it
            torch.manual_seed(seed)
            for _ in range(4):
                torch.randint(0, N, (B,))
    sd = net.state_dict()
    names = [k for k, v in sd.items() if v.dtype.is_floating_point]
    save('step_dgcnn.npz', data=data, data_t=data_t, label=lab, label_t=lab_t, seed=seed,
         losses=np.array(losses, dtype=np.float64), names=np.array(names),
         p_sum=np.array([sd[k].double().sum().item() for k in names]),
         p_abs=np.array([sd[k].double().abs().sum().item() for k in names]))


def gen_step_seeds():
    """The two-step run of gen_step for 8 more seeds (data = O.synth_clouds from the seed, so the fixture holds only the
    reference's six loss values per seed): how far apart are the reference's own second-step losses on two CPUs, next
    to how far the HIP path is from the oracle on the same machine (tests/test_gpu_step.py)."""
    B, N = 4, 1024
    LR, WD = 1e-3, 5e-5
    geo = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 50, 'GEO_SCALE': 1}
    sem = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'none', 'LABEL_WEIGHT': 0.5, 'SEM_SCALE': 1}
    seeds = [701 + i for i in range(8)]
    all_losses = []
    for seed in seeds:
        g = torch.Generator().manual_seed(seed)
        data, data_t = O.synth_clouds(B, N, g), O.synth_clouds(B, N, g)
        lab, lab_t = torch.randint(0, 10, (B,), generator=g), torch.randint(0, 10, (B,), generator=g)
        net = r_M.Net_MDA('DGCNN')
        _load(net, seed)
        _no_dropout(net)
        net.train()
        params = [{'params': v} for k, v in net.g.named_parameters() if 'pred_offset' not in k]
        opt_g = torch.optim.Adam(params, lr=LR, weight_decay=WD)
        opt_c = torch.optim.Adam([{'params': net.c1.parameters()}, {'params': net.c2.parameters()}], lr=LR, weight_decay=WD)
        opt_d = torch.optim.Adam([{'params': net.g.parameters()}, {'params': net.attention_s.parameters()},
                                  {'params': net.attention_t.parameters()}], lr=LR, weight_decay=WD)
        crit = torch.nn.CrossEntropyLoss()
        losses = []
        torch.manual_seed(seed)
        for step in range(2):
            ps1, ps2, fs1, fs2 = net(data, semantic_adaption=True)
            pt1, pt2, ft1, ft2 = net(data_t, semantic_adaption=True)
            loss_cls = 0.5 * crit(ps1, lab) + 0.5 * crit(ps2, lab)
            node_s = net(data, node_adaptation_s=True)
            node_t = net(data_t, node_adaptation_t=True)
            l_geo = r_mmd.mmd_cal(lab, node_s, lab_t, node_t, geo, data_s=data, data_t=data_t)
            l1 = r_mmd.mmd_cal(lab, fs1, lab_t, ft1, sem, data_s=ps1, data_t=pt1)
            l2 = r_mmd.mmd_cal(lab, fs2, lab_t, ft2, sem, data_s=ps2, data_t=pt2)
            loss = loss_cls + l_geo + (0.5 * l1 + 0.5 * l2)
            loss.backward()
            opt_d.step(); opt_g.step(); opt_c.step()
            opt_g.zero_grad(); opt_c.zero_grad(); opt_d.zero_grad()
            losses.append([loss_cls.item(), l_geo.item(), (0.5 * l1 + 0.5 * l2).item()])
        print('seed', seed, losses)
        all_losses.append(losses)
    save('step_dgcnn_seeds.npz', seeds=np.array(seeds, dtype=np.int32), losses=np.array(all_losses, dtype=np.float64),
         B=np.int32(B), N=np.int32(N))


def gen_eval(name, B, N, seed, fname):
    """Eval-mode forwards (utils/eval_utils.py:5-88 calls model.eval() and then `pred1, pred2 = model(data)`;
    train_dg_single_gpu.py:364 deep-copies the best model for it): ONE train-mode forward on a first batch so that the
    BatchNorm running statistics are not the initial ones, then net.eval() and every Net_MDA.forward mode on a second
    batch.  Eval-mode farthest-point sampling still draws its random start (point_utils.py:17): seeded per call."""
    g = torch.Generator().manual_seed(seed)
    x_tr, x = O.synth_clouds(B, N, g), O.synth_clouds(B, N, g)
    net = r_M.Net_MDA(name)
    p0 = _load(net, seed)
    _no_dropout(net)            # (eval mode ignores dropout anyway; the train-mode warm-up forward must not draw from it)
    fps_ranges = {'Pointnet2': [N, 512], 'PTran': [N, 256, 64, 16]}.get(name, [N])
    multi = len(fps_ranges) > 1
    net.train()
    torch.manual_seed(seed + 1)
    with torch.no_grad():
        net(x_tr, semantic_adaption=True)
    sd1 = {k: v.clone() for k, v in net.state_dict().items()}
    net.eval()
    out = {'x_train': x_tr, 'x': x, 'seed': seed}
    with torch.no_grad():
        torch.manual_seed(seed + 2)
        y1, y2 = net(x)
        torch.manual_seed(seed + 3)
        z1, z2, s1, s2 = net(x, semantic_adaption=True)
        torch.manual_seed(seed + 4)
        node_s = net(x, node_adaptation_s=True)
        torch.manual_seed(seed + 5)
        node_t = net(x, node_adaptation_t=True)
        torch.manual_seed(seed + 6)
        feat, node = net(x, mid_feat=True)
    out.update(y1=y1, y2=y2, z1=z1, z2=z2, s1=s1, s2=s2, node_s=node_s, node_t=node_t, mid_feat=feat,
               mid_node=node.reshape(B, -1))
    for k, v in net.state_dict().items():           # eval mode must not touch any buffer
        assert torch.equal(v, sd1[k]), k
    bn_names = [k for k in sd1 if k.endswith('running_mean') or k.endswith('running_var')]
    out['bn_names'] = np.array(bn_names)
    out['bn_sum'] = np.array([sd1[k].double().sum().item() for k in bn_names])
    for i in range(6):
        torch.manual_seed(seed + 1 + i)
        out['start%d' % i] = torch.stack([torch.randint(0, r, (B,)) for r in fps_ranges]) if multi else torch.randint(0, N, (B,))

    # restatement check: same train-mode warm-up, then training=False
    p = {k: v.clone() for k, v in p0.items()}
    st = lambda i: (tuple(out['start%d' % i]) if multi else [out['start%d' % i]])
    with torch.no_grad():
        O.net_mda(p, name, x_tr, True, st(0), semantic_adaption=True)
        for k in bn_names:
            same(p[k], sd1[k], '%s warm-up %s' % (name, k), 2e-6)
        o = O.net_mda(p, name, x, False, st(1))
        same(o[0], y1, name + ' eval y1', 2e-6), same(o[1], y2, name + ' eval y2', 2e-6)
        o = O.net_mda(p, name, x, False, st(2), semantic_adaption=True)
        for a, b, nm in zip(o, (z1, z2, s1, s2), ('z1', 'z2', 's1', 's2')):
            same(a, b, '%s eval %s' % (name, nm), 2e-6)
        same(O.net_mda(p, name, x, False, st(3), node_adaptation_s=True), node_s, name + ' eval node_s', 2e-5)
        same(O.net_mda(p, name, x, False, st(4), node_adaptation_t=True), node_t, name + ' eval node_t', 2e-5)
        of, on = O.net_mda(p, name, x, False, st(5), mid_feat=True)
        same(of, feat, name + ' eval mid_feat', 2e-6), same(on.reshape(B, -1), node.reshape(B, -1), name + ' eval mid_node', 2e-6)
        for k in bn_names:
            same(p[k], sd1[k], '%s eval left %s alone' % (name, k), 0.0)
    if name == 'DGCNN':
        # the four neighbour lists of the plain eval forward (teacher forcing where feature-space near-ties are CPU-dependent)
        with torch.no_grad():
            torch.manual_seed(seed + 2)
            _, _, (x1, x2, x3, x4) = O.dgcnn_g({k: v.clone() for k, v in sd1.items()}, 'g.', x, False, out['start1'])
            for i, t in enumerate((x.squeeze(-1), x1, x2, x3)):
                out['knn%d' % (i + 1)] = r_mu.knn(t, 20)
            # ... and of the train-mode warm-up forward (a near-tie that falls differently there moves the running statistics
            # by ~1e-4, which is all the eval logits' tolerance: the GPU test teacher-forces the warm-up as well)
            _, _, (x1, x2, x3, x4) = O.dgcnn_g({k: v.clone() for k, v in p0.items()}, 'g.', x_tr, True, out['start0'])
            for i, t in enumerate((x_tr.squeeze(-1), x1, x2, x3)):
                out['knn_train%d' % (i + 1)] = r_mu.knn(t, 20)
    save(fname, **out)


def gen_eval_cls():
    """The three source-only classifiers of train_source.py (model/model_pointnet.py) in eval mode, after one train-mode
    forward: logits of a second batch."""
    out = {}
    for tag, ctor, ofn, B, N, seed in (('pointnet', r_mp.Pointnet_cls, O.pointnet_cls, 4, 1024, 41),
                                       ('pointnet2', r_mp.Pointnet2_cls, O.pointnet2_cls, 4, 2048, 42),
                                       ('dgcnn', r_mp.DGCNN, O.dgcnn_cls, 2, 1024, 43)):
        g = torch.Generator().manual_seed(seed)
        x_tr, x = O.synth_clouds(B, N, g), O.synth_clouds(B, N, g)
        net = ctor()
        p0 = _load(net, seed)
        _no_dropout(net)
        net.train()
        with torch.no_grad():
            torch.manual_seed(seed + 1)
            net(x_tr)
            sd1 = {k: v.clone() for k, v in net.state_dict().items()}
            net.eval()
            torch.manual_seed(seed + 2)
            y = net(x)
        for k, v in net.state_dict().items():
            assert torch.equal(v, sd1[k]), k
        rec = {'x_train': x_tr, 'x': x, 'y': y, 'seed': seed}
        p = {k: v.clone() for k, v in p0.items()}
        with torch.no_grad():
            if tag == 'pointnet2':
                for i in range(2):
                    torch.manual_seed(seed + 1 + i)
                    rec['start%d' % i] = torch.stack([torch.randint(0, r, (B,)) for r in (N, 512)])
                ofn(p, x_tr, True, tuple(rec['start0']))
                o = ofn(p, x, False, tuple(rec['start1']))
            elif tag == 'dgcnn':
                ofn(p, x_tr, True)
                o, (x1, x2, x3, x4) = ofn(p, x, False)
                for i, t in enumerate((x.squeeze(-1), x1, x2, x3)):
                    rec['knn%d' % (i + 1)] = r_mu.knn(t, 20)
            else:
                ofn(p, x_tr, True)
                o = ofn(p, x, False)
        same(o, y, 'eval cls ' + tag, 2e-6)
        bn_names = [k for k in sd1 if k.endswith('running_mean') or k.endswith('running_var')]
        rec['bn_names'] = np.array(bn_names)
        rec['bn_sum'] = np.array([sd1[k].double().sum().item() for k in bn_names])
        out.update({tag + '_' + k: v for k, v in rec.items()})
    save('eval_cls.npz', **out)


def gen_entropy():
    """ENTROPY_WEIGHTS (model/mmd.py:47-48, :155-166; dataset_splitter.py:234-245): per-sample weights = symmetric KL between
    the prediction ENTROPIES of paired source / target samples.  What the reference can actually run: weighting 'none' and
    'mean2one' on PROBABILITY inputs (raw logits give NaN through log of a negative number; 'exp_inverse' -- its default --,
    'naive_inverse' and 'hist' raise inside distance2weights).  Reached through mmd_cal only together with SEM_WEIGHTS
    (:28 tests GEO / SEM, :47 prefers ENTROPY)."""
    g = torch.Generator().manual_seed(61)
    m = 8
    ps, pt = torch.softmax(torch.randn(m, 10, generator=g) * 2, 1), torch.softmax(torch.randn(m, 10, generator=g) * 2, 1)
    ls, lt = torch.randint(0, 10, (m,), generator=g), torch.randint(0, 10, (m,), generator=g)
    fs, ft = torch.randn(m, 256, generator=g), torch.randn(m, 256, generator=g) + 0.1
    out = {'ps': ps, 'pt': pt, 'ls': ls, 'lt': lt, 'fs': fs, 'ft': ft}
    for w in ('none', 'mean2one'):
        out['w_' + w] = r_mmd.entropy_weights(ps, pt, weighting=w).float()
        same(O.entropy_weights(ps, pt, w), out['w_' + w], 'entropy_weights ' + w, 1e-6)
        args = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'mean2one', 'ENTROPY_WEIGHTS': w, 'LABEL_WEIGHT': 0.5}
        out['mmd_' + w] = r_mmd.mmd_cal(ls, fs, lt, ft, args, data_s=ps, data_t=pt)
        same(O.mmd_cal(ls, fs, lt, ft, args, ps, pt), out['mmd_' + w], 'mmd_cal ENTROPY_WEIGHTS ' + w, 1e-5)
    # a saturated (one-hot) probability row: its entropy is exactly 0; scipy's kl_div(0, y) = y and kl_div(y, 0) = +inf, so
    # the reference's distance of that pair is +inf ('none'), and 'mean2one' (1 / mean = 0, truncated) multiplies it by 0:
    # NaN in that slot, 0 elsewhere -- recorded as the reference produces it (the naive x log(x/y) gives NaN for both)
    ps1 = ps.clone()
    ps1[2] = torch.nn.functional.one_hot(torch.tensor(4), 10).float()
    out['ps1'] = ps1
    for w in ('none', 'mean2one'):
        out['w1_' + w] = r_mmd.entropy_weights(ps1, pt, weighting=w).float()
        got = O.entropy_weights(ps1, pt, w).reshape(-1)
        ref = out['w1_' + w].reshape(-1)
        fin = torch.isfinite(ref)
        assert torch.equal(torch.isnan(got), torch.isnan(ref)) and torch.equal(torch.isinf(got), torch.isinf(ref)), (w, got, ref)
        same(got[fin], ref[fin], 'entropy_weights (one-hot row) ' + w, 1e-6)
    print('one-hot row: none ->', out['w1_none'].reshape(-1)[:4].tolist(), ' mean2one ->', out['w1_mean2one'].reshape(-1)[:4].tolist())
    raised = []
    for w in ('exp_inverse', 'naive_inverse', 'hist'):
        try:
            r_mmd.entropy_weights(ps, pt, weighting=w)
            raised.append(w + ':ok')
        except Exception as e:
            raised.append(w + ':' + type(e).__name__)
    out['reference_raises'] = np.array(raised)
    print('reference behaviour of the other weightings:', raised)
    save('entropy.npz', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['ops', 'mmd', 'pointnet_cls', 'dgcnn', 'pointnet', 'pointnet2', 'ptran', 'step', 'focal', 'ptran2048',
                             'pointnet2_b4', 'pointnet2_cls', 'dgcnn_cls', 'step_seeds', 'eval', 'entropy']
    if 'ops' in which:
        gen_ops()
    if 'mmd' in which:
        gen_mmd()
    if 'pointnet_cls' in which:
        gen_pointnet_cls()
    if 'dgcnn' in which:
        gen_model('DGCNN', 2, 1024, 11, 'model_dgcnn.npz')
    if 'pointnet' in which:
        gen_model('Pointnet', 4, 1024, 12, 'model_pointnet.npz')
    if 'pointnet2' in which:
        gen_model('Pointnet2', 2, 2048, 13, 'model_pointnet2.npz')
    if 'ptran' in which:
        gen_model('PTran', 2, 1024, 14, 'model_ptran.npz')
    if 'focal' in which:
        gen_focal()
    if 'ptran2048' in which:            # BASELINE config 5's cloud size (the FPS schedule stays 256/64/16/4)
        gen_model('PTran', 2, 2048, 15, 'model_ptran_n2048.npz')
    if 'pointnet2_b4' in which:         # config 3 shape at a batch that exercises the 8-cloud XCD grouping with pairs
        gen_model('Pointnet2', 4, 2048, 16, 'model_pointnet2_b4.npz')
    if 'step' in which:
        gen_step()
    if 'step_seeds' in which:
        gen_step_seeds()
    if 'pointnet2_cls' in which:
        gen_pointnet2_cls()
    if 'dgcnn_cls' in which:
        gen_dgcnn_cls()
    if 'entropy' in which:
        gen_entropy()
    if 'eval' in which:                 # eval-mode forwards (utils/eval_utils.py:5-88) after one train-mode forward
        gen_eval('DGCNN', 2, 1024, 51, 'eval_dgcnn.npz')
        gen_eval('Pointnet', 4, 1024, 52, 'eval_pointnet.npz')
        gen_eval('Pointnet2', 2, 2048, 53, 'eval_pointnet2.npz')
        gen_eval('PTran', 2, 1024, 54, 'eval_ptran.npz')
        gen_eval_cls()

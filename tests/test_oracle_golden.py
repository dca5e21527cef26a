"""CPU: the oracle (oracle/ref_cpu.py, what travels to the GPU box) reproduces the golden
vectors produced by the reference, and the host-side mirror has the reference's parameters."""
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu as O


def test_ops_oracle_vs_golden(ops_golden):
    G = ops_golden
    assert torch.equal(O.knn_idx(G['knn_grid_x'], 20), G['knn_grid_idx'])
    assert torch.equal(O.knn_idx(G['knn_rand_x'], 20), G['knn_rand_idx'])
    assert torch.equal(O.graph_feature(G['gf_x'], 4, G['gf_idx']), G['gf_out'])
    assert torch.equal(O.fps_cf(G['fps_cf_xyz'], 64, G['fps_cf_start']), G['fps_cf_idx'])
    assert torch.equal(O.fps_cl(G['fps_cl_xyz'], 512, G['fps_cl_start']), G['fps_cl_idx'])
    assert torch.equal(O.fps_cf(G['fps_dup_xyz'], 32, G['fps_dup_start']), G['fps_dup_idx'])
    torch.manual_seed(777)                      # the restatement draws the start like the reference
    assert torch.equal(O.fps_cf(G['fps_cf_xyz'], 64), G['fps_cf_idx'])
    xyz, new = G['fps_cf_xyz'], G['bq_cf_new']
    assert torch.equal(O.ball_query_cf(0.3, 64, xyz, new), G['bq_cf_r03'])
    assert torch.equal(O.ball_query_cf(0.05, 64, xyz, new), G['bq_cf_small_r'])
    assert torch.equal(O.ball_query_cf(None, 64, xyz, G['bq_cf_moved']), G['bq_cf_knn'])
    x1 = G['fps_cl_xyz'][:1]
    assert torch.equal(O.ball_query_cl(0.2, 32, x1, O.gather_cl(x1, G['fps_cl_idx'][:1])), G['bq_cl_r02'])
    torch.testing.assert_close(O.upsample_inter(xyz, G['bq_cf_moved'], G['up_p1'], G['up_p2'], 3), G['up_out'],
                               rtol=1e-6, atol=1e-6)
    assert torch.equal(O.sqdist_cf(new, xyz)[:, :4], G['sqd_cf'])
    xs = G['fps_cl_xyz'][:, :512].contiguous()
    nxyz, npts = O.sample_and_group(128, 0.4, 64, xs, G['sag_pts'], G['sag_start'])
    assert torch.equal(nxyz, G['sag_new_xyz'])
    assert torch.equal(npts[:, :, 0], G['sag_new_points_row0'])


@pytest.mark.parametrize('tag,lsc', [('sem', 5.0), ('geo', 50.0), ('sem32', 5.0)])
def test_mmd_oracle_vs_golden(mmd_golden, tag, lsc):
    G = mmd_golden
    X, Y = G[tag + '_X'].requires_grad_(True), G[tag + '_Y'].requires_grad_(True)
    ls, lt, w = G[tag + '_ls'], G[tag + '_lt'], G[tag + '_w']
    tol = dict(rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(O.mix_rbf_mmd2(X, Y), G[tag + '_plain'], **tol)
    torch.testing.assert_close(O.mix_rbf_mmd2(X, Y, sample_weights=w), G[tag + '_weighted'], **tol)
    v = O.soft_mmd(ls, X, lt, Y, lsc, w)
    torch.testing.assert_close(v, G[tag + '_soft'], **tol)
    gx, gy = torch.autograd.grad(v, (X, Y))
    # fp32 autograd through the diagonal K_ii (gamma up to 5000) cancels only approximately, so
    # the reference's own gradient is noise-limited at this level (MMD backward)
    m = X.shape[0]
    noise = 5050.5 * (4.0 / (m * m)) * max(float(X.abs().max()), lsc) * 4 * torch.finfo(torch.float32).eps
    torch.testing.assert_close(gx, G[tag + '_soft_gx'], rtol=1e-3, atol=noise)
    torch.testing.assert_close(O.mmd_cal(ls, X, ls.clone(), Y, {'NAME': 'HARD_MMD'}), G[tag + '_hard'], **tol)
    torch.testing.assert_close(O.mmd_cal(ls, X, lt, Y, {'NAME': 'MAX_HARD_MMD'}), G[tag + '_maxhard'], **tol)
    for meth in ('mean2one', 'none'):
        torch.testing.assert_close(O.prob_weights_soft(G[tag + '_ps'], G[tag + '_pt'], ls, lt, 0.5, meth),
                                   G[tag + '_pw_' + meth], rtol=1e-5, atol=1e-8)


def test_pointnet_forward_oracle_vs_golden():
    """One full encoder on CPU (PointNet, B=4): the restatement equals the reference output."""
    G = load_golden('model_pointnet.npz')
    from sug_amd.model.Model import Net_MDA
    shapes = {k: tuple(v.shape) for k, v in Net_MDA('Pointnet').state_dict().items()}
    p = O.fill_params(shapes, G['seed'])
    torch.manual_seed(G['seed'] + 1)
    with torch.no_grad():
        y1, y2, s1, s2 = O.net_mda(p, 'Pointnet', G['x'], True, semantic_adaption=True)
    torch.testing.assert_close(y1, G['y1'], rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(s2, G['s2'], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('name,fname', [('DGCNN', 'model_dgcnn.npz'), ('Pointnet', 'model_pointnet.npz'),
                                        ('Pointnet2', 'model_pointnet2.npz'), ('PTran', 'model_ptran.npz'),
                                        ('PTran', 'model_ptran_n2048.npz'), ('Pointnet2', 'model_pointnet2_b4.npz')])
def test_mirror_has_reference_parameters(name, fname):
    """Same parameter / buffer names as the reference (so its checkpoints load): every name the
    reference produced a gradient or BN buffer for exists here, and the FPS start draws
    recorded in the golden are what the CPU generator yields."""
    from sug_amd.model.Model import Net_MDA
    G = load_golden(fname)
    net = Net_MDA(name)
    names = set(net.state_dict().keys())
    for k in G['grad_names'] + G['bn_names']:
        assert k in names, k
    B, N = G['x'].shape[0], G['x'].shape[2]
    torch.manual_seed(G['seed'] + 1)
    first = torch.randint(0, N, (B,))
    want = G['start0'][0] if name in ('Pointnet2', 'PTran') else G['start0']
    assert torch.equal(first, want)


@pytest.mark.parametrize('cls,fname', [('Pointnet2_cls', 'pointnet2_cls.npz'), ('DGCNN', 'dgcnn_cls.npz'),
                                       ('Pointnet_cls', 'pointnet_cls.npz')])
def test_source_only_classifiers_oracle_and_names_vs_golden(cls, fname):
    """model/model_pointnet.py:5-161 (what train_source.py:5,76-77 imports): the mirror module exists under the
    reference's class name with the reference's parameter names, and the oracle restatement reproduces the reference's
    logits (and, for DGCNN, its four neighbour graphs) on CPU."""
    from sug_amd.model import model_pointnet as MP
    G = load_golden(fname)
    net = getattr(MP, cls)()
    names = set(net.state_dict().keys())
    for k in G.get('grad_names', []) + G.get('bn_names', []):
        assert k in names, k
    p = O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, G['seed'])
    with torch.no_grad():
        if cls == 'Pointnet2_cls':
            y = O.pointnet2_cls(p, G['x'], True, (G['start0'][0], G['start0'][1]))
        elif cls == 'DGCNN':
            y, (x1, x2, x3, _) = O.dgcnn_cls(p, G['x'], True)
            assert torch.equal(O.knn_idx(G['x'].squeeze(-1), 20), G['knn1'])
            for f, key in ((x1, 'knn2'), (x2, 'knn3'), (x3, 'knn4')):
                got = O.knn_idx(f, 20)
                # feature-space graphs: same machine as the golden run -> equal up to the rare sgemm-order tie
                assert (got != G[key]).float().mean() < 1e-3, key
        else:
            y = O.pointnet_cls(p, G['x'], True)
    torch.testing.assert_close(y, G['y'], rtol=1e-5, atol=2e-5)
    loss = torch.nn.functional.cross_entropy(y, G['label'])
    assert abs(loss.item() - float(G['loss'])) < 1e-5


def test_mirror_refuses_to_run_without_gpu():
    from sug_amd.model.Model import Net_MDA
    net = Net_MDA('DGCNN')
    with pytest.raises(RuntimeError, match='HIP device'):
        net(torch.zeros(2, 3, 1024, 1), semantic_adaption=True)


def test_host_helpers():
    from sug_amd.utils.common_utils import create_one_hot_labels, get_most_overlapped_element
    from sug_amd.model import mmd
    lab = torch.tensor([3, 0, 9, 3])
    assert torch.equal(create_one_hot_labels(lab), O.one_hot(lab))
    a, b = torch.tensor([1, 3, 3, 0, 7, 1]), torch.tensor([3, 1, 1, 1, 2, 0])
    assert get_most_overlapped_element(a, b) == O.most_overlapped(a, b)
    d = torch.tensor([0.2, 0.3, 0.1])
    for meth in ('mean2one', 'none', 'naive_inverse', 'exp_inverse'):
        torch.testing.assert_close(mmd.distance2weights(d, meth), O.distance2weights(d, meth))
    assert float(mmd.distance2weights(torch.tensor([2.0, 4.0]), 'mean2one').abs().sum()) == 0.0   # int(1/3) == 0


@pytest.mark.parametrize('tag,avg', [('uni', True), ('cls', True), ('sum', False)])
def test_focal_loss_oracle_and_mirror_vs_golden(tag, avg):
    """focal_loss (model/model_utils.py:131-176): the oracle restatement and the host mirror (pure
    torch, device-agnostic) reproduce the reference's value and gradient."""
    from sug_amd.model.model_utils import focal_loss
    G = load_golden('focal.npz')
    pr, lab, alpha, gamma = G[tag + '_pred'], G[tag + '_label'], G[tag + '_alpha'], float(G[tag + '_gamma'])
    tol = dict(rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(O.focal_loss(pr, lab, alpha, gamma, avg), G[tag + '_loss'], **tol)
    crit = focal_loss(alpha=alpha.tolist(), gamma=gamma, num_classes=10, size_average=avg)
    x = pr.clone().requires_grad_(True)
    v = crit(x, lab)
    torch.testing.assert_close(v, G[tag + '_loss'], **tol)
    (g,) = torch.autograd.grad(v, x)
    torch.testing.assert_close(g, G[tag + '_grad'], **tol)
    # unlike the reference module (which overwrites self.alpha, model_utils.py:168), a second call
    # of the same module gives the same value
    torch.testing.assert_close(crit(pr, lab), G[tag + '_loss'], **tol)


def test_step_driver_lr_schedules():
    """SUGStep.set_epoch reproduces the reference's three schedules (train_dg_single_gpu.py:194-203,
    :210-212; utils/train_utils.py:39-48): CosineAnnealingLR stepped with an explicit epoch for
    optimizer_g / optimizer_c, the 5/10-epoch halving for optimizer_dis."""
    import math
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    LR, scaler, T = 1e-3, 2.0, 40
    tr = SUGStep(Net_MDA('Pointnet'), lr=LR, lr_scaler=scaler)
    w = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.Adam([w], lr=LR)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(ref_opt, T_max=T)
    import warnings
    for epoch in (0, 1, 2, 5, 6, 30, 31, 39):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            sched.step(epoch=epoch)             # the reference's call (closed form of the cosine)
        want_cos = ref_opt.param_groups[0]['lr']
        want_dis = LR * scaler if epoch == 0 else LR * scaler * (0.5 ** (epoch // 5 if epoch <= 30 else epoch // 10))
        lr_g, lr_c, lr_dis = tr.set_epoch(epoch, T)
        assert math.isclose(lr_g, want_cos, rel_tol=1e-12, abs_tol=1e-18) and lr_c == lr_g
        assert math.isclose(lr_dis, want_dis, rel_tol=1e-12)
        assert all(g['lr'] == lr_dis for g in tr.optimizer_dis.param_groups)
        assert all(g['lr'] == lr_g for g in tr.optimizer_g.param_groups)


_EVAL_FPS = {'Pointnet2': lambda N: [N, 512], 'PTran': lambda N: [N, 256, 64, 16]}


def _eval_starts(G, name, i):
    s = G['start%d' % i]
    return tuple(s) if name in _EVAL_FPS else [s]


@pytest.mark.parametrize('name,fname', [('DGCNN', 'eval_dgcnn.npz'), ('Pointnet', 'eval_pointnet.npz'),
                                        ('Pointnet2', 'eval_pointnet2.npz'), ('PTran', 'eval_ptran.npz')])
def test_eval_mode_oracle_vs_golden(name, fname):
    """Eval-mode forwards (utils/eval_utils.py:5-88: model.eval(), then `pred1, pred2 = model(data)`): the restatement with
    training=False after ONE train-mode forward reproduces the reference's eval outputs and leaves the buffers alone."""
    from sug_amd.model.Model import Net_MDA
    G = load_golden(fname)
    p = O.fill_params({k: tuple(v.shape) for k, v in Net_MDA(name).state_dict().items()}, G['seed'])
    B = G['x'].shape[0]
    tol = dict(rtol=1e-5, atol=2e-5)
    with torch.no_grad():
        O.net_mda(p, name, G['x_train'], True, _eval_starts(G, name, 0), semantic_adaption=True)
        for k, v in zip(G['bn_names'], G['bn_sum'].tolist()):
            assert abs(p[k].double().sum().item() - v) <= 1e-5 * max(1.0, abs(v)), k
        before = {k: p[k].clone() for k in G['bn_names']}
        y1, y2 = O.net_mda(p, name, G['x'], False, _eval_starts(G, name, 1))
        torch.testing.assert_close(y1, G['y1'], **tol)
        torch.testing.assert_close(y2, G['y2'], **tol)
        z1, z2, s1, s2 = O.net_mda(p, name, G['x'], False, _eval_starts(G, name, 2), semantic_adaption=True)
        torch.testing.assert_close(z2, G['z2'], **tol)
        torch.testing.assert_close(s1, G['s1'], **tol)
        node_s = O.net_mda(p, name, G['x'], False, _eval_starts(G, name, 3), node_adaptation_s=True)
        torch.testing.assert_close(node_s, G['node_s'], rtol=1e-4, atol=1e-4)
        node_t = O.net_mda(p, name, G['x'], False, _eval_starts(G, name, 4), node_adaptation_t=True)
        torch.testing.assert_close(node_t, G['node_t'], rtol=1e-4, atol=1e-4)
        feat, node = O.net_mda(p, name, G['x'], False, _eval_starts(G, name, 5), mid_feat=True)
        torch.testing.assert_close(feat, G['mid_feat'], **tol)
        torch.testing.assert_close(node.reshape(B, -1), G['mid_node'], **tol)
    for k in G['bn_names']:
        assert torch.equal(p[k], before[k]), 'eval mode changed ' + k
    # the recorded start draws are what the CPU generator yields for the reference's seeds
    N = G['x'].shape[2]
    torch.manual_seed(G['seed'] + 2)
    first = torch.randint(0, N, (B,))
    assert torch.equal(first, G['start1'][0] if name in _EVAL_FPS else G['start1'])


def test_eval_mode_classifiers_oracle_vs_golden():
    """model/model_pointnet.py classifiers in eval mode after one train-mode forward (train_source.py's eval loop)."""
    G = load_golden('eval_cls.npz')
    from sug_amd.model import model_pointnet as MP
    for tag, cls, fn in (('pointnet', 'Pointnet_cls', O.pointnet_cls), ('pointnet2', 'Pointnet2_cls', O.pointnet2_cls),
                         ('dgcnn', 'DGCNN', O.dgcnn_cls)):
        net = getattr(MP, cls)()
        p = O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, G[tag + '_seed'])
        with torch.no_grad():
            if tag == 'pointnet2':
                fn(p, G[tag + '_x_train'], True, tuple(G[tag + '_start0']))
                y = fn(p, G[tag + '_x'], False, tuple(G[tag + '_start1']))
            elif tag == 'dgcnn':
                fn(p, G[tag + '_x_train'], True)
                y, _ = fn(p, G[tag + '_x'], False)
            else:
                fn(p, G[tag + '_x_train'], True)
                y = fn(p, G[tag + '_x'], False)
        torch.testing.assert_close(y, G[tag + '_y'], rtol=1e-5, atol=2e-5)
        for k, v in zip(G[tag + '_bn_names'], G[tag + '_bn_sum'].tolist()):
            assert abs(p[k].double().sum().item() - v) <= 1e-5 * max(1.0, abs(v)), k


def test_oracle_dropout_with_a_given_mask_is_torch_dropout():
    """oracle._drop(y, p, training, keep) (used to hold the HIP heads' dropout against the oracle on the SAME mask) is
    F.dropout's arithmetic bit for bit: recover the mask torch draws under a seed, feed it back."""
    import torch.nn.functional as F
    y = torch.randn(64, 512)
    torch.manual_seed(9)
    want = F.dropout(y, 0.4, True)
    torch.manual_seed(9)
    keep = F.dropout(torch.ones_like(y), 0.4, True) > 0
    assert torch.equal(O._drop(y, 0.4, True, keep), want)
    assert 0.55 < float(keep.float().mean()) < 0.65
    assert torch.equal(O._drop(y, 0.4, False, keep), y)


def test_entropy_weights_oracle_and_mirror_vs_golden():
    """ENTROPY_WEIGHTS (model/mmd.py:47-48, :155-166): oracle and host mirror (pure torch, device-agnostic) reproduce what the
    reference can run ('none', 'mean2one' on probability inputs); 'hist' raises as it does there."""
    from sug_amd.model import mmd
    G = load_golden('entropy.npz')
    assert G['reference_raises'] == ['exp_inverse:AttributeError', 'naive_inverse:AttributeError', 'hist:TypeError']
    for w in ('none', 'mean2one'):
        torch.testing.assert_close(O.entropy_weights(G['ps'], G['pt'], w), G['w_' + w], rtol=1e-4, atol=2e-7)      # (x log(x/y) - x + y of nearby entropies cancels: 6e-8 absolute between scipy and torch)
        torch.testing.assert_close(mmd.entropy_weights(G['ps'], G['pt'], w), G['w_' + w], rtol=1e-4, atol=2e-7)      # (x log(x/y) - x + y of nearby entropies cancels: 6e-8 absolute between scipy and torch)
        args = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'mean2one', 'ENTROPY_WEIGHTS': w, 'LABEL_WEIGHT': 0.5}
        torch.testing.assert_close(O.mmd_cal(G['ls'], G['fs'], G['lt'], G['ft'], args, G['ps'], G['pt']), G['mmd_' + w],
                                   rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(mmd.cal_sample_weights(G['ps'], G['pt'], args, G['ls'], G['lt']), G['w_' + w], rtol=1e-4, atol=2e-7)      # (x log(x/y) - x + y of nearby entropies cancels: 6e-8 absolute between scipy and torch)
    with pytest.raises(TypeError):
        mmd.entropy_weights(G['ps'], G['pt'], 'hist')
    assert torch.isnan(mmd.entropy_weights(torch.randn(4, 10), torch.randn(4, 10), 'none')).any()      # logits: NaN, as there

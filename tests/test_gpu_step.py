"""GPU parity of two full SUG training steps (DGCNN, B=4) against the reference run recorded
in tests/golden/step_dgcnn.npz: per-step (loss_cls, loss_geo_mmd, loss_sem_mmd) and
post-step parameter checksums."""
import os

import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu


def test_two_training_steps_match_reference():
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    G = load_golden('step_dgcnn.npz')
    seed = G['seed']
    net = Net_MDA('DGCNN')
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    net = net.cuda().train()
    methods = {'GEO_MMD': [{'NAME': 'SOFT_MMD', 'LABEL_SCALE': 50, 'GEO_SCALE': 1}],
               'SEM_MMD': [{'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'none', 'LABEL_WEIGHT': 0.5, 'SEM_SCALE': 1}]}
    tr = SUGStep(net, lr=1e-3, weight_decay=5e-5, methods=methods)        # defaults: paired domains, sug_amd.optim.Adam
    data, data_t = G['data'].cuda(), G['data_t'].cuda()
    lab, lab_t = G['label'].cuda(), G['label_t'].cuda()
    torch.manual_seed(seed)
    got = []
    for _ in range(2):
        lc, lg, ls = tr.step(data, lab, data_t, lab_t)
        got.append([lc.item(), lg.item(), ls.item()])
    want = G['losses'].tolist()

    # The oracle run on THIS machine's CPU (same algorithm as the reference, proven equal to it
    # where the golden was produced).  Free-running feature-space kNN makes the reference's own
    # losses CPU-dependent at the 1e-3 level (sgemm rounding flips near-tied neighbours), so the
    # tight comparison is against the oracle here and the golden gets the looser bound.
    p = O.as_params(O.fill_params({k: tuple(v.shape) for k, v in Net_MDA('DGCNN').state_dict().items()}, seed))
    names = list(p.keys())
    og = torch.optim.Adam([p[k] for k in names if k.startswith('g.') and p[k].requires_grad and 'pred_offset' not in k], lr=1e-3, weight_decay=5e-5)
    oc = torch.optim.Adam([p[k] for k in names if k.startswith(('c1.', 'c2.')) and p[k].requires_grad], lr=1e-3, weight_decay=5e-5)
    od = torch.optim.Adam([p[k] for k in names if k.startswith(('g.', 'attention')) and p[k].requires_grad], lr=1e-3, weight_decay=5e-5)
    torch.manual_seed(seed)
    ora = []
    for _ in range(2):
        lc, lg, ls = O.sug_losses(p, 'DGCNN', G['data'], G['label'], G['data_t'], G['label_t'],
                                  methods['GEO_MMD'][0], methods['SEM_MMD'][0])
        (lc + lg + ls).backward()
        od.step(); og.step(); oc.step()
        for o in (og, oc, od):
            o.zero_grad()
        ora.append([lc.item(), lg.item(), ls.item()])
    print('losses gpu', got)
    print('losses oracle(here)', ora)
    print('losses golden(reference, build container)', want)
    for a, b in zip(got[0], ora[0]):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (got, ora)
    # after one Adam update of every parameter: Adam turns rounding-level gradients into +-lr
    # moves and arg-max / kNN near-ties flip, so trajectories separate quickly -- the reference
    # itself gives 1.7815 (build container CPU) vs 1.7900 (this CPU) for this loss
    for a, b in zip(got[1], ora[1]):
        assert abs(a - b) <= 5e-3 * max(1.0, abs(b)), (got, ora)
    for a, b in zip(got[0] + got[1], want[0] + want[1]):
        assert abs(a - b) <= 1e-2 * max(1.0, abs(b)), (got, want)
    # Adam's first updates are ~ lr*sign(g): an element whose gradient is rounding noise (e.g. a
    # conv bias in front of BatchNorm, whose true gradient is 0) may move the other way, so the
    # checksums are compared with an allowance of 4% of elements flipping (2 steps of lr each).
    sd = net.state_dict()
    worst = 0.0
    for k in names:
        if not p[k].dtype.is_floating_point:
            continue
        v, r = sd[k].double().cpu(), p[k].detach().double()
        allow = 0.04 * 2 * 2 * 1e-3 * v.numel() + 1e-4 * float(r.abs().sum()) + 1e-6
        if O.is_buffer(k):     # running stats absorb the +-lr moves of zero-gradient biases in front of BN
            allow = 1e-3 * v.numel() + 1e-3 * float(r.abs().sum())
        assert abs(float(v.sum() - r.sum())) <= allow, (k, float(v.sum()), float(r.sum()), allow)
        worst = max(worst, abs(float(v.sum() - r.sum())) / allow)
    for k, ps, pa in zip(G['names'], G['p_sum'].tolist(), G['p_abs'].tolist()):
        v = sd[k].double()
        allow = 0.05 * 2 * 2 * 1e-3 * v.numel() + 1e-3 * pa + 1e-6
        if O.is_buffer(k):
            allow = 2e-3 * v.numel() + 2e-3 * pa
        assert abs(v.sum().item() - ps) <= allow, (k, v.sum().item(), ps, allow)
    print('worst checksum deviation / allowance vs oracle = %.3f' % worst)


_STEP_METHODS = {'GEO_MMD': [{'NAME': 'SOFT_MMD', 'LABEL_SCALE': 50, 'GEO_SCALE': 1}],
                 'SEM_MMD': [{'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'none', 'LABEL_WEIGHT': 0.5, 'SEM_SCALE': 1}]}


def _two_steps_hip_and_oracle(seed, data, lab, data_t, lab_t):
    """Two SUG steps (DGCNN, dropout off, the golden's METHODS) on the HIP path and by the oracle on this machine's CPU."""
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    shapes = {k: tuple(v.shape) for k, v in Net_MDA('DGCNN').state_dict().items()}
    net = Net_MDA('DGCNN')
    net.load_state_dict(O.fill_params(shapes, seed))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    from sug_amd import ops
    tr = SUGStep(net.cuda().train(), lr=1e-3, weight_decay=5e-5, methods=_STEP_METHODS)
    torch.manual_seed(seed)
    got, lists, real_knn = [], [], ops.knn

    def spy(f, k):
        idx = real_knn(f, k)
        lists.append(idx.cpu().long())
        return idx
    for i in range(2):
        ops.knn = spy if i == 0 else real_knn            # the neighbour graphs of step 1 (paired pass: [2B, N, k] each)
        try:
            got.append([float(v) for v in tr.step(data.cuda(), lab.cuda(), data_t.cuda(), lab_t.cuda())])
        finally:
            ops.knn = real_knn
    B = data.shape[0]
    assert len(lists) >= 4 and lists[0].shape[0] == 2 * B
    forced = ([l[:B] for l in lists[:4]], [l[B:] for l in lists[:4]])
    p = O.as_params(O.fill_params(shapes, seed))
    with torch.no_grad():                                # step 1 of the oracle on the HIP path's neighbour graphs
        torch.manual_seed(seed)
        tf = [float(v) for v in O.sug_losses(p, 'DGCNN', data, lab, data_t, lab_t, _STEP_METHODS['GEO_MMD'][0],
                                             _STEP_METHODS['SEM_MMD'][0], knn_override=forced)]
    names = list(p.keys())
    og = torch.optim.Adam([p[k] for k in names if k.startswith('g.') and p[k].requires_grad and 'pred_offset' not in k], lr=1e-3, weight_decay=5e-5)
    oc = torch.optim.Adam([p[k] for k in names if k.startswith(('c1.', 'c2.')) and p[k].requires_grad], lr=1e-3, weight_decay=5e-5)
    od = torch.optim.Adam([p[k] for k in names if k.startswith(('g.', 'attention')) and p[k].requires_grad], lr=1e-3, weight_decay=5e-5)
    torch.manual_seed(seed)
    ora = []
    for _ in range(2):
        lc, lg, ls = O.sug_losses(p, 'DGCNN', data, lab, data_t, lab_t, _STEP_METHODS['GEO_MMD'][0], _STEP_METHODS['SEM_MMD'][0])
        (lc + lg + ls).backward()
        od.step(); og.step(); oc.step()
        for o in (og, oc, od):
            o.zero_grad()
        ora.append([lc.item(), lg.item(), ls.item()])
    return got, ora, tf


def test_second_step_spread_over_8_seeds():
    """the second training step is compared at 5e-3 with the oracle on the same machine and 1e-2 with
    the golden on ONE seed -- here the same two-step run on 8 more seeds (tests/golden/step_dgcnn_seeds.npz: the reference's
    losses on the build container's CPU; inputs and weights are regenerated from the seed).  Per seed, the relative
    deviation d(a,b) = max over the three loss terms of |a-b| / max(1,|b|) is formed for (HIP, oracle here) and for
    (oracle here, reference there), for both steps, and all four spreads are printed.
    Measured (r04): step 1 -- HIP vs oracle median 3e-6, but 1.5e-4 and 1.1e-3 on two of the eight seeds; the oracle here vs
    the reference there 3e-4 / 4e-4 on two OTHER seeds: a feature-space neighbour whose score gap is below the fp32
    rounding of the score enters or leaves a list, on either pair of machines.  What is asserted:
      (a) step 1 of the oracle evaluated ON THE HIP PATH'S NEIGHBOUR GRAPHS equals the HIP losses to 1e-4 on EVERY seed
          (everything but the rank decisions meets the north-star bar, no exceptions);
      (b) free-running step 1: median <= 1e-4, worst seed <= 5e-3;
      (c) step 2: the HIP path's deviation from the oracle on this machine is not larger than what two CPUs running the
          reference's own arithmetic differ by -- median <= max(2 x the CPU-to-CPU median, 1e-3), worst seed <=
          max(2 x the CPU-to-CPU worst seed, 5e-3)."""
    import statistics
    G = load_golden('step_dgcnn_seeds.npz')
    B, N = int(G['B']), int(G['N'])
    dev = lambda a, b: max(abs(x - y) / max(1.0, abs(y)) for x, y in zip(a, b))
    d_hip1, d_hip2, d_ref1, d_ref2, d_tf = [], [], [], [], []
    for seed, want in zip(G['seeds'].tolist(), G['losses'].tolist()):
        g = torch.Generator().manual_seed(seed)
        data, data_t = O.synth_clouds(B, N, g), O.synth_clouds(B, N, g)
        lab, lab_t = torch.randint(0, 10, (B,), generator=g), torch.randint(0, 10, (B,), generator=g)
        got, ora, tf = _two_steps_hip_and_oracle(seed, data, lab, data_t, lab_t)
        d_hip1.append(dev(got[0], ora[0])); d_hip2.append(dev(got[1], ora[1]))
        d_ref1.append(dev(ora[0], want[0])); d_ref2.append(dev(ora[1], want[1]))
        d_tf.append(dev(got[0], tf))
        print('seed %d: step1 HIP-vs-oracle %.2e (on the HIP graphs %.2e), oracle-vs-golden %.2e | step2 HIP-vs-oracle %.2e, '
              'oracle-vs-golden %.2e' % (seed, d_hip1[-1], d_tf[-1], d_ref1[-1], d_hip2[-1], d_ref2[-1]))
    med, mx = statistics.median, max
    print('step 1 over 8 seeds: HIP vs oracle(here) median %.2e max %.2e; on the HIP graphs max %.2e | oracle(here) vs '
          'reference(build container) median %.2e max %.2e' % (med(d_hip1), mx(d_hip1), mx(d_tf), med(d_ref1), mx(d_ref1)))
    print('step 2 over 8 seeds: HIP vs oracle(here) median %.2e max %.2e | oracle(here) vs reference(build container) '
          'median %.2e max %.2e' % (med(d_hip2), mx(d_hip2), med(d_ref2), mx(d_ref2)))
    assert mx(d_tf) <= 1e-4, d_tf
    assert med(d_hip1) <= 1e-4 and mx(d_hip1) <= 5e-3, d_hip1
    assert med(d_hip2) <= max(2.0 * med(d_ref2), 1e-3), (d_hip2, d_ref2)
    assert mx(d_hip2) <= max(2.0 * mx(d_ref2), 5e-3), (d_hip2, d_ref2)


@pytest.mark.parametrize('model_name', ['DGCNN', 'PTran', 'Pointnet'])
def test_prefix_sharing_is_exact(model_name):
    """SUGStep(share_prefix=True) (the semantic and node pass of a batch share the stage in front of the
    first random sampling: kNN+conv1/conv2 for DGCNN, fc1 + transformer1 for the Point Transformer, both T-Nets +
    conv1/conv2 for PointNet)
    gives bit-identical losses and BN buffers, and the same gradients, as four independent passes."""
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    G = load_golden('step_dgcnn.npz')
    seed = G['seed']
    res = []
    for share in (False, True):
        net = Net_MDA(model_name)
        net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        net = net.cuda().train()
        tr = SUGStep(net, share_prefix=share, fused_adam=False, pair_domains=False)
        torch.manual_seed(seed)
        lc, lg, ls = tr.losses(G['data'].cuda(), G['label'].cuda(), G['data_t'].cuda(), G['label_t'].cuda())
        (lc + lg + ls).backward()
        res.append(([lc.item(), lg.item(), ls.item()],
                    {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                    {k: v.clone() for k, v in net.state_dict().items() if 'running' in k or 'num_batches' in k}))
    if model_name == 'DGCNN':
        assert res[0][0] == res[1][0]
    else:
        assert all(abs(a - b) <= 1e-6 * max(1.0, abs(a)) for a, b in zip(res[0][0], res[1][0])), (res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys()
    gmax = max(float(g.abs().max()) for g in res[0][1].values())
    for k in res[0][1]:     # sharing only changes the order in which upstream gradients are summed
        torch.testing.assert_close(res[1][1][k], res[0][1][k], rtol=1e-4, atol=3e-6 * gmax)
    for k in res[0][2]:
        if model_name == 'DGCNN':
            assert torch.equal(res[0][2][k], res[1][2][k]), k
        else:       # library GEMMs in the transformer blocks: equal to rounding, not bit for bit
            torch.testing.assert_close(res[0][2][k].float(), res[1][2][k].float(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('model_name', ['DGCNN', 'Pointnet', 'Pointnet2', 'PTran'])
def test_pair_domains_match_separate_passes(model_name):
    """SUGStep(pair_domains=True) sends cat(source, target) through the encoder once per pass kind
    with per-domain BatchNorm statistics; losses, gradients and BN buffers must be those of the
    reference's separate forward calls (GEMM shapes differ, so equality is to fp32 rounding)."""
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    G = load_golden('step_dgcnn.npz')
    seed = G['seed']
    res = []
    for pair in (False, True):
        net = Net_MDA(model_name)
        net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        net = net.cuda().train()
        tr = SUGStep(net, fused_adam=False, pair_domains=pair)
        assert tr.pair_domains == pair
        torch.manual_seed(seed)
        lc, lg, ls = tr.losses(G['data'].cuda(), G['label'].cuda(), G['data_t'].cuda(), G['label_t'].cuda())
        (lc + lg + ls).backward()
        res.append(([lc.item(), lg.item(), ls.item()],
                    {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                    {k: v.clone() for k, v in net.state_dict().items() if 'running' in k or 'num_batches' in k}))
    print(res[0][0], res[1][0])
    for a, b in zip(res[0][0], res[1][0]):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys()
    gmax = max(float(g.abs().max()) for g in res[0][1].values())
    worst = max(float((res[1][1][k] - res[0][1][k]).abs().max()) for k in res[0][1])
    print('max grad diff %.3e of gmax %.3e' % (worst, gmax))
    for k in res[0][1]:
        a, b = res[0][1][k], res[1][1][k]
        rel = float((a - b).norm() / (a.norm() + 1e-12))
        if model_name == 'PTran':
            # ReLU stacks + atomically accumulated gather gradients: bounded in norm; a bias that only
            # feeds a BatchNorm (transformer fc2 -> transition-down conv -> BN) has true gradient 0
            if float(a.norm()) < 1e-3 * gmax:
                continue
            assert rel <= 2e-2, (k, rel)
        elif model_name == 'Pointnet2':
            # ReLU heads: on this input one LayerNorm output of c2.mlp1 is +-1e-7, a GEMM of twice
            # the rows rounds it to the other side of the ReLU kink (tools/diag_pair4.py) and that
            # single element carries 5% of the gradient norm; the grouped BN kernels themselves are
            # bit-identical to separate calls (test_gpu_edgeconv.py::test_*_groups).  Bounded in norm.
            if 'mlp_convs' in k and k.endswith('.bias'):
                continue            # a bias in front of BatchNorm: true gradient 0, only rounding noise
            assert rel <= 1e-1, (k, rel)
        else:
            torch.testing.assert_close(b, a, rtol=2e-3, atol=2e-4 * gmax)
    for k in res[0][2]:
        torch.testing.assert_close(res[1][2][k].float(), res[0][2][k].float(), rtol=1e-5, atol=1e-6)


def test_step_with_tuned_library_gemms_matches_default():
    """bench.py's default switches on the recorded TunableOp choices (sug_amd/tuning) and sends the
    weight gradients of the listed shapes to the library: same losses as the untuned step (first
    step 1e-4; the second only loosely -- see test_two_training_steps_match_reference)."""
    import torch.cuda.tunable as tn
    from sug_amd import ops
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    from sug_amd.tuning import enable_tuned_gemms
    G = load_golden('step_dgcnn.npz')
    seed = G['seed']
    data, data_t = G['data'].cuda(), G['data_t'].cuda()
    lab, lab_t = G['label'].cuda(), G['label_t'].cuda()
    res = []
    try:
        for tuned in (False, True):
            if tuned:
                enable_tuned_gemms()        # the table may be ignored on another ROCm stack; the step must still be right
                ops.DW_FORCE_LIBRARY = True  # every weight gradient through the library (the golden batch is not a tuned shape)
            net = Net_MDA('DGCNN')
            net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
            for m in net.modules():
                if isinstance(m, torch.nn.Dropout2d):
                    m.p = 0.0
            tr = SUGStep(net.cuda().train())
            torch.manual_seed(seed)
            res.append([[float(v) for v in tr.step(data, lab, data_t, lab_t)] for _ in range(2)])
    finally:
        tn.enable(False)
        ops.DW_FORCE_LIBRARY = False
        ops.DW_LIBRARY_SHAPES = set()
    for a, b in zip(res[0][0], res[1][0]):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), res
    for a, b in zip(res[0][1], res[1][1]):
        assert abs(a - b) <= 5e-3 * max(1.0, abs(a)), res


def test_graph_mode_follows_eager_through_an_lr_change():
    """hipGraph mode (experimental): eager planning step, capture, replays -- and a learning-rate
    change selects a new graph (the kernels take lr by value) -- follow the eager trajectory; the
    Adam step count kept on the device comes back through state_dict()."""
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    G = load_golden('step_dgcnn.npz')
    seed = G['seed']
    data, data_t = G['data'].cuda(), G['data_t'].cuda()
    lab, lab_t = G['label'].cuda(), G['label_t'].cuda()
    res, steps = [], None
    for use_graph in (False, True):
        net = Net_MDA('DGCNN')
        net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        tr = SUGStep(net.cuda().train(), use_graph=use_graph)
        torch.manual_seed(seed)
        out = []
        for i in range(7):
            if i == 4:
                tr.set_epoch(3, 10)              # cosine for g / c, unchanged for dis (< 5 epochs)
            out.append([float(v) for v in tr.step(data, lab, data_t, lab_t)])
        res.append(out)
        if use_graph:
            sd = tr.optimizer_c.state_dict()
            steps = {float(v['step']) for v in sd['state'].values()}
    assert steps == {7.0}, steps
    assert res[0] == res[1], res          # all seven steps, through the lr change: bit for bit (one fixed summation order, round 4)


def test_fp16_mode_weight_copies_follow_the_optimizer():
    """Point Transformer 16-bit mode: from the second step on the 16-bit weight copies are refreshed by one
    multi-tensor copy into the first step's buffers (ops.w16_prefill).  Three steps with that plan must equal three
    steps that rebuild every copy (plan dropped before each step): a stale buffer would show after the first update."""
    from sug_amd.model.Model import Net_MDA
    from sug_amd.model import Ptran_transformer as PT
    from sug_amd.train_step import SUGStep
    g = torch.Generator().manual_seed(7)
    B, N = 2, 256
    data = (torch.rand(B, 3, N, 1, generator=g) * 2 - 1).cuda()
    data_t = (torch.rand(B, 3, N, 1, generator=g) * 2 - 1).cuda()
    label = torch.randint(0, 10, (B,), generator=g).cuda()
    label_t = torch.randint(0, 10, (B,), generator=g).cuda()
    out = []
    try:
        PT.GEMM_DTYPE, PT.PROJ_16BIT = torch.float16, True
        for keep_plan in (True, False):
            torch.manual_seed(5)
            net = Net_MDA('PTran').cuda().train()
            for m in net.modules():
                if isinstance(m, torch.nn.Dropout2d):
                    m.p = 0.0
            tr = SUGStep(net, lr=2e-3, use_graph=False)
            torch.manual_seed(9)
            losses = []
            for _ in range(3):
                if not keep_plan:
                    tr._w16_plan = None
                losses.append([float(v) for v in tr.step(data, label, data_t, label_t) if v is not None])
            if keep_plan:
                assert tr._w16_plan, 'the step recorded no 16-bit copies'
            out.append((losses, {k: v.clone() for k, v in net.state_dict().items()}))
    finally:
        PT.GEMM_DTYPE, PT.PROJ_16BIT = None, False
    # (a stale copy after the first update would repeat the first step's losses.)  Round 4: the step is reproducible bit for
    # bit -- index_points' backward no longer uses float atomics -- so the two trainers agree exactly, weights included
    assert abs(out[0][0][1][0] - out[0][0][0][0]) > 0.05, 'the first update should move the classification loss'
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k


@pytest.mark.parametrize('model_name', ['DGCNN', 'Pointnet', 'Pointnet2', 'PTran', 'PTran_fp16'])
def test_training_step_issues_no_device_memsets(model_name):
    """A replayed step graph must not contain memset nodes on this stack (they were not reliably
    ordered under replay -- garbage MMD values, NaN weights).  One step under the profiler: no memset on the device.
    The profiled step runs in a child process (tests/memset_probe.py): the profiler's own teardown crashes the process now
    and then on this stack (segmentation fault inside stop_trace), which must not take the test run down."""
    import json
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'memset_probe.py')
    cmd = [sys.executable, probe] + model_name.split('_')
    res = None
    for attempt in range(6):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith('PROBE ')]
        if lines:
            res = json.loads(lines[-1][6:])
            break
        assert r.returncode < 0, 'probe failed (rc %d): %s' % (r.returncode, r.stderr.decode()[-800:])     # killed by a signal: retry
    if res is None:
        pytest.skip('torch.profiler crashed the probe process six times in a row')
    assert res['kernels'] > 0, 'the profiler saw no kernels'
    assert not res['memsets'], res['memsets']


def test_automatic_prefix_sharing_is_exact_and_safe():
    """A plain Net_MDA (share_prefix = 'auto', no SUGStep): the four forwards of train_dg_single_gpu.py:260-310 followed
    by ONE backward reuse the kNN + conv1 + conv2 stage of each batch -- same losses / gradients / BatchNorm buffers as
    with sharing off.  And it is safe outside that pattern: once a backward has consumed the cached graph, the next
    forward on the same input recomputes (no 'backward through the graph a second time')."""
    from sug_amd import ops
    from sug_amd.model.Model import Net_MDA
    from sug_amd.model import mmd
    G = load_golden('step_dgcnn.npz')
    seed = G['seed']
    data, data_t = G['data'].cuda(), G['data_t'].cuda()
    lab, lab_t = G['label'].cuda(), G['label_t'].cuda()
    cfg = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 50, 'GEO_SCALE': 1}
    res, calls = [], []
    real_knn = ops.knn
    for mode in (False, 'auto'):
        net = Net_MDA('DGCNN')
        net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        net = net.cuda().train()
        assert net.g.share_prefix == 'auto'
        net.g.share_prefix = mode
        n = [0]

        def spy(f, k):
            n[0] += 1
            return real_knn(f, k)
        ops.knn = spy
        try:
            torch.manual_seed(seed)
            ys = net(data, semantic_adaption=True)
            yt = net(data_t, semantic_adaption=True)
            fs = net(data, node_adaptation_s=True)
            ft = net(data_t, node_adaptation_t=True)
            loss = torch.nn.functional.cross_entropy(ys[0], lab) + mmd.mmd_cal(lab, fs, lab_t, ft, cfg) + \
                mmd.mmd_cal(lab, ys[2], lab_t, yt[2], cfg)
            loss.backward()
            calls.append(n[0])
            res.append((float(loss), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                        {k: v.clone() for k, v in net.state_dict().items() if 'running' in k}))
            if mode == 'auto':
                # the cached graph is consumed: a further forward + backward on the same input must still work
                n[0] = 0
                y2 = net(data, semantic_adaption=True)
                y2[0].sum().backward()
                assert n[0] == 4, 'after a backward the prefix must be recomputed (%d kNN calls)' % n[0]
        finally:
            ops.knn = real_knn
    assert calls == [16, 12], calls                       # 4 passes x 4 graphs; the node passes reuse 2 each
    assert res[0][0] == res[1][0]
    gmax = max(float(g.abs().max()) for g in res[0][1].values())
    assert res[0][1].keys() == res[1][1].keys()
    for k in res[0][1]:
        torch.testing.assert_close(res[1][1][k], res[0][1][k], rtol=1e-4, atol=3e-6 * gmax)
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


@pytest.mark.parametrize('model_name', ['DGCNN', 'Pointnet', 'PTran'])
def test_prefix_cache_is_keyed_on_tensor_identity_not_on_the_address(model_name):
    """a train-mode forward under no_grad leaves a live cache entry (no backward ever marks it dead);
    the next batch may be allocated at the SAME address with _version 0 again (normal caching-allocator reuse) and must
    not be served the previous batch's prefix.  Two different batches through one address, sharing 'auto', against a
    model with sharing off: bit-identical outputs."""
    from sug_amd.model.Model import Net_MDA
    seed = 21
    nets = []
    for mode in ('auto', False):
        net = Net_MDA(model_name)
        net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
        net = net.cuda().train()
        net.g.share_prefix = mode
        nets.append(net)
    g = torch.Generator().manual_seed(seed)
    xa, xb = O.synth_clouds(2, 1024, g), O.synth_clouds(2, 1024, g)
    outs = []
    with torch.no_grad():
        a = xa.cuda()
        addr = a.data_ptr()
        torch.manual_seed(1)
        nets[0](a, mid_feat=True)
        del a
        b = xb.cuda()
        if b.data_ptr() != addr:
            pytest.skip('the allocator did not reuse the address (%x vs %x)' % (b.data_ptr(), addr))
        assert b._version == 0
        for net in nets:
            torch.manual_seed(2)
            outs.append(net(b, mid_feat=True))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), \
        'a batch at a reused address was served the previous batch\'s cached prefix'


@pytest.mark.parametrize('model_name,N', [('Pointnet2', 2048), ('DGCNN', 1024), ('Pointnet', 1024)])
def test_planned_geometry_of_both_passes_is_the_same_step(monkeypatch, model_name, N):
    """The farthest-point sampling / ball query of the semantic and the node pass of a step (PointNet++'s two SA layers; the
    SA-node module of DGCNN / PointNet) run as ONE set of launches over both passes (plan_geometry) -- same start draws in the same order, same kernels on the same
    coordinates: the classification / semantic losses and the BatchNorm buffers are bit-identical to the pass-by-pass form
    (the geometric MMD term carries the Chamfer weights, whose float atomics are not ordered: 1e-6)."""
    from bench import BENCH_METHODS, synth
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    res = []
    for planned in ('1', '0'):
        monkeypatch.setenv('SUG_PLAN_GEOMETRY', planned)
        net = Net_MDA(model_name)
        net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, 4))
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        net = net.cuda().train()
        tr = SUGStep(net, lr=0.0, weight_decay=0.0, methods=BENCH_METHODS)
        data = synth(4, N, 11, 'cuda')
        torch.manual_seed(123)
        lc, lg, ls = tr.losses(*data)
        (lc + lg + ls).backward()
        nxt = int(torch.randint(0, 1 << 30, (1,)))               # the CPU generator is left where the other form leaves it
        res.append(([float(lc), float(lg), float(ls)], {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                    {k: v.clone() for k, v in net.state_dict().items() if 'running' in k}, nxt))
    assert res[0][0][0] == res[1][0][0] and res[0][0][2] == res[1][0][2], (res[0][0], res[1][0])
    assert abs(res[0][0][1] - res[1][0][1]) <= 1e-6 * abs(res[1][0][1]), (res[0][0], res[1][0])
    assert res[0][3] == res[1][3]
    for k in res[1][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
    gmax = max(float(g.abs().max()) for g in res[1][1].values())
    for k in res[1][1]:
        # (the first SA layer's backward sums over unsorted reverse lists: order not fixed run to run; conv biases in front
        # of BatchNorm have true gradient 0 and carry only that noise)
        torch.testing.assert_close(res[0][1][k], res[1][1][k], rtol=1e-4, atol=1e-5 * gmax)

"""GPU parity of two full SUG training steps (DGCNN, B=4) against the reference run recorded
in tests/golden/step_dgcnn.npz: per-step (loss_cls, loss_geo_mmd, loss_sem_mmd) and
post-step parameter checksums."""
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
    tr = SUGStep(net, lr=1e-3, weight_decay=5e-5, methods=methods, fused_adam=False)
    data, data_t = G['data'].cuda(), G['data_t'].cuda()
    lab, lab_t = G['label'].cuda(), G['label_t'].cuda()
    torch.manual_seed(seed)
    got = []
    for _ in range(2):
        lc, lg, ls = tr.step(data, lab, data_t, lab_t)
        got.append([lc.item(), lg.item(), ls.item()])
    want = G['losses'].tolist()
    print('losses', got, want)
    for a, b in zip(got[0], want[0]):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (got, want)
    for a, b in zip(got[1], want[1]):      # second step sees parameters after one Adam update
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (got, want)
    # Adam's first updates are ~ lr*sign(g): an element whose gradient is rounding noise (e.g. a
    # conv bias in front of BatchNorm, whose true gradient is 0) may move the other way, so the
    # checksums are compared with an allowance of 2% of elements flipping (2 steps of lr each).
    sd = net.state_dict()
    worst = 0.0
    for k, ps, pa in zip(G['names'], G['p_sum'].tolist(), G['p_abs'].tolist()):
        v = sd[k].double()
        allow = 0.02 * 2 * 2 * 1e-3 * v.numel() + 1e-4 * pa + 1e-6
        assert abs(v.sum().item() - ps) <= allow, (k, v.sum().item(), ps, allow)
        assert abs(v.abs().sum().item() - pa) <= allow, (k, v.abs().sum().item(), pa, allow)
        worst = max(worst, abs(v.sum().item() - ps) / allow)
    print('worst checksum deviation / allowance = %.3f' % worst)

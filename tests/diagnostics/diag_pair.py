"""Which loss term carries the pair-vs-separate gradient difference of PointNet++ (diagnostic)."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')  # run from the repo root
from conftest import load_golden
from oracle import ref_cpu as O
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
G = load_golden('step_dgcnn.npz'); seed = G['seed']
for which in ('cls', 'geo', 'sem'):
    res = []
    for pair in (False, True):
        net = Net_MDA('Pointnet2')
        net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0.0
        net = net.cuda().train()
        tr = SUGStep(net, fused_adam=False, pair_domains=pair)
        torch.manual_seed(seed)
        lc, lg, ls = tr.losses(G['data'].cuda(), G['label'].cuda(), G['data_t'].cuda(), G['label_t'].cuda())
        {'cls': lc, 'geo': lg, 'sem': ls}[which].backward()
        res.append({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    for k in ('g.sa3.mlp_bns.2.weight', 'g.sa3.mlp_convs.2.weight', 'g.sa1.mlp_convs.0.weight', 'c1.mlp1.fc.0.weight'):
        if k in res[0]:
            a, b = res[0][k], res[1][k]
            print(which, k, 'rel %.2e  norm %.2e' % (float((a - b).norm() / (a.norm() + 1e-20)), float(a.norm())))

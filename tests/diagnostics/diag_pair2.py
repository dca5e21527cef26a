"""Where does the PointNet++ pair-vs-separate gradient difference enter? (diagnostic)"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')  # run from the repo root
from conftest import load_golden
from oracle import ref_cpu as O
from sug_amd import ops
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
G = load_golden('step_dgcnn.npz'); seed = G['seed']
orig = ops.bn_act_rows
rec = None
def patched(y, bn, slope):
    y.retain_grad()
    out = orig(y, bn, slope)
    out.retain_grad()
    rec.append((y, out))
    return out
ops.bn_act_rows = patched
import sug_amd.model.pointnet2_utils as pu
runs = []
for pair in (False, True):
    rec = []
    net = Net_MDA('Pointnet2')
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    net = net.cuda().train()
    tr = SUGStep(net, fused_adam=False, pair_domains=pair)
    torch.manual_seed(seed)
    lc, lg, ls = tr.losses(G['data'].cuda(), G['label'].cuda(), G['data_t'].cuda(), G['label_t'].cuda(), mmd_on=False)
    lc.backward()
    runs.append(rec)
sep, par = runs
print(len(sep), len(par))
# separate: 9 calls for source sem pass then 9 for target; pair: 9 calls with 2B
for L in range(9):
    ys = torch.cat((sep[L][0], sep[9 + L][0]), 0); os_ = torch.cat((sep[L][1], sep[9 + L][1]), 0)
    z = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)
    gy = torch.cat((z(sep[L][0]), z(sep[9 + L][0])), 0); go = torch.cat((z(sep[L][1]), z(sep[9 + L][1])), 0)
    yp, op = par[L]
    r = lambda a, b: float((a - b).norm() / (a.norm() + 1e-20))
    print('layer %d shape %s: y %.1e out %.1e | grad_out %.1e grad_y %.1e' % (L, tuple(yp.shape), r(ys, yp), r(os_, op), r(go, op.grad), r(gy, yp.grad)))
L = 8
z = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)
for half, (a, b) in enumerate(((z(sep[L][1]), par[L][1].grad[:4]), (z(sep[9 + L][1]), par[L][1].grad[4:]))):
    print('half', half, 'norm sep %.4e pair %.4e  diff %.3e  nnz sep %d pair %d  same-pos %d' % (
        float(a.norm()), float(b.norm()), float((a - b).norm()), int((a != 0).sum()), int((b != 0).sum()),
        int(((a != 0) & (b != 0)).sum())))
    fa, fb = a.sum(dim=2), b.sum(dim=2)      # = dL/dfeat
    print('   dL/dfeat diff %.3e of %.3e' % (float((fa - fb).norm()), float(fa.norm())))

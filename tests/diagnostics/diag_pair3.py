import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')  # run from the repo root
from conftest import load_golden
from oracle import ref_cpu as O
from sug_amd import ops
from sug_amd.model.Model import Net_MDA, Pointnet_c
from sug_amd.train_step import SUGStep
G = load_golden('step_dgcnn.npz'); seed = G['seed']
rec = None
orig_c = Pointnet_c.forward
def pc(self, x, adapt=False):
    x.retain_grad()
    out = orig_c(self, x, adapt)
    out[0].retain_grad(); out[1].retain_grad()
    rec.append((x, out[0], out[1]))
    return out
Pointnet_c.forward = pc
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
    runs.append((rec, float(lc), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
(sep, l0, g0), (par, l1, g1) = runs
print('loss', l0, l1, 'calls', len(sep), len(par))
r = lambda a, b: float((a - b).norm() / (a.norm() + 1e-20))
for h in range(2):
    xs, ys, fs = sep[h]; xp, yp, fp = par[h]
    print('head', h, 'x %.1e y %.1e f %.1e | gy %.2e gf %s gx %.2e' % (r(xs, xp[:4]), r(ys, yp[:4]), r(fs, fp[:4]),
          r(ys.grad, yp.grad[:4]), 'none' if fs.grad is None else '%.2e' % r(fs.grad, fp.grad[:4]), r(xs.grad, xp.grad[:4])))
    print('    x is same tensor across heads:', sep[0][0] is sep[1][0], ' gx norms', float(xs.grad.norm()), float(xp.grad[:4].norm()))
for k in g0:
    if k.startswith(('c1', 'c2')):
        print(k, '%.2e' % r(g0[k], g1[k]))
# fp64 reference of the heads' gradients from the recorded encoder output
import copy
lab = G['label']
c1 = copy.deepcopy(net.c1).cpu().double(); c2 = copy.deepcopy(net.c2).cpu().double()
x64 = sep[0][0].detach().cpu().double().requires_grad_(True)
ce = torch.nn.CrossEntropyLoss()
M = tr.methods
y1, _ = orig_c(c1, x64, True); y2, _ = orig_c(c2, x64, True)
loss = M['CLS_WEIGHT'] * M['SRC_LOSS_WEIGHT'] * (0.5 * ce(y1, lab) + 0.5 * ce(y2, lab))
loss.backward()
print('weights', M['CLS_WEIGHT'], M['SRC_LOSS_WEIGHT'], 'ref loss', float(loss))
ref = {('c1.' + k): p.grad for k, p in c1.named_parameters()}
ref.update({('c2.' + k): p.grad for k, p in c2.named_parameters()})
for k in ('c1.mlp1.fc.0.weight', 'c2.mlp1.fc.0.weight', 'c2.mlp1.fc.1.weight', 'c2.mlp2.fc.0.weight'):
    print(k, 'sep vs fp64 %.2e   pair vs fp64 %.2e' % (r(ref[k], g0[k].cpu().double()), r(ref[k], g1[k].cpu().double())))
print('dfeat sep vs fp64 %.2e  pair vs fp64 %.2e' % (r(x64.grad, sep[0][0].grad.cpu().double()), r(x64.grad, par[0][0].grad[:4].cpu().double())))

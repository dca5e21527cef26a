import sys, copy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')  # run from the repo root
from oracle import ref_cpu as O
from sug_amd.model.Model import Net_MDA
net = Net_MDA('Pointnet2')
net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, 1234))
torch.manual_seed(0)
f = torch.randn(8, 1024)
g = torch.randn(8, 512); g[4:] = 0
for name in ('c1', 'c2'):
    m = getattr(net, name).mlp1
    ref = copy.deepcopy(m).double()
    xr = f.double().requires_grad_(True); ref(xr).backward(g.double())
    mg = copy.deepcopy(m).cuda()
    res = {}
    for B in (4, 8):
        x = f[:B].cuda().requires_grad_(True)
        for p in mg.parameters():
            p.grad = None
        out = mg(x); out.backward(g[:B].cuda())
        res[B] = (x.grad[:4].cpu(), [p.grad.cpu() for p in mg.parameters()], out[:4].detach().cpu())
    r = lambda a, b: float((a.double() - b).norm() / (b.norm() + 1e-30))
    print(name, 'LN weight min|w| %.2e' % float(m.fc[1].weight.abs().min()), 'out rows equal', torch.equal(res[4][2], res[8][2]))
    for B in (4, 8):
        print('  B=%d: dX vs fp64 %.2e ; params vs fp64 %s' % (B, r(res[B][0], xr.grad[:4]),
              ['%.1e' % r(a, b.grad) for a, b in zip(res[B][1], ref.parameters())]))

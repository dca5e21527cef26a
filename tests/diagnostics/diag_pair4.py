import sys, copy, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')  # run from the repo root
from conftest import load_golden
from oracle import ref_cpu as O
from sug_amd.model.Model import Net_MDA, Pointnet_c
from sug_amd.train_step import SUGStep
G = load_golden('step_dgcnn.npz'); seed = G['seed']
def hook_all(head, store):
    mods = [('mlp1.lin', head.mlp1.fc[0]), ('mlp1.ln', head.mlp1.fc[1]), ('mlp1.act', head.mlp1.fc[2]), ('drop1', head.dropout1),
            ('mlp2.lin', head.mlp2.fc[0]), ('mlp2.ln', head.mlp2.fc[1]), ('mlp2.act', head.mlp2.fc[2])]
    for name, m in mods:
        def fh(mod, inp, out, name=name):
            store.setdefault(name + '.outs', []).append(out.detach().clone())
        def bh(mod, gin, gout, name=name):
            store[name + '.gout'] = gout[0].detach().clone()
            if gin[0] is not None:
                store[name + '.gin'] = gin[0].detach().clone()
        m.register_forward_hook(fh); m.register_full_backward_hook(bh)
stores = []
feat = None
for pair in (False, True):
    net = Net_MDA('Pointnet2')
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    net = net.cuda().train()
    st = {}
    hook_all(net.c2, st)
    tr = SUGStep(net, fused_adam=False, pair_domains=pair)
    torch.manual_seed(seed)
    lc, lg, ls = tr.losses(G['data'].cuda(), G['label'].cuda(), G['data_t'].cuda(), G['label_t'].cuda(), mmd_on=False)
    lc.backward()
    stores.append(st)
r = lambda a, b: float((a.double() - b.double()).norm() / (a.double().norm() + 1e-30))
# in separate mode hooks fire twice (source, then target pass): forward hooks keep the last (target) -> compare grads only
for k in sorted(stores[1]):
    a, b = stores[0][k], stores[1][k]
    if 'out' in k:
        continue
    # separate mode: backward hooks fire for the target pass first? keep whichever is nonzero
    print('%-12s sep shape %s norm %.3e | pair[:4] norm %.3e | rel %.2e' % (k, tuple(a.shape), float(a.norm()), float(b[:4].norm()), r(a, b[:4])))

a = stores[0]['mlp1.ln.outs'][0]; b = stores[1]['mlp1.ln.outs'][0][:4]
flip = (a > 0) != (b > 0)
print('LN out sign flips:', int(flip.sum()), 'of', a.numel(), ' max|a-b| %.3e' % float((a - b).abs().max()))
print('values at flips (sep):', a[flip][:10].tolist())
print('values at flips (pair):', b[flip][:10].tolist())
x0 = stores[0]['mlp1.lin.outs'][0]; x1 = stores[1]['mlp1.lin.outs'][0][:4]
print('lin out max diff %.3e (scale %.3e)' % (float((x0 - x1).abs().max()), float(x0.abs().max())))

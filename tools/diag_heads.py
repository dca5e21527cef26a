"""Heads on [8,1024] (rows 4.. unused by the loss) vs on [4,1024]: is dL/dfeat the same? (diagnostic)"""
import sys, torch
sys.path.insert(0, '.')
from sug_amd.model.Model import Pointnet_c
torch.manual_seed(0)
for dg in (False, True):
    c = Pointnet_c(dgcnn_flag=dg).cuda().train()
    for m in c.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    f = torch.randn(8, 1024, device='cuda')
    lab = torch.randint(0, 10, (4,), device='cuda')
    ce = torch.nn.CrossEntropyLoss()
    fa = f[:4].clone().requires_grad_(True)
    y, _ = c(fa, adapt=True); ce(y, lab).backward()
    fb = f.clone().requires_grad_(True)
    y2, _ = c(fb, adapt=True); ce(y2[:4], lab).backward()
    print('dgcnn_flag', dg, 'out diff %.2e' % float((y - y2[:4]).abs().max()),
          'dfeat rel diff %.3e' % float((fa.grad - fb.grad[:4]).norm() / fa.grad.norm()), 'unused rows grad %.2e' % float(fb.grad[4:].norm()))
    # layer by layer
    x = f.clone()
    for name, mod in (('mlp1', c.mlp1), ('drop1', c.dropout1), ('mlp2', c.mlp2), ('drop2', c.dropout2), ('mlp3', c.mlp3)):
        xa = x[:4].clone().requires_grad_(True); xb = x.clone().requires_grad_(True)
        oa, ob = mod(xa), mod(xb)
        g = torch.randn_like(ob)
        oa.backward(g[:4]); gb = g.clone(); gb[4:] = 0; ob.backward(gb)
        print('   %-6s fwd %.2e bwd rel %.3e' % (name, float((oa - ob[:4]).abs().max()), float((xa.grad - xb.grad[:4]).norm() / xa.grad.norm())))
        x = ob.detach()

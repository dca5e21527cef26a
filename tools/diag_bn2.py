import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from sug_amd import ops
torch.manual_seed(0)
B, N = 4, 1024
x = torch.randn(B, N, 3)
W1, W2 = torch.randn(64, 3) * 0.5, torch.randn(128, 64) * 0.1
g1, b1 = 1 + 0.2 * torch.randn(64), 0.1 * torch.randn(64)
g2, b2 = 1 + 0.2 * torch.randn(128), 0.1 * torch.randn(128)
probe = torch.randn(B, N, 128)

def ref64():
    xd = x.double().requires_grad_(True)
    h = F.relu(F.batch_norm(F.linear(xd, W1.double()).reshape(-1, 64), None, None, g1.double(), b1.double(), True)).view(B, N, 64)
    o = F.relu(F.batch_norm(F.linear(h, W2.double()).reshape(-1, 128), None, None, g2.double(), b2.double(), True)).view(B, N, 128)
    (o * probe.double()).sum().backward()
    return xd.grad

def run(own1, own2):
    xg = x.cuda().requires_grad_(True)
    bn1, bn2 = torch.nn.BatchNorm1d(64).cuda(), torch.nn.BatchNorm1d(128).cuda()
    with torch.no_grad():
        bn1.weight.copy_(g1.cuda()); bn1.bias.copy_(b1.cuda()); bn2.weight.copy_(g2.cuda()); bn2.bias.copy_(b2.cuda())
    def layer(h, W, bn, own):
        y = F.linear(h, W.cuda())
        if own:
            return ops.bn_act_rows(y, bn, 0.0)
        C = y.shape[-1]
        return F.relu(F.batch_norm(y.reshape(-1, C), bn.running_mean, bn.running_var, bn.weight, bn.bias, True, 0.1, 1e-5)).view(y.shape)
    o = layer(layer(xg, W1, bn1, own1), W2, bn2, own2)
    (o * probe.cuda()).sum().backward()
    return xg.grad.cpu().double()

r = ref64()
for own1 in (False, True):
    for own2 in (False, True):
        g = run(own1, own2)
        print('own64=%s own128=%s  rel err of dx vs fp64: %.3e' % (own1, own2, float((g - r).norm() / r.norm())))

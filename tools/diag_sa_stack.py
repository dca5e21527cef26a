"""Diagnostic: a set-abstraction layer with point features (D = 128: PointNet++'s sa2) step by step -- every intermediate of
the product path and its gradient against fp64 torch on the same index sets.  usage: python tools/diag_sa_stack.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from sug_amd import ops
from sug_amd.model.pointnet2_utils import sample_and_group_idx
rel = lambda a, b: float((a.detach().double().cpu().reshape(b.shape) - b.detach().cpu()).norm() / (b.detach().cpu().norm() + 1e-300))
torch.manual_seed(0)
B, N, D, S, ns, radius = 4, 512, 128, 64, 64, 0.45
mlp = [128, 128, 256]
xyz = (torch.rand(B, N, 3) - 0.5).cuda()
pts = torch.relu(torch.randn(B, N, D) * 0.7 + 0.3).cuda().requires_grad_(True)
if len(sys.argv) > 1 and sys.argv[1] == 'module':          # the parameters of a real PointNetSetAbstraction (default init)
    from sug_amd.model.pointnet2_utils import PointNetSetAbstraction
    g = torch.Generator().manual_seed(D + 1)
    xyz = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()
    pts = (torch.relu(torch.randn(B, N, D, generator=g) * 0.7 + 0.3)).cuda().requires_grad_(True)
    sa = PointNetSetAbstraction(S, radius, ns, 3 + D, mlp, False)
    torch.manual_seed(3)
    for m in sa.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.uniform_(-0.2, 0.2)
    sa = sa.cuda().train()
    Ws = [c.weight.detach().view(c.weight.shape[0], -1).clone().requires_grad_(True) for c in sa.mlp_convs]
    bs = [c.bias.detach().clone().requires_grad_(True) for c in sa.mlp_convs]
    bns = list(sa.mlp_bns)
else:
    Ws = [(torch.randn(mlp[0], 3 + D) / 8).cuda().requires_grad_(True), (torch.randn(mlp[1], mlp[0]) / 11).cuda().requires_grad_(True),
          (torch.randn(mlp[2], mlp[1]) / 11).cuda().requires_grad_(True)]
    bs = [(torch.randn(c) * 0.1).cuda().requires_grad_(True) for c in mlp]
    bns = [torch.nn.BatchNorm2d(c).cuda().train() for c in mlp]
    for bn in bns:
        bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-0.2, 0.2)
torch.manual_seed(11)
new_xyz, idx = sample_and_group_idx(S, radius, ns, xyz)
# product path
P = ops.linear_rows(torch.cat((xyz, pts), -1), Ws[0])
Q = ops.sub_row_bias(ops.linear_rows(new_xyz, Ws[0][:, :3]), bs[0])
g0 = ops.sa_first_layer(P, Q, idx, bns[0])
g1 = ops.bn_act_rows(ops.linear_rows(g0, Ws[1], bs[1]), bns[1], 0.0)
out = ops.pointmlp_max(g1, Ws[2], bs[2], bns[2], 0.0, ns).view(B, S, -1)
for t in (P, Q, g0, g1):
    t.retain_grad()
# fp64
xd, cd, pd = xyz.double().cpu(), new_xyz.double().cpu(), pts.detach().double().cpu().requires_grad_(True)
Wd = [w.detach().double().cpu().requires_grad_(True) for w in Ws]
bd = [b.detach().double().cpu().requires_grad_(True) for b in bs]
gam = [bn.weight.detach().double().cpu().requires_grad_(True) for bn in bns]
bet = [bn.bias.detach().double().cpu().requires_grad_(True) for bn in bns]
bi = torch.arange(B).view(B, 1, 1)
il = idx.long().cpu()
Pd = torch.cat((xd, pd), -1) @ Wd[0].t()
Qd = cd @ Wd[0][:, :3].t() - bd[0]
Pd.retain_grad(); Qd.retain_grad()
z0 = Pd[bi, il] - Qd.unsqueeze(2)
g0d = torch.relu(F.batch_norm(z0.reshape(-1, mlp[0]), None, None, gam[0], bet[0], True, 0.1, 1e-5)).view(B, S, ns, -1)
g0d.retain_grad()
g1d = torch.relu(F.batch_norm(g0d.reshape(-1, mlp[0]) @ Wd[1].t() + bd[1], None, None, gam[1], bet[1], True, 0.1, 1e-5)).view(B, S, ns, -1)
g1d.retain_grad()
A = torch.relu(F.batch_norm(g1d.reshape(-1, mlp[1]) @ Wd[2].t() + bd[2], None, None, gam[2], bet[2], True, 0.1, 1e-5)).view(B, S, ns, -1)
top = A.topk(2, dim=2)[0]
clear = ((top[:, :, 0] - top[:, :, 1]) > 1e-4).double()
probe = torch.randn(B, S, mlp[2], dtype=torch.float64) * clear
(top[:, :, 0] * probe).sum().backward()
(out * probe.float().cuda()).sum().backward()
print('forward:  P %.1e  Q %.1e  g0 %.1e  g1 %.1e  out %.1e' % (rel(P, Pd), rel(Q, Qd), rel(g0, g0d), rel(g1, g1d),
                                                               float(((out.double().cpu() - top[:, :, 0]) * clear).norm() / top[:, :, 0].norm())))
print('gradient: g1 %.1e  g0 %.1e  P %.1e  Q %.1e  points %.1e' % (rel(g1.grad, g1d.grad), rel(g0.grad, g0d.grad), rel(P.grad, Pd.grad),
                                                                 rel(Q.grad, Qd.grad), rel(pts.grad, pd.grad)))
print('weights:  ' + '  '.join('W%d %.1e' % (i, rel(Ws[i].grad, Wd[i].grad)) for i in range(3)) + '  ' +
      '  '.join('gamma%d %.1e beta%d %.1e' % (i, rel(bns[i].weight.grad, gam[i].grad), i, rel(bns[i].bias.grad, bet[i].grad)) for i in range(3)))
ko = int(((g1.detach().cpu() > 0) != (g1d.detach() > 0)).sum())
print('ReLU masks of layer 1 that differ: %d of %d; of layer 0: %d' % (ko, g1d.numel(), int(((g0.detach().cpu() > 0) != (g0d.detach() > 0)).sum())))

"""Diagnostic: forward and backward of the first set-abstraction layer on the neighbour lists (ops.sa_first_layer:
relu(bn(P[idx] - Q))) against fp64 torch, for the two channel widths in use.  usage: python tools/diag_sa_first.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from sug_amd import ops
rel = lambda a, b: float((a.double().cpu() - b.cpu()).norm() / (b.cpu().norm() + 1e-300))
for C, N, S, ns, feat in ((64, 512, 128, 32, 0.0), (128, 512, 64, 64, 1.0), (128, 512, 64, 64, 0.0), (64, 512, 64, 64, 1.0)):
    torch.manual_seed(C + ns)
    B = 4
    # P per point (a feature part of size `feat` on top of an xyz part), Q per centroid
    P = (torch.randn(B, N, C) * (0.3 + feat) + feat * 0.5).cuda().requires_grad_(True)
    Q = (torch.randn(B, S, C) * 0.3).cuda().requires_grad_(True)
    idx = torch.randint(0, N, (B, S, ns), dtype=torch.int32).cuda()
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-0.2, 0.2)
    out = ops.sa_first_layer(P, Q, idx, bn)                       # [B,S,ns,C]
    probe = torch.randn_like(out)
    got = torch.autograd.grad((out * probe).sum(), [P, Q, bn.weight, bn.bias])
    Pd, Qd = P.detach().double().cpu().requires_grad_(True), Q.detach().double().cpu().requires_grad_(True)
    wd, bd = bn.weight.detach().double().cpu().requires_grad_(True), bn.bias.detach().double().cpu().requires_grad_(True)
    bi = torch.arange(B).view(B, 1, 1)
    z = Pd[bi, idx.long().cpu()] - Qd.unsqueeze(2)
    outd = torch.relu(F.batch_norm(z.reshape(-1, C), None, None, wd, bd, True, 0.1, bn.eps)).view(B, S, ns, C)
    ref = torch.autograd.grad((outd * probe.double().cpu()).sum(), [Pd, Qd, wd, bd])
    print('C %3d ns %2d feat %.0f: forward %.1e | dP %.1e dQ %.1e dgamma %.1e dbeta %.1e' % (
        C, ns, feat, rel(out, outd), *[rel(a, b) for a, b in zip(got, ref)]))

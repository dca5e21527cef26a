"""Diagnostic: precision of the BatchNorm + ReLU backward over rows (sug_bn_act_rows_bwd) for a SPARSE incoming gradient (the
gradient of a max over groups of rows: one non-zero row per group and channel), against fp64 torch.  usage: python tools/diag_bn_sparse.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops
torch.manual_seed(0)
for rows, C, seg in ((512, 1024, 128), (512, 512, 128), (65536, 64, 32)):
    y = (torch.randn(rows, C) * 0.7 + 0.3).cuda().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-0.2, 0.2)
    out = ops.bn_act_rows(y, bn, 0.0)
    pooled = out.view(rows // seg, seg, C).max(1)[0]
    probe = torch.randn_like(pooled)
    (gy, gw, gb) = torch.autograd.grad((pooled * probe).sum(), [y, bn.weight, bn.bias])
    yd = y.detach().double().requires_grad_(True)
    wd, bd = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    outd = torch.relu(torch.nn.functional.batch_norm(yd, None, None, wd, bd, True, 0.1, bn.eps))
    pd = outd.view(rows // seg, seg, C).max(1)[0]
    (ry, rw, rb) = torch.autograd.grad((pd * probe.double()).sum(), [yd, wd, bd])
    y32 = y.detach().clone().requires_grad_(True)
    w32, b32 = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    out32 = torch.relu(torch.nn.functional.batch_norm(y32, None, None, w32, b32, True, 0.1, bn.eps))
    (ty, tw, tb) = torch.autograd.grad((out32.view(rows // seg, seg, C).max(1)[0] * probe).sum(), [y32, w32, b32])
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    print('rows %6d C %4d seg %3d:  dy own %.2e torch-fp32 %.2e | column sums of dy: own %.2e torch-fp32 %.2e (|sum| / |dy| %.1e) | dgamma %.2e %.2e dbeta %.2e %.2e'
          % (rows, C, seg, rel(gy, ry), rel(ty, ry), float((gy.double().sum(0) - ry.sum(0)).norm() / ry.norm()),
             float((ty.double().sum(0) - ry.sum(0)).norm() / ry.norm()), float(ry.sum(0).norm() / ry.norm()),
             rel(gw, rw), rel(tw, rw), rel(gb, rb), rel(tb, rb)))

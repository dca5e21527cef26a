"""Achieved HBM rate of the streaming BatchNorm kernels (bn_act_rows forward / backward) at the PointNet++ config-3
layer shapes.  usage: python tools/bench_stream.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops

torch.manual_seed(0)
for rows, C in ((2097152, 64), (1048576, 128), (65536, 64), (65536, 512)):
    y = torch.randn(rows, C, device='cuda', requires_grad=True)
    g = torch.randn(rows, C, device='cuda')
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(7):
        with ops.bn_groups(2):
            ev[0].record()
            z = ops.bn_act_rows(y, bn, 0.0)
            ev[1].record()
            z.backward(g)
            ev[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf += ev[0].elapsed_time(ev[1]) / 5
            tb += ev[1].elapsed_time(ev[2]) / 5
    nb = rows * C * 4
    # forward: read y (stats), read y + write z (apply) = 3 passes; backward: read g, y + write a; read a, y + write dy = 6
    print('[%8d, %3d]  fwd %6.0f us = %.2f TB/s (3 passes)   bwd %6.0f us = %.2f TB/s (6 passes)' % (
        rows, C, tf * 1e3, 3 * nb / tf / 1e9, tb * 1e3, 6 * nb / tb / 1e9))

"""Time the set-abstraction first-layer kernels (sug_sa_first_fwd / bwd) at the config-3 shapes.
usage: python tools/bench_sagroup.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops

torch.manual_seed(0)
for name, (B, N, S, ns, D, C, r) in {'sa1': (128, 2048, 512, 32, 0, 64, 0.2), 'sa2': (128, 512, 128, 64, 128, 128, 0.4)}.items():
    xyz = torch.rand(B, N, 3, device='cuda') * 2 - 1
    new_xyz = xyz[:, :S].contiguous()
    idx = ops.ball_query(xyz, new_xyz, r, ns)
    P = torch.randn(B, N, C, device='cuda', requires_grad=True)
    Q = torch.randn(B, S, C, device='cuda', requires_grad=True)
    bn = torch.nn.BatchNorm1d(C).cuda().train()
    gz = torch.randn(B, S, ns, C, device='cuda')
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(6):
        with ops.bn_groups(2):
            ev[0].record()
            z = ops.sa_first_layer(P, Q, idx, bn)
            ev[1].record()
            z.backward(gz)
            ev[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf += ev[0].elapsed_time(ev[1]) / 4
            tb += ev[1].elapsed_time(ev[2]) / 4
    rows = B * S * ns
    print('%s: rows %d C %d  fwd %.0f us  bwd %.0f us  (gz %.0f MB)' % (name, rows, C, tf * 1e3, tb * 1e3, rows * C * 4 / 1e6))

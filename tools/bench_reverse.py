"""Timing of sug_knn_reverse on neighbour lists of low and high hubness (64 clouds x 1024 points, k = 20).
usage: python tools/bench_reverse.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops

torch.manual_seed(0)
B, N = 64, 1024
cases = {'xyz': torch.rand(B, N, 3, device='cuda'),
         'feat64': torch.randn(B, N, 64, device='cuda') * 0.3 + torch.randn(B, 1, 64, device='cuda')}
for name, x in cases.items():
    idx = ops.knn(x, 20)
    off, _ = ops.knn_reverse(idx)
    cnt = (off[:, 1:] - off[:, :-1])
    for _ in range(3):
        ops.knn_reverse(idx)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        ops.knn_reverse(idx)
    t1.record()
    torch.cuda.synchronize()
    print('%s: %.1f us per call; list length max %d, mean of per-cloud max %.1f, share > 32: %.3f' % (
        name, t0.elapsed_time(t1) * 50, int(cnt.max()), float(cnt.max(dim=1)[0].float().mean()),
        float((cnt > 32).float().mean())))

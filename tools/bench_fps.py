"""Time sug_fps alone (GPU time under hipGraph replay).  usage: python tools/bench_fps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops

def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    g.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3

for B, N, S in ((64, 1024, 64), (128, 2048, 512), (128, 512, 128), (32, 2048, 512), (32, 512, 128), (32, 128, 32)):
    xyz = torch.rand(B, N, 3, device='cuda')
    start = torch.zeros(B, dtype=torch.int32, device='cuda')
    us = t(lambda: ops.fps(xyz, S, start))
    print('B=%4d N=%5d npoint=%4d  %7.1f us  %.3f us / iteration' % (B, N, S, us, us / S))

"""Time sug_knn (current library) at the benchmark shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops

torch.manual_seed(0)
for B in (32, 64):
    for C in (3, 64, 128):
        x = torch.randn(B, 1024, C, device='cuda')
        for _ in range(3):
            ops.knn(x, 20)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            ops.knn(x, 20)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        fl = B * 1024 * 1024 * (2 * C + 3)
        print('B=%d C=%3d  %7.1f us  %.1f TFLOP/s (%.2f of 157.3)' % (B, C, us, fl / us / 1e6, fl / us / 1e6 / 157.3))

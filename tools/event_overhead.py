"""What does a HIP event pair read around a kernel that runs back to back with its neighbours?  Queues, behind a spin kernel,
(a) empty event pairs, (b) pairs around a tiny kernel, (c) pairs around sug_knn at C = 128 with and without a large
elementwise kernel in front (dirty L2 lines), and compares with the live (un-queued) reading."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops
x = torch.randn(64, 1024, 128, device='cuda')
big = torch.randn(64 * 1024 * 256, device='cuda')
one = torch.zeros(1, device='cuda')
def pair(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record()
    return a, b
def run(queued, fn, pre=None, n=10):
    ev = []
    torch.cuda.synchronize()
    if queued:
        torch.cuda._sleep(int(0.02 * 2.1e9))
    for _ in range(n):
        if pre: pre()
        ev.append(pair(fn))
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return v[len(v) // 2]
for _ in range(3): ops.knn(x, 20)
for q in (False, True):
    print('queued' if q else 'live  ', 'empty pair %.1f us | tiny kernel %.1f | knn C=128 %.1f | knn after a 67 MB elementwise kernel %.1f' % (
        run(q, lambda: None), run(q, lambda: one.add_(1)), run(q, lambda: ops.knn(x, 20)), run(q, lambda: ops.knn(x, 20), pre=lambda: big.mul_(1.0))))

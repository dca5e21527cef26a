"""Time sug_pointmlp_max forward / backward at the benchmark shapes (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, B, seg, K, Co in (('pointnet conv5 64 clouds', 64, 1024, 128, 1024), ('sa1 last 128 clouds-eq', 128 * 512, 32, 64, 128),
                            ('sa2 last', 128 * 128, 64, 128, 256)):
    x = torch.randn(B * seg, K, device='cuda')
    W = torch.randn(Co, K, device='cuda') / K ** 0.5
    b = torch.randn(Co, device='cuda') * 0.1
    bn = torch.nn.BatchNorm1d(Co).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(-1, 1, Co))
    probe = torch.randn(B, Co, device='cuda')
    with torch.no_grad():
        t_f = timeit(lambda: ops.pointmlp_max(x, W, b, bn, 0.0, seg))
    xi, Wi = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    out = ops.pointmlp_max(xi, Wi, b, bn, 0.0, seg)
    t_b = timeit(lambda: torch.autograd.grad((out * probe).sum(), (xi, Wi), retain_graph=True))
    fl = 2.0 * B * seg * K * Co
    # the unfused composition
    def unf():
        y = torch.nn.functional.relu(bn(x @ W.t() + b))
        return y.view(B, seg, Co).max(dim=1)[0]
    with torch.no_grad():
        t_u = timeit(unf, 5)
    print('%-26s rows %8d K %3d Co %4d: fwd %8.1f us (%.1f TFLOP/s), bwd %8.1f us, unfused torch fwd %8.1f us' %
          (name, B * seg, K, Co, t_f, fl / t_f / 1e6, t_b, t_u))

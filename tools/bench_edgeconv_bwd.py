"""Diagnostic: device time of the EdgeConv layer backward kernels (reverse lists, BN sums, LDS-resident scatter) at the DGCNN\nlayer shapes, 64 clouds x 1024 points, k = 20 (torch.profiler kernel timestamps).  usage: python tools/bench_edgeconv_bwd.py"""
import os, sys, time, torch, ctypes
sys.path.insert(0, os.getcwd())
from sug_amd import ops
from sug_amd.model.model_utils import conv_2d
torch.manual_seed(0)
for C, Co in ((64, 64), (64, 128), (128, 256)):
    # spatially smooth features of a 3-D cloud (what the step feeds these layers): in-degrees of the kNN graph like the real ones
    pts = torch.rand(64, 1024, 3, device='cuda') * 2 - 1
    x = torch.tanh(pts @ torch.randn(3, C, device='cuda') + 0.1 * torch.randn(64, 1024, C, device='cuda')).requires_grad_(True)
    idx = ops.knn(x.detach(), 20)
    layer = conv_2d(2 * C, Co, 1, activation='leakyrelu', bias=False).cuda().train()
    y = layer.edge_rows(x, idx)
    g = torch.randn_like(y)
    for _ in range(3):
        (gx,) = torch.autograd.grad(y, x, g, retain_graph=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            (gx,) = torch.autograd.grad(y, x, g, retain_graph=True)
        torch.cuda.synchronize()
    t = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            t.setdefault(e.name[:40], []).append(e.device_time)
    print('C=%d Co=%d' % (C, Co), {k: round(sum(v) / len(v), 1) for k, v in t.items() if 'edgeconv' in k or 'reverse' in k or 'col_reduce' in k})
    print('   checksum', float(gx.double().sum()), float(gx.double().abs().sum()))

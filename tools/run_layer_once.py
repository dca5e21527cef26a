"""A few launches of one hot kernel at the C2 paired shape (64 clouds x 1024 points, k = 20) for rocprofv3 passes.
usage: python tools/run_layer_once.py knn C | edgeconv C Co | edgeconv_fwd C Co | pointmlp_sa K Co seg | pointmlp"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops
from sug_amd.model.model_utils import conv_2d

what = sys.argv[1]
torch.manual_seed(0)
B, N = 64, 1024
if what == 'knn':
    C = int(sys.argv[2])
    x = torch.randn(B, N, C, device='cuda')
    for _ in range(3):
        ops.knn(x, 20)
elif what == 'edgeconv':
    C, Co = int(sys.argv[2]), int(sys.argv[3])
    x = (torch.randn(B, N, C, device='cuda') * 0.3 + torch.randn(B, 1, C, device='cuda')).requires_grad_(True)
    idx = ops.knn(x.detach(), 20)
    m = conv_2d(2 * C, Co, 1, activation='leakyrelu', bias=False).cuda().train()
    with ops.bn_groups(2):
        for _ in range(3):
            y = m.edge_rows(x, idx)
            y.square().sum().backward()
elif what == 'edgeconv_fwd':          # forward only, no autograd: the fused layer without the [P|Q] side output
    C, Co = int(sys.argv[2]), int(sys.argv[3])
    x = torch.randn(B, N, C, device='cuda') * 0.3 + torch.randn(B, 1, C, device='cuda')
    idx = ops.knn(x, 20)
    m = conv_2d(2 * C, Co, 1, activation='leakyrelu', bias=False).cuda().train()
    with torch.no_grad(), ops.bn_groups(2):
        for _ in range(5):
            m.edge_rows(x, idx)
elif what == 'pointmlp_sa':           # last set-abstraction layer of config 3, one domain group: K, Co, segment length
    K, Co, seg = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rows = 64 * 512 * 32 if seg == 32 else 64 * 128 * 64
    x = torch.randn(rows, K, device='cuda')
    W = torch.randn(Co, K, device='cuda') / K ** 0.5
    bn = torch.nn.BatchNorm1d(Co).cuda().train()
    with torch.no_grad():
        for _ in range(3):
            ops.pointmlp_max(x, W, None, bn, 0.0, seg)
else:
    x = torch.randn(B * N, 128, device='cuda', requires_grad=True)
    W = (torch.randn(1024, 128, device='cuda') / 11).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(1024).cuda().train()
    with ops.bn_groups(2):
        for _ in range(3):
            ops.pointmlp_max(x, W, None, bn, 0.0, N).square().sum().backward()
torch.cuda.synchronize()

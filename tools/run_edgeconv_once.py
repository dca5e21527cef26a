import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops
from sug_amd.model.model_utils import conv_2d
C, Co = 128, 256
torch.manual_seed(0)
x = torch.randn(32, 1024, C, device='cuda') * 0.3 + torch.randn(32, 1, C, device='cuda')
idx = ops.knn(x, 20)
m = conv_2d(2 * C, Co, 1, activation='leakyrelu', bias=False).cuda().train()
with torch.no_grad():
    for _ in range(3):
        y = m.edge_rows(x, idx)
torch.cuda.synchronize()

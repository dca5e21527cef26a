"""Diagnostic: a handful of sug_knn launches for profiling (rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
x = torch.randn(B, 1024, C, device='cuda')
for _ in range(3):
    ops.knn(x, 20)
torch.cuda.synchronize()

"""Diagnostic: do the FPS / ball-query index sets of PointNet++'s sa1 / sa2 on the gradient-error test's clouds agree between
the HIP kernels, the fp32 oracle and the fp64 oracle?  (A differing group member is a discrete decision: it moves the gradients
downstream by far more than rounding.)  usage: python tests/diagnostics/diag_pn2_groups.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import ref_cpu as O
from sug_amd import ops
g0 = torch.Generator().manual_seed(3)
x = O.synth_clouds(4, 2048, g0)                       # [B,3,N,1]
xyz = x.squeeze(-1).permute(0, 2, 1).contiguous()      # [B,N,3]
B = xyz.shape[0]
res = {}
for name, dt in (('fp32', torch.float32), ('fp64', torch.float64)):
    torch.manual_seed(6)
    X = xyz.to(dt)
    s1 = torch.randint(0, 2048, (B,), dtype=torch.long)
    f1 = O.fps_cl(X, 512, s1)
    c1 = O.gather_cl(X, f1)
    b1 = O.ball_query_cl(0.2, 32, X, c1)
    s2 = torch.randint(0, 512, (B,), dtype=torch.long)
    f2 = O.fps_cl(c1, 128, s2)
    c2 = O.gather_cl(c1, f2)
    b2 = O.ball_query_cl(0.4, 64, c1, c2)
    res[name] = (f1, b1, f2, b2)
torch.manual_seed(6)
X = xyz.cuda()
s1 = torch.randint(0, 2048, (B,), dtype=torch.long)
f1 = ops.fps(X, 512, s1.cuda())
c1 = ops.gather_rows(X, f1)
b1 = ops.ball_query(X, c1, 0.2, 32)
s2 = torch.randint(0, 512, (B,), dtype=torch.long)
f2 = ops.fps(c1, 128, s2.cuda())
c2 = ops.gather_rows(c1, f2)
b2 = ops.ball_query(c1, c2, 0.4, 64)
res['hip'] = tuple(t.cpu().long() for t in (f1, b1, f2, b2))
for a, b in (('hip', 'fp32'), ('hip', 'fp64'), ('fp32', 'fp64')):
    d = [int((u != v).sum()) for u, v in zip(res[a], res[b])]
    print('%s vs %s: differing entries  fps1 %d / %d   ball1 %d / %d   fps2 %d / %d   ball2 %d / %d' % (
        a, b, d[0], res[a][0].numel(), d[1], res[a][1].numel(), d[2], res[a][2].numel(), d[3], res[a][3].numel()))

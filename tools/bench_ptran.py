"""Microbenchmark of the sug_ptran_* kernels at the block-1 shape of BASELINE config 5's per-GPU share
(32 clouds x 2048 points x k=16, 512 channels): time per launch and achieved rate of the compulsory bytes.
usage: python tools/bench_ptran.py [fp16|fp32] [B] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops
from sug_amd._lib import lib, check

lo = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == 'fp16') else torch.float32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
k, d = 16, 512
code = 1 if lo == torch.float16 else 0
es = 2 if code else 4
dev = torch.device('cuda')
torch.manual_seed(0)
xyz = torch.rand(B, n, 3, device=dev)
nbr = ops.knn_query_direct(k, xyz, xyz) if hasattr(ops, 'knn_query_direct') else None
if nbr is None:
    dist = torch.cdist(xyz, xyz)
    nbr = dist.topk(k, largest=False).indices.int()
nbr = nbr.int().contiguous()
R = B * n * k
P = B * n
_p = lambda t: t.data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream
L_ = lib()
w1 = torch.randn(d, 3, device=dev)
b1 = torch.randn(d, device=dev)
q, kf, vf = (torch.randn(B, n, d, device=dev) for _ in range(3))
T0 = torch.empty(R, d, dtype=lo, device=dev)
delta = (torch.randn(R, d, device=dev) * 0.5).to(lo)
U = torch.empty(R, d, dtype=lo, device=dev)
Lg = (torch.randn(R, d, device=dev) * 2).to(lo)
mixed = torch.empty(B, n, d, device=dev)
mx, sm = torch.empty_like(mixed), torch.empty_like(mixed)
g = torch.randn(B, n, d, device=dev)
off, ent = ops.knn_reverse(nbr)
dL, da = torch.empty_like(Lg), torch.empty_like(Lg)
dv, dq, dk = (torch.empty(B, n, d, device=dev) for _ in range(3))
cws = torch.empty(L_.sug_ptran_colsum_workspace(R), device=dev)
dbv = torch.empty(d, device=dev)
dU = (torch.randn(R, d, device=dev) * 0.1).to(lo)
dw1 = torch.empty(d, 3, device=dev)
db1 = torch.empty(d, device=dev)
ws = torch.empty(1024 * 4 * d, device=dev)
scale = 1.0 / d ** 0.5
G = 1e9
big, pt = R * d * es, P * d * 4

cases = [
    ('pos1_fwd', lambda: L_.sug_ptran_pos1_fwd(_p(xyz), _p(nbr), _p(w1), _p(b1), B, n, k, d, code, _p(T0), st()), big),
    ('qk_fwd', lambda: L_.sug_ptran_qk_fwd(_p(q), _p(kf), _p(delta), _p(nbr), B, n, k, d, code, _p(U), st()), 2 * big + 2 * pt),
    ('attn_fwd', lambda: L_.sug_ptran_attn_fwd(_p(Lg), _p(delta), _p(vf), _p(nbr), B, n, k, d, code, scale, _p(mixed), _p(mx),
                                               _p(sm), st()), 2 * big + 4 * pt),
    ('attn_bwd', lambda: L_.sug_ptran_attn_bwd(_p(g), _p(mixed), _p(Lg), _p(delta), _p(vf), _p(nbr), _p(mx), _p(sm), _p(off), _p(ent),
                                                      B, n, k, d, code, scale, _p(dL), _p(da), _p(dv), _p(dbv), _p(cws), st()),
     5 * big + 5 * pt),      # attn_bwd proper 4 big + rev_sum 1 big
    ('relu_bwd_db', lambda: L_.sug_ptran_relu_bwd_db(_p(dL), _p(Lg), R, d, code, _p(dbv), _p(cws), st()), 3 * big),
    ('qk_bwd', lambda: L_.sug_ptran_qk_bwd(_p(dU), _p(da), _p(off), _p(ent), B, n, k, d, code, _p(dq), _p(dk), _p(dbv),
                                                  _p(cws), st()), 4 * big + 2 * pt),   # rev_sum 1 big + qk_bwd 3 big
    ('pos1_bwd', lambda: L_.sug_ptran_pos1_bwd(_p(dU), _p(xyz), _p(nbr), _p(w1), _p(b1), B, n, k, d, code, _p(dw1), _p(db1),
                                               _p(ws), st()), big),
]
print('%s  B=%d n=%d k=%d: one k-expanded tensor = %.2f GB' % (lo, B, n, k, big / G))
for name, fn, nbytes in cases:
    for _ in range(2):
        check(fn(), name)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        check(fn(), name)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e3
    print('%-12s %8.1f us  %6.2f GB compulsory  %5.2f TB/s' % (name, t, nbytes / G, nbytes / t / 1e6))

"""[P|Q] of the fused EdgeConv layer (sug_edgeconv_fused_layer_fwd's side output) against the library GEMM x . Wcat^T of
the other path, element by element.  The fused kernel's MFMA chain runs through the input features in ascending order
(a plain dot product's fma chain); the library kernels order theirs differently for C >= 64, so this prints how many
elements differ and by how much (rounding: ~1e-6).
usage: python tools/check_fused_pq_bits.py [tuned]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from sug_amd import ops
from sug_amd._lib import lib

if len(sys.argv) > 1 and sys.argv[1] == 'tuned':
    from sug_amd.tuning import enable_tuned_gemms
    enable_tuned_gemms()
L = lib()
p = lambda t: 0 if t is None else t.data_ptr()
torch.manual_seed(0)
B, N, k = 64, 1024, 20
for C, Co, bias in ((3, 64, False), (64, 64, False), (64, 128, False), (64, 64, True), (128, 256, False)):
    x = torch.randn(B, N, C, device='cuda') * 0.5 + torch.randn(B, 1, C, device='cuda')
    w = torch.randn(2 * Co, C, device='cuda') / C ** 0.5
    b = torch.randn(Co, device='cuda') if bias else None       # the conv bias: enters Q only
    idx = torch.randint(0, N, (B, N, k), device='cuda', dtype=torch.int32)
    gamma, beta = torch.rand(Co, device='cuda') + 0.5, torch.randn(Co, device='cuda')
    rm, rv = torch.zeros(Co, device='cuda'), torch.ones(Co, device='cuda')
    z = torch.empty(B, N, Co, device='cuda'); arg = torch.empty(B, N, Co, dtype=torch.uint8, device='cuda')
    s1 = torch.empty(B, N, Co, device='cuda'); pq = torch.empty(B, N, 2 * Co, device='cuda')
    ws = torch.empty(ops.STATS_BLOCKS * 2 * Co, device='cuda'); coef = torch.empty(2, 5, Co, device='cuda')
    out = torch.empty(B, N, Co, device='cuda')
    rc = L.sug_edgeconv_fused_layer_fwd(p(x), C, C, p(w), p(b), p(idx), p(gamma), p(beta), B, N, k, Co, 2, 1, 1e-5, 0.1, 0.2,
                                        p(rm), p(rv), p(z), p(arg), p(s1), p(pq), 2 * Co, p(coef), p(out), Co, p(ws), ops._st())
    assert rc == 0, rc
    # (the library adds a bias in its epilogue, the fused kernel after its chain too)
    ref = F.linear(x.view(B * N, C), w, None if b is None else torch.cat([torch.zeros_like(b), b])).view(B, N, 2 * Co)
    d = (pq != ref)
    print('C=%3d Co=%3d bias=%d: %d of %d elements differ, max |diff| %.3e' % (C, Co, bias, int(d.sum()), d.numel(),
                                                                                  float((pq - ref).abs().max())))

"""Forward of one Point Transformer attention block in the fp16 mode at config 5's block-1 shape (32 clouds x 2048 points x 16
neighbours = 1 M k-expanded rows): the one-kernel form (sug_ptran_fused_fwd, with and without the tensors a backward needs)
against the composed chain (pos1 + 3 library GEMMs + qk + attn).  usage: python tools/bench_ptran_fused.py [CLOUDS] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops
from sug_amd.model import Ptran_transformer as PT
from sug_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
torch.manual_seed(0)
blk = PT.TransformerBlock(32, 512, 16).cuda()
xyz = torch.rand(B, n, 3, device='cuda')
nbr = ops.knn_query(xyz, xyz, 16, direct=True)
q, kf, vf = (torch.randn(B, n, 512, device='cuda') for _ in range(3))
R = B * n * 16
flop = 3 * 2.0 * R * 512 * 512


def run(fused, grad):
    ops.PTRAN_FUSED = bool(fused)
    qq = q.clone().requires_grad_(grad)
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx:
        for _ in range(3):
            out = ops.ptran_attention(xyz, nbr, qq, kf, vf, blk.fc_delta, blk.fc_gamma, torch.float16)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            out = ops.ptran_attention(xyz, nbr, qq, kf, vf, blk.fc_delta, blk.fc_gamma, torch.float16)
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / 10, out


ref = None
for fused, grad in ((False, True), (True, True), (True, False), (False, False)):
    ms, out = run(fused, grad)
    if ref is None:
        ref = out.detach()
    print('%-9s %-22s %8.3f ms   %6.1f TFLOP/s on the three linears   max |diff| vs composed %.2e'
          % ('one-kernel' if fused else 'composed', 'saving for a backward' if grad else 'no backward', ms, flop / ms * 1e-9,
             float((out.detach() - ref).abs().max())), flush=True)

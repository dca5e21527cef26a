"""Config 5 (Point Transformer, 16 clouds per domain, N = 2048, fp16 mode) with the one-kernel fp16 forward of the vector attention
(sug_ptran_fused_fwd, ops.PTRAN_FUSED) against the composed chain (pos1 + 3 library GEMMs + qk + attn): ms per step of the
captured two-pass step (median of 3 windows of 10 replays), each form in its own trainer.  usage: python tools/ab_ptran_fused.py [B] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth, BENCH_METHODS
from sug_amd import ops
from sug_amd.model import Ptran_transformer as PT
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
from sug_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
PT.GEMM_DTYPE, PT.PROJ_16BIT = torch.float16, True
dev = torch.device('cuda')
batch = synth(B, N, 666, dev)
for fused in (True, False, True, False):
    ops.PTRAN_FUSED = bool(fused)
    torch.manual_seed(666)
    tr = SUGStep(Net_MDA('PTran').to(dev).train(), lr=1e-3, weight_decay=5e-5, use_graph=True, methods=BENCH_METHODS)
    for _ in range(4):
        losses = tr.step(*batch)
    torch.cuda.synchronize()
    w = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            losses = tr.step(*batch)
        torch.cuda.synchronize()
        w.append(1e3 * (time.perf_counter() - t0) / 10)
    print('fused forward %-5s  %.3f ms per step  (windows %s)  losses %s' % (fused, sorted(w)[1], ['%.3f' % x for x in w],
                                                                           ['%.4f' % float(l) for l in losses]), flush=True)
    tr.drop_graphs()
    del tr
    torch.cuda.empty_cache()

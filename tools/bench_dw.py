"""Time sug_linear_dw_bias alone for the weight-gradient shapes of the C2 step (rows, M, N); the tuned-library
figures to compare with are in sug_amd/tuning/dw_choice_gfx950.json.  usage: python tools/bench_dw.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sug_amd import ops
from sug_amd._lib import lib

def t(fn, n=20):
    """GPU time per call: n calls captured in a hipGraph (two small launches per call are host-bound when enqueued live)."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    graph.replay()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): graph.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * n) * 1e3

L = lib()
if os.environ.get('TUNED', '1') == '1':
    from sug_amd.tuning import enable_tuned_gemms
    enable_tuned_gemms()
for R, M, N in ((65536, 3, 64), (65536, 64, 64), (65536, 64, 128), (65536, 128, 3), (65536, 128, 64), (65536, 256, 64),
                (65536, 128, 128), (65536, 512, 128), (65536, 512, 512), (2097152, 64, 64), (1048576, 128, 128)):
    g = torch.randn(R, M, device='cuda'); x = torch.randn(R, N, device='cuda')
    dw = torch.empty(M, N, device='cuda'); db = torch.empty(M, device='cuda')
    ws = torch.empty(int(L.sug_linear_dw_workspace(R, M, N)), device='cuda')
    us = t(lambda: L.sug_linear_dw_bias(g.data_ptr(), M, x.data_ptr(), N, R, M, N, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ops._st()))
    lib_us = t(lambda: torch.mm(g.t(), x))
    bmm = {}
    if M * N >= 128 * 128:
        for S in (8, 16, 32, 64):       # batched library GEMM over row chunks = split-K without a memset, + ordered sum
            if R % S == 0:
                bmm[S] = t(lambda: torch.bmm(g.view(S, R // S, M).transpose(1, 2), x.view(S, R // S, N)).sum(0))
    ref = (g.double().t() @ x.double()).float()
    err = float((dw - ref).abs().max() / ref.abs().max())
    print('R=%8d M=%4d N=%4d  %7.1f us  %6.1f TFLOP/s  %5.2f TB/s  rel err %.1e   library %7.1f us  bmm+sum %s' % (R, M, N, us, 2.0 * R * M * N / us / 1e6, 4.0 * R * (M + N) / us / 1e6, err, lib_us, {k: round(v, 1) for k, v in bmm.items()}))

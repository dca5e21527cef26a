"""Weight-gradient GEMMs of the wide layers as ops.linear_rows_backward runs them under capture -- K = rows split into S
batches, torch.bmm(g^T chunks, x chunks) -- with the library's default heuristic (what the step uses: the call is made with
TunableOp off) against TunableOp-tuned solutions.  usage (GPU box): python tools/bench_dw_bmm.py"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tn
from sug_amd import ops


def timeit(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


R = 65536
for M, N in ((512, 512), (256, 64), (512, 128), (128, 128), (256, 128)):
    g = torch.randn(R, M, device='cuda')
    x = torch.randn(R, N, device='cuda')
    S = ops._dw_bmm_chunks(R, M, N, g, x)
    if not S:
        print('M %d N %d: not on the bmm path' % (M, N))
        continue
    f = lambda: torch.bmm(g.view(S, R // S, M).transpose(1, 2), x.view(S, R // S, N))
    tn.enable(False)
    t0 = timeit(f)
    tn.enable(True)
    tn.tuning_enable(True)
    tn.set_max_tuning_duration(60)
    tn.set_filename(os.path.join(tempfile.gettempdir(), 'dw_bmm_%d.csv' % os.getpid()))
    t1 = timeit(f)
    tn.enable(False)
    fl = 2.0 * R * M * N
    print('dW %3d x %3d over %d rows, S = %2d chunks: default %7.1f us (%5.1f TF) | tuned %7.1f us (%5.1f TF)' %
          (M, N, R, S, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6), flush=True)
print([r for r in tn.get_results() if 'Batched' in r[0]])

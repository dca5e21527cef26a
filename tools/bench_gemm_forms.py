"""Does the operand layout matter for the library's fp32 GEMM on the step's big shapes?  y = x . W^T computed as
F.linear(x, W) (W [out, in] row-major: the "TN" library call) against x @ Wt with Wt = W^T stored contiguously ("NN"), both
with TunableOp tuning switched on for this run (results go to a scratch file).  usage (GPU box): python tools/bench_gemm_forms.py"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.cuda.tunable as tn

tn.enable(True)
tn.tuning_enable(True)
tn.set_max_tuning_duration(int(os.environ.get('SUG_TUNE_MS', '30')))
tn.set_filename(os.path.join(tempfile.gettempdir(), 'gemm_forms_%d.csv' % os.getpid()))


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


for rows, K, Co in ((65536, 512, 512), (65536, 128, 512), (65536, 128, 64), (65536, 64, 128)):
    x = torch.randn(rows, K, device='cuda')
    W = torch.randn(Co, K, device='cuda') / K ** 0.5
    Wt = W.t().contiguous()
    g = torch.randn(rows, Co, device='cuda')
    gt = g.t().contiguous()
    fl = 2.0 * rows * K * Co
    t_tn = timeit(lambda: torch.nn.functional.linear(x, W))
    t_nn = timeit(lambda: x @ Wt)
    y = torch.empty(Co, rows, device='cuda')
    t_out_t = timeit(lambda: torch.mm(W, x.t(), out=y))            # y^T = W . x^T
    t_dx = timeit(lambda: g @ W)                                   # input gradient (NN)
    t_dx_t = timeit(lambda: torch.nn.functional.linear(g, Wt))     # the same as TN with W^T stored
    print('rows %6d K %3d Co %3d:  x.W^T (F.linear) %7.1f us %5.1f TF | x@Wt %7.1f us %5.1f TF | (W.x^T) %7.1f us || dx = g@W %7.1f us | F.linear(g, Wt) %7.1f us' %
          (rows, K, Co, t_tn, fl / t_tn / 1e6, t_nn, fl / t_nn / 1e6, t_out_t, t_dx, t_dx_t), flush=True)

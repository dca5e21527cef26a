"""Diagnostic: TunableOp-tuned library GEMM vs sug_linear_dw for the large weight gradients."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(PYTORCH_TUNABLEOP_ENABLED='1', PYTORCH_TUNABLEOP_TUNING='1', PYTORCH_TUNABLEOP_FILENAME='/tmp/dw_tune.csv')
import torch
from sug_amd import ops
def t(fn, n=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for R, M, N in ((65536, 512, 512), (65536, 512, 128), (65536, 256, 64), (65536, 128, 128)):
    g = torch.randn(R, M, device='cuda'); x = torch.randn(R, N, device='cuda')
    W = torch.randn(M, N, device='cuda', requires_grad=True)
    lib_us = t(lambda: g.t() @ x)
    def mine():
        y = ops.linear_rows(x.detach(), W)
        torch.autograd.grad(y, W, g)
    # time only the dW kernel path: call the C function directly
    from sug_amd._lib import lib, check
    dw = torch.empty(M, N, device='cuda'); ws = torch.empty(int(lib().sug_linear_dw_workspace(R, M, N)), device='cuda')
    st = ops._st()
    own_us = t(lambda: lib().sug_linear_dw(g.data_ptr(), M, x.data_ptr(), N, R, M, N, dw.data_ptr(), ws.data_ptr(), st))
    print('R=%d M=%d N=%d  tuned library %.1f us   sug_linear_dw %.1f us' % (R, M, N, lib_us, own_us))

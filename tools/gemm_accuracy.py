"""fp32 accuracy of torch (rocBLAS / hipBLASLt) GEMMs at the shapes the heads use (diagnostic)."""
import torch
torch.manual_seed(0)
print('allow_tf32', torch.backends.cuda.matmul.allow_tf32, 'precision', torch.get_float32_matmul_precision())
try:
    print('preferred blas', torch.backends.cuda.preferred_blas_library())
except Exception as e:
    print('blas query failed', e)
def rel(a, b):
    return float((a.double() - b).norm() / b.norm())
for M in (1, 2, 4, 8, 16, 32, 64, 256):
    for (K, N) in ((512, 1024), (1024, 512), (256, 512), (4096, 512), (512, 4096)):
        a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda')
        ref = a.double() @ w.double().t()
        r1 = rel(a @ w.t(), ref)
        r2 = rel(torch.nn.functional.linear(a, w), ref)
        g = torch.randn(M, N, device='cuda')
        r3 = rel(g @ w, g.double() @ w.double())          # grad_input
        r4 = rel(g.t() @ a, g.double().t() @ a.double())  # grad_weight
        flag = ' <<<' if max(r1, r2, r3, r4) > 1e-5 else ''
        print('M=%4d K=%5d N=%5d  mm %.1e linear %.1e dX %.1e dW %.1e%s' % (M, K, N, r1, r2, r3, r4, flag))

"""A/B of two builds of pointmlp.hip (the product library against tools/ab/libsug_amd_il.so = -DSUG_POINTMLP_INTERLEAVE, the
round 2-3 form with the epilogue issued between the MFMAs): sug_pointmlp_max_fwd alone, interleaved rounds, median.
Build the variant first (repo root):
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -DSUG_POINTMLP_INTERLEAVE \
        -c sug_amd/csrc/pointmlp.hip -o /tmp/pm_il.o
  and link it with the other objects of sug_amd/csrc into tools/ab/libsug_amd_il.so
usage (GPU box): python tools/ab_pointmlp.py [variant.so]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

variant = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tools', 'ab', 'libsug_amd_il.so')
libs = [('product', ctypes.CDLL(os.path.join(ROOT, 'sug_amd', 'libsug_amd.so'))),
        ('variant', ctypes.CDLL(variant))]
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
for _, L in libs:
    L.sug_pointmlp_max_fwd.argtypes = [vp, i64, i64, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]


def run(L, x, W, b, gamma, seg, zext, arg, ws, iters):
    rows, K = x.shape
    Co = W.shape[0]
    nblk = ctypes.c_int(0)
    st = vp(torch.cuda.current_stream().cuda_stream)
    call = lambda: L.sug_pointmlp_max_fwd(x.data_ptr(), K, rows, K, W.data_ptr(), b.data_ptr(), gamma.data_ptr(), Co, seg,
                                           zext.data_ptr(), arg.data_ptr(), ws.data_ptr(), ctypes.byref(nblk), st)
    for _ in range(2):
        assert call() == 0
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        call()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / iters * 1e3


for name, S, seg, K, Co in (('pointnet conv5, 64 clouds', 64, 1024, 128, 1024), ('pointnet conv5, 16 clouds', 16, 1024, 128, 1024),
                            ('sa1 last layer, one domain group', 64 * 512, 32, 64, 128),
                            ('sa2 last layer, one domain group', 64 * 128, 64, 128, 256)):
    torch.manual_seed(0)
    x = torch.randn(S * seg, K, device='cuda')
    W = torch.randn(Co, K, device='cuda') / K ** 0.5
    b = torch.randn(Co, device='cuda') * 0.1
    gamma = torch.linspace(-1, 1, Co, device='cuda')
    zext = torch.empty(S, Co, device='cuda')
    arg = torch.empty(S, Co, dtype=torch.int32, device='cuda')
    ws = torch.empty(1024 * 2 * Co * 4, device='cuda')
    res = {n: [] for n, _ in libs}
    outs = {}
    for _ in range(7):
        for n, L in libs:
            res[n].append(run(L, x, W, b, gamma, seg, zext, arg, ws, 20))
            outs[n] = (zext.clone(), arg.clone())
    same = torch.equal(outs['product'][0], outs['variant'][0]) and torch.equal(outs['product'][1], outs['variant'][1])
    fl = 2.0 * S * seg * K * Co
    print('%-34s rows %8d K %3d Co %4d: ' % (name, S * seg, K, Co) +
          '  '.join('%s %7.1f us (%.1f TFLOP/s)' % (n, sorted(v)[3], fl / sorted(v)[3] / 1e6) for n, v in res.items()) +
          '  results %s' % ('identical' if same else 'DIFFER'))

"""Where does sug_ptran_fused_fwd spend its time?  Timing-only builds of csrc/ptran_fused.hip with phases compiled out
(-DPF_SKIP bitmask: 1 pos1, 2 qk pass, 4 attention pass, 8 the three GEMMs; results of such builds are wrong).  The switches are
not in the product source: tools/ubench/ptran_fused_phases.patch adds them to a scratch copy, which this script compiles on the
GPU box.  usage: python tools/bench_ptran_phases.py [CLOUDS] [N]"""
import ctypes, glob, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sug_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
torch.manual_seed(0)
xyz = torch.rand(B, n, 3, device='cuda')
nbr = ops.knn_query(xyz, xyz, 16, direct=True).contiguous()
q, kf, vf = (torch.randn(B, n, 512, device='cuda') for _ in range(3))
w1, b1 = torch.randn(512, 3, device='cuda'), torch.randn(512, device='cuda')
W = [torch.randn(512, 512, device='cuda').half() * 0.05 for _ in range(3)]
bs = [torch.randn(512, device='cuda').half() * 0.1 for _ in range(3)]
R = B * n * 16
T = [torch.empty(R, 512, dtype=torch.float16, device='cuda') for _ in range(5)]
mixed, mx, sm = (torch.empty(B, n, 512, device='cuda') for _ in range(3))
vp = ctypes.c_void_p
p = lambda t: vp(t.data_ptr())
names = {0: 'product', 1: 'no pos1', 2: 'no qk pass', 4: 'no attention pass', 6: 'no qk, no attention pass', 7: 'GEMMs + LDS epilogues only',
         8: 'no GEMMs (row passes only)', 15: 'empty loop'}
top = os.path.join(tempfile.gettempdir(), 'sug_ptran_phase_src')
shutil.rmtree(top, ignore_errors=True)
os.makedirs(os.path.join(top, 'sug_amd'))
shutil.copytree(os.path.join(ROOT, 'sug_amd', 'csrc'), os.path.join(top, 'sug_amd', 'csrc'), ignore=shutil.ignore_patterns('*.o', '*.so'))
shutil.copytree(os.path.join(ROOT, 'include'), os.path.join(top, 'include'))        # (common.h includes ../../include/sug_amd.h)
subprocess.run(['patch', '-p1', '-s', '-i', os.path.join(ROOT, 'tools', 'ubench', 'ptran_fused_phases.patch')], cwd=top, check=True)
src = os.path.join(top, 'sug_amd', 'csrc')
for mask in (0, 1, 2, 4, 6, 7, 8, 15):
    f = os.path.join(top, 'pf_%d.so' % mask)
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-shared',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + src, '-DPF_SKIP=%d' % mask, os.path.join(src, 'ptran_fused.hip'),
                    os.path.join(src, 'capi.cpp'), '-o', f], check=True)
    L = ctypes.CDLL(f)
    fn = L.sug_ptran_fused_fwd
    fn.argtypes = [vp] * 13 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int] + [vp] * 9
    for save in (1, 0):
        def call():
            return fn(p(xyz), p(nbr), p(q), p(kf), p(vf), p(w1), p(b1), p(W[0]), p(bs[0]), p(W[1]), p(bs[1]), p(W[2]), p(bs[2]),
                      B, n, 16, 512, 512 ** -0.5, save, p(T[0]), p(T[1]), p(T[2]), p(T[3]), p(T[4]), p(mixed), p(mx), p(sm),
                      vp(torch.cuda.current_stream().cuda_stream))
        for _ in range(2):
            assert call() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            call()
        b.record()
        torch.cuda.synchronize()
        print('%-32s save=%d  %7.3f ms' % (names.get(mask, mask), save, a.elapsed_time(b) / 5), flush=True)

"""Diagnostic (round 1): time ablation builds of the single-wave kNN kernels (tools/ubench/knn_mfma_legacy.hip, no longer part
of the product library) on the GPU box.
Usage: python tools/bench_knn.py   (needs hipcc + a GPU)"""
import ctypes, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

SRC = os.path.join(ROOT, 'sug_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-shared']


def build(tag, defs):
    out = os.path.join(tempfile.gettempdir(), 'libknn_%s.so' % tag)
    files = [os.path.join(ROOT, 'tools', 'ubench', 'knn_mfma_legacy.hip'), os.path.join(SRC, 'capi.cpp')]
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-I' + SRC, '-I' + os.path.join(ROOT, 'include')] + defs + files + ['-o', out], check=True)
    return ctypes.CDLL(out)


def time_knn(L, x, k, iters=20):
    B, N, C = x.shape
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.sug_knn_legacy.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                          ctypes.c_void_p, ctypes.c_void_p]
    for _ in range(3):
        L.sug_knn_legacy(x.data_ptr(), C, B, N, C, k, idx.data_ptr(), st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        L.sug_knn_legacy(x.data_ptr(), C, B, N, C, k, idx.data_ptr(), st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


_prev = [0] * 16


def stamps(L):
    buf = (ctypes.c_ulonglong * 16)()
    L.sug_debug_read_stamps(buf)
    v = list(buf)
    names = ['query staging', 'sweep (mfma+med3+ring)', 'final compact', 'build keys', 'rank+store']
    global _prev
    d = [a - b for a, b in zip(v, _prev)]
    _prev = v
    L6 = 6.0   # accumulators add up over the 3 warm-up + 3 timed launches
    acc = 'per launch: step %d, tile_store %d, barrier_or %d, refresh %d (x%.1f), compact %d (x%.1f)' % (d[8] / L6, d[9] / L6, d[10] / L6, d[11] / L6, d[13] / L6, d[12] / L6, d[14] / L6)
    return ', '.join('%s %d' % (n, v[i + 1] - v[i]) for i, n in enumerate(names)) + ' | ' + acc


if __name__ == '__main__':
    torch.manual_seed(0)
    variants = [('default', []), ('two-pass', ['-DSUG_KNN_TWO_PASS=2']), ('single-pass', ['-DSUG_KNN_TWO_PASS=0'])]
    libs = [(t, build(t, d)) for t, d in variants]
    for B in (32, 64):
        for C in (3, 64, 128):
            x = torch.randn(B, 1024, C, device='cuda')
            print('B=%d C=%3d ' % (B, C) + '  '.join('%s %7.1f us' % (t, time_knn(L, x, 20)) for t, L in libs))
    S = build('stamp', ['-DSUG_KNN_STAMP=1'])
    print('two-pass kernel: resident workgroups per CU  C=64: %d  C=3: %d' % (S.sug_debug_knn2p_occupancy(64), S.sug_debug_knn2p_occupancy(3)))
    for C in (3, 64, 128):
        x = torch.randn(32, 1024, C, device='cuda')
        us = time_knn(S, x, 20, iters=3)
        print('C=%3d stamped build %.1f us; cycles (s_memtime @100MHz?) of block 0: %s' % (C, us, stamps(S)))

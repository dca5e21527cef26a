"""Diagnostic: where does the fused EdgeConv layer's time go?  Builds ablation variants of edgeconv_fused.hip on the GPU box
(timing only: their results are wrong) and times them next to the product build at the four DGCNN layer shapes
(64 clouds x 1024 points, k = 20).  usage: python tools/bench_edgeconv_fused.py"""
import ctypes, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

SRC = os.path.join(ROOT, 'sug_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-shared', '-I' + os.path.join(ROOT, 'include')]
vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


def lib_path(tag):
    return os.path.join(tempfile.gettempdir(), 'libef_%s.so' % tag)


def compile_variant(tag, defs):
    """hipcc in a child process: NEVER call this from a process that runs under rocprofv3 (the profiler's preload
    initialises the GPU in every child, and hipcc execs clang / lld: the forbidden exec from a GPU-initialised process).
    The --pmc flow is two commands: `--build-only "<defs>"` un-profiled, then EF_ONLY under the profiler (load only)."""
    out = lib_path(tag)
    files = [os.path.join(SRC, f) for f in ('edgeconv_fused.hip', 'edgeconv.hip', 'capi.cpp')]
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + defs + files + ['-o', out], check=True)
    return out


def build(tag, defs, compile=True):
    out = compile_variant(tag, defs) if compile else lib_path(tag)
    if not os.path.exists(out):
        raise RuntimeError('%s is missing: run `python tools/bench_edgeconv_fused.py --build-only "%s"` first '
                           '(outside rocprofv3)' % (out, ' '.join(defs)))
    L = ctypes.CDLL(out)
    L.sug_edgeconv_fused_layer_fwd.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, f32, f32,
                                               vp, vp, vp, vp, vp, vp, i64, vp, vp, i64, vp, vp]
    return L


def time_layer(L, C, Co, train, iters=20, B=64, N=1024, k=20):
    dev = 'cuda'
    x = torch.randn(B, N, C, device=dev) * 0.5
    idx = torch.randint(0, N, (B, N, k), device=dev, dtype=torch.int32)
    idx[:, :, 0] = torch.arange(N, device=dev)
    w = torch.randn(2 * Co, C, device=dev) * 0.1
    gamma, beta = torch.randn(Co, device=dev), torch.randn(Co, device=dev)
    rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    z, out = torch.empty(B, N, Co, device=dev), torch.empty(B, N, Co, device=dev)
    arg = torch.empty(B, N, Co, device=dev, dtype=torch.uint8)
    s1 = torch.empty(B, N, Co, device=dev) if train else None
    pq = torch.empty(B, N, 2 * Co, device=dev) if train else None
    coef = torch.empty(2, 5, Co, device=dev)
    ws = torch.empty(1024 * 2 * Co, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None

    fresh = os.environ.get('EF_FRESH') == '1'

    def call():
        nonlocal z, arg, out
        if fresh:           # new output buffers per launch, as the autograd op allocates them
            z, out = torch.empty(B, N, Co, device=dev), torch.empty(B, N, Co, device=dev)
            arg = torch.empty(B, N, Co, device=dev, dtype=torch.uint8)
        rc = L.sug_edgeconv_fused_layer_fwd(p(x), C, C, p(w), None, p(idx), p(gamma), p(beta), B, N, k, Co, 2, 1, 1e-5, 0.1, 0.01,
                                            p(rm), p(rv), p(z), p(arg), p(s1), p(pq), 2 * Co, p(coef), p(out), Co, p(ws), st)
        assert rc == 0, rc
    for _ in range(3):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        call()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


if __name__ == '__main__':
    torch.manual_seed(0)
    variants = [('product', []), ('main kernel', ['-DSUG_EF_ABL_NOACT']),
                ('no gather', ['-DSUG_EF_ABL_NOACT', '-DSUG_EF_ABL_NOGATHER']),
                ('no MFMA', ['-DSUG_EF_ABL_NOACT', '-DSUG_EF_ABL_NOMFMA']),
                ('neither', ['-DSUG_EF_ABL_NOACT', '-DSUG_EF_ABL_NOGATHER', '-DSUG_EF_ABL_NOMFMA']),
                ('neither, no x loads', ['-DSUG_EF_ABL_NOACT', '-DSUG_EF_ABL_NOGATHER', '-DSUG_EF_ABL_NOMFMA', '-DSUG_EF_ABL_NOLOADX']),
                ('no x loads', ['-DSUG_EF_ABL_NOACT', '-DSUG_EF_ABL_NOLOADX'])] + \
               [(t, d.split()) for t, d in (a.split('=', 1) for a in sys.argv[1:] if '=' in a)]
    if len(sys.argv) >= 2 and sys.argv[1] == '--build-only':     # un-profiled step of the --pmc flow: compile, touch no GPU
        print(compile_variant('only', (sys.argv[2] if len(sys.argv) > 2 else '').split()))
        sys.exit(0)
    only = os.environ.get('EF_ONLY')           # one variant, layer 2 shape, few launches: for a rocprofv3 --pmc pass
    if only is not None:
        L = build('only', only.split(), compile=False)           # load the prebuilt library only: no child process here
        print(time_layer(L, 64, 64, 0, iters=5))
        sys.exit(0)
    libs = [(t, build(t.replace(' ', '_').replace(',', ''), d)) for t, d in variants]
    for train in (0, 1):
        for C, Co in ((3, 64), (64, 64), (64, 128), (128, 256)):
            print('train=%d C=%3d Co=%3d  ' % (train, C, Co) + '  '.join('%s %6.1f' % (t, time_layer(L, C, Co, train)) for t, L in libs), flush=True)

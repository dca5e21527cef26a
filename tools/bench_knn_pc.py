"""Diagnostic: where does knn_pc_kernel's time go?  Builds ablation variants of knn_pc.hip on the GPU box (timing only:
their results are wrong) and times them next to the product build.  The ablation switches (-DSUG_KNN_ABL_*, SUG_KNN_SERIAL,
SUG_KNN_STAGE_MID) are NOT in the product source since round 6: tools/ubench/knn_pc_ablations.patch adds them to a scratch
copy of sug_amd/csrc, which is what this script compiles.  usage: python tools/bench_knn_pc.py"""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

def _ablation_sources():
    """A scratch copy of the repository's sources with the ablation patch applied to knn_pc.hip."""
    top = os.path.join(tempfile.gettempdir(), 'sug_knn_ablation_src')
    shutil.rmtree(top, ignore_errors=True)
    os.makedirs(os.path.join(top, 'sug_amd'))
    shutil.copytree(os.path.join(ROOT, 'sug_amd', 'csrc'), os.path.join(top, 'sug_amd', 'csrc'),
                    ignore=shutil.ignore_patterns('*.o', '*.so'))
    shutil.copytree(os.path.join(ROOT, 'include'), os.path.join(top, 'include'))        # (common.h includes ../../include/sug_amd.h)
    subprocess.run(['patch', '-p1', '-s', '-i', os.path.join(ROOT, 'tools', 'ubench', 'knn_pc_ablations.patch')], cwd=top, check=True)
    return os.path.join(top, 'sug_amd', 'csrc')


SRC = _ablation_sources()
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-shared', '-I' + os.path.join(ROOT, 'include')]


def build(tag, defs):
    out = os.path.join(tempfile.gettempdir(), 'libknnpc_%s.so' % tag)
    files = [os.path.join(SRC, f) for f in ('knn.hip', 'knn_pc.hip', 'capi.cpp')]
    subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + defs + files + ['-o', out], check=True)
    return ctypes.CDLL(out)


def time_knn(L, x, k, iters=20):
    B, N, C = x.shape
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.sug_knn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                          ctypes.c_void_p, ctypes.c_void_p]
    for _ in range(3):
        L.sug_knn(x.data_ptr(), C, B, N, C, k, idx.data_ptr(), st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        L.sug_knn(x.data_ptr(), C, B, N, C, k, idx.data_ptr(), st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


if __name__ == '__main__':
    torch.manual_seed(0)
    variants = [('product', []), ('staging inside the chain', ['-DSUG_KNN_STAGE_MID']), ('serial', ['-DSUG_KNN_SERIAL']), ('no insertion', ['-DSUG_KNN_ABL_NOINSERT']), ('no consumer', ['-DSUG_KNN_ABL_NOCONS']),
                ('no MFMA', ['-DSUG_KNN_ABL_NOMFMA']), ('no MFMA, no consumer', ['-DSUG_KNN_ABL_NOMFMA', '-DSUG_KNN_ABL_NOCONS']),
                ('no MFMA, no insertion', ['-DSUG_KNN_ABL_NOMFMA', '-DSUG_KNN_ABL_NOINSERT'])]
    if len(sys.argv) > 1 and sys.argv[1] == 'producer':       # the producer side alone, piece by piece
        NC = ['-DSUG_KNN_ABL_NOCONS']
        variants = [('no consumer', NC), ('+ no score writes', NC + ['-DSUG_KNN_ABL_NOSCOREWRITE']),
                    ('+ no staging', NC + ['-DSUG_KNN_ABL_NOSCOREWRITE', '-DSUG_KNN_ABL_NOSTAGE']),
                    ('+ no A reads', NC + ['-DSUG_KNN_ABL_NOSCOREWRITE', '-DSUG_KNN_ABL_NOSTAGE', '-DSUG_KNN_ABL_NOAREAD']),
                    ('+ no barrier', NC + ['-DSUG_KNN_ABL_NOSCOREWRITE', '-DSUG_KNN_ABL_NOSTAGE', '-DSUG_KNN_ABL_NOAREAD', '-DSUG_KNN_ABL_NOBARRIER']),
                    ('no barrier only', NC + ['-DSUG_KNN_ABL_NOBARRIER']), ('no A reads only', NC + ['-DSUG_KNN_ABL_NOAREAD']),
                    ('no staging only', NC + ['-DSUG_KNN_ABL_NOSTAGE'])]
    if len(sys.argv) > 1 and sys.argv[1] == 'ab':             # careful A/B of two builds: interleaved rounds, median and minimum
        variants = [('product', []), ('staging inside the chain', ['-DSUG_KNN_STAGE_MID'])]
    if len(sys.argv) > 1 and sys.argv[1] == 'symm':           # ceiling of a symmetric (block-pair) form: interleaved A/B
        variants = [('product', []), ('mirrored tiles skip their MFMA chain (ceiling)', ['-DSUG_KNN_ABL_SYMM']),
                    ('no MFMA at all', ['-DSUG_KNN_ABL_NOMFMA'])]
        sys.argv[1] = 'ab'
    if len(sys.argv) > 1 and sys.argv[1] == 'seed':
        # Upper bound of threshold seeding: the consumer's threshold starts at the ORACLE value (the exact
        # K-th best score of every query, computed here with torch), and the candidates that pass the scan are counted.
        # Inputs: Gaussian rows (as the other ablations) and the features of a random-init DGCNN forward (layers 2-4 of the
        # benchmark: spatially smooth, what the step really feeds the kernel).
        Lc = build('count', ['-DSUG_KNN_ABL_COUNT', '-DSUG_KNN_ABL_SEED'])
        Lp = build('product', [])
        Lc.sug_knn_abl_set_seed.argtypes = [ctypes.c_void_p]
        Lc.sug_knn_abl_accepted.restype = ctypes.c_longlong
        from sug_amd import ops
        from sug_amd.model.Model import Net_MDA
        from bench import synth
        net = Net_MDA('DGCNN').cuda().train()
        feats = []
        real_knn = ops.knn
        ops.knn = lambda f, k: (feats.append(f.detach().clone()), real_knn(f, k))[1]
        with torch.no_grad():
            net(synth(64, 1024, 666, 'cuda')[0], semantic_adaption=True)
        ops.knn = real_knn
        inputs = [('gaussian C=%d' % C, torch.randn(64, 1024, C, device='cuda')) for C in (3, 64, 128)] + \
                 [('DGCNN layer %d input, C=%d' % (i + 1, f.shape[-1]), f.contiguous()) for i, f in enumerate(feats[:4])]
        for tag, x in inputs:
            B, N, C = x.shape
            xd = x.double()
            d = (xd * xd).sum(-1, keepdim=True) - 2 * xd @ xd.transpose(1, 2) + (xd * xd).sum(-1).unsqueeze(1)
            dK = d.topk(20, largest=False)[0][:, :, 19]
            seed = (-(dK * (1 + 1e-5)) - 1e-5).float().contiguous()          # score threshold, slightly loose
            row = []
            for name, sd in (('running threshold (product)', None), ('oracle seed', seed)):
                Lc.sug_knn_abl_set_seed(None if sd is None else ctypes.c_void_p(sd.data_ptr()))
                Lc.sug_knn_abl_accepted(1)
                t = time_knn(Lc, x, 20, 20)
                torch.cuda.synchronize()
                acc = Lc.sug_knn_abl_accepted(1) / (23 * B * N)
                row.append('%s: %.1f us, %.1f accepted candidates per query' % (name, t, acc))
            Lc.sug_knn_abl_set_seed(None)
            print('%-34s product build %.1f us | counting build: %s' % (tag, time_knn(Lp, x, 20, 20), ' | '.join(row)), flush=True)
        sys.exit(0)
    libs = [(t, build(t.replace(' ', '_').replace(',', '').replace('+', 'p'), d)) for t, d in variants]
    for C in (3, 64, 128):
        x = torch.randn(64, 1024, C, device='cuda')
        if len(sys.argv) > 1 and sys.argv[1] == 'ab':
            res = {t: [] for t, _ in libs}
            for _ in range(9):
                for t, L in libs:
                    res[t].append(time_knn(L, x, 20, 50))
            print('C=%3d  ' % C + '  '.join('%s median %6.1f min %6.1f' % (t, sorted(v)[len(v) // 2], min(v)) for t, v in res.items()))
        else:
            print('C=%3d  ' % C + '  '.join('%s %6.1f' % (t, time_knn(L, x, 20)) for t, L in libs))

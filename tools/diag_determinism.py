"""Diagnostic: is the training step reproducible bit for bit?  Runs the same K steps twice in eager mode and once under
hipGraph replay (same weights, same batch, same generator seed), and reports per step whether the six losses and a hash of
all parameters are identical.  With --grads the first step's gradients are compared tensor by tensor (names the modules whose
backward is order-dependent).   usage: python tools/diag_determinism.py [--model DGCNN] [--batch 4] [--steps 4] [--grads]"""
import argparse
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from oracle import ref_cpu as O


def make(model, B, N, wseed=5, seed=11):
    from sug_amd.model.Model import Net_MDA
    net = Net_MDA(model)
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, wseed))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    g = torch.Generator().manual_seed(seed)
    data, data_t = O.synth_clouds(B, N, g), O.synth_clouds(B, N, g)
    lab, lab_t = torch.randint(0, 10, (B,), generator=g), torch.randint(0, 10, (B,), generator=g)
    return net.cuda().train(), [t.cuda() for t in (data, lab, data_t, lab_t)]


def phash(net):
    h = hashlib.sha256()
    for k, v in sorted(net.state_dict().items()):
        h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:12]


def run(args, use_graph):
    from sug_amd.train_step import SUGStep
    net, batch = make(args.model, args.batch, args.npoints)
    tr = SUGStep(net, lr=1e-3, weight_decay=5e-5, use_graph=use_graph, methods=bench.BENCH_METHODS)
    if args.fp16:
        from sug_amd.model import Ptran_transformer as PT
        PT.GEMM_DTYPE = torch.float16
    torch.manual_seed(3)
    out = []
    for _ in range(args.steps):
        l = [float(v) for v in tr.step(*batch)]
        out.append((l, phash(net)))
        if args.which:
            STATES.setdefault(len(out) - 1, []).append({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    return out


STATES = {}


def report_which():
    for step, runs in sorted(STATES.items()):
        names = ('eager', 'eager2', 'graph')
        for i in range(1, len(runs)):
            bad = [(k, float((runs[0][k].double() - runs[i][k].double()).abs().max())) for k in runs[0] if not torch.equal(runs[0][k], runs[i][k])]
            if bad:
                print('step %d: %s vs %s: %d of %d tensors differ' % (step, names[0], names[i], len(bad), len(runs[0])))
                for k, d in bad[:80]:
                    print('     %-56s %.3e' % (k, d))
                return


def grads(args):
    from sug_amd.train_step import SUGStep
    res, vals = [], []
    for _ in range(2):
        net, batch = make(args.model, args.batch, args.npoints)
        tr = SUGStep(net, lr=0.0, weight_decay=0.0, use_graph=False, methods=bench.BENCH_METHODS)
        torch.manual_seed(3)
        parts = [v for v in tr.losses(*batch, combine=False) if v is not None]
        vals.append([float(v) for v in parts])
        sum(parts).backward()
        if hasattr(net.g, 'clear_prefix_cache'):
            net.g.clear_prefix_cache()
        res.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
    print('forward of two eager runs:', 'same' if vals[0] == vals[1] else 'DIFFER %s %s' % (vals[0], vals[1]))
    bad = [(k, float((res[0][k] - res[1][k]).abs().max()), float(res[0][k].abs().max())) for k in res[0] if not torch.equal(res[0][k], res[1][k])]
    print('gradients: %d tensors, %d differ between two eager runs' % (len(res[0]), len(bad)))
    for k, d, m in bad:
        print('   %-50s max |diff| %.3e  (max |g| %.3e)' % (k, d, m))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='DGCNN')
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--npoints', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--grads', action='store_true')
    ap.add_argument('--fp16', action='store_true')
    ap.add_argument('--which', action='store_true', help='name the tensors of the first differing step')
    ap.add_argument('--tuned', action='store_true', help='the recorded TunableOp choices on, as bench.py runs')
    args = ap.parse_args()
    if args.tuned:
        from sug_amd.tuning import enable_tuned_gemms
        enable_tuned_gemms()
    if args.grads:
        grads(args)
    a, b, c = run(args, False), run(args, False), run(args, True)
    if args.which:
        report_which()
    for i in range(args.steps):
        same_l = a[i][0] == b[i][0]
        print('step %d  eager/eager: losses %s params %s | eager/graph: losses %s params %s' % (
            i, 'same' if same_l else 'DIFFER', 'same' if a[i][1] == b[i][1] else 'DIFFER',
            'same' if a[i][0] == c[i][0] else 'DIFFER', 'same' if a[i][1] == c[i][1] else 'DIFFER'))
        if not same_l:
            print('     ', a[i][0], '\n     ', b[i][0])
        if a[i][0] != c[i][0]:
            print('   e ', a[i][0], '\n   g ', c[i][0])

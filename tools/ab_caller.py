"""Interleaved in-process timing of the UNCHANGED-CALLER form of a step (four separate model(...) calls, nothing shared by
the caller: what train_dg_single_gpu.py:260-310 does after the two-import swap), eager launches, with ops-level switches
flipped.  usage: python tools/ab_caller.py [ATTR ...]   (module attributes of sug_amd.ops set to False for variant B)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth, BENCH_METHODS
from sug_amd import ops
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
from sug_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device('cuda')
attrs = sys.argv[1:] or ['CALAYER_FUSED']
data = synth(32, 1024, 666, dev)
torch.manual_seed(666)
tr = SUGStep(Net_MDA('DGCNN').to(dev).train(), lr=1e-3, weight_decay=5e-5, use_graph=False, methods=BENCH_METHODS,
             pair_domains=False, share_prefix=False)
tr.model.g.share_prefix = 'auto'
variants = [('shipped', None)] + [('no_' + a, a) for a in attrs]
res = {k: [] for k, _ in variants}
for rnd in range(6):
    for name, a in variants:
        keep = getattr(ops, a) if a else None
        if a:
            setattr(ops, a, False)
        for _ in range(2 if rnd else 4):
            tr.step(*data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.step(*data)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 10 * 1e3)
        if a:
            setattr(ops, a, keep)
for k, v in res.items():
    print('%-22s caller-form ms/step: min %.3f median %.3f  all %s' % (k, min(v), sorted(v)[len(v) // 2], ['%.2f' % x for x in v]))

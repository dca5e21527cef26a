"""Interleaved in-process A/B of the captured DGCNN step (bench configuration) with one ops-level switch flipped:
usage: python tools/ab_toggle.py ATTR [ATTR ...]   e.g. CALAYER_FUSED HEADS_FUSED   (module attributes of sug_amd.ops that are
read when a step is built / captured; variant A = as shipped, variant B = the attribute set to False)."""
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
trainers = {}
for name, val in [('shipped', True)] + [('no_' + a, a) for a in attrs]:
    torch.manual_seed(666)
    keep = {}
    if val is not True:
        keep[val] = getattr(ops, val)
        setattr(ops, val, False)
    tr = SUGStep(Net_MDA('DGCNN').to(dev).train(), lr=1e-3, weight_decay=5e-5, use_graph=True, methods=BENCH_METHODS)
    for _ in range(4):
        tr.step(*data)
    torch.cuda.synchronize()
    for k, v in keep.items():
        setattr(ops, k, v)
    trainers[name] = tr
res = {k: [] for k in trainers}
for rnd in range(7):
    for name, tr in trainers.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            tr.step(*data)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 20 * 1e3)
for k, v in res.items():
    print('%-22s ms/step: min %.3f median %.3f  all %s' % (k, min(v), sorted(v)[len(v) // 2], ['%.3f' % x for x in v]))

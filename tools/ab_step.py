"""Diagnostic: interleaved in-process A/B timing of SUGStep variants (cdna guide rule 24)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep

dev = torch.device('cuda')
variants = {"graph": dict(use_graph=True), "eager": dict(use_graph=False), "eager_noshare": dict(use_graph=False, share_prefix=False)}
if len(sys.argv) > 1:
    variants = {k: v for k, v in variants.items() if k in sys.argv[1:]}
tr = {}
for name, kw in variants.items():
    torch.manual_seed(666)
    tr[name] = SUGStep(Net_MDA('DGCNN').to(dev).train(), **kw)
data = synth(32, 1024, 666, dev)
for name, t in tr.items():
    for i in range(3):
        t.step(*data)
        torch.cuda.synchronize()
        print('warmup', name, i, 'ok', flush=True)
res = {k: [] for k in tr}
for rnd in range(5):
    for name, t in tr.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            t.step(*data)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 5 * 1e3)
for k, v in res.items():
    print('%-8s ms/step: min %.2f median %.2f  all %s' % (k, min(v), sorted(v)[len(v) // 2], ['%.2f' % x for x in v]))

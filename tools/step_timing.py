"""Diagnostic: ms/step of SUGStep over different loop lengths, with / without the kNN event instrumentation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from sug_amd import ops
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
dev = torch.device('cuda')
torch.manual_seed(666)
tr = SUGStep(Net_MDA('DGCNN').to(dev).train(), lr=1e-3, weight_decay=5e-5)
data = synth(32, 1024, 666, dev)
for _ in range(5):
    tr.step(*data)
torch.cuda.synchronize()
def run(n, instrument):
    ops.PROFILE_ONLY = {'knn'}
    ops.PROFILE = {} if instrument else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(*data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ops.PROFILE = None
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
for n, ins in ((5, False), (30, False), (30, True), (5, True), (30, False), (100, False)):
    e, t = run(n, ins)
    print('steps %3d instrument %-5s enqueue %.2f total %.2f ms/step' % (n, ins, e, t))

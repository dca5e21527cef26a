"""Diagnostic: aten::copy_ / contiguous launches of one step with shapes and the enclosing ops."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
dev = torch.device('cuda')
tr = SUGStep(Net_MDA('DGCNN').to(dev).train())
data = synth(32, 1024, 666, dev)
for _ in range(3):
    tr.step(*data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(*data)
    torch.cuda.synchronize()
agg = collections.Counter()
tim = collections.Counter()
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or e.name != 'aten::copy_' or not e.kernels:
        continue
    chain, a = [], e.cpu_parent
    while a is not None and len(chain) < 4:
        chain.append(a.name.replace('aten::', '').replace('autograd::engine::evaluate_function: ', 'bw:'))
        a = a.cpu_parent
    key = (str(e.input_shapes[:2]), ' < '.join(chain))
    agg[key] += 1
    tim[key] += sum(k.duration for k in e.kernels)
for k, n in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:25]:
    print('%2d x %7.1f us  %-44s %s' % (n, tim[k], k[0], k[1][:110]))

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
for _ in range(3): tr.step(*data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(*data); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::matmul', 'aten::linear') and e.device_time_total > 0 and e.name in ('aten::mm', 'aten::addmm', 'aten::bmm'):
        k = (e.name, str(e.input_shapes))
        agg[k][0] += 1; agg[k][1] += e.device_time_total
tot = sum(v[1] for v in agg.values())
print('GEMM device time per step: %.3f ms' % (tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print('%7.1f us x%2d  %s %s' % (v[1] / v[0], v[0], k[0], k[1]))

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
dev = torch.device('cuda')
net = Net_MDA('DGCNN').to(dev).train()
tr = SUGStep(net, use_graph=False)
data = synth(4, 1024, 666, dev)
tr.step(*data); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(*data); torch.cuda.synchronize()
for e in prof.events():
    if 'scatter' in e.name or 'index_put' in e.name or 'index_add' in e.name:
        print(e.name, e.input_shapes, [s for s in (e.stack or [])[:6]])

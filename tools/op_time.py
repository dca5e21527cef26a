"""GPU time of one training step by aten / custom op and input shapes (torch profiler; diagnostic).
usage: python tools/op_time.py [MODEL] [BATCH] [NPOINTS]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep

model = sys.argv[1] if len(sys.argv) > 1 else 'DGCNN'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device('cuda')
if len(sys.argv) > 4 and sys.argv[4] == 'fp16':
    from sug_amd.model import Ptran_transformer as PT
    PT.GEMM_DTYPE = torch.float16
    PT.PROJ_16BIT = True
torch.manual_seed(666)
net = Net_MDA(model).to(dev).train()
tr = SUGStep(net, use_graph=False)
data = synth(B, N, 666, dev)
for _ in range(3):
    tr.step(*data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(*data)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and not any(c.kernels for c in e.cpu_children):
        t = sum(k.duration for k in e.kernels)
        key = (e.name, str(e.input_shapes)[:110])
        agg[key][0] += len(e.kernels)
        agg[key][1] += t
tot = sum(v[1] for v in agg.values())
print('total kernel time %.2f ms' % (tot / 1e3))
by_count = os.environ.get('BY_COUNT') == '1'
print('launches %d' % sum(v[0] for v in agg.values()))
for (name, shp), (n, t) in sorted(agg.items(), key=lambda kv: -(kv[1][0] if by_count else kv[1][1]))[:45]:
    print('%8.1f us %3d  %-34s %s' % (t, n, name[:34], shp))

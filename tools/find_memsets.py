"""Which ops of a training step issue device memsets (hipMemsetAsync -> __amd_rocclr_fillBufferAligned)?  Memset nodes
of a replayed step graph were not reliably ordered on ROCm 7: the captured step should have none.
usage: python tools/find_memsets.py [MODEL] [BATCH] [NPOINTS] [fp16]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
from sug_amd.tuning import enable_tuned_gemms
model = sys.argv[1] if len(sys.argv) > 1 else 'DGCNN'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
if len(sys.argv) > 4 and sys.argv[4] == 'fp16':
    from sug_amd.model import Ptran_transformer as PT
    PT.GEMM_DTYPE = torch.float16
    PT.PROJ_16BIT = True
enable_tuned_gemms()
dev = torch.device('cuda')
torch.manual_seed(666)
tr = SUGStep(Net_MDA(model).to(dev).train(), use_graph=False)
data = synth(B, N, 666, dev)
for _ in range(3):
    tr.step(*data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(*data)
    torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        n = sum(1 for k in e.kernels if 'emset' in k.name or 'fillBuffer' in k.name)
        if n and not any(any('emset' in k.name or 'fillBuffer' in k.name for k in c.kernels) for c in e.cpu_children):
            agg[(e.name, str(e.input_shapes)[:100])] += n
print('memsets per step: %d' % sum(agg.values()))
cp = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        n = sum(1 for k in e.kernels if 'emcpy' in k.name)
        if n and not any(any('emcpy' in k.name for k in c.kernels) for c in e.cpu_children):
            cp[(e.name, str(e.input_shapes)[:100])] += n
print('device memcpys per step: %d' % sum(cp.values()))
for (name, shp), n in cp.most_common(12):
    print('%3d  %-32s %s' % (n, name, shp))
for (name, shp), n in agg.most_common():
    print('%3d  %-32s %s' % (n, name, shp))

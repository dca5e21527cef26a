"""Child process of test_training_step_issues_no_device_memsets: one eager training step under torch.profiler, prints
{"kernels": n, "memsets": [names]}.  A child because the profiler's teardown (kineto / roctracer, stop_trace) now and then
takes the whole process down with a segmentation fault on this stack; the test retries, the test run survives.
usage: python tests/memset_probe.py MODEL [fp16]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from sug_amd.model.Model import Net_MDA
from sug_amd.model import Ptran_transformer as PT
from sug_amd.train_step import SUGStep

model_name = sys.argv[1]
if len(sys.argv) > 2 and sys.argv[2] == 'fp16':
    PT.GEMM_DTYPE, PT.PROJ_16BIT = torch.float16, True
g = torch.Generator().manual_seed(3)
B, N = 4, 1024
data = (torch.rand(B, 3, N, 1, generator=g) * 2 - 1).cuda()
data_t = (torch.rand(B, 3, N, 1, generator=g) * 2 - 1).cuda()
label = torch.randint(0, 10, (B,), generator=g).cuda()
label_t = torch.randint(0, 10, (B,), generator=g).cuda()
torch.manual_seed(1)
tr = SUGStep(Net_MDA(model_name).cuda().train(), use_graph=False)
for _ in range(2):
    tr.step(data, label, data_t, label_t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(data, label, data_t, label_t)
    torch.cuda.synchronize()
names = [k.name for e in prof.events() for k in (e.kernels or [])]
print('PROBE ' + json.dumps({'kernels': len(names), 'memsets': [n for n in names if 'emset' in n or 'fillBuffer' in n][:5]}), flush=True)
os._exit(0)      # results are out: skip the interpreter's teardown of the profiler state

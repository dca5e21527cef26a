"""aten-op and kernel-launch counts of one training step, with the Python source of the top ones."""
import sys, collections, torch
sys.path.insert(0, '.')
import bench
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
torch.manual_seed(666)
net = Net_MDA('DGCNN').to(dev).train()
tr = SUGStep(net, lr=1e-3, weight_decay=5e-5)
data = bench.synth(32, 1024, 666, dev)
for _ in range(3):
    tr.step(*data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(*data)
    torch.cuda.synchronize()
ev = prof.events()
launch_src = collections.Counter()
kern_by_op = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::') and e.kernels:
        # innermost aten op that owns kernels
        if any(c.name.startswith('aten::') and c.kernels for c in e.cpu_children):
            continue
        src = next((s for s in e.stack if '/repo/' in s or 'sug_amd' in s), (e.stack[0] if e.stack else '?'))
        launch_src[(e.name, src.split('/repo/')[-1][:70])] += len(e.kernels)
        kern_by_op[e.name] += len(e.kernels)
print('launches by aten op:')
for k, v in kern_by_op.most_common(25):
    print('  %-40s %d' % (k, v))
print('launches by (op, source):')
for (k, s), v in launch_src.most_common(60):
    print('  %4d %-32s %s' % (v, k, s))

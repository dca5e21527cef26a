"""Kernel launches of one training step by phase and aten op (diagnostic)."""
import sys, collections, torch
sys.path.insert(0, '.')
import bench
from sug_amd import ops
from sug_amd.model.Model import Net_MDA
from sug_amd.model import mmd
from sug_amd.train_step import SUGStep
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
torch.manual_seed(666)
net = Net_MDA('DGCNN').to(dev).train()
tr = SUGStep(net, lr=1e-3, weight_decay=5e-5)
data, lab, data_t, lab_t = bench.synth(32, 1024, 666, dev)
for _ in range(3):
    tr.step(data, lab, data_t, lab_t)
torch.cuda.synchronize()

def phase(name, fn):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    cnt = collections.Counter()
    total = 0
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
            if any(c.kernels for c in e.cpu_children):
                continue
            cnt[e.name] += len(e.kernels)
            total += len(e.kernels)
    print('== %-28s %4d launches: %s' % (name, total, ', '.join('%s %d' % (k.replace('aten::', ''), v) for k, v in cnt.most_common(14))))
    if name == 'backward':
        by_node = collections.Counter()
        detail = collections.defaultdict(collections.Counter)
        for e in prof.events():
            if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and not any(c.kernels for c in e.cpu_children):
                a = e
                top = None
                while a is not None:
                    if 'evaluate_function' in a.name:
                        top = a.name.split(': ')[-1]
                    a = a.cpu_parent
                by_node[top] += len(e.kernels)
                detail[top][e.name.replace('aten::', '')] += len(e.kernels)
        for k, v in by_node.most_common(40):
            print('      %-36s %3d  %s' % (k, v, dict(detail[k])))
    return out

M = tr.methods
pair = torch.cat((data, data_t), 0)
with ops.bn_groups(2):
    enc = phase('encoder sem (g only)', lambda: net.g(pair, node=True))
s, t = phase('forward_pair sem (g+heads)', lambda: net.forward_pair(pair))
ns, nt = phase('forward_pair node (shared)', lambda: net.forward_pair(pair, node_adaptation=True))
geo, sem = M['GEO_MMD'][0], M['SEM_MMD'][0]
lg = phase('mmd geo', lambda: tr._mmd(lab, ns, lab_t, nt, geo, data, data_t))
l1 = phase('mmd sem (one of two)', lambda: tr._mmd(lab, s[2], lab_t, t[2], sem, s[0], t[0]))
l2 = tr._mmd(lab, s[3], lab_t, t[3], sem, s[1], t[1])
lc = phase('cls loss', lambda: 0.5 * tr.criterion(s[0], lab) + 0.5 * tr.criterion(s[1], lab))
loss = lc + lg + 0.5 * l1 + 0.5 * l2
phase('backward', lambda: loss.backward())
net.g.clear_prefix_cache()
phase('3 x adam + zero_grad', lambda: [o.step() for o in (tr.optimizer_dis, tr.optimizer_g, tr.optimizer_c)] + [o.zero_grad() for o in (tr.optimizer_dis, tr.optimizer_g, tr.optimizer_c)])

"""Which gradients differ between the eager caller form and the graphed one at the first captured step?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('SUG_CALL_GRAPHS_STRICT', '1')
import torch
import test_gpu_call_graphs as T
from sug_amd.model.Model import Net_MDA
from sug_amd.model import mmd
model = sys.argv[1] if len(sys.argv) > 1 else 'Pointnet'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
terms = sys.argv[4] if len(sys.argv) > 4 else 'cgs'
res = {}
for graphs in (False, True):
    net, batches = T._make(model, B, N)
    Net_MDA.call_graphs = 'auto' if graphs else False
    torch.manual_seed(3)
    crit = torch.nn.CrossEntropyLoss()
    for s in range(3):
        data, label, data_t, label_t = batches[0]
        ps1, ps2, ss1, ss2 = net(data, semantic_adaption=True)
        pt1, pt2, st1, st2 = net(data_t, semantic_adaption=True)
        loss = 0
        if 'c' in terms:
            loss = loss + 0.5 * crit(ps1, label) + 0.5 * crit(ps2, label)
        if 'g' in terms:
            ns = net(data, node_adaptation_s=True)
            nt = net(data_t, node_adaptation_t=True)
            loss = loss + mmd.mmd_cal(label, ns, label_t, nt, T.GEO, data_s=data, data_t=data_t)
        if 's' in terms:
            loss = loss + 0.5 * mmd.mmd_cal(label, ss1, label_t, st1, T.SEM, data_s=ps1, data_t=pt1) + 0.5 * mmd.mmd_cal(label, ss2, label_t, st2, T.SEM, data_s=ps2, data_t=pt2)
        loss.backward()
        res[(graphs, s)] = (float(loss), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                            {k: v.clone() for k, v in net.state_dict().items() if 'running' in k or 'num_batches' in k})
        net.zero_grad(set_to_none=True)
    mgr = net.__dict__.get('_call_graph_mgr')
    if mgr:
        print(mgr.stats)
        for k, ks in mgr.keys.items():
            print('  key flags', k[0], 'instances', [(id(i) % 10000, None if i.dep is None else id(i.dep) % 10000, i.generation, i.busy) for i in ks.instances], ks.why)
Net_MDA.call_graphs = False
for s in range(3):
    a, b = res[(False, s)], res[(True, s)]
    print('step', s, 'loss', a[0], b[0], 'grad keys equal', a[1].keys() == b[1].keys())
    bad = [(k, float((a[1][k] - b[1][k]).abs().max()), float(a[1][k].abs().max())) for k in a[1] if k in b[1] and not torch.equal(a[1][k], b[1][k])]
    print('   differing grads: %d of %d' % (len(bad), len(a[1])), bad[:12])
    badb = [k for k in a[2] if not torch.equal(a[2][k], b[2][k])]
    print('   differing buffers:', badb[:10])

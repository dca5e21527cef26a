import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zlib
from tests.conftest import load_golden
from oracle import ref_cpu as O
from sug_amd.model import model_utils
from sug_amd.model.Model import Net_MDA

def probe(shape, tag):
    g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % (2 ** 31)); return torch.randn(shape, generator=g)

G = load_golden('model_pointnet.npz'); seed = G['seed']
from sug_amd import ops
REC = []
_kq, _t3 = ops.knn_query, ops.three_nn_raw
def kq(*a, **k):
    r = _kq(*a, **k); REC.append(('knn_query', (r[0] if isinstance(r, tuple) else r).clone())); return r
def t3(*a, **k):
    r = _t3(*a, **k); REC.append(('three_nn', r[0].clone())); return r
ops.knn_query, ops.three_nn_raw = kq, t3
BASE = {}
def run(rule):
    REC.clear()
    model_utils.OWN_BN = rule
    net = Net_MDA('Pointnet')
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d): m.p = 0.0
    net = net.cuda().train()
    torch.manual_seed(seed + 1)
    outs = net(G['x'].cuda(), semantic_adaption=True)
    loss = sum((t * probe(t.shape, 'probe%d' % i).cuda()).sum() for i, t in enumerate(outs))
    loss.backward()
    got = dict(net.named_parameters())
    worst = []
    gmax = max(G['grad_norm'].tolist())
    for k, gn, gd in zip(G['grad_names'], G['grad_norm'].tolist(), G['grad_dot'].tolist()):
        if gn < 1e-3 * gmax:
            continue
        g = got[k].grad
        d = (g.cpu() * probe(g.shape, 'g' + k)).sum().item()
        worst.append((abs(d - gd) / max(gn, 1e-9), k, d, gd, gn))
    worst.sort(reverse=True)
    for i, (nm, t) in enumerate(REC):
        if i not in BASE:
            BASE[i] = t
        else:
            print('      %s #%d: %d of %d indices differ from the torch-BN run' % (nm, i, int((BASE[i] != t).sum()), t.numel()))
    return worst[:2]
for name, rule in (('torch BN everywhere', lambda bn: False), ('own BN everywhere', lambda bn: True),
                   ('own BN only C<=128', lambda bn: bn.num_features <= 128), ('own BN only C==1024', lambda bn: bn.num_features == 1024),
                   ('own BN only C==64', lambda bn: bn.num_features == 64)):
    print(name)
    for w in run(rule): print('   rel %.2e %s got %.5g want %.5g norm %.4g' % w)

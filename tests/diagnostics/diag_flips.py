"""Diagnostic (not a test): where does the free-running DGCNN deviate from the oracle?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.conftest import load_golden
from oracle import ref_cpu as O
from sug_amd.model.Model import Net_MDA
from sug_amd import ops

G = load_golden('step_dgcnn.npz')
seed = G['seed']
net = Net_MDA('DGCNN')
sd = O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed)
net.load_state_dict(sd)
for m in net.modules():
    if isinstance(m, torch.nn.Dropout2d):
        m.p = 0.0
net = net.cuda().train()
x = G['data']
p = {k: v.clone() for k, v in sd.items()}
torch.manual_seed(seed)
with torch.no_grad():
    feat_o, node_o, (x1, x2, x3, x4) = O.dgcnn_g(p, 'g.', x, True, None)
# GPU: layer by layer with the oracle's features as inputs -> isolates kNN flips
for name, f in (('xyz', x.squeeze(-1)), ('x1', x1), ('x2', x2), ('x3', x3)):
    idx_o = O.knn_idx(f, 20)
    idx_g = ops.knn(f.transpose(1, 2).contiguous().cuda(), 20).cpu().long()
    score = O.knn_neg_dist(f)
    neq = (idx_o != idx_g)
    sh, sr = torch.gather(score, 2, idx_g), torch.gather(score, 2, idx_o)
    gap = (sh - sr).abs()
    setdiff = sum(len(set(a.tolist()) ^ set(b.tolist())) > 0 for a, b in zip(idx_o.reshape(-1, 20), idx_g.reshape(-1, 20)))
    print('%s: C=%d mismatched slots %d / %d, rows with different SETS %d, max score gap %.3e (score scale %.3e)'
          % (name, f.shape[1], int(neq.sum()), neq.numel(), setdiff, float(gap.max()), float(score.abs().max())))
torch.manual_seed(seed)
with torch.no_grad():
    feat_g, node_g, _ = net.g(x.cuda(), node=True)
print('free-running feat err %.3e  node err %.3e' % ((feat_g.cpu() - feat_o).abs().max(), (node_g.cpu() - node_o).abs().max()))

# ---- heads / losses
import torch.nn.functional as F
lab = G['label']
torch.manual_seed(seed)
with torch.no_grad():
    o = O.net_mda(p, 'DGCNN', x, True, None, semantic_adaption=True)
torch.manual_seed(seed)
with torch.no_grad():
    g_ = net(x.cuda(), semantic_adaption=True)
for a, b, nm in zip(g_, o, ('y1', 'y2', 's1', 's2')):
    print(nm, 'err %.3e' % (a.cpu() - b).abs().max())
print('CE oracle', (0.5 * F.cross_entropy(o[0], lab) + 0.5 * F.cross_entropy(o[1], lab)).item(),
      'gpu', (0.5 * F.cross_entropy(g_[0].cpu(), lab) + 0.5 * F.cross_entropy(g_[1].cpu(), lab)).item(), 'golden', G['losses'][0])

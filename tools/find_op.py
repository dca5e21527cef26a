"""Where do the small aten launches of a step's forward come from?  usage: python tools/find_op.py [MODEL] [OP ...]
(ops by aten packet name: cat mul add ...).  Counts calls per (op, innermost sug_amd / bench call site) in one eager step."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
model = sys.argv[1] if len(sys.argv) > 1 else 'DGCNN'
names = set(sys.argv[2:])        # empty: every op that is not a view / metadata op
dev = torch.device('cuda')
net = Net_MDA(model).to(dev).train()
tr = SUGStep(net, use_graph=False)
data = synth(32, 1024, 666, dev)
for _ in range(2):
    tr.step(*data)
torch.cuda.synchronize()
agg = collections.Counter()
SKIP = {'view', '_unsafe_view', 'reshape', 't', 'transpose', 'permute', 'slice', 'select', 'expand', 'unsqueeze', 'squeeze', 'detach', 'alias',
        'empty', 'empty_like', 'empty_strided', 'as_strided', 'narrow', 'split', 'chunk', 'unbind', 'size', 'stride', 'is_same_size',
        'split_with_sizes', 'lift_fresh', '_local_scalar_dense', 'new_empty', 'view_as', 'contiguous', 'unflatten', 'flatten'}


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if (name in names) if names else name not in SKIP:
            site = '?'
            for fr in reversed(traceback.extract_stack()):
                if ('sug_amd' in fr.filename or 'bench' in fr.filename) and 'find_op' not in fr.filename:
                    site = '%s:%d %s' % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            agg[(name, site, str(shp)[:50])] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    tr.step(*data)
torch.cuda.synchronize()
for (n, site, shp), c in sorted(agg.items(), key=lambda kv: -kv[1]):
    print('%3d  %-8s %-60s %s' % (c, n, site, shp))

"""Which torch (aten) ops still launch kernels in the BENCHED step (paired domains, shared prefix, fused heads, BENCH_METHODS)?
usage: python tools/aten_launches.py [MODEL] [BATCH] [NPOINTS]
Part 1: kernel launches of one eager step grouped by (aten / custom op, input shapes), at:: kernels only unless ALL=1.
Part 2: the same ops by call site (innermost sug_amd frame; backward ops show the autograd node that issued them)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from torch.utils._python_dispatch import TorchDispatchMode
from bench import synth, BENCH_METHODS
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep

model = sys.argv[1] if len(sys.argv) > 1 else 'DGCNN'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device('cuda')
if len(sys.argv) > 4 and sys.argv[4] == 'fp16':
    from sug_amd.model import Ptran_transformer as PT
    PT.GEMM_DTYPE, PT.PROJ_16BIT = torch.float16, True
torch.manual_seed(666)
net = Net_MDA(model).to(dev).train()
tr = SUGStep(net, use_graph=False, methods=BENCH_METHODS)
tr.fused_heads = True                    # what the captured step runs
if os.environ.get('CALLER') == '1':      # the unchanged-caller form (four separate model(...) calls), launched eagerly
    from sug_amd.model.Model import Net_MDA as _N
    _N.call_graphs = False
    tr.pair_domains = tr.share_prefix = False
    net.g.share_prefix = 'auto'
    for m_ in tr._split_layers:
        m_.cache_weight_split = False
if os.environ.get('TUNED') == '1':
    from sug_amd.tuning import enable_tuned_gemms
    enable_tuned_gemms()
data = synth(B, N, 666, dev)
for _ in range(3):
    tr.step(*data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(*data)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, set()])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and not any(c.kernels for c in e.cpu_children):
        for k in e.kernels:
            if os.environ.get('ALL') == '1' or k.name.startswith(('at::', 'void at::', '__amd_rocclr', 'Cijk')) or 'at::native' in k.name:
                key = (e.name, str(e.input_shapes)[:100])
                agg[key][0] += 1
                agg[key][1] += k.duration
                agg[key][2].add(k.name[:50])
print('library / torch kernel launches %d, %.1f us' % (sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values())))
for (name, shp), (n, t, ks) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%8.1f us %3d  %-30s %-100s %s' % (t, n, name[:30], shp, sorted(ks)[0]))

SKIP = {'view', '_unsafe_view', 'reshape', 't', 'transpose', 'permute', 'slice', 'select', 'expand', 'unsqueeze', 'squeeze', 'detach', 'alias',
        'empty', 'empty_like', 'empty_strided', 'as_strided', 'narrow', 'split', 'chunk', 'unbind', 'size', 'stride', 'is_same_size',
        'split_with_sizes', 'lift_fresh', '_local_scalar_dense', 'new_empty', 'view_as', 'contiguous', 'unflatten', 'flatten', 'set_',
        'new_empty_strided', '_reshape_alias', 'unsafe_split', 'resize_'}
sites = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in SKIP:
            site = '?'
            for fr in reversed(traceback.extract_stack()):
                if ('sug_amd' in fr.filename or 'bench' in fr.filename) and 'aten_launches' not in fr.filename:
                    site = '%s:%d %s' % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            sites[(name, site, str(shp)[:60])] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    tr.step(*data)
torch.cuda.synchronize()
print('\n--- aten ops by call site (one step) ---')
for (n, site, shp), c in sorted(sites.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print('%3d  %-22s %-62s %s' % (c, n, site, shp))

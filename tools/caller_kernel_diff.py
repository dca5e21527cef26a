"""Where does the unchanged-caller form (four separate Net_MDA calls per step, train_dg_single_gpu.py:260-310) spend more GPU
time than the paired step?  Per-kernel microseconds per step of both forms (torch.profiler kernel timestamps, eager
launches, 3 steps each), sorted by the difference.  usage: python tools/caller_kernel_diff.py [rows]"""
import collections, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synth, BENCH_METHODS
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
from sug_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device('cuda')


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    m = re.match(r'at::native::(\w+)<.*?at::native::(\w+)', n)
    if m:
        return 'at::%s<%s>' % (m.group(1), m.group(2))
    n = re.sub(r'\(.*', '', n)
    return n[:70]


tab = {}
for mode in ('caller', 'paired'):
    torch.manual_seed(666)
    tr = SUGStep(Net_MDA('DGCNN').to(dev).train(), methods=BENCH_METHODS)
    if mode == 'caller':
        tr.pair_domains = tr.share_prefix = False
        tr.model.g.share_prefix = 'auto'
        for m_ in tr._split_layers:
            m_.cache_weight_split = False
    data = synth(32, 1024, 666, dev)
    for _ in range(3):
        tr.step(*data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(*data)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            tr.step(*data)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            a = agg[short(e.name)]
            a[0] += 1
            a[1] += e.device_time
    tab[mode] = {k: (v[0] / 3.0, v[1] / 3.0) for k, v in agg.items()}
    print('%-7s wall %.2f ms/step (eager), GPU kernel time %.2f ms/step in %d launches'
          % (mode, wall, sum(v[1] for v in tab[mode].values()) / 1e3, sum(v[0] for v in tab[mode].values())))
rows = []
for k in set(tab['caller']) | set(tab['paired']):
    c, p = tab['caller'].get(k, (0, 0.0)), tab['paired'].get(k, (0, 0.0))
    rows.append((c[1] - p[1], k, c, p))
rows.sort(reverse=True)
print('%-72s %8s %8s   %8s %8s   %8s' % ('kernel', 'n call', 'us call', 'n pair', 'us pair', 'diff us'))
for d, k, c, p in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 40]:
    print('%-72s %8.1f %8.1f   %8.1f %8.1f   %8.1f' % (k, c[0], c[1], p[0], p[1], d))

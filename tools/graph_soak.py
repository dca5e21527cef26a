"""Soak test of SUGStep graph mode: N replays, loss finiteness, parameter agreement with an eager twin
after the same number of steps on the same data (dropout off so both see the same arithmetic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
dev = torch.device('cuda')
if os.environ.get('FIXED_START'):          # same FPS start (index 0) in every call: graph and eager see identical arithmetic
    _ri = torch.randint
    torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=kw.get('dtype', torch.long))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32


def make(use_graph):
    torch.manual_seed(666)
    net = Net_MDA('DGCNN').to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    return net, SUGStep(net, lr=1e-3, weight_decay=5e-5, use_graph=use_graph)


data = synth(B, 1024, 666, dev)
res = {}
for mode in (True, False, None):        # graph, eager, eager again (run-to-run noise of the eager path)
    net, tr = make(bool(mode))
    torch.manual_seed(1234)                      # FPS start draws
    t0 = time.perf_counter()
    for i in range(N if mode else min(N, int(os.environ.get('EAGER_STEPS', '60')))):
        l = tr.step(*data)
        if i % 100 == 99:
            torch.cuda.synchronize()
            print('%s step %d loss %.4f %.4f %.4f  %.2f ms/step' % ('graph' if mode else 'eager', i + 1, float(l[0]), float(l[1]), float(l[2]),
                                                                    (time.perf_counter() - t0) / (i + 1) * 1e3), flush=True)
    torch.cuda.synchronize()
    res[mode] = [float(x) for x in l]
    assert all(x == x for x in res[mode]), 'NaN loss'
print('final losses graph(%d steps) %s | eager(%d steps) %s | eager again %s' % (N, res[True], min(N, int(os.environ.get('EAGER_STEPS', '60'))), res[False], res[None]))

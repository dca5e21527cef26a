"""Diagnostic: per-step wall times of a long run (GC collected + frozen as in bench.py); prints outliers."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
from sug_amd.tuning import enable_tuned_gemms
dev = torch.device('cuda')
enable_tuned_gemms()
torch.manual_seed(666)
tr = SUGStep(Net_MDA('DGCNN').to(dev).train(), lr=1e-3, weight_decay=5e-5)
data = synth(32, 1024, 666, dev)
for _ in range(5):
    tr.step(*data)
torch.cuda.synchronize()
gc.collect(); gc.freeze()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ts = []
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info.get('generation'), time.perf_counter())))
for i in range(N):
    t0 = time.perf_counter()
    tr.step(*data)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts_sorted = sorted(ts)
print('median %.2f  p90 %.2f  max %.2f  mean %.2f' % (ts_sorted[N // 2], ts_sorted[int(N * 0.9)], ts_sorted[-1], sum(ts) / N))
for i, t in enumerate(ts):
    if t > 1.5 * ts_sorted[N // 2]:
        print('  spike at step %d: %.2f ms' % (i, t))
g2 = [g for g in gcs if g[0] == 'start' and g[1] == 2]
print('gen2 collections during the run: %d; all gc starts: %d' % (len(g2), len([g for g in gcs if g[0] == 'start'])))

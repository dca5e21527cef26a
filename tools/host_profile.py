import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
dev = torch.device('cuda')
tr = SUGStep(Net_MDA('DGCNN').to(dev).train())
if len(sys.argv) > 1 and sys.argv[1] == 'caller':          # the unchanged-caller form bench.py times: four separate model(...) calls
    tr.pair_domains = tr.share_prefix = False
    tr.model.g.share_prefix = 'auto'
    for m_ in tr._split_layers:
        m_.cache_weight_split = False
data = synth(32, 1024, 666, dev)
for _ in range(3):
    tr.step(*data)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    tr.step(*data)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('enqueue ms/step %.2f, total ms/step %.2f' % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(*data)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
pstats.Stats(pr).sort_stats('cumtime').print_stats(45)

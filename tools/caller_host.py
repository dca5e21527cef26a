"""Host time vs wall time of the unchanged-caller form (four graphed Net_MDA calls + SUGStep's eager tail): the host's
enqueue time for K steps (before the final synchronize) against the wall time including it.  usage: caller_host.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth, BENCH_METHODS
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
from sug_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device('cuda')
torch.manual_seed(666)
tr = SUGStep(Net_MDA('DGCNN').to(dev).train(), methods=BENCH_METHODS)
tr.pair_domains = tr.share_prefix = False
tr.model.g.share_prefix = 'auto'
for m_ in tr._split_layers:
    m_.cache_weight_split = False
data = synth(32, 1024, 666, dev)
for _ in range(5):
    tr.step(*data)
torch.cuda.synchronize()
import gc
gc.collect(); gc.disable()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K):
        tr.step(*data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host enqueue %.3f ms/step, wall %.3f ms/step' % (1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K), flush=True)
mgr = tr.model.__dict__.get('_call_graph_mgr')
print(mgr.stats if mgr else None)
if os.environ.get('CPROF'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20):
        tr.step(*data)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(35)

"""Diagnostic: wall time of each of the first 60 steps (sync after each), allocator stats, gc counts."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
dev = torch.device('cuda')
torch.manual_seed(666)
tr = SUGStep(Net_MDA('DGCNN').to(dev).train(), lr=1e-3, weight_decay=5e-5)
data = synth(32, 1024, 666, dev)
ts = []
for i in range(60):
    st = torch.cuda.memory_stats()
    a0, g0 = st.get('num_device_alloc', 0), gc.get_count()
    t0 = time.perf_counter()
    tr.step(*data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3, st.get('num_device_alloc', 0) - a0, st.get('num_alloc_retries', 0), gc.get_count(), st['reserved_bytes.all.current'] >> 20))
for i, t in enumerate(ts):
    print('step %2d enqueue %6.2f total %6.2f  hipMallocs %d retries %d gc %s reserved %d MiB' % ((i,) + t))

"""GPU kernel time of one training step in the unchanged-caller form (four separate Net_MDA calls of one domain each, as
train_dg_single_gpu.py:260-310 makes them) next to its wall time: is that form host-bound or bound by the half-batch launches?
usage: python tools/caller_gpu_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
from sug_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device('cuda')
for mode in ('caller', 'paired eager'):
    torch.manual_seed(666)
    tr = SUGStep(Net_MDA('DGCNN').to(dev).train())
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
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    gpu = sum(e.device_time for e in ev) / 3 / 1e3
    print('%-13s wall %.2f ms/step, GPU kernel time %.2f ms/step in %d launches' % (mode, wall, gpu, len(ev) // 3))

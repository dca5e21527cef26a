#!/usr/bin/env python3
"""Benchmark of the SUG hot path on MI355X: full training steps of the DGCNN-backbone
Net_MDA with MSA + SDA losses (BASELINE.json configs[1]: N=1024, k=20, batch 32 per domain).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = 2 semantic + 2 node forwards (4 encoder passes over 2B clouds), 3 soft-MMD losses
with SDA weights, one backward, 3 Adam updates (train_dg_single_gpu.py:246-335).  Inputs are
synthetic (U(-1,1)^3 -> normal_pc, labels randint(0,10)) and resident in HBM before the
timed region; weights are random-init.  Prints ONE JSON line on rank 0.

`roofline` describes the dominant hand-written kernel of the step, timed live with events
on the stream it is launched on; `cpu_baseline` times the CPU oracle (oracle/ref_cpu.py,
the restatement of the reference's algorithm) on a bounded sample of the same workload.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP32_PEAK_TFLOPS = 157.3       # fp32 vector = fp32-input MFMA rate


def synth(B, N, seed, device):
    g = torch.Generator().manual_seed(seed)
    def clouds():
        pc = torch.rand(B, N, 3, generator=g) * 2 - 1
        pc = pc - pc.mean(dim=1, keepdim=True)
        pc = pc / pc.pow(2).sum(-1).sqrt().max(dim=1)[0].view(B, 1, 1)     # normal_pc, data/data_utils.py:5-15
        return pc.permute(0, 2, 1).unsqueeze(-1).contiguous()
    data, data_t = clouds(), clouds()
    lab, lab_t = torch.randint(0, 10, (B,), generator=g), torch.randint(0, 10, (B,), generator=g)
    return [t.to(device) for t in (data, lab, data_t, lab_t)]


BACKBONE = {'DGCNN': 'DGCNN EdgeConv backbone k=20', 'Pointnet': 'PointNet backbone', 'Pointnet2': 'PointNet++ backbone',
            'PTran': 'Point Transformer backbone k=16'}


def kernel_model(name, shape):
    """Algorithmic bytes / FLOPs of one launch (per-kernel table)."""
    B, N, k = shape['B'], shape['N'], shape['k']
    if name.startswith('knn'):
        C = shape['C']
        return {'bytes': B * (4 * C * N + 4 * N * k), 'flops': B * N * N * (2 * C + 3)}
    if name.startswith('edgeconv_fused_fwd'):       # fused layer call (2 launches): SURVEY 8d's fused-layer bytes
        C, Co = shape['C'], shape['Co']              # x, idx in; activations out; + arg-max bytes when a backward follows
        return {'bytes': B * N * (4 * C + 4 * Co + 4 * k + (Co if shape.get('train') else 0)),
                'flops': B * N * (2 * C * 2 * Co + 6 * k * Co)}
    if name.startswith('ptran_'):                   # k-expanded [B*n*k, d] operands, e bytes per element
        R, d, e = B * N * k, shape['d'], shape['e']
        nops = {'ptran_pos1_fwd': 1, 'ptran_qk_fwd': 2, 'ptran_attn_fwd': 2, 'ptran_attn_bwd': 4, 'ptran_qk_bwd': 2,
                'ptran_relu_bwd': 2, 'ptran_pos1_bwd': 1}.get(name.rsplit('_n', 1)[0], 1)
        return {'bytes': nops * R * d * e, 'flops': 0}
    if name.startswith(('edgeconv_fwd', 'edgeconv_layer_fwd')):     # layer call: + stats fold + affine pass
        Co = shape['Co']            # read PQ (8Co) + idx (4k); write z (4Co) + arg (Co) + s1 (4Co)
        return {'bytes': B * N * (8 * Co + 4 * k + 9 * Co), 'flops': B * N * k * Co * 6}
    if name.startswith(('edgeconv_bwd', 'edgeconv_layer_bwd')):     # layer call: + reverse lists + BN sums
        Co = shape['Co']            # read a, arg, s1, PQ, rev lists; write dPQ (8Co)
        return {'bytes': B * N * (4 * Co + Co + 4 * Co + 8 * Co + 8 * Co + 4 * k + 4), 'flops': B * N * k * Co * 4}
    if name.startswith('pointmlp_max'):             # shape: B = rows, N = segment length, k = input channels
        R, L, K, Co = shape['B'], shape['N'], shape['k'], shape['Co']
        return {'bytes': 4 * R * K + 4 * Co * K + 8 * (R // L) * Co, 'flops': 2 * R * K * Co}
    return {'bytes': 0, 'flops': 0}


# The shipped METHODS block (tools/cfgs/cfgs_local/DG_unified_loss.yaml:13-30): soft MMD on both levels,
# SDA weights on both ('mean2one'); TARGET_LOSS / class weighting are data-loader-side options left off.
BENCH_METHODS = {'GEO_MMD': [{'NAME': 'SOFT_MMD', 'LABEL_SCALE': 50, 'GEO_WEIGHTS': 'mean2one', 'GEO_SCALE': 1}],
                 'SEM_MMD': [{'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'mean2one', 'LABEL_WEIGHT': 0.5,
                              'SEM_SCALE': 1}]}


KERNEL_SOURCES = {'knn': ('knn_pc.hip', 'mfma_tile.h', 'common.h'), 'edgeconv_fused': ('edgeconv_fused.hip', 'common.h'),
                  'edgeconv': ('edgeconv.hip', 'common.h'), 'pointmlp': ('pointmlp.hip', 'mfma_tile.h', 'common.h'),
                  'ptran': ('ptran.hip', 'common.h')}


def kernel_source_hash(name):
    """sha256 (12 hex digits) of the sources of the kernel family `name` belongs to: the PMC traffic file of a kernel
    carries this hash in its name (tools/pmc_traffic.py), so numbers of an older kernel version are never reported."""
    import hashlib
    fam = max((f for f in KERNEL_SOURCES if name.startswith(f)), key=len, default=None)
    if fam is None:
        return None
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[fam]:
        with open(os.path.join(ROOT, 'sug_amd', 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return fam + '_' + h.hexdigest()[:12]


def pmc_traffic(name, shape):
    """HBM bytes per launch of kernel `name` from the committed PMC passes, or None when there is no file for the
    CURRENT sources of that kernel or the shape differs."""
    tag = kernel_source_hash(name)
    if tag is None:
        return None
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic_%s.json' % tag))).get(name)
    except (OSError, ValueError):
        return None
    if rec and all(rec.get(k) == shape.get(k) for k in ('B', 'N', 'k')):
        return rec
    return None


def measured_copy_gbps(dev, mib=1024, iters=10):
    """What this box's HBM actually sustains (report the achievable rate beside the 8 TB/s vendor peak): a device-to-
    device copy of `mib` MiB, read + write counted, HIP events around `iters` back-to-back copies."""
    n = mib * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    a.fill_(1.0)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    del a, b
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def hold_gpu(ms):
    """Occupy the current stream for about `ms` milliseconds (a spin kernel), so that everything enqueued meanwhile queues
    up behind it and then runs back to back."""
    torch.cuda._sleep(int(ms * 1e-3 * 2.1e9))          # cycles at ~2.1 GHz


# ops._timed name -> substring of the ONE device kernel behind that C-ABI call (ops that are a single launch): their
# durations are read from the kernels' own begin / end timestamps (torch.profiler = roctracer), which is what
# `rocprofv3 --kernel-trace` reports; an event pair around the call reads 5-12 us more (tools/event_overhead.py).
DEVICE_KERNEL_OF = {'knn_C3': 'knn_pc_kernel<4,', 'knn_C64': 'knn_pc_kernel<64,', 'knn_C128': 'knn_pc_kernel<128,'}


def device_kernel_durations(run_steps):
    """{kernel name: [duration in ms, ...]} of every device kernel launched by run_steps(), from the kernel timestamps."""
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run_steps()
        torch.cuda.synchronize()
    out = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            out.setdefault(e.name, []).append(e.device_time * 1e-3)
    return out


def kernel_timestamps_in_child(args, want_events=False):
    """A child `bench.py --timestamps-child` (same model / batch / launch switches, ONE rank, no other pass) replays the
    captured step a few times under torch.profiler and returns {'exact': {kernel: [ms, ...]} or None, 'events': ...}.
    want_events: the child first times the hand-written families with HIP events on queued eager steps (what a one-rank
    parent does in-process) -- {name: [[ms, shape], ...]} -- and writes that out before the profiler pass, so it survives a
    crash of the latter.  Nothing usable -> {'exact': None, 'events': None}."""
    import subprocess, tempfile
    out = os.path.join(tempfile.gettempdir(), 'sug_bench_timestamps_%d.json' % os.getpid())
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--model', args.model, '--batch', str(args.batch),
           '--npoints', str(args.npoints), '--warmup', '3', '--profile-steps', str(max(args.profile_steps, 1)),
           '--timestamps-child', out]
    for flag, on in (('--fp16', args.fp16), ('--no-share-prefix', args.no_share_prefix), ('--no-tuned-gemms', args.no_tuned_gemms),
                     ('--no-pair', args.no_pair), ('--child-events', want_events), ('--single-pass', args.single_pass)):
        if on:
            cmd.append(flag)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',
                                                               'GROUP_RANK', 'ROLE_RANK', 'LOCAL_WORLD_SIZE', 'TORCHELASTIC_RUN_ID')}
    res = {'exact': None, 'events': None}
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=180)
        if r.returncode != 0:
            print('bench.py: kernel-timestamp child failed (rc %s); HIP-event timings stand' % r.returncode, file=sys.stderr)
        if os.path.exists(out):
            with open(out) as f:
                got = json.load(f)
            res['exact'] = got.get('exact') or None
            res['events'] = got.get('events') or None
    except Exception as e:
        print('bench.py: kernel-timestamp child skipped (%s)' % str(e).splitlines()[0], file=sys.stderr)
    finally:
        if os.path.exists(out):
            os.remove(out)
    return res


def segmented_rehearsal(args):
    """The MULTI-rank launch form on this one GPU: a child `bench.py --segmented` (five captured graph segments with the four
    RCCL collectives between them, on a one-rank process group) -> {'ms_per_step', 'collectives'} or None.  No multi-GPU
    node is needed for it, and it bounds what the launch form itself costs against the whole-step graph of the headline
    (link time comes on top on a real node)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--segmented', '--model', args.model, '--batch', str(args.batch),
           '--npoints', str(args.npoints), '--warmup', '5', '--steps', '40', '--no-cpu-baseline', '--no-other-workloads',
           '--caller-steps', '0', '--profile-steps', '3']
    for flag, on in (('--fp16', args.fp16), ('--no-tuned-gemms', args.no_tuned_gemms)):
        if on:
            cmd.append(flag)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',
                                                               'GROUP_RANK', 'ROLE_RANK', 'LOCAL_WORLD_SIZE', 'TORCHELASTIC_RUN_ID')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    try:
        import socket
        sk = socket.socket()
        sk.bind(('127.0.0.1', 0))                    # a free rendezvous port for the one-rank group
        env['MASTER_ADDR'], env['MASTER_PORT'] = '127.0.0.1', str(sk.getsockname()[1])
        sk.close()
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)     # ~20 s when healthy
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
        if r.returncode != 0 or not lines:
            print('bench.py: segmented rehearsal failed (rc %s)' % r.returncode, file=sys.stderr)
            return None
        d = json.loads(lines[-1])
        return {'ms_per_step': d['ms_per_step'], 'launch': d['config']['launch'], 'collectives': d['config']['collectives'],
                'note': 'the multi-rank launch form on ONE rank (RCCL group of size 1): what the five graph segments and four '
                        'collective calls cost against the whole-step graph; no link time'}
    except Exception as e:
        print('bench.py: segmented rehearsal skipped (%s)' % str(e).splitlines()[0], file=sys.stderr)
        return None


def kernel_table(profs, exact=None):
    """{name: [(ev0, ev1, shape), ...]} -> {name: launches, avg / total ms, GB/s, TFLOP/s, bound, frac}."""
    kern = {}
    for name, recs in profs.items():
        ms = [a if b is None else a.elapsed_time(b) for a, b, _ in recs]      # (ms, None, shape): timed by the child process
        ev_total = sum(ms)                 # the ONE timing source every candidate has: used for ranking
        timing = 'HIP events around the C-ABI call'
        needle = DEVICE_KERNEL_OF.get(name)
        if exact and needle:
            # every instantiation behind this op (e.g. PW=2 and PW=4 of knn_pc_kernel<C,...>): durations concatenated
            hit = [d for k, v in exact.items() if needle in k for d in v]
            if hit:
                ms, timing = hit, 'kernel timestamps (torch.profiler / roctracer) in the measured launch mode'
        mdl = kernel_model(name, recs[0][2])
        avg = sum(ms) / len(ms)
        gbps = mdl['bytes'] / avg / 1e6 if avg > 0 else 0.0
        tfl = mdl['flops'] / avg / 1e9 if avg > 0 else 0.0
        cb = mdl['flops'] / max(mdl['bytes'], 1) > FP32_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS
        kern[name] = {'launches': len(ms), 'avg_ms': avg, 'total_ms': sum(ms), 'rank_ms': ev_total,
                      'GBps': gbps, 'TFLOPs': tfl,
                      'bytes': mdl['bytes'], 'flops': mdl['flops'], 'bound': 'mfma' if cb else 'hbm', 'timing': timing,
                      'frac': tfl / FP32_PEAK_TFLOPS if cb else gbps / HBM_PEAK_GBS}
    return kern


def other_workload(model_name, B, N, fp16, dev, steps=10, warmup=3, profile_steps=3, single_pass_too=True):
    """One more BASELINE configuration in the same process: `steps` hipGraph-replayed SUG steps after `warmup`,
    then a few eager steps with kernel events for the dominant hand-written kernel."""
    from sug_amd import ops
    from sug_amd.model import Ptran_transformer as PT
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    keep = (PT.GEMM_DTYPE, PT.PROJ_16BIT)
    try:
        if fp16:
            PT.GEMM_DTYPE, PT.PROJ_16BIT = torch.float16, True
        torch.manual_seed(666)
        model = Net_MDA(model_name).to(dev).train()
        tr = SUGStep(model, lr=1e-3, weight_decay=5e-5, use_graph=True, methods=BENCH_METHODS)
        batch = synth(B, N, 666, dev)
        torch.manual_seed(666)
        for _ in range(max(warmup, 3)):
            tr.step(*batch)
        torch.cuda.synchronize()

        def timed_windows(trainer, n_windows=3):
            """median over `n_windows` windows of `steps` replayed steps (a 10-step window of a 2.7 ms step is 27 ms: one host
            hiccup -- the interpreter's collector, a scheduler pause -- doubles it; the median window is insensitive to one)"""
            gc.collect()
            gc.disable()
            try:
                w = []
                for _ in range(n_windows):
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        out_ = trainer.step(*batch)
                    torch.cuda.synchronize()
                    w.append(1e3 * (time.perf_counter() - t0) / steps)
            finally:
                gc.enable()
            return sorted(w)[len(w) // 2], w, out_

        ms, windows, losses = timed_windows(tr)
        vals = [None if l is None else float(l) for l in losses]
        # the opt-in single-pass step of the same model, same batch: captured and replayed the same way
        sp_ms = None
        if single_pass_too:
            tr1 = SUGStep(model, lr=1e-3, weight_decay=5e-5, use_graph=True, methods=BENCH_METHODS, single_pass=True)
            for _ in range(max(warmup, 3)):
                tr1.step(*batch)
            torch.cuda.synchronize()
            sp_ms, _, _ = timed_windows(tr1)
            tr1.drop_graphs()
            tr1 = None
        # the same workload the way train_dg_single_gpu.py:260-310 calls the API (four separate model(...) calls, replayed
        # per call by sug_amd.call_graphs; SUGStep's own losses / backward / Adam launched eagerly around them)
        caller_ms = None
        try:
            trc = SUGStep(model, lr=1e-3, weight_decay=5e-5, use_graph=False, methods=BENCH_METHODS, pair_domains=False,
                          share_prefix=False)
            if hasattr(model.g, 'share_prefix'):
                model.g.share_prefix = 'auto'
            for _ in range(4):          # step 1 eager (planning), step 2 captures the four calls, then replays
                trc.step(*batch)
            torch.cuda.synchronize()
            caller_ms, _, _ = timed_windows(trc)
            from sug_amd import call_graphs as _cg
            _cg.drop(model)
        except RuntimeError as e:
            print('bench.py: caller form of %s failed: %s' % (model_name, str(e).splitlines()[0][:160]), file=sys.stderr)
        trc = None
        if hasattr(model.g, 'share_prefix'):        # back to the timed trainer's settings for the eager profile steps below
            model.g.share_prefix = tr.share_prefix
        for m_ in tr._split_layers:
            m_.cache_weight_split = True
        tr.use_graph = False
        tr.fused_heads = False
        tr.step(*batch)
        torch.cuda.synchronize()
        keep_prof = (ops.CTX.profile, ops.CTX.profile_only)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr.step(*batch)
        torch.cuda.synchronize()
        e_ms = 1e3 * (time.perf_counter() - t1)
        ops.CTX.profile_only, ops.CTX.profile = {'edgeconv', 'pointmlp', 'knn', 'ptran'}, {}
        for _ in range(profile_steps):
            hold_gpu(2.0 * e_ms)                 # kernels queue up behind the spin kernel: event pairs bracket kernels, not host gaps
            tr.step(*batch)
            torch.cuda.synchronize()
        prof, (ops.CTX.profile, ops.CTX.profile_only) = ops.CTX.profile, keep_prof
        kern = kernel_table(prof)
        out = {'workload': '%s, N=%d, batch=%d per domain, MSA+SDA losses on' % (BACKBONE.get(model_name, model_name), N, B)
                           + (', fp16 transformer linears' if fp16 else ''),
               'dtype': 'f16' if fp16 else 'f32', 'ms_per_step': ms, 'clouds_per_sec': 2 * B / (ms * 1e-3), 'steps': steps,
               'warmup': max(warmup, 3), 'launch': 'hipGraph replay of the whole step', 'losses': vals,
               'timing': 'median of 3 windows of %d steps' % steps, 'window_ms': [round(w, 4) for w in windows]}
        import bench_work
        n_params, adam_elems = param_counts(model)
        out['step_roofline'] = bench_work.step_roofline(model_name, B, N, ms, fp16=fp16, n_params=n_params, adam_elems=adam_elems)
        if sp_ms is not None:
            out['single_pass_ms_per_step'] = sp_ms
            out['single_pass_clouds_per_sec'] = 2 * B / (sp_ms * 1e-3)
        out['unchanged_caller_ms_per_step'] = caller_ms
        if kern:
            dom = max(kern, key=lambda n: kern[n]['rank_ms'])
            kd = kern[dom]
            out['dominant_kernel'] = {'kernel': dom, 'bound': kd['bound'], 'frac': round(kd['frac'], 4),
                                      'avg_launch_ms': round(kd['avg_ms'], 5),
                                      'ms_per_step': round(kd['total_ms'] / profile_steps, 4),
                                      'unit': 'TFLOP/s' if kd['bound'] == 'mfma' else 'GB/s',
                                      'achieved': round(kd['TFLOPs'] if kd['bound'] == 'mfma' else kd['GBps'], 2)}
        return out
    finally:
        PT.GEMM_DTYPE, PT.PROJ_16BIT = keep
        tr = model = None
        gc.collect()
        torch.cuda.empty_cache()


def param_counts(model):
    """(parameter elements of the model, parameter elements summed over the three Adam optimizers of a step)."""
    n_g = sum(p.numel() for p in model.g.parameters())
    n_c = sum(p.numel() for m in (model.c1, model.c2) for p in m.parameters())
    n_a = sum(p.numel() for m in (model.attention_s, model.attention_t) for p in m.parameters())
    n_off = sum(p.numel() for k, p in model.g.named_parameters() if 'pred_offset' in k)
    return n_g + n_c + n_a, (n_g - n_off) + n_c + (n_g + n_a)       # optimizer_g, optimizer_c, optimizer_dis


def cpu_model_string():
    """Model name of the host CPU the baseline ran on (/proc/cpuinfo), with the number of logical CPUs the process may use."""
    name = None
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    name = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return '%s (%d logical CPUs visible)' % (name or 'unknown CPU', avail)


def cpu_baseline(B, N, steps=1, model_name='DGCNN'):
    """The CPU oracle's full SUG step (same algorithm as the reference: materialised [B,N,N]
    distances + topk / sort, k-expanded EdgeConv and attention tensors, 4 encoder passes, 3 MMDs, backward, 3 Adam)
    -> (clouds per second, seconds)."""
    from oracle import ref_cpu as O
    from sug_amd.model.Model import Net_MDA
    # the GPU box exposes all host CPUs but grants a 16-core share per GPU: more threads than
    # that only oversubscribe (measured: 256 threads -> 20x slower than 16)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(int(os.environ.get('SUG_CPU_THREADS', min(avail, 16))))
    net = Net_MDA(model_name)
    p = O.as_params(net.state_dict())
    g = [v for k, v in p.items() if k.startswith('g.') and v.requires_grad]
    opt = [torch.optim.Adam([v for k, v in p.items() if k.startswith('g.') and v.requires_grad and 'pred_offset' not in k], lr=1e-3, weight_decay=5e-5),
           torch.optim.Adam([v for k, v in p.items() if k.startswith(('c1.', 'c2.')) and v.requires_grad], lr=1e-3, weight_decay=5e-5),
           torch.optim.Adam(g + [v for k, v in p.items() if k.startswith('attention') and v.requires_grad], lr=1e-3, weight_decay=5e-5)]
    geo, sem = dict(BENCH_METHODS['GEO_MMD'][0]), dict(BENCH_METHODS['SEM_MMD'][0])

    def run(batch, n):
        data, lab, data_t, lab_t = synth(batch, N, 666, 'cpu')
        for _ in range(n):
            lc, lg, ls = O.sug_losses(p, model_name, data, lab, data_t, lab_t, geo, sem, drop_p=0.4)
            (lc + lg + ls).backward()
            opt[2].step(); opt[0].step(); opt[1].step()
            for o in opt:
                o.zero_grad()

    if model_name != 'PTran':
        run(2, 1)                               # untimed: thread pool / allocator warm-up (PTran: a step is ~20 s, no warm-up)
    t0 = time.perf_counter()
    run(B, steps)
    dt = time.perf_counter() - t0
    return 2 * B * steps / dt, dt


def cpu_record(model_name, B, N, steps, note=''):
    cps, secs = cpu_baseline(B, N, steps, model_name)
    return {'value': cps, 'unit': 'point-clouds/sec', 'cores': torch.get_num_threads(), 'kind': 'port', 'cpu': cpu_model_string(),
            'sample': '%d full SUG step%s (oracle/ref_cpu.py, fp32 = the reference arithmetic), %s N=%d, batch %d per domain '
                      '(%d clouds per step), %.1f s on %d torch threads%s'
                      % (steps, '' if steps == 1 else 's', model_name, N, B, 2 * B, secs, torch.get_num_threads(), note)}


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per
    GPU (python -m torch.distributed.run, rendezvous on 127.0.0.1), BEFORE this process touches
    the GPU; relay the ranks' output (rank 0 prints the JSON line) and return the child's exit
    code.  The reference's DDP seam is train_dg.py:59-66, :216-217 (torch.distributed.launch)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC for RCCL on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)       # ~0.7 s timed: a one-off host hiccup of tens of ms
    ap.add_argument('--warmup', type=int, default=10)       # weighs little (20 steps gave rare 8-9 ms outliers)
    ap.add_argument('--batch', type=int, default=32, help='clouds per domain per GPU')
    ap.add_argument('--npoints', type=int, default=1024)
    ap.add_argument('--model', default='DGCNN')
    ap.add_argument('--cpu-batch', type=int, default=32, help='per-domain batch of the CPU baseline sample (default: the GPU workload)')
    ap.add_argument('--cpu-steps', type=int, default=3)
    ap.add_argument('--caller-steps', type=int, default=10,
                    help='extra timed steps in the unchanged-caller form (four separate model(...) calls, no sharing); 0 = skip')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='skip the extra configurations timed after the headline region (BASELINE configs 1, 3, 5 -> config.other_workloads)')
    ap.add_argument('--fp16', action='store_true',
                    help='Point Transformer only (BASELINE config 5): k-expanded attention tensors and their 512x512 linears in '
                         'fp16 (MFMA, fp32 accumulation); default fp32 = the reference arithmetic')
    ap.add_argument('--eager', action='store_true',
                    help='launch every kernel of the timed steps from Python instead of replaying a captured hipGraph '
                         '(the default on one GPU: an eager step is host-bound on this path -- ~450 launches in ~6.5 ms -- '
                         'and a single host hiccup inside a 20-step window moves the result by several per cent)')
    ap.add_argument('--segmented', action='store_true',
                    help='one GPU: run the multi-rank launch form (5 graph segments + 4 RCCL collectives on a 1-rank group) -- a '
                         'rehearsal of what --gpus N > 1 executes')
    ap.add_argument('--rendezvous-only', action='store_true',
                    help='start the ranks, form the process group, all-reduce one CPU/GPU scalar, print a JSON line on rank 0 '
                         'and exit BEFORE any model / kernel call (launch + rendezvous + relay check; works without a GPU '
                         'with SUG_BENCH_BACKEND=gloo)')
    ap.add_argument('--plain', action='store_true', help='only warm-up + timed steps (for rocprofv3 kernel traces): no extra passes')
    ap.add_argument('--timestamps-child', default=None, metavar='OUT.json',
                    help='(internal) warm up, replay --profile-steps captured steps under torch.profiler, write the device '
                         'kernel durations to OUT.json and exit: bench.py runs this pass in a CHILD process because roctracer '
                         'under hipGraph replay occasionally crashes the process (segmentation fault in ~1 of 12 runs)')
    ap.add_argument('--child-events', action='store_true', help='(internal, with --timestamps-child) also HIP-event timings of eager steps')
    ap.add_argument('--eager-steps', type=int, default=10,
                    help='graph mode: extra eager steps after the timed region (per-kernel HIP-event timings, eager ms/step)')
    ap.add_argument('--profile-steps', type=int, default=5,
                    help='eager steps with HIP events around the hand-written kernels (roofline / kernels), each queued behind a spin kernel')
    ap.add_argument('--no-share-prefix', action='store_true',
                    help='recompute the kNN+conv1/conv2 stage in the node passes instead of sharing it (identical results)')
    ap.add_argument('--no-tuned-gemms', action='store_true',
                    help='library GEMMs by the default heuristic instead of the recorded TunableOp choices (sug_amd/tuning)')
    ap.add_argument('--single-pass', action='store_true',
                    help='time the opt-in single-pass dual-output step (SURVEY 8 f2) instead of the exact two-pass step; the line '
                         'then says so in config.workload (not the headline configuration)')
    ap.add_argument('--no-pair', action='store_true',
                    help='separate encoder passes for the source and the target batch (identical results)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        print('bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus=%d'
              % (args.gpus, world, world), file=sys.stderr)
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.rendezvous_only:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29534')
        backend = os.environ.get('SUG_BENCH_BACKEND', 'nccl')
        have_gpu = backend == 'nccl'
        if have_gpu:
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
            dist.init_process_group('nccl', rank=rank, world_size=world,
                                    device_id=torch.device('cuda', local % max(torch.cuda.device_count(), 1)))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], device='cuda' if have_gpu else 'cpu')
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({'rendezvous': 'ok', 'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                              'rank_sum': float(t.item()), 'expected_rank_sum': world * (world + 1) / 2.0}))
        dist.destroy_process_group()
        return
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # SUG_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks
        # (ranks then share devices; RCCL refuses that).  The measured configuration is nccl = RCCL.
        backend = os.environ.get('SUG_BENCH_BACKEND', 'nccl')
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device('cuda', local)
    if world == 1 and args.segmented:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        torch.cuda.set_device(local)
        dist.init_process_group(os.environ.get('SUG_BENCH_BACKEND', 'nccl'), rank=0, world_size=1, device_id=dev)

    from sug_amd import ops, _lib
    from sug_amd.model.Model import Net_MDA
    from sug_amd.train_step import SUGStep
    _lib.lib()                                          # fail loudly if the HIP library is missing
    tuned = False
    if not args.no_tuned_gemms:
        from sug_amd.tuning import enable_tuned_gemms
        tuned = enable_tuned_gemms()

    if args.fp16:
        from sug_amd.model import Ptran_transformer as PT
        PT.GEMM_DTYPE = torch.float16
        PT.PROJ_16BIT = True               # every 512-wide linear of the transformer blocks on the fp16 MFMA path
    torch.manual_seed(666)                              # train_dg_single_gpu.py:65
    model = Net_MDA(args.model).to(dev).train()
    if world > 1:                                       # same initial weights on every rank
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)
    want_graph = not args.eager

    def make_trainer(use_graph):
        return SUGStep(model, lr=1e-3, weight_decay=5e-5, share_prefix=not args.no_share_prefix, use_graph=use_graph,
                       pair_domains=not args.no_pair, methods=BENCH_METHODS, force_segmented=args.segmented,
                       single_pass=args.single_pass)

    trainer = make_trainer(want_graph)
    B, N = args.batch, args.npoints
    data, lab, data_t, lab_t = synth(B, N, 666 + rank, dev)
    torch.manual_seed(666 + rank)                       # FPS start draws, per rank (train_dg.py:78)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Launch mode.  Default on one GPU: the step (4 encoder passes, losses, backward, 3 Adam updates) is captured
    # once into a hipGraph and the timed steps replay it -- same kernels, same arithmetic, one host call per step.
    # Kernel events cannot be captured on ROCm, so the per-kernel HIP-event timings (roofline, `kernels`) come from
    # eager steps of the same trainer right after the timed region.  --eager: everything launched from Python; the
    # kernels of the dominant family are then timed inside the timed region itself.
    timed_family = {'DGCNN': {'knn'}, 'Pointnet': {'pointmlp'}, 'Pointnet2': {'pointmlp'}}.get(args.model, {'knn'})
    all_families = {'edgeconv', 'pointmlp', 'knn'}
    ops.CTX.profile_only = set(timed_family)
    try:
        for i in range(max(args.warmup, 3 if trainer.use_graph else 1)):
            trainer.step(data, lab, data_t, lab_t)
        sync()
    except RuntimeError as e:                              # capture refused on this stack: fall back to eager launches
        if not trainer.use_graph:
            raise
        print('bench.py: hipGraph capture failed (%s); falling back to eager launches' % str(e).splitlines()[0], file=sys.stderr)
        trainer = make_trainer(False)
        for i in range(max(args.warmup, 1)):
            trainer.step(data, lab, data_t, lab_t)
        sync()
    graph_mode = trainer.use_graph
    if args.timestamps_child:
        # Stage A (only when asked: the multi-rank parent has no eager pass of its own): HIP-event readings of every
        # hand-written family on eager steps queued behind a spin kernel, written out BEFORE stage B can crash.
        res = {'exact': None, 'events': None}
        if args.child_events:
            trainer.use_graph = False
            for _ in range(2):
                trainer.step(data, lab, data_t, lab_t)
            sync()
            t1 = time.perf_counter()
            for _ in range(3):
                trainer.step(data, lab, data_t, lab_t)
            sync()
            ems = 1e3 * (time.perf_counter() - t1) / 3
            ops.CTX.profile_only, ops.CTX.profile = set(all_families), {}
            for _ in range(max(args.profile_steps, 1)):
                hold_gpu(2.5 * ems)
                trainer.step(data, lab, data_t, lab_t)
                sync()
            res['events'] = {k: [[a.elapsed_time(b), shp] for a, b, shp in v] for k, v in ops.CTX.profile.items()}
            ops.CTX.profile, ops.CTX.profile_only = None, set(timed_family)
            trainer.use_graph = graph_mode
            with open(args.timestamps_child, 'w') as f:
                json.dump(res, f)
        # Stage B: the device kernels' own timestamps over a few replays of the captured step
        if graph_mode:
            def _replays():
                for _ in range(max(args.profile_steps, 1)):
                    trainer.step(data, lab, data_t, lab_t)
                sync()
            res['exact'] = device_kernel_durations(_replays)
        with open(args.timestamps_child, 'w') as f:
            json.dump(res, f)
        return
    # The interpreter's full (generation-2) collection walks every object torch has created so
    # far: a ~70 ms pause that otherwise lands somewhere in the first 20 steps.  Collect now and
    # freeze the survivors (what a long-running training loop reaches after its first minutes).
    gc.collect()
    gc.freeze()
    if not graph_mode:
        ops.CTX.profile = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = trainer.step(data, lab, data_t, lab_t)
    sync()
    dt = time.perf_counter() - t0
    loss_vals = [None if l is None else float(l) for l in losses]
    prof, ops.CTX.profile = ({} if graph_mode else ops.CTX.profile), None
    eager_ms = None
    exact_ms = None
    graph_kernel_ms = graph_launches = None
    child_events = None
    if graph_mode and not args.plain:
        # kernel durations INSIDE the measured launch mode: the device kernels' own timestamps (torch.profiler = roctracer,
        # what `rocprofv3 --kernel-trace` reads) over a few more replays of the captured step
        # -- in a CHILD process (rank 0 only, a one-rank replica of the same step): roctracer under hipGraph replay crashed
        # the process now and then (segmentation fault after the profiler started, ~1 of 12 runs on ROCm 7.2), and a crash
        # here must not take the measurement with it; if the child fails, the HIP-event readings below stand
        child = kernel_timestamps_in_child(args, want_events=(world > 1 or args.segmented)) if rank == 0 else {'exact': None, 'events': None}
        exact_ms = child['exact']
        if exact_ms is not None and not any('knn_pc_kernel' in k for k in exact_ms):
            exact_ms = None                                      # the tracer did not see inside the graph launches
        child_events = child['events']
        if exact_ms is not None:
            # step-level accounting in the measured launch mode: device time of ALL kernels of a replayed step (their own
            # timestamps) against the step's wall time -- the difference is launch gaps / dependency bubbles inside the graph
            nrep = max(args.profile_steps, 1)
            graph_kernel_ms = sum(sum(v) for v in exact_ms.values()) / nrep
            graph_launches = sum(len(v) for v in exact_ms.values()) / nrep
    collectives = None
    if (world > 1 or args.segmented) and graph_mode and getattr(trainer, 'segmented', False):
        # per-collective milliseconds (events on the compute stream around the eager RCCL calls between the graph replays)
        # over a few more steps, outside the timed region; bucket 1 additionally alone (synchronous, nothing beside it)
        trainer.collective_events = {}
        for _ in range(max(args.profile_steps, 1)):
            trainer.step(data, lab, data_t, lab_t)
        sync()
        collectives = trainer.collective_summary()
        trainer.collective_events = None
        flat1 = None
        for st_ in (trainer._graphs or {}).values():
            flat1 = ((st_.get('S') or {}).get('static') or {}).get('flat1', flat1)
        if flat1 is not None:
            scratch = flat1.clone()
            dist.all_reduce(scratch)
            sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dist.all_reduce(scratch)
            e1.record()
            torch.cuda.synchronize()
            collectives['ms']['bucket1_alone'] = round(e0.elapsed_time(e1) / 5, 4)
        collectives.update(backend=dist.get_backend(), world_size=dist.get_world_size(),
                           note='ms per step, mean over %d steps after the timed region; events on the compute stream around '
                                'each eager collective; bucket1_exposed = wait for bucket 1 after the encoder backward it '
                                'runs under; bucket1_alone = the same message all-reduced with nothing beside it'
                                % max(args.profile_steps, 1))
    if args.plain or ((world > 1 or args.segmented) and graph_mode):
        # multi-rank: nothing but the timed region in this process; the per-kernel timings of the hand-written families
        # come from rank 0's one-rank child (the kernels are the same: the per-GPU share is what every rank runs)
        extra_prof = {}
        if not args.plain and graph_mode and rank == 0 and child_events:
            extra_prof = {k: [(ms, None, {kk: vv for kk, vv in shp.items()}) for ms, shp in v] for k, v in child_events.items()}
    elif graph_mode:
        # eager steps of the same trainer: per-kernel event timings of every hand-written family + eager ms/step
        trainer.use_graph = False
        for _ in range(2):
            trainer.step(data, lab, data_t, lab_t)
        sync()
        gc.collect()
        gc.disable()
        try:
            t1 = time.perf_counter()
            for _ in range(max(args.eager_steps, 1)):
                trainer.step(data, lab, data_t, lab_t)
            sync()
            eager_ms = 1e3 * (time.perf_counter() - t1) / max(args.eager_steps, 1)
        finally:
            gc.enable()
        # per-kernel event timings: a few more eager steps, each queued BEHIND a spin kernel that holds the GPU while the
        # host enqueues the whole step -- the kernels then run back to back and an event pair brackets the kernel alone
        # (launched live, an event pair also spans the host's gap to the next launch whenever the host is the slower side)
        ops.CTX.profile_only, ops.CTX.profile = set(all_families), {}
        for _ in range(max(args.profile_steps, 1)):
            hold_gpu(2.5 * eager_ms)
            trainer.step(data, lab, data_t, lab_t)
            sync()
        extra_prof, ops.CTX.profile, ops.CTX.profile_only = ops.CTX.profile, None, set(timed_family)
        # (no in-process profiler pass as a fall-back: if the child failed, the HIP-event readings of the queued steps stand)
    else:
        # the other hand-written layer kernels (EdgeConv layer calls, per-point MLP + max): a few extra steps
        # outside the timed region, for the `kernels` table only
        ops.CTX.profile_only, ops.CTX.profile = all_families - timed_family, {}
        for _ in range(5):
            trainer.step(data, lab, data_t, lab_t)
        sync()
        extra_prof, ops.CTX.profile, ops.CTX.profile_only = ops.CTX.profile, None, set(timed_family)
    # The same workload the way train_dg_single_gpu.py:260-310 calls the API: four separate model(...)
    # calls per step, nothing shared between them (the headline uses the exact restructurings of
    # paired domains + shared prefix).
    caller_ms = None
    if args.caller_steps > 0 and not args.plain and world == 1 and not args.segmented and (trainer.pair_domains or trainer.share_prefix):
        keep = (trainer.pair_domains, trainer.share_prefix)
        trainer.pair_domains = trainer.share_prefix = False
        if hasattr(model.g, 'share_prefix'):
            model.g.share_prefix = 'auto'          # what a plain Net_MDA(...) does by itself (liveness-checked reuse)
        for m_ in trainer._split_layers:
            m_.cache_weight_split = False
        for _ in range(3):
            trainer.step(data, lab, data_t, lab_t)
        sync()
        # (an eager step creates ~10^4 Python objects; a generation-2 collection of the interpreter inside these few steps
        # reads as +3 ms per step -- collected before, held off during the window, as for the headline region)
        gc.collect()
        gc.disable()
        try:
            t1 = time.perf_counter()
            for _ in range(args.caller_steps):
                trainer.step(data, lab, data_t, lab_t)
            sync()
            caller_ms = 1e3 * (time.perf_counter() - t1) / args.caller_steps
        finally:
            gc.enable()
        trainer.pair_domains, trainer.share_prefix = keep
    # The opt-in single-pass step (SUGStep(single_pass=True)): one encoder evaluation per domain feeds heads and
    # attention layers.  Reported BESIDE the headline, never as it: it differs from the reference's step in one documented
    # way (one FPS start draw per sampling stage instead of two).
    single_ms = None
    if args.caller_steps > 0 and not args.plain and world == 1 and not args.segmented and graph_mode and not args.single_pass:
        tr1 = SUGStep(model, lr=1e-3, weight_decay=5e-5, share_prefix=not args.no_share_prefix, use_graph=True,
                      pair_domains=not args.no_pair, methods=BENCH_METHODS, single_pass=True)
        for _ in range(4):
            tr1.step(data, lab, data_t, lab_t)
        sync()
        t1 = time.perf_counter()
        for _ in range(max(args.caller_steps, 20)):
            tr1.step(data, lab, data_t, lab_t)
        sync()
        single_ms = 1e3 * (time.perf_counter() - t1) / max(args.caller_steps, 20)
        tr1.drop_graphs()
        tr1 = None
    n_params, adam_elems = param_counts(model)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    share_prefix_on, pair_domains_on = trainer.share_prefix, trainer.pair_domains
    if rank == 0:
        clouds = world * 2 * B * args.steps
        value = clouds / dt
        # ---- per-kernel live timings (events on the launch stream) -> roofline of the dominant one
        timed_names = set(prof)
        allprof = dict(extra_prof)
        allprof.update(prof)
        kern = kernel_table(allprof, exact_ms)
        roofline = None
        if kern:
            fam = [n for n in kern if n.startswith(tuple(timed_family))]
            cands = timed_names or fam or set(kern)
            # strictly the largest summed time, ranked on ONE timing source for all candidates (the event pairs of the
            # queued eager steps; the reported durations of the single-launch ops are the kernel timestamps)
            dom = max(cands, key=lambda n: kern[n]['rank_ms'])
            kd = kern[dom]
            if kd['bound'] == 'mfma':
                roofline = {'kernel': dom, 'bound': 'mfma', 'achieved': kd['TFLOPs'], 'peak': FP32_PEAK_TFLOPS,
                            'unit': 'TFLOP/s', 'frac': kd['TFLOPs'] / FP32_PEAK_TFLOPS, 'traffic': None,
                            'note': 'fp32 kernel; peak = fp32 vector/MFMA rate; HBM GB/s at algorithmic bytes: %.1f' % kd['GBps']}
            else:
                roofline = {'kernel': dom, 'bound': 'hbm', 'achieved': kd['GBps'], 'peak': HBM_PEAK_GBS,
                            'unit': 'GB/s', 'frac': kd['GBps'] / HBM_PEAK_GBS, 'traffic': None}
            roofline['avg_launch_ms'] = kd['avg_ms']
            roofline['timing'] = kd['timing']
            # time-weighted fraction over the whole kernel family of the named kernel (e.g. kNN at C = 3, 64, 128)
            famname = dom.split('_')[0]
            members = [n for n in kern if n.split('_')[0] == famname and kern[n]['bound'] == kd['bound']]
            tt = sum(kern[n]['total_ms'] for n in members)
            if tt > 0:
                work = sum((kern[n]['flops'] if kd['bound'] == 'mfma' else kern[n]['bytes']) * kern[n]['launches'] for n in members)
                rate = work / tt / (1e9 if kd['bound'] == 'mfma' else 1e6)
                roofline['family'] = {'kernels': sorted(members), 'achieved': rate,
                                      'frac': rate / (FP32_PEAK_TFLOPS if kd['bound'] == 'mfma' else HBM_PEAK_GBS)}
            # HBM bytes per launch from the committed PMC passes of THIS kernel version (file name = source hash;
            # tools/pmc_traffic.py); not collectable inside this process
            rec = pmc_traffic(dom, allprof[dom][0][2])
            if rec:
                roofline['traffic'] = rec['traffic_bytes']
                roofline['traffic_note'] = 'bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, profiles/pmc_traffic_%s.json; algorithmic %d' % (kernel_source_hash(dom), kd['bytes'])
            else:
                roofline['traffic_note'] = 'no PMC passes on file for this kernel version (profiles/pmc_traffic_%s.json)' % kernel_source_hash(dom)
        if roofline is not None:
            try:
                roofline['hbm_copy_gbps_measured'] = round(measured_copy_gbps(dev), 1)
                roofline['hbm_note'] = 'device-to-device copy of 1 GiB on this box (read + write) beside the %.0f GB/s vendor peak the fractions use' % HBM_PEAK_GBS
            except RuntimeError:
                pass
            import bench_work
            roofline['step'] = bench_work.step_roofline(args.model, B, N, 1e3 * dt / args.steps, single_pass=args.single_pass,
                                                        fp16=args.fp16 and args.model == 'PTran', n_params=n_params,
                                                        adam_elems=adam_elems)
            if single_ms is not None:
                roofline['step_single_pass'] = {k: v for k, v in bench_work.step_roofline(
                    args.model, B, N, single_ms, single_pass=True, n_params=n_params, adam_elems=adam_elems).items()
                    if k in ('gflop', 'gbytes', 'ms', 'achieved_tflops', 'achieved_gbps', 'frac_mfma', 'frac_hbm')}
        others = None
        if world == 1 and not args.plain and not args.segmented and not args.no_other_workloads and args.model == 'DGCNN':
            # the other BASELINE configurations, driver-timed in the same line (configs 1, 3, 5 and N = 2048)
            trainer = model = None
            gc.collect()
            torch.cuda.empty_cache()
            others = []
            for nm, b_, n_, f16 in (('Pointnet', 8, 1024, False), ('Pointnet2', 64, 2048, False), ('PTran', 16, 2048, True)):
                try:
                    others.append(other_workload(nm, b_, n_, f16, dev))
                except RuntimeError as e:
                    others.append({'workload': '%s N=%d batch=%d' % (nm, n_, b_), 'error': str(e).splitlines()[0][:200]})
        seg1 = None
        if world == 1 and not args.plain and not args.segmented and not args.no_other_workloads and args.model == 'DGCNN' \
                and not args.single_pass:
            seg1 = segmented_rehearsal(args)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_record(args.model, args.cpu_batch, N, args.cpu_steps)
            if others:
                # the CPU oracle beside every single-GPU configuration, on bounded samples of the same
                # workloads: config 1 at its full batch, configs 3 / 5 at a reduced batch (the oracle's step time is linear
                # in the batch: every cloud is independent through the encoder)
                for rec, (nm, b_, n_, st_) in zip(others, (('Pointnet', 8, 1024, 3), ('Pointnet2', 8, 2048, 2), ('PTran', 2, 2048, 1))):
                    if 'error' in rec:
                        continue
                    try:
                        note = '' if nm == 'Pointnet' else '; sample batch reduced from the GPU workload\'s (clouds/s is batch-independent on the CPU)'
                        rec['cpu_baseline'] = cpu_record(nm, b_, n_, st_, note)
                        rec['gpu_over_cpu'] = round(rec['clouds_per_sec'] / rec['cpu_baseline']['value'], 1)
                    except Exception as e:                       # a baseline must never take the measurement with it
                        rec['cpu_baseline'] = {'error': str(e).splitlines()[0][:200]}
        out = {'metric': 'point-clouds/sec (train step, N=%d)' % N, 'value': value, 'unit': 'point-clouds/sec',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f16' if (args.fp16 and args.model == 'PTran') else 'f32',
               'data': 'synthetic',
               'config': {'workload': '%s, N=%d, batch=%d per domain per GPU, MSA+SDA losses on '
                                      '(%s, 3 soft-MMD, backward, 3 Adam)' % (BACKBONE.get(args.model, args.model), N, B,
                                      'OPT-IN SINGLE PASS: one dual-output forward per domain' if args.single_pass else '2 sem + 2 node forwards'),
                          'global_batch': world * B, 'global_batch_note': 'clouds per domain x world (the reference\'s batch_size); '
                                                                           'a step runs clouds_per_step = 2 x that (source + target)',
                          'parallelism': 'dp%d' % world, 'collectives': collectives,
                          'launch': ('segmented hipGraph (5 captured segments around the all-gather, the 9-double all-reduce and the two '
                                     'gradient-bucket all-reduces, bucket 1 in flight under the encoder backward)' if (graph_mode and (world > 1 or args.segmented)) else
                                     'hipGraph replay of the whole step' if graph_mode else 'eager'),
                          'eager_ms_per_step': eager_ms,
                          'graph_kernel_ms_per_step': graph_kernel_ms, 'graph_launches_per_step': graph_launches,
                          'share_prefix': share_prefix_on, 'pair_domains': pair_domains_on,
                          'tuned_gemms': tuned, 'geo_weights': 'mean2one', 'sem_weights': 'mean2one',
                          'weight_gradients': ('outputs below 128x128: own split-K MFMA kernels (sug_linear_dw); 128x128 and wider: batched '
                                               'library GEMMs over row chunks + own ordered fp64 fold (no library split-K, hence no '
                                               'memset nodes, inside the captured graph)'
                                               if graph_mode else 'own kernels, tuned library GEMMs for the listed shapes'),
                          'unchanged_caller_ms_per_step': caller_ms,
                          'single_pass_ms_per_step': single_ms,
                          'single_pass_clouds_per_sec': None if single_ms is None else world * 2 * B / (single_ms * 1e-3),
                          'single_pass_note': 'opt-in SUGStep(single_pass=True), SURVEY 8 f2: one encoder evaluation per domain feeds '
                                              'heads and attention layers; same losses / gradients / BatchNorm buffers as the two-pass step whose '
                                              'node pass draws the semantic pass\'s FPS starts (one draw per stage instead of two); NOT the headline',
                          'clouds_per_step': world * 2 * B,
                          'other_workloads': others,
                          'segmented_one_rank_rccl': seg1,
                          **({'fp16_linears': 'k-expanded and per-point 512-wide linears of the transformer blocks'}
                             if (args.fp16 and args.model == 'PTran') else {})},
               'roofline': roofline, 'cpu_baseline': cpu, 'losses': loss_vals,
               'kernels': {k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != 'flops'} for k, v in kern.items()}}
        # the driver's parser keeps scalars and drops nested objects -- the per-config and step-level numbers
        # again as flat keys beside the objects they summarise
        cfg = out['config']
        for tag, rec in zip(('c1', 'c3', 'c5'), others or ()):
            if rec and 'error' not in rec:
                cfg['%s_ms_per_step' % tag] = rec.get('ms_per_step')
                cfg['%s_clouds_per_sec' % tag] = rec.get('clouds_per_sec')
                cfg['%s_single_pass_ms_per_step' % tag] = rec.get('single_pass_ms_per_step')
                cfg['%s_unchanged_caller_ms_per_step' % tag] = rec.get('unchanged_caller_ms_per_step')
                sr = rec.get('step_roofline') or {}
                cfg['%s_step_frac_mfma' % tag] = sr.get('frac_mfma')
                cfg['%s_step_frac_hbm' % tag] = sr.get('frac_hbm')
                cb = rec.get('cpu_baseline') or {}
                cfg['%s_cpu_clouds_per_sec' % tag] = cb.get('value')
        if seg1:
            cfg['segmented_one_rank_rccl_ms'] = seg1.get('ms_per_step')
        if roofline is not None:
            st_ = roofline.get('step') or {}
            roofline['step_frac_mfma'] = st_.get('frac_mfma')
            roofline['step_frac_hbm'] = st_.get('frac_hbm')
            roofline['step_gflop'] = st_.get('gflop')
            roofline['step_gbytes'] = st_.get('gbytes')
            fam_ = roofline.get('family') or {}
            roofline['family_frac'] = fam_.get('frac')
            for kn, tag in (('knn_C3', 'knn_C3'), ('knn_C64', 'knn_C64'), ('knn_C128', 'knn_C128')):
                if kn in kern:
                    roofline['%s_us' % tag] = round(1e3 * kern[kn]['avg_ms'], 2)
                    roofline['%s_frac' % tag] = kern[kn].get('frac')
        print(json.dumps(out))
    if world > 1 or args.segmented:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""GPU twin of tests/test_dist_gloo.py: a 2-rank SUGStep on HIP tensors (one process per rank, batch
sharded, bucketed gradient all-reduce overlapped with backward, packed differentiable all-gather for
the global-batch MMD) against single-process runs of the same shards.

  * backend 'nccl' (= RCCL over xGMI): needs >= 2 devices, skipped otherwise (the driver's 8-GPU node);
  * backend 'gloo' with both ranks on cuda:0: the same code path minus RCCL, runs on the 1-GPU box.

Checked per rank: loss_cls equals a single-process step on that rank's shard (local BatchNorm, as the
reference's DDP, train_dg.py:216-217); the MMD terms are identical on all ranks and equal the MMD of
the gathered features; after the optimizer step every rank holds the same parameters."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed):
    from oracle import ref_cpu as O
    from sug_amd.model.Model import Net_MDA
    net = Net_MDA('DGCNN')
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    return net


def _data(B, N, seed):
    from oracle import ref_cpu as O
    g = torch.Generator().manual_seed(seed)
    data, data_t = O.synth_clouds(B, N, g), O.synth_clouds(B, N, g)
    lab, lab_t = torch.randint(0, 10, (B,), generator=g), torch.randint(0, 10, (B,), generator=g)
    return data, lab, data_t, lab_t


def _worker(rank, world, port, backend, q, mode='eager', steps=2, B=4, single_pass=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    dev = torch.device('cuda', rank if backend == 'nccl' else 0)
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from sug_amd.train_step import SUGStep
    N = 1024                                    # B = global batch per domain; each rank owns B / world clouds
    data, lab, data_t, lab_t = _data(B, N, 5)
    lo, hi = rank * B // world, (rank + 1) * B // world
    shard = [t[lo:hi].to(dev) for t in (data, lab, data_t, lab_t)]
    net = _make(3).to(dev).train()
    if mode == 'segmented_eager':
        os.environ['SUG_SEGMENTED_EAGER'] = '1'    # the segment / collective sequence of graph mode, uncaptured
    tr = SUGStep(net, global_mmd=True, use_graph=(mode == 'segmented_graph'), single_pass=single_pass)
    torch.manual_seed(100 + rank)                  # FPS start draws, per rank (train_dg.py:78)
    out = []
    for _ in range(steps):
        out.append([float(v) for v in tr.step(*shard)])
    if mode == 'segmented_graph':
        assert tr.segmented and len(tr._graphs) == 1 and len(next(iter(tr._graphs.values()))['graphs']) == 5
    chk = torch.stack([p.detach().double().sum() for p in net.parameters()]).cpu()
    q.put((rank, out, chk.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(backend, mode='eager', steps=2, world=2, B=4, single_pass=False):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q, mode, steps, B, single_pass)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _check(res):
    from sug_amd.train_step import SUGStep
    (r0, out0, chk0), (r1, out1, chk1) = res
    # the MMD terms are computed on the gathered global batch: identical on both ranks
    for s in range(2):
        assert abs(out0[s][1] - out1[s][1]) <= 1e-6 * max(1.0, abs(out0[s][1])), (out0, out1)
        assert abs(out0[s][2] - out1[s][2]) <= 1e-6 * max(1.0, abs(out0[s][2])), (out0, out1)
    # averaged gradients + identical optimizers: the replicas stay in step
    assert abs(chk0 - chk1).max() <= 1e-6 * max(1.0, abs(chk0).max()), 'parameters diverged between ranks'
    # per-rank classification loss = a single-process step on that shard (first step: same weights)
    B, N = 4, 1024
    data, lab, data_t, lab_t = _data(B, N, 5)
    for rank, out in ((0, out0), (1, out1)):
        lo, hi = rank * B // 2, (rank + 1) * B // 2
        net = _make(3).cuda().train()
        tr = SUGStep(net, global_mmd=False)
        torch.manual_seed(100 + rank)
        lc, _, _ = tr.losses(*[t[lo:hi].cuda() for t in (data, lab, data_t, lab_t)])
        assert abs(float(lc) - out[0][0]) <= 1e-4 * max(1.0, abs(out[0][0])), (rank, float(lc), out)
    # and the first step's global MMD = MMD of the two shards' features put together
    feats = []
    for rank in range(2):
        lo, hi = rank * B // 2, (rank + 1) * B // 2
        net = _make(3).cuda().train()
        torch.manual_seed(100 + rank)
        d, l, dt, lt = [t[lo:hi].cuda() for t in (data, lab, data_t, lab_t)]
        pair = torch.cat((d, dt))
        with torch.no_grad():
            (ps1, ps2, ss1, ss2), (pt1, pt2, st1, st2) = net.forward_pair(pair)
            fs, ft = net.forward_pair(pair, node_adaptation=True)
        feats.append((l, lt, fs, ft))
    from sug_amd.model import mmd
    from sug_amd.train_step import GEO_MMD
    cat = lambda i: torch.cat([f[i] for f in feats])
    geo = float(mmd.mmd_cal(cat(0), cat(2), cat(1), cat(3), GEO_MMD))
    assert abs(geo - out0[0][1]) <= 1e-4 * max(1.0, abs(geo)), (geo, out0)


def test_two_rank_step_gloo_on_one_gpu():
    _check(_run('gloo'))


def _segmented(backend):
    """Graph mode on two ranks = five captured segments around the four collectives (SUGStep._segments): the
    uncaptured segment sequence and the captured one follow the eager multi-rank step (bucketed all-reduce from
    autograd hooks) -- first step to rounding, later steps within the trajectory noise of two eager runs -- and the
    replicas stay identical."""
    runs = {m: _run(backend, m, steps=5) for m in ('eager', 'segmented_eager', 'segmented_graph')}
    for m, res in runs.items():
        (_, out0, chk0), (_, out1, chk1) = res
        assert (chk0 == chk1).all(), '%s: parameters diverged between ranks' % m      # same averaged gradients, same updates
        for s in range(5):
            for t in (1, 2):                      # the global MMD terms are identical on both ranks
                assert out0[s][t] == out1[s][t], (m, out0, out1)
    # the same arithmetic in two launch forms: bit for bit, all five steps, both ranks (every sum has one fixed order, round 4)
    for r in range(2):
        assert runs['segmented_eager'][r][1] == runs['segmented_graph'][r][1], (r, runs['segmented_eager'][r][1], runs['segmented_graph'][r][1])
    # against the eager multi-rank step (another summation FORM: bucketed all-reduce from autograd hooks, MMD on gathered
    # rows instead of row blocks): first step to rounding, then the two forms' trajectories drift apart as two
    # different-but-valid roundings do
    ref = runs['eager']
    for m in ('segmented_eager', 'segmented_graph'):
        for r in range(2):
            for a, b in zip(ref[r][1][0], runs[m][r][1][0]):
                assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (m, ref[r][1], runs[m][r][1])
            for sa, sb in zip(ref[r][1], runs[m][r][1]):
                for a, b in zip(sa, sb):
                    assert abs(a - b) <= 2e-2 * max(1.0, abs(a)), (m, ref[r][1], runs[m][r][1])
    print({m: res[0][1] for m, res in runs.items()})


def test_two_rank_segmented_graph_gloo_on_one_gpu():
    _segmented('gloo')


def test_two_rank_segmented_graph_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 HIP devices (RCCL over xGMI)')
    _segmented('nccl')


def _worker_one_rank_rccl(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    from sug_amd.train_step import SUGStep
    shard = [t.to(dev) for t in _data(4, 1024, 5)]
    out = {}
    for mode in ('whole', 'segmented'):
        if mode == 'segmented':
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        tr = SUGStep(_make(3).to(dev).train(), use_graph=True, force_segmented=(mode == 'segmented'))
        assert tr.segmented == (mode == 'segmented')
        torch.manual_seed(100)
        out[mode] = [[float(v) for v in tr.step(*shard)] for _ in range(5)]
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_segmented_graph_over_rccl_on_one_rank():
    """The multi-rank launch form -- five graph replays with RCCL collectives between them -- on a one-rank RCCL
    group: this is the only way the 1-GPU box can execute RCCL calls between hipGraph replays (all-gather of values,
    fp64 all-reduce, async all-reduce under a replay, wait).  Losses follow the whole-step graph of the same trainer."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker_one_rank_rccl, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    print(out)
    for a, b in zip(out['whole'][0], out['segmented'][0]):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), out
    for sa, sb in zip(out['whole'], out['segmented']):
        for a, b in zip(sa, sb):
            assert abs(a - b) <= 2e-2 * max(1.0, abs(a)), out


def test_two_rank_step_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 HIP devices (RCCL over xGMI)')
    _check(_run('nccl'))


def test_bench_two_ranks_end_to_end_gloo_on_one_gpu():
    """`python bench.py --gpus 2` the way the driver starts a multi-GPU run, with gloo standing in for RCCL and both ranks on
    the one device of this box: bench.launch_ranks -> torch.distributed.run -> two ranks -> five graph segments around the
    four collectives -> per-collective milliseconds -> rank 0's one-rank child process for the kernel timings -> ONE JSON
    line on stdout with roofline, kernels and config.collectives filled in.  (The step time itself means nothing here: two
    processes share a GPU and gloo moves the gradients through the host.)"""
    import json
    import subprocess
    env = dict(os.environ, SUG_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '3', '--batch', '4',
                        '--no-cpu-baseline', '--no-other-workloads', '--caller-steps', '0', '--eager-steps', '1', '--profile-steps', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['scaling'] == 'weak' and d['config']['parallelism'] == 'dp2'
    assert d['config']['clouds_per_step'] == 16 and d['config']['launch'].startswith('segmented hipGraph')
    col = d['config']['collectives']
    assert col['backend'] == 'gloo' and col['world_size'] == 2
    assert set(col['ms']) >= {'all_gather_packed', 'sums_all_reduce', 'bucket1_exposed', 'bucket2_all_reduce', 'bucket1_alone'}
    assert all(v >= 0 for v in col['ms'].values()) and col['bytes']['bucket1'] > col['bytes']['bucket2'] > 0
    assert d['roofline'] is not None and d['roofline']['kernel'].startswith('knn') and 0 < d['roofline']['frac'] < 1
    assert any(k.startswith('edgeconv') for k in d['kernels'])
    assert all(v is not None for v in d['losses'])


def test_four_rank_segmented_graph_gloo_on_one_gpu():
    """World-size-specific arithmetic (row0 = rank * m_local, M = world * m_local rows in every MMD block, the packed
    all-gather's layout, / world in the gradient buckets) at a world size other than 2: FOUR ranks on the one device of this
    box (the box allows at most 6 processes on its card; the first real 8-rank run is the driver's), 2 clouds per domain and
    rank, five captured segments around the four collectives.  Replicas identical after every step; the global MMD terms
    identical on all ranks and equal to the single-process MMD of the gathered batch; per-rank CE = the shard's own."""
    from sug_amd.model import mmd
    from sug_amd.train_step import SUGStep, GEO_MMD
    world, B, N = 4, 8, 1024
    res = _run('gloo', 'segmented_graph', steps=3, world=world, B=B)
    outs, chks = [r[1] for r in res], [r[2] for r in res]
    for r in range(1, world):
        assert (chks[r] == chks[0]).all(), 'parameters diverged between ranks 0 and %d' % r
        for s in range(3):
            assert outs[r][s][1] == outs[0][s][1] and outs[r][s][2] == outs[0][s][2], (r, outs)
    data, lab, data_t, lab_t = _data(B, N, 5)
    feats = []
    for rank in range(world):
        lo, hi = rank * B // world, (rank + 1) * B // world
        net = _make(3).cuda().train()
        torch.manual_seed(100 + rank)
        d, l, dt, lt = [t[lo:hi].cuda() for t in (data, lab, data_t, lab_t)]
        tr = SUGStep(net, global_mmd=False)
        lc, _, _ = tr.losses(d, l, dt, lt)
        assert abs(float(lc) - outs[rank][0][0]) <= 1e-4 * max(1.0, abs(outs[rank][0][0])), (rank, float(lc), outs[rank][0])
        net = _make(3).cuda().train()
        torch.manual_seed(100 + rank)
        pair = torch.cat((d, dt))
        with torch.no_grad():
            net.forward_pair(pair)
            fs, ft = net.forward_pair(pair, node_adaptation=True)
        feats.append((l, lt, fs, ft))
    cat = lambda i: torch.cat([f[i] for f in feats])
    geo = float(mmd.mmd_cal(cat(0), cat(2), cat(1), cat(3), GEO_MMD))
    assert abs(geo - outs[0][0][1]) <= 1e-4 * max(1.0, abs(geo)), (geo, outs[0][0])


def test_bench_three_ranks_end_to_end_gloo_on_one_gpu():
    """`bench.py --gpus 3 --batch 2` on one device over gloo: a world size that is not a power of two (M = 6 rows per domain in
    the global MMD, / 3 in the gradient buckets) through the launcher, the five graph segments and rank 0's kernel-timestamp
    child.  (This test process + 3 ranks + the child = 5 processes on the card; the box allows 6 -- the first 8-rank run is
    the driver's.)"""
    import json
    import math
    import subprocess
    env = dict(os.environ, SUG_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--steps', '2', '--warmup', '3', '--batch', '2',
                        '--no-cpu-baseline', '--no-other-workloads', '--caller-steps', '0', '--eager-steps', '1', '--profile-steps', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1100)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 3 and d['config']['parallelism'] == 'dp3' and d['config']['clouds_per_step'] == 12
    assert d['config']['launch'].startswith('segmented hipGraph') and d['value'] > 0
    assert d['config']['collectives']['world_size'] == 3
    assert all(v is not None and math.isfinite(v) for v in d['losses']), d['losses']
    assert d['roofline'] is not None and d['roofline']['step']['gflop'] > 0


def test_two_rank_single_pass_step_gloo_on_one_gpu():
    """The opt-in single-pass step on two ranks: eager (bucketed all-reduce from autograd hooks), the uncaptured
    segment sequence and the five captured segments agree -- first step to rounding, segmented-eager == segmented-graph bit
    for bit -- the replicas stay identical, and the first step's losses equal the two-pass multi-rank step's CE exactly
    (same semantic pass) while its geometric MMD differs only through the node pass's own FPS draw."""
    runs = {m: _run('gloo', m, steps=3, single_pass=True) for m in ('eager', 'segmented_eager', 'segmented_graph')}
    for m, res in runs.items():
        (_, out0, chk0), (_, out1, chk1) = res
        assert (chk0 == chk1).all(), '%s: parameters diverged between ranks' % m
        for s in range(3):
            assert out0[s][1] == out1[s][1] and out0[s][2] == out1[s][2], (m, out0, out1)
    for r in range(2):
        assert runs['segmented_eager'][r][1] == runs['segmented_graph'][r][1]
        for a, b in zip(runs['eager'][r][1][0], runs['segmented_graph'][r][1][0]):
            assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (runs['eager'][r][1], runs['segmented_graph'][r][1])
    two = _run('gloo', 'segmented_graph', steps=2)      # (>= 2 steps: the worker asserts that the capture has happened)
    for r in range(2):
        a, b = runs['segmented_graph'][r][1][0], two[r][1][0]
        assert abs(a[0] - b[0]) <= 1e-6 * max(1.0, abs(b[0])), (a, b)          # CE: the same semantic pass
        assert abs(a[2] - b[2]) <= 1e-6 * max(1.0, abs(b[2])), (a, b)          # semantic MMD likewise
        assert abs(a[1] - b[1]) <= 5e-2 * max(1.0, abs(b[1])), (a, b)          # geometric MMD: another FPS draw of the nodes
    print({m: res[0][1] for m, res in runs.items()})

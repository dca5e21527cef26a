"""CPU, world_size 2, gloo: the batch-sharded path (differentiable feature all-gather for the
global MMD + flat gradient all-reduce) reproduces the single-process gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sug_amd.train_step import gather_rows_ddp, allreduce_grads_
    torch.manual_seed(0)
    m, D = 6, 16                       # global batch; each rank owns m/world rows of each domain
    Xs, Xt = torch.randn(m, D), torch.randn(m, D) + 0.2
    W = torch.nn.Parameter(torch.randn(D, D) * 0.3)
    w = torch.rand(m) + 0.5
    lo, hi = rank * m // world, (rank + 1) * m // world
    fs, ft = Xs[lo:hi] @ W, Xt[lo:hi] @ W
    cls = (fs ** 2).mean()                                            # a per-rank mean loss, like CE
    loss = cls + O.mix_rbf_mmd2(gather_rows_ddp(fs), gather_rows_ddp(ft), sample_weights=gather_rows_ddp(w[lo:hi]))
    loss.backward()
    allreduce_grads_([W], world)
    q.put((rank, W.grad.clone(), loss.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mmd_and_grad_allreduce_match_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the global batch
    torch.manual_seed(0)
    m, D = 6, 16
    Xs, Xt = torch.randn(m, D), torch.randn(m, D) + 0.2
    W = torch.nn.Parameter(torch.randn(D, D) * 0.3)
    w = torch.rand(m) + 0.5
    fs, ft = Xs @ W, Xt @ W
    loss = (fs ** 2).mean() + O.mix_rbf_mmd2(fs, ft, sample_weights=w)
    loss.backward()
    for rank, g, _ in res:
        torch.testing.assert_close(g, W.grad, rtol=1e-4, atol=1e-6)
    assert torch.equal(res[0][1], res[1][1])

"""optim.AdamChain (sug_adam_chain_step): the three optimizer steps of train_dg_single_gpu.py:333-335 in one launch, against
the optimizers stepped one after the other (bit for bit) and against torch.optim.Adam (fp32 rounding)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(AdamCls, seed=0, **kw):
    """Parameters laid out as the reference's three optimizers are (:193-203): `enc` sits in optimizer_dis AND in
    optimizer_g; sizes cover whole chunks, tails, unaligned views and one-element tensors."""
    torch.manual_seed(seed)
    dev = 'cuda'
    big = torch.randn(3 * 2048 + 4096 + 5, device=dev)
    enc = [big[:3 * 2048].clone(), torch.randn(4096 + 7, device=dev), torch.randn(64, device=dev), torch.randn(1, device=dev),
           torch.randn(33, 65, device=dev)]
    heads = [torch.randn(512, 256, device=dev), torch.randn(10, device=dev)]
    att = [torch.randn(64, 64, device=dev), torch.randn(3, device=dev)]
    nograd = torch.randn(100, device=dev)                 # never gets a gradient: must be skipped
    for t in enc + heads + att + [nograd]:
        t.requires_grad_(True)
    opt_g = AdamCls([{'params': [p]} for p in enc], lr=1e-3, weight_decay=5e-5, **kw)
    opt_c = AdamCls([{'params': heads[:1]}, {'params': heads[1:] + [nograd]}], lr=1e-3, weight_decay=5e-5, **kw)
    opt_dis = AdamCls([{'params': enc}, {'params': att}], lr=3e-3, weight_decay=5e-5, **kw)
    return enc + heads + att, (opt_dis, opt_g, opt_c)


def _grads(params, step):
    g = torch.Generator(device='cuda').manual_seed(100 + step)
    return [torch.randn(p.shape, device='cuda', generator=g) for p in params]


def _run(mode, capturable, steps=5, lr_schedule=None):
    from sug_amd.optim import Adam, AdamChain
    if mode == 'torch':
        params, opts = _setup(torch.optim.Adam)
    else:
        params, opts = _setup(Adam, graph_capturable=capturable)
    chain = AdamChain(opts) if mode == 'chain' else None
    for s in range(steps):
        if lr_schedule is not None:
            for o, scale in zip(opts, (3.0, 1.0, 1.0)):
                for g in o.param_groups:
                    g['lr'] = lr_schedule[s] * scale
        for p, g in zip(params, _grads(params, s)):
            p.grad = g
        if chain is not None:
            chain.step()
        else:
            for o in opts:
                o.step()
    torch.cuda.synchronize()
    return params, opts, chain


@pytest.mark.parametrize('capturable', [False, True])
def test_chain_equals_sequential_steps_bit_for_bit(capturable):
    lrs = [1e-3, 1e-3, 4e-4, 4e-4, 1e-3]
    pa, oa, chain = _run('chain', capturable, lr_schedule=lrs)
    pb, ob, _ = _run('seq', capturable, lr_schedule=lrs)
    assert not chain._joint['fallback'] and chain._joint['T'] == len(pa)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
    for x, y in zip(oa, ob):
        sx, sy = x.state_dict()['state'], y.state_dict()['state']
        assert sx.keys() == sy.keys()
        for k in sx:
            assert float(sx[k]['step']) == float(sy[k]['step']) == 5.0
            assert torch.equal(sx[k]['exp_avg'], sy[k]['exp_avg']) and torch.equal(sx[k]['exp_avg_sq'], sy[k]['exp_avg_sq'])
    pt, _, _ = _run('torch', False, lr_schedule=lrs)
    for a, t in zip(pa, pt):
        torch.testing.assert_close(a, t, rtol=2e-5, atol=2e-6)


def test_chain_replays_from_a_graph_and_follows_plan_changes():
    """Captured: the step counts, bias corrections and learning rates are device values, so replays continue the eager
    trajectory; a plain o.step() and load_state_dict() in between keep working (the chain rebuilds its joint table)."""
    from sug_amd.optim import Adam, AdamChain
    res = []
    for graphed in (False, True):
        params, opts = _setup(Adam, graph_capturable=True)
        chain = AdamChain(opts)
        static = [torch.zeros_like(p) for p in params]
        for p, g in zip(params, static):
            p.grad = g
        def load(s):
            for dst, src in zip(static, _grads(params, s)):
                dst.copy_(src)
        load(0)
        chain.step()                                        # eager: builds the plans outside any capture
        gen0 = chain.plan_generation
        graph = None
        if graphed:
            graph = torch.cuda.CUDAGraph()
            load(1)
            with torch.cuda.graph(graph):
                chain.step()
            graph.replay()                                  # a capture records, it does not run
        else:
            load(1)
            chain.step()
        for s in (2, 3):
            if s == 3:
                for g in opts[1].param_groups:
                    g['lr'] = 2e-4
                for o in opts:
                    o.refresh_device_scalars()              # what SUGStep._graph_step does before a replay
            load(s)
            graph.replay() if graphed else chain.step()
        assert chain.plan_generation == gen0
        load(4)
        for o in opts:                                      # the optimizers on their own: same device scalars
            o.step()
        opts[0].load_state_dict(copy.deepcopy(opts[0].state_dict()))
        load(5)
        chain.step()                                        # joint table rebuilt on the new moments
        assert chain.plan_generation == gen0 + 1
        torch.cuda.synchronize()
        steps = {float(v['step']) for o in opts for v in o.state_dict()['state'].values()}
        assert steps == {6.0}, steps
        res.append([p.detach().clone() for p in params])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_chain_falls_back_when_the_optimizers_do_not_fit():
    from sug_amd.optim import Adam, AdamChain
    torch.manual_seed(0)
    w = torch.randn(300, device='cuda', requires_grad=True)
    ref = w.detach().clone().requires_grad_(True)
    mk = lambda p: [Adam([p], lr=1e-3 * (i + 1)) for i in range(3)]     # one parameter in three optimizers
    a, b = mk(w), mk(ref)
    chain = AdamChain(a)
    for s in range(3):
        g = torch.randn(300, device='cuda')
        w.grad, ref.grad = g.clone(), g.clone()
        chain.step()
        for o in b:
            o.step()
    assert chain._joint['fallback']
    assert torch.equal(w, ref)
    # two chains over the same (capturable) optimizers: each re-homes the optimizers' device scalars when it builds its
    # table; the other one notices and rebuilds instead of advancing stale step counts
    params, opts = _setup(Adam, graph_capturable=True)
    c1, c2 = AdamChain(opts), AdamChain(opts)
    for s, c in enumerate((c1, c2, c1, c2, c1)):
        for p, g in zip(params, _grads(params, s)):
            p.grad = g
        c.step()
    steps = {float(v['step']) for o in opts for v in o.state_dict()['state'].values()}
    assert steps == {5.0}, steps
    with pytest.raises(RuntimeError, match='sug_amd.optim.Adam optimizers only'):
        AdamChain([torch.optim.Adam([w])])

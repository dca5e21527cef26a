"""Round 6: per-call hipGraph replay behind Net_MDA.forward (sug_amd/call_graphs.py) == the eager call-by-call form.

The caller is the reference's loop body (train_dg_single_gpu.py:260-264, :269-283, :309-335): four separate `model(...)` calls,
CE on the source logits of both heads, three `mmd_cal` terms, ONE backward, three torch.optim.Adam optimizers with the
reference's overlapping parameter groups (:191-203), zero_grad of all three -- nothing of SUGStep.  Run once with
`model.call_graphs = False` (every kernel launched from Python) and once with call graphs on: the losses of every step and a
sha256 over the whole state_dict are identical, and the CPU generator ends in the same state (same draws, same order)."""
import copy
import hashlib

import pytest
import torch

from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

GEO = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 50, 'GEO_SCALE': 1}
SEM = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'mean2one', 'LABEL_WEIGHT': 0.5, 'SEM_SCALE': 1}


def _state_hash(net):
    h = hashlib.sha256()
    for k, v in sorted(net.state_dict().items()):
        h.update(v.detach().cpu().numpy().tobytes())
    return h.hexdigest()


def _make(model, B, N, wseed=5, seed=11, p_drop=0.0):
    from sug_amd.model.Model import Net_MDA
    net = Net_MDA(model)
    net.load_state_dict(O.fill_params({k: tuple(v.shape) for k, v in net.state_dict().items()}, wseed))
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = p_drop
    g = torch.Generator().manual_seed(seed)
    batches = []
    for _ in range(2):          # two different batches, alternated: a replay must read the CURRENT batch
        data, data_t = O.synth_clouds(B, N, g), O.synth_clouds(B, N, g)
        lab, lab_t = torch.randint(0, 10, (B,), generator=g), torch.randint(0, 10, (B,), generator=g)
        batches.append([t.cuda() for t in (data, lab, data_t, lab_t)])
    return net.cuda().train(), batches


def _reference_loop(net, batches, steps, mmd_from=0):
    """train_dg_single_gpu.py:191-203 (optimizers) and :246-335 (loop body), MMD_WEIGHT = CLS_WEIGHT = SRC_LOSS_WEIGHT = 1."""
    from sug_amd.model import mmd
    crit = torch.nn.CrossEntropyLoss()
    params = [{'params': v} for k, v in net.g.named_parameters() if 'pred_offset' not in k]
    opt_g = torch.optim.Adam(params, lr=1e-3, weight_decay=5e-5)
    opt_c = torch.optim.Adam([{'params': net.c1.parameters()}, {'params': net.c2.parameters()}], lr=1e-3, weight_decay=5e-5)
    opt_dis = torch.optim.Adam([{'params': net.g.parameters()}, {'params': net.attention_s.parameters()},
                                {'params': net.attention_t.parameters()}], lr=1e-3, weight_decay=5e-5)
    out = []
    for s in range(steps):
        data, label, data_t, label_t = batches[s % len(batches)]
        pred_s1, pred_s2, sem_s1, sem_s2 = net(data, semantic_adaption=True)
        pred_t1, pred_t2, sem_t1, sem_t2 = net(data_t, semantic_adaption=True)
        loss_cls = 0.5 * crit(pred_s1, label) + 0.5 * crit(pred_s2, label)
        if s < mmd_from:
            loss = loss_cls
            vals = [float(loss_cls)]
        else:
            node_s = net(data, node_adaptation_s=True)
            node_t = net(data_t, node_adaptation_t=True)
            loss_geo = mmd.mmd_cal(label, node_s, label_t, node_t, GEO, data_s=data, data_t=data_t)
            l1 = mmd.mmd_cal(label, sem_s1, label_t, sem_t1, SEM, data_s=pred_s1, data_t=pred_t1)
            l2 = mmd.mmd_cal(label, sem_s2, label_t, sem_t2, SEM, data_s=pred_s2, data_t=pred_t2)
            loss_sem = 0.5 * l1 + 0.5 * l2
            loss = loss_cls + loss_geo + loss_sem
            vals = [float(loss_cls), float(loss_geo), float(loss_sem)]
        loss.backward()
        opt_dis.step()
        opt_g.step()
        opt_c.step()
        opt_g.zero_grad()
        opt_c.zero_grad()
        opt_dis.zero_grad()
        out.append((vals, _state_hash(net)))
    return out


def _run(model, B, N, steps, graphs, **kw):
    from sug_amd import call_graphs
    from sug_amd.model.Model import Net_MDA
    net, batches = _make(model, B, N, **{k: v for k, v in kw.items() if k in ('p_drop',)})
    Net_MDA.call_graphs = 'auto' if graphs else False
    torch.manual_seed(3)
    try:
        out = _reference_loop(net, batches, steps, mmd_from=kw.get('mmd_from', 0))
    finally:
        Net_MDA.call_graphs = False
    rng = torch.get_rng_state().clone()
    mgr = net.__dict__.get('_call_graph_mgr')
    return out, rng, (dict(mgr.stats) if mgr is not None else None), (mgr, net)


@pytest.mark.parametrize('model,B,N', [('DGCNN', 4, 1024), ('Pointnet', 4, 1024), ('Pointnet2', 2, 2048), ('PTran', 2, 1024)])
def test_graphed_calls_equal_the_eager_caller_form_bit_for_bit(model, B, N):
    """Six optimizer steps of the reference's loop body: step 1 eager (planning), step 2 captures the four calls, steps 3-6
    replay.  Losses and sha256(state_dict) of every step, and the CPU generator's final state, equal the all-eager run."""
    a, rng_a, _, _ = _run(model, B, N, 6, False)
    b, rng_b, stats, (mgr, net) = _run(model, B, N, 6, True)
    assert stats is not None and stats['refused'] == 0, (stats, [ks.why for ks in mgr.keys.values()])
    assert stats['captured'] == 4 and stats['replayed'] == 5 * 4, stats
    assert [x[0] for x in a] == [x[0] for x in b], 'losses differ: %s vs %s' % ([x[0] for x in a], [x[0] for x in b])
    assert [x[1] for x in a] == [x[1] for x in b], 'parameters / buffers differ after some step'
    assert torch.equal(rng_a, rng_b), 'the CPU generator was advanced differently (FPS start draws)'


def test_fp16_point_transformer_calls_are_graphed_with_and_without_a_step_cache():
    """BASELINE config 5's arithmetic (fp16 linears, fp32 accumulation) through call graphs: bit-identical to the eager calls;
    and a caller that keeps a step-scoped cache of 16-bit weight copies (ops.CTX.w16_cache, as SUGStep does) no longer
    turns the captures away -- a captured call casts its weights inside its own graph."""
    from sug_amd import ops
    from sug_amd.model import Ptran_transformer as PT
    keep = (PT.GEMM_DTYPE, PT.PROJ_16BIT)
    PT.GEMM_DTYPE, PT.PROJ_16BIT = torch.float16, True
    try:
        a, rng_a, _, _ = _run('PTran', 2, 1024, 5, False)
        b, rng_b, stats, (mgr, _) = _run('PTran', 2, 1024, 5, True)
        assert stats['refused'] == 0 and stats['captured'] == 4 and stats['replayed'] == 4 * 4, (stats, [ks.why for ks in mgr.keys.values()])
        assert a == b and torch.equal(rng_a, rng_b)
        ops.CTX.w16_cache = {}
        try:
            c, rng_c, stats_c, _ = _run('PTran', 2, 1024, 5, True)
        finally:
            ops.CTX.w16_cache = None
        assert stats_c['refused'] == 0 and stats_c['captured'] == 4, stats_c
        assert [x[0] for x in c[2:]] == [x[0] for x in a[2:]] or all(
            abs(u - v) <= 1e-6 * max(1.0, abs(v)) for x, y in zip(c, a) for u, v in zip(x[0], y[0]))
    finally:
        PT.GEMM_DTYPE, PT.PROJ_16BIT = keep


def test_outputs_survive_the_next_replay_and_unused_parameters_keep_no_gradient():
    """What a caller may rely on: the tensors a call returns are its own (not overwritten by the instance's next replay);
    parameters the forward never uses (DGCNN.input_transform_net, adapt_layer_off.trans -- model/Model.py:61, model_utils.py:97)
    keep .grad None, as under eager autograd (Adam skips them: no weight decay on parameters without a gradient)."""
    from sug_amd.model.Model import Net_MDA
    net, batches = _make('DGCNN', 4, 1024)
    Net_MDA.call_graphs = 'auto'
    try:
        torch.manual_seed(0)
        kept = []
        for s in range(4):
            data, label, data_t, label_t = batches[s % 2]
            y = net(data, semantic_adaption=True)
            kept.append((y[0], y[0].detach().clone()))
            (y[0].sum() + y[2].sum()).backward()
            unused = [k for k, p in net.named_parameters() if p.grad is None]
            assert any('input_transform_net' in k for k in unused) and any('node_fea_adapt.trans' in k for k in unused), unused
            assert all(p.grad is not None for k, p in net.c1.named_parameters())
            net.zero_grad(set_to_none=True)
        for live, snap in kept:
            assert torch.equal(live, snap)
        st = net.__dict__['_call_graph_mgr'].stats
        assert st['captured'] == 1 and st['replayed'] == 3 and st['refused'] == 0, st
    finally:
        Net_MDA.call_graphs = False


def test_gradient_accumulation_and_mixed_eager_calls():
    """Two forward/backward rounds WITHOUT zero_grad in between accumulate (as .grad does under eager autograd), also when
    one of the rounds is an eager call (call graphs switched off for it)."""
    from sug_amd.model.Model import Net_MDA
    res = []
    for pattern in ((False, False, False, False), (True, True, True, True), (True, True, False, True)):
        net, batches = _make('Pointnet', 4, 1024)
        torch.manual_seed(0)
        try:
            for s, on in enumerate(pattern):
                Net_MDA.call_graphs = 'auto' if on else False
                data, label, data_t, label_t = batches[s % 2]
                y = net(data, semantic_adaption=True)
                torch.nn.functional.cross_entropy(y[0], label).backward()
                if s == 1:
                    net.zero_grad(set_to_none=True)         # rounds 3 and 4 accumulate
        finally:
            Net_MDA.call_graphs = False
        res.append({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    for r in res[1:]:
        assert r.keys() == res[0].keys()
        for k in r:
            assert torch.equal(r[k], res[0][k]), k


def test_deepcopy_and_eval_are_untouched():
    """copy.deepcopy(model) (train_dg_single_gpu.py:364) works with captured graphs in place; eval-mode and no_grad calls
    never go through a graph."""
    from sug_amd.model.Model import Net_MDA
    net, batches = _make('Pointnet', 4, 1024)
    Net_MDA.call_graphs = 'auto'
    try:
        data, label, data_t, label_t = batches[0]
        for _ in range(3):
            y = net(data, semantic_adaption=True)
            y[0].sum().backward()
            net.zero_grad()
        mgr = net.__dict__['_call_graph_mgr']
        assert mgr.stats['captured'] == 1
        twin = copy.deepcopy(net)
        assert twin.__dict__.get('_call_graph_mgr') is None
        before = dict(mgr.stats)
        net.eval()
        with torch.no_grad():
            torch.manual_seed(9)            # (the SA-node module draws its FPS start in eval mode as well)
            e1 = net(data, semantic_adaption=True)
            torch.manual_seed(9)
            e2 = twin.eval()(data, semantic_adaption=True)
        assert dict(mgr.stats) == before
        for u, v in zip(e1, e2):
            assert torch.equal(u, v)
    finally:
        Net_MDA.call_graphs = False


def _grads(net):
    return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}


def test_many_calls_in_flight_eval_interludes_and_new_shapes():
    """What a real loop does besides the steady state: (a) SIX forwards of one kind before the single backward (more than the
    four graph instances a key may hold: the surplus runs eagerly), (b) an eval-mode validation pass between training steps
    (graphs untouched, reused afterwards), (c) a last batch of another size (a new key: eager once, then captured).  Losses,
    BatchNorm buffers and the CPU generator's state equal the all-eager run bit for bit (no parameter update between the
    steps: each is judged from the same weights); gradient norms to 1e-5: with THREE calls sharing one batch's prefix (two importers of one exporter) autograd adds the importers' prefix
    gradients to each other before the exporter's backward graph adds its own, where the eager graph adds them one node at a
    time -- another association of the same fp32 sums (the reference's pattern, one importer per prefix, is exact:
    test_graphed_calls_equal_the_eager_caller_form_bit_for_bit)."""
    from sug_amd.model.Model import Net_MDA
    res = {}
    for graphs in (False, True):
        net, batches = _make('Pointnet', 4, 1024)
        small = [t[:2].clone() for t in batches[1]]
        torch.manual_seed(5)
        log = []
        try:
            Net_MDA.call_graphs = 'auto' if graphs else False
            for step in range(5):
                data, label, data_t, label_t = batches[step % 2] if step != 3 else small       # step 3: the odd-sized batch
                outs = [net(data if i % 2 == 0 else data_t, semantic_adaption=True) for i in range(6 if step >= 2 else 2)]
                loss = sum(torch.nn.functional.cross_entropy(o[0], label if i % 2 == 0 else label_t) + o[2].square().mean()
                           for i, o in enumerate(outs))
                loss.backward()
                log.append((float(loss), _state_hash(net), {k: float(v.double().norm()) for k, v in _grads(net).items()}))
                net.zero_grad(set_to_none=True)           # (no parameter update: every step is judged from the same weights)
                if step == 1:                           # validation interlude
                    net.eval()
                    with torch.no_grad():
                        torch.manual_seed(77)
                        log.append(float(net(data, semantic_adaption=True)[0].sum()))
                    net.train()
                    torch.manual_seed(6)
        finally:
            Net_MDA.call_graphs = False
        res[graphs] = (log, torch.get_rng_state().clone())
        if graphs:
            st = net.__dict__['_call_graph_mgr'].stats
            assert st['refused'] == 0 and st['captured'] >= 2 and st['replayed'] >= 4 and st['eager'] >= 2, st
    for i, (a, b) in enumerate(zip(res[True][0], res[False][0])):
        if isinstance(a, float):                            # the eval-mode interlude's output
            assert a == b
            continue
        assert a[0] == b[0] and a[1] == b[1], (i, a[0], b[0])                  # loss, sha256(parameters + BatchNorm buffers)
        assert a[2].keys() == b[2].keys()
        gmax = max(abs(v) for v in b[2].values())
        for k in a[2]:
            assert abs(a[2][k] - b[2][k]) <= 1e-5 * abs(b[2][k]) + 1e-6 * gmax, (i, k, a[2][k], b[2][k])      # (gradient norms)
    assert torch.equal(res[True][1], res[False][1])

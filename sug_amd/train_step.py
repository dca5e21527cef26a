"""One SUG training step (the caller of the hot path), mirroring the batch loop of the
reference's train_dg_single_gpu.py:246-335 and its optimiser set-up (:191-203):

  2 semantic forwards (source, target) -> CE on both heads -> 2 node forwards ->
  geometric MMD on attention features + semantic MMD on both heads' 256-d features
  (SDA weights from the head logits) -> one backward -> Adam steps (dis, g, c).

Not reproduced on purpose: the unconditional KPConv import (:27) and the `loss_s`
read-before-assignment under ADV_WEIGHT > 0 (:274-276); ADV_WEIGHT > 0 is supported with
the evident intent (loss_s += loss_adv after loss_s is formed).

Multi-GPU (one process per GPU, RCCL): batch-sharded; gradients are averaged with one flat
all-reduce per step; with `global_mmd=True` the MMD is the single-GPU loss of the *global*
batch (differentiable all-gather of the [m, D+10] features and of logits/labels for the SDA
weights; backward = reduce-scatter), otherwise reference-style local MMD (train_dg.py).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .model import mmd

GEO_MMD = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 50, 'GEO_SCALE': 1}
SEM_MMD = {'NAME': 'SOFT_MMD', 'LABEL_SCALE': 5, 'SEM_WEIGHTS': 'mean2one', 'LABEL_WEIGHT': 0.5, 'SEM_SCALE': 1}
METHODS = {'MMD_WEIGHT': 1.0, 'CLS_WEIGHT': 1.0, 'ADV_WEIGHT': 0.0, 'TARGET_LOSS': 0.0, 'SRC_LOSS_WEIGHT': 1.0,
           'PURE_CLS_EPOCH': 0, 'GRL': False, 'GEO_MMD': [GEO_MMD], 'SEM_MMD': [SEM_MMD]}


def discrepancy(out1, out2):
    """utils/train_utils.py:51-54."""
    return torch.mean(torch.abs(F.softmax(out1, dim=-1) - F.softmax(out2, dim=-1)))


def _all_gather_rows(x):
    world = dist.get_world_size()
    x = x.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if dist.get_backend() == 'gloo' and x.is_cuda:        # rehearsal only: gloo gathers device tensors as a list
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, x)
    else:
        dist.all_gather_into_tensor(out, x)
    return out


class _AllGatherRows(torch.autograd.Function):
    """cat over ranks along dim 0; backward sums every rank's gradient for the local slice."""

    @staticmethod
    def forward(ctx, x):
        return _all_gather_rows(x)

    @staticmethod
    def backward(ctx, g):
        world = dist.get_world_size()
        g = g.contiguous()
        m = g.shape[0] // world
        if dist.get_backend() == 'gloo':          # CPU tests: gloo has no reduce-scatter
            g = g.clone()
            dist.all_reduce(g)
            r = dist.get_rank()
            return g[r * m:(r + 1) * m]
        out = torch.empty((m,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
        return out


def gather_rows_ddp(x):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x
    if x.requires_grad:
        return _AllGatherRows.apply(x)
    return _all_gather_rows(x)


def gather_rows_packed(tensors):
    """gather_rows_ddp of several [m, w_i] tensors with ONE collective (one all-gather forward, one
    reduce-scatter backward): the columns are packed side by side, gathered, and split again.
    Integer tensors ride along as exact small floats.  Returns the gathered [world*m, w_i] tensors."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(tensors)
    cols, meta = [], []
    for t in tensors:
        t2 = t.reshape(t.shape[0], -1)
        meta.append((t2.shape[1], t.dtype, tuple(t.shape[1:])))
        cols.append(t2 if t2.dtype == torch.float32 else t2.detach().to(torch.float32))
    G = gather_rows_ddp(torch.cat(cols, dim=1))
    out, off = [], 0
    for w, dt, tail in meta:
        g = G[:, off:off + w]
        off += w
        if dt != torch.float32:
            g = g.round().to(dt)
        out.append(g.reshape((G.shape[0],) + tail))
    return out


def allreduce_grads_(params, world):
    """Average gradients over ranks with a single flat all-reduce (RCCL over xGMI: one large
    message keeps all 7 links busy; SURVEY 2.2 -- 42 MiB for DGCNN).  Only parameters that
    received a gradient take part; that set is the same on every rank (same graph)."""
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat)
    flat.div_(world)
    off = 0
    for p in ps:                    # the averaged gradients stay in the flat buffer: no copy back
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n


class GradReducer:
    """Gradient averaging over ranks, overlapped with backward.

    Parameters are split into buckets in the order their gradients become ready (for Net_MDA: heads
    and attention layers first -- 38 of the 46 MB --, the encoder last).  A bucket's flat all-reduce
    (one large message per bucket: RCCL spreads it over the 7 xGMI links) is issued from the
    post-accumulate hook of its last parameter, while the rest of backward still runs; finish()
    waits, divides by the world size and leaves every .grad as a view of the reduced flat buffer.
    Which parameters receive a gradient is learned per `key` (e.g. MMD on / off) from a first,
    non-overlapped step; any deviation later falls back to the synchronous path for that bucket."""

    def __init__(self, buckets, world):
        self.world = world
        self.buckets = [[p for p in b if p.requires_grad] for b in buckets]
        self.expected = {}                       # key -> [set of param ids per bucket]
        self.key = None
        self.active = False
        for bi, ps in enumerate(self.buckets):
            for p in ps:
                p.register_post_accumulate_grad_hook(lambda q, bi=bi: self._ready(bi, q))

    def begin(self, key):
        """Call right before backward."""
        n = len(self.buckets)
        self.key = key
        self.seen = [set() for _ in range(n)]
        self.handles = [None] * n
        self.flats = [None] * n
        self.members = [None] * n
        self.dirty = [False] * n
        self.active = True

    def _launch(self, bi, async_op):
        ps = [p for p in self.buckets[bi] if p.grad is not None]
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        self.flats[bi], self.members[bi] = flat, ps
        self.handles[bi] = dist.all_reduce(flat, async_op=async_op)

    def _ready(self, bi, p):
        if not self.active:
            return
        self.seen[bi].add(id(p))
        exp = self.expected.get(self.key)
        if exp is None:
            return                                # learning step: everything happens in finish()
        if self.flats[bi] is not None:
            self.dirty[bi] = True                 # a gradient after the launch: redo this bucket in finish()
        elif self.seen[bi] == exp[bi]:
            self._launch(bi, async_op=True)

    def finish(self):
        """Call after backward, before the optimizers."""
        self.active = False
        for bi in range(len(self.buckets)):
            if self.handles[bi] is not None and hasattr(self.handles[bi], 'wait'):
                self.handles[bi].wait()
            if self.flats[bi] is None or self.dirty[bi]:
                self.flats[bi] = None
                self._launch(bi, async_op=False)
            flat = self.flats[bi]
            if flat is None:
                continue
            flat.div_(self.world)
            off = 0
            for p in self.members[bi]:            # the averaged gradients stay in the flat buffer
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n
        if self.key not in self.expected:
            self.expected[self.key] = [set(s) for s in self.seen]
        self.flats = self.members = self.handles = None


from .call_graphs import StartFeeder as _StartFeeder      # FPS start draws of a replayable step (moved there in round 6)


class SUGStep:
    def __init__(self, model, lr=1e-3, weight_decay=5e-5, lr_scaler=1.0, methods=None, criterion=None,
                 global_mmd=True, fused_adam=None, share_prefix=True, use_graph=False, pair_domains=True,
                 force_segmented=False, single_pass=False):
        self.model = model
        # ONE encoder evaluation per domain feeds the heads and the attention layers, instead of the
        # semantic + node pass of train_dg_single_gpu.py:260-264, :309-310.  Same losses and gradients as the two-pass step
        # whose node pass draws the FPS starts of its semantic pass, BatchNorm running statistics included (the pass's update
        # is applied twice, Net_MDA.dual_updates_bn_twice); the one difference: one start draw per sampling stage instead of two.
        self.single_pass = bool(single_pass)
        self.base_lr, self.lr_scaler = float(lr), float(lr_scaler)
        # source and target batch go through the encoder as one 2B-cloud batch with per-domain
        # BatchNorm statistics (Net_MDA.forward_pair): same results, half the launches
        self.pair_domains = bool(pair_domains) and hasattr(model, 'forward_pair')
        # one backward over all four forwards of a step -> the encoder may share the stage in
        # front of the SA-node module between the semantic and node pass of a batch (exact)
        self.share_prefix = bool(share_prefix) and hasattr(model.g, 'share_prefix')
        if hasattr(model.g, 'share_prefix'):
            model.g.share_prefix = self.share_prefix
        self._split_layers = [m for m in model.modules() if hasattr(m, 'cache_weight_split')]
        for m in self._split_layers:
            m.cache_weight_split = bool(share_prefix)
        self.methods = dict(METHODS)
        if methods:
            self.methods.update(methods)
        self.criterion = criterion if criterion is not None else nn.CrossEntropyLoss()
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # force_segmented: the multi-rank launch form (graph segments + collectives) even on ONE rank of an initialised
        # process group -- a one-GPU rehearsal of the RCCL calls between graph replays (bench.py --segmented)
        force_segmented = bool(force_segmented) and dist.is_available() and dist.is_initialized()
        self.global_mmd = global_mmd and (self.world > 1 or force_segmented)
        self.reducer = None
        if self.world > 1 and not use_graph:     # eager multi-rank steps: bucketed all-reduce from autograd hooks
            early = [p for m in (model.c1, model.c2, model.attention_s, model.attention_t) for p in m.parameters()]
            self.reducer = GradReducer([early, list(model.g.parameters())], self.world)
        # fused_adam: None/True -> sug_amd.optim.Adam (one launch per optimizer) on a HIP device;
        # False -> torch.optim.Adam's default path (the parity tests' reference update)
        kw = {}
        on_gpu = next(model.parameters()).is_cuda
        own_adam = on_gpu and (fused_adam is None or fused_adam)
        # hipGraph mode: the whole step (forwards, losses, backward, 3 Adam updates) is captured once and replayed, FPS
        # start indices fed through a static buffer.  One rank: one graph per configuration key;
        # more ranks: five captured segments around the four collectives (_segmented_step).
        self.use_graph = bool(use_graph) and next(model.parameters()).is_cuda
        # more than one rank: the step is captured as FIVE graph segments around its four collectives (_segmented_step)
        self.segmented = self.use_graph and (self.world > 1 or force_segmented)
        self._graphs = None
        self._total = None
        self._combine_tail = os.environ.get('SUG_FUSED_LOSS', '1') != '0'
        self._mmd_multi = os.environ.get('SUG_MMD_MULTI', '1') != '0'   # the step's MMD terms stage by stage in one launch
        self.max_graphs = 4                             # captured steps kept (each owns a private memory pool)
        # segmented steps: {collective name: [(event before, event after), ...]} when a dict is assigned (bench.py
        # --gpus N reads it for config.collectives); None = no events
        self.collective_events = None
        # SUG_GRAPH_GUARD=1 restores the historical guard (one eager op between two replays)
        self._tick = torch.zeros(1, device=next(model.parameters()).device) \
            if (self.use_graph and os.environ.get('SUG_GRAPH_GUARD') == '1') else None
        from .optim import Adam as _SugAdam
        AdamCls = _SugAdam if own_adam else torch.optim.Adam
        # fused small ops (LayerNorm + activation of the heads in one launch); scoped to this trainer's forwards (set and
        # restored around self.losses()).  On in eager AND in graph mode since round 4: both modes then run the same kernels,
        # and a captured step equals its eager twin bit for bit (tests/test_gpu_determinism.py)
        self.fused_heads = on_gpu
        # opt-in (SUG_PARALLEL_BRANCHES=1): independent small-kernel chains (heads, attention layers, MMD terms) on forked
        # streams inside the captured graph.  Measured SLOWER on ROCm 7.2 / MI355X (5.35 vs 5.26 ms per step: a cross-stream
        # edge of a hipGraph costs more than the ~5 us kernels it lets overlap), hence off.
        self.parallel_branches = self.use_graph and os.environ.get('SUG_PARALLEL_BRANCHES', '0') == '1'
        if self.use_graph:
            if own_adam:
                kw['graph_capturable'] = True           # step count / bias corrections / lr on the device
            else:
                kw['capturable'] = True
                kw['fused'] = True
        # train_dg_single_gpu.py:191-203
        params = [{'params': v} for k, v in model.g.named_parameters() if 'pred_offset' not in k]
        self.optimizer_g = AdamCls(params, lr=lr, weight_decay=weight_decay, **kw)
        self.optimizer_c = AdamCls([{'params': model.c1.parameters()}, {'params': model.c2.parameters()}],
                                            lr=lr, weight_decay=weight_decay, **kw)
        self.optimizer_dis = AdamCls([{'params': model.g.parameters()},
                                               {'params': model.attention_s.parameters()},
                                               {'params': model.attention_t.parameters()}],
                                              lr=lr * lr_scaler, weight_decay=weight_decay, **kw)
        # the three steps of train_dg_single_gpu.py:333-335 (dis, g, c -- the encoder's parameters get two updates) in one
        # launch, bit-identical to the calls in sequence (optim.AdamChain; SUG_ADAM_CHAIN=0: the three calls)
        self._adam_chain = None
        if own_adam and os.environ.get('SUG_ADAM_CHAIN', '1') != '0':
            from .optim import AdamChain
            self._adam_chain = AdamChain([self.optimizer_dis, self.optimizer_g, self.optimizer_c])

    # ------------------------------------------------------------------ learning-rate schedules
    def set_epoch(self, epoch, max_epoch_num):
        """Learning rates of the three optimizers at the start of `epoch`, as the reference sets them
        (train_dg_single_gpu.py:194-203, :210-212): CosineAnnealingLR(T_max=max_epoch_num, eta_min=0)
        stepped with an explicit epoch (= its closed form) for optimizer_g / optimizer_c, and
        utils/train_utils.py:39-48 `adjust_learning_rate` for optimizer_dis (halved every 5 epochs up
        to epoch 30, every 10 afterwards; untouched at epoch 0).  Returns (lr_g, lr_c, lr_dis)."""
        import math
        cos = (1.0 + math.cos(math.pi * epoch / max_epoch_num)) / 2.0
        for opt in (self.optimizer_g, self.optimizer_c):
            for g in opt.param_groups:
                g['lr'] = self.base_lr * cos
        if epoch > 0:
            lr = self.base_lr * self.lr_scaler * (0.5 ** (epoch // 5 if epoch <= 30 else epoch // 10))
            for g in self.optimizer_dis.param_groups:
                g['lr'] = lr
        return (self.optimizer_g.param_groups[0]['lr'], self.optimizer_c.param_groups[0]['lr'],
                self.optimizer_dis.param_groups[0]['lr'])

    # ------------------------------------------------------------------ losses
    def _mmd(self, label, feat_s, label_t, feat_t, cfg, data_s, data_t):
        if self.global_mmd:
            label, label_t = gather_rows_ddp(label), gather_rows_ddp(label_t)
            feat_s, feat_t = gather_rows_ddp(feat_s), gather_rows_ddp(feat_t)
            if cfg.get('GEO_WEIGHTS') or cfg.get('SEM_WEIGHTS'):
                data_s, data_t = gather_rows_ddp(data_s.detach()), gather_rows_ddp(data_t.detach())
        return mmd.mmd_cal(label, feat_s, label_t, feat_t, cfg, data_s=data_s, data_t=data_t)

    def _global_soft_mmd(self, data, label, data_t, label_t, node_s, node_t, head1, head2):
        """The three soft-MMD terms of the GLOBAL batch on a batch-sharded step.  Everything
        the terms need from the other ranks -- node features, semantic features, logits, labels and the
        per-pair Chamfer distances behind GEO_WEIGHTS -- crosses the ranks as VALUES in ONE packed
        all-gather; each rank then evaluates its own row block of every kernel matrix
        (mmd.soft_mmd_sharded: one 3-double all-reduce per term, no collective in the backward).
        The SDA weights are computed from the gathered batch, identically on every rank."""
        M_, geo, sem = self.methods, self.methods['GEO_MMD'][0], self.methods['SEM_MMD'][0]
        rank = dist.get_rank()
        mloc = label.shape[0]
        (s1, t1, p1s, p1t), (s2, t2, p2s, p2t) = head1, head2
        vals = [label, label_t, node_s.detach(), node_t.detach(), s1.detach(), t1.detach(), s2.detach(), t2.detach(),
                p1s.detach(), p1t.detach(), p2s.detach(), p2t.detach()]
        if geo.get('GEO_WEIGHTS'):
            vals.append(mmd.chamfer_distances(data, data_t).reshape(-1, 1))
        g = gather_rows_packed(vals)
        label_g, label_tg, fn_s, fn_t, g_s1, g_t1, g_s2, g_t2, g_p1s, g_p1t, g_p2s, g_p2t = g[:12]
        row0 = rank * mloc
        w_geo = None
        if geo.get('GEO_WEIGHTS'):
            w_geo = mmd.distance2weights(g[12].reshape(-1), geo['GEO_WEIGHTS']).reshape(1, -1)
        loss_geo = M_['MMD_WEIGHT'] * geo['GEO_SCALE'] * mmd.soft_mmd_sharded(
            label, node_s, label_t, node_t, label_g, fn_s, label_tg, fn_t, float(geo['LABEL_SCALE']), row0, w_geo,
            self.world)
        terms = []
        for (fs, ft, gs, gt, gps, gpt) in ((s1, t1, g_s1, g_t1, g_p1s, g_p1t), (s2, t2, g_s2, g_t2, g_p2s, g_p2t)):
            w = mmd.sda_weights_of(sem, gps, gpt, label_g, label_tg)      # mmd_cal's rules (GEO > ENTROPY > SEM)
            terms.append(mmd.soft_mmd_sharded(label, fs, label_t, ft, label_g, gs, label_tg, gt,
                                              float(sem['LABEL_SCALE']), row0, w, self.world))
        return loss_geo, (0.5 * M_['MMD_WEIGHT'] * sem['SEM_SCALE']) * (terms[0] + terms[1])

    def losses(self, data, label, data_t, label_t, mmd_on=True, combine=False):
        """(loss_cls, loss_geo, loss_sem), each differentiable.  combine=True (the step's own call): the total is formed in
        one launch each way and kept in self._total; the returned geo / sem parts are then values for reporting only."""
        M = self.methods
        model = self.model
        pair = None
        fused_ce = None
        nodes = None                # single_pass: the attention features came out of the semantic forward already
        if self.pair_domains and data.shape == data_t.shape and model.training:
            pair = torch.cat((data, data_t), dim=0)
            dual = self.single_pass and mmd_on and M['MMD_WEIGHT'] > 0
            if mmd_on and M['MMD_WEIGHT'] > 0 and not dual and hasattr(model, 'plan_pair_geometry') \
                    and os.environ.get('SUG_PLAN_GEOMETRY', '1') != '0':
                model.plan_pair_geometry(pair, passes=2)    # FPS / ball query of the semantic AND the node pass in one set of launches
            plain_ce = isinstance(self.criterion, nn.CrossEntropyLoss) and self.criterion.weight is None \
                and self.criterion.reduction == 'mean' and self.criterion.label_smoothing == 0.0 \
                and M['ADV_WEIGHT'] <= 0 and M['TARGET_LOSS'] <= 0
            if plain_ce and os.environ.get('SUG_FUSED_LOSS', '1') != '0':
                # the paired logits / mid features stay paired: CE of both heads in one launch each way (ops.ce_pair writes the
                # whole pair's gradient), the halves of the mid features split with a copy-free backward
                if dual:
                    y1, y2, f1, f2, nodes = model.forward_pair(pair, paired_out=True, dual=True)
                else:
                    y1, y2, f1, f2 = model.forward_pair(pair, paired_out=True)
                if y1.dim() == 2 and ops.ce_pair_supported(y1, y2, label):
                    Bs = label.shape[0]
                    fused_ce = ops.ce_pair(y1, y2, label, 0.5 * M['SRC_LOSS_WEIGHT'] * M['CLS_WEIGHT'], self.criterion.ignore_index)
                    (sem_s1, sem_t1), (sem_s2, sem_t2) = ops.split_halves(f1), ops.split_halves(f2)
                    yd1, yd2 = y1.detach(), y2.detach()            # the SDA weights read the logits' values only
                    pred_s1, pred_t1, pred_s2, pred_t2 = yd1[:Bs], yd1[Bs:], yd2[:Bs], yd2[Bs:]
                else:
                    (pred_s1, pred_t1), (pred_s2, pred_t2) = ops.split_halves(y1), ops.split_halves(y2)
                    (sem_s1, sem_t1), (sem_s2, sem_t2) = ops.split_halves(f1), ops.split_halves(f2)
            elif dual:
                (pred_s1, pred_s2, sem_s1, sem_s2), (pred_t1, pred_t2, sem_t1, sem_t2), nodes = \
                    model.forward_pair(pair, dual=True)
            else:
                (pred_s1, pred_s2, sem_s1, sem_s2), (pred_t1, pred_t2, sem_t1, sem_t2) = model.forward_pair(pair)
        elif self.single_pass and mmd_on and M['MMD_WEIGHT'] > 0:
            pred_s1, pred_s2, sem_s1, sem_s2, node_s = model(data, semantic_adaption=True, node_adaptation_s=True)
            pred_t1, pred_t2, sem_t1, sem_t2, node_t = model(data_t, semantic_adaption=True, node_adaptation_t=True)
            nodes = (node_s, node_t)
        else:
            pred_s1, pred_s2, sem_s1, sem_s2 = model(data, semantic_adaption=True)
            pred_t1, pred_t2, sem_t1, sem_t2 = model(data_t, semantic_adaption=True)
        # scalar algebra with the constant factors multiplied on the host: every tensor-scalar op is a launch forward
        # and one backward (0.5*a + 0.5*b = 0.5*(a + b) exactly; the folded weights differ from the reference's
        # left-to-right products by an ulp at most)
        if fused_ce is None and pair is None and self._combine_tail and model.training and pred_s1.is_cuda \
                and isinstance(self.criterion, nn.CrossEntropyLoss) and self.criterion.weight is None \
                and self.criterion.reduction == 'mean' and self.criterion.label_smoothing == 0.0 \
                and M['ADV_WEIGHT'] <= 0 and M['TARGET_LOSS'] <= 0 and pred_s1.dim() == 2 \
                and ops.ce_pair_supported(pred_s1, pred_s2, label):
            # the separate-calls form (the four model(...) calls of an unchanged caller): CE of both heads in one launch each
            # way as well (ops.ce_pair scores the first len(label) rows: here all of them)
            fused_ce = ops.ce_pair(pred_s1, pred_s2, label, 0.5 * M['SRC_LOSS_WEIGHT'] * M['CLS_WEIGHT'], self.criterion.ignore_index)
        if fused_ce is not None:
            loss_cls = fused_ce
        elif M['ADV_WEIGHT'] > 0 or M['TARGET_LOSS'] > 0:
            loss_s = 0.5 * (self.criterion(pred_s1, label) + self.criterion(pred_s2, label))
            if M['ADV_WEIGHT'] > 0:
                loss_s = loss_s - M['ADV_WEIGHT'] * discrepancy(pred_t1, pred_t2)
            if M['TARGET_LOSS'] > 0:    # the reference scores the target predictions against `label` (:284-285)
                loss_t = 0.5 * (self.criterion(pred_t1, label) + self.criterion(pred_t2, label))
                loss = 0.5 * (loss_s + loss_t)
            else:
                loss = M['SRC_LOSS_WEIGHT'] * loss_s
            loss_cls = M['CLS_WEIGHT'] * loss
        else:
            loss_cls = (0.5 * M['SRC_LOSS_WEIGHT'] * M['CLS_WEIGHT']) * (self.criterion(pred_s1, label) +
                                                                          self.criterion(pred_s2, label))
        if not mmd_on or M['MMD_WEIGHT'] <= 0:
            return loss_cls, None, None
        if nodes is not None:
            feat_node_s, feat_node_t = nodes
        elif pair is not None:
            feat_node_s, feat_node_t = model.forward_pair(pair, node_adaptation=True)
        else:
            feat_node_s = model(data, node_adaptation_s=True)
            feat_node_t = model(data_t, node_adaptation_t=True)
        geo, sem = M['GEO_MMD'][0], M['SEM_MMD'][0]
        if self.global_mmd and geo['NAME'] == 'SOFT_MMD' and sem['NAME'] == 'SOFT_MMD' and sem['SEM_SCALE'] > 0 \
                and pred_s1.dim() == 2:
            return (loss_cls,) + self._global_soft_mmd(data, label, data_t, label_t, feat_node_s, feat_node_t,
                                                        (sem_s1, sem_t1, pred_s1, pred_t1),
                                                        (sem_s2, sem_t2, pred_s2, pred_t2))
        # the three MMD terms are independent chains of small kernels: side by side under graph replay (ops.run_parallel)
        cs, ct = data, data_t
        if pair is not None and geo.get('GEO_WEIGHTS') and not self.global_mmd and pair.shape[2] != 3:   # (N == 3: [m,3,3] rows would read as channel-first, mmd.py:110)
            rows = ops.cloud_rows(pair)                  # the encoder's own [2B,N,3] rows: no second transpose for Chamfer
            cs, ct = rows[:data.shape[0], :, :3], rows[data.shape[0]:, :, :3]
        if self._mmd_multi and not self.global_mmd and not ops.CTX.parallel_branches and feat_node_s.is_cuda \
                and geo['NAME'] == 'SOFT_MMD' and (sem['SEM_SCALE'] <= 0 or sem['NAME'] == 'SOFT_MMD') \
                and feat_node_s.dim() == 2 and feat_node_s.shape == feat_node_t.shape \
                and (sem['SEM_SCALE'] <= 0 or (sem_s1.dim() == 2 and sem_s1.shape == sem_t1.shape == sem_s2.shape == sem_t2.shape
                                               and sem_s1.shape[0] == feat_node_s.shape[0])):
            # the terms share the batch and its labels: every stage of the three in one launch (mmd.soft_mmd_multi; each
            # term's value and gradient are those of its own mmd_cal call bit for bit)
            tl = [(feat_node_s, feat_node_t, geo, cs, ct)]
            if sem['SEM_SCALE'] > 0:
                tl += [(sem_s1, sem_t1, sem, pred_s1, pred_t1), (sem_s2, sem_t2, sem, pred_s2, pred_t2)]
            vals = mmd.soft_mmd_multi(label, label_t, tl)
        else:
            terms = [lambda: self._mmd(label, feat_node_s, label_t, feat_node_t, geo, cs, ct)]
            if sem['SEM_SCALE'] > 0:
                terms += [lambda: self._mmd(label, sem_s1, label_t, sem_t1, sem, pred_s1, pred_t1),
                          lambda: self._mmd(label, sem_s2, label_t, sem_t2, sem, pred_s2, pred_t2)]
            vals = ops.run_parallel(terms)
        if combine and self._combine_tail and loss_cls.is_cuda:
            # total and its two reported parts in one launch each way (ops.loss_combine): _eager_step takes the total
            wg, wsem = M['MMD_WEIGHT'] * geo['GEO_SCALE'], 0.5 * M['MMD_WEIGHT'] * sem['SEM_SCALE']
            has_sem = sem['SEM_SCALE'] > 0
            tot, lg, ls = ops.loss_combine(loss_cls, vals[0], vals[1] if has_sem else None, vals[2] if has_sem else None, wg, wsem)
            self._total = tot
            return loss_cls, lg, (ls if has_sem else None)
        loss_geo = M['MMD_WEIGHT'] * geo['GEO_SCALE'] * vals[0]
        loss_sem = None
        if sem['SEM_SCALE'] > 0:
            loss_sem = (0.5 * M['MMD_WEIGHT'] * sem['SEM_SCALE']) * (vals[1] + vals[2])
        return loss_cls, loss_geo, loss_sem

    # ------------------------------------------------------------------ step
    def step(self, data, label, data_t, label_t, epoch=0):
        """Returns (loss_cls, loss_geo_mmd, loss_sem_mmd) as 0-d device tensors (no host sync).
        In graph mode the returned tensors are static buffers overwritten by the next step."""
        if self.use_graph:
            return self._graph_step(data, label, data_t, label_t, epoch)
        if self.world > 1 and os.environ.get('SUG_SEGMENTED_EAGER') == '1':
            return self._run_segments(self._segments(data, label, data_t, label_t, epoch), {})
        return self._eager_step(data, label, data_t, label_t, epoch)

    def _opts(self):
        return (self.optimizer_g, self.optimizer_c, self.optimizer_dis)

    def _graph_key(self, mmd_on, tensors):
        """Everything a captured step bakes in by value.  sug_amd.optim.Adam keeps lr on the device (one graph serves a
        whole schedule); any other optimizer takes lr by value, so its lr of every group is part of the key."""
        hyp = []
        for o in self._opts():
            if hasattr(o, 'graph_key'):
                hyp.append(o.graph_key())
            else:
                hyp.append(tuple((g['lr'], tuple(g['betas']), g['eps'], g['weight_decay']) for g in o.param_groups))
        return (mmd_on, self.single_pass, tuple(tuple(t.shape) for t in tensors), tuple(hyp))

    def _plan_generations(self):
        return tuple(getattr(o, 'plan_generation', 0) for o in self._opts() + (self._adam_chain,))

    def _optimizers_step(self):
        if self._adam_chain is not None:
            self._adam_chain.step()
        else:
            self.optimizer_dis.step()
            self.optimizer_g.step()
            self.optimizer_c.step()

    def drop_graphs(self):
        """Forget every captured step (their private pools are released).  Called automatically when an optimizer's
        device-side plan changed under a graph (load_state_dict, another set of parameters with gradients)."""
        self._graphs = None

    def _graph_step(self, data, label, data_t, label_t, epoch):
        """hipGraph mode (bench.py's default launch mode on one GPU; opt-in for callers: use_graph=True).  One captured
        graph per configuration key = (MMD on/off, input shapes, the by-value hyper-parameters of the optimizers).  The
        learning rates are NOT part of the key: sug_amd.optim.Adam reads them from device memory, so one graph serves a
        whole schedule (`set_epoch`).  The first step of a key runs eagerly (it records the FPS start-draw plan and
        builds the Adam update plans OUTSIDE any capture), the second captures, later ones replay.  At most
        `max_graphs` captured steps are kept (least recently used first out); a graph whose optimizers have rebuilt
        their device-side plan since the capture (load_state_dict, ...) is dropped and captured again."""
        mmd_on = epoch >= self.methods['PURE_CLS_EPOCH']
        key = self._graph_key(mmd_on, (data, label, data_t, label_t))
        if self._graphs is None:
            self._graphs = {}
        for o in self._opts():                              # a schedule step since the last replay: new lr -> device
            if hasattr(o, 'refresh_device_scalars'):
                o.refresh_device_scalars()
        st = self._graphs.get(key)
        if st is not None and st.get('gens') is not None and st['gens'] != self._plan_generations():
            del self._graphs[key]                           # raw pointers into a freed Adam plan: never replay
            st = None
        if self.segmented:
            if st is None:
                while len(self._graphs) >= self.max_graphs:
                    self._graphs.pop(next(iter(self._graphs)))
            return self._segmented_step(st, key, data, label, data_t, label_t, epoch)
        if st is None:
            while len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))  # dicts keep insertion order; a hit re-inserts (below)
            st = {'feeder': _StartFeeder(data.device), 'graph': None, 'gens': None}
            self._graphs[key] = st
            ops.CTX.start_provider = st['feeder'].record
            try:
                out = self._eager_step(data, label, data_t, label_t, epoch)
            finally:
                ops.CTX.start_provider = None
            st['feeder'].build()
            return out
        self._graphs[key] = self._graphs.pop(key)           # most recently used last
        if st['graph'] is None:
            st['in'] = [t.clone() for t in (data, label, data_t, label_t)]
            for o in self._opts():
                o.zero_grad(set_to_none=True)
            st['feeder'].cursor = 0
            st['graph'] = torch.cuda.CUDAGraph()
            if os.environ.get('SUG_GRAPH_DUMP'):
                st['graph'].enable_debug_mode()
            ops.CTX.start_provider = st['feeder'].provide
            try:
                with ops.capture_guard(), torch.cuda.graph(st['graph']):
                    st['out'] = self._eager_step(*st['in'], epoch)
            finally:
                ops.CTX.start_provider = None
            st['gens'] = self._plan_generations()
            if os.environ.get('SUG_GRAPH_DUMP'):
                st['graph'].debug_dump(os.environ['SUG_GRAPH_DUMP'])
        for dst, src in zip(st['in'], (data, label, data_t, label_t)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        st['feeder'].refill()
        if self._tick is not None:
            # Historical guard (SUG_GRAPH_GUARD=1): an early round-1 version of the step faulted in its second
            # back-to-back replay (a torch scatter kernel that has since left the step read out-of-range indices; an
            # eager op between replays avoided it).  Never reproduced after the memset nodes were removed (DESIGN
            # section 5); tests/test_gpu_graph.py soaks 300 back-to-back replays without it.
            self._tick.add_(1)
        st['graph'].replay()
        return st['out']

    # ------------------------------------------------------------------ multi-rank step in graph segments
    def _seg_check(self, data, data_t, mmd_on):
        M = self.methods
        geo, sem = M['GEO_MMD'][0], M['SEM_MMD'][0]
        ok = self.pair_domains and data.shape == data_t.shape and self.model.training and M['ADV_WEIGHT'] <= 0 and \
            M['TARGET_LOSS'] <= 0
        if mmd_on and M['MMD_WEIGHT'] > 0:
            ok = ok and self.global_mmd and geo['NAME'] == 'SOFT_MMD' and sem['NAME'] == 'SOFT_MMD' and sem['SEM_SCALE'] > 0
        if not ok:
            raise RuntimeError('SUGStep(use_graph=True) on %d ranks captures the step in segments around its collectives; '
                               'that form covers paired domains + global soft MMD (or MMD off) with ADV_WEIGHT = '
                               'TARGET_LOSS = 0 -- use use_graph=False for other METHODS' % self.world)

    def _segments(self, data, label, data_t, label_t, epoch):
        """The step of a batch-sharded run as five device segments (closures over a state dict S) with the four
        collectives BETWEEN them, so that each segment can be captured into a hipGraph and the collectives stay
        eager RCCL calls (no collective inside a capture):
            A  both paired encoder passes, heads, CE loss; everything the global MMD needs from the other ranks packed
               column-wise into one [m_local, W] buffer                         -> all-gather (values)
            B  SDA weights of the gathered batch, label-augmented operands, this rank's row block of the three kernel
               matrices -> 9 partial sums                                        -> all-reduce (9 doubles)
            C1 the three MMD values, total loss, backward of everything above the encoder (heads, attention layers): their
               gradients packed into flat bucket 1                               -> all-reduce (async: runs under C2)
            C2 the encoder's backward from the gradients at its outputs, bucket 2 -> wait for bucket 1, all-reduce
            D  / world, gradients as views of the flat buffers, three Adam updates.
        Same arithmetic as the eager multi-rank step (train_dg.py:216-217, :357-368 is the reference's seam); what
        changes is the launch: five replays + four collectives per step instead of ~350 launches from Python."""
        M_ = self.methods
        mmd_on = epoch >= M_['PURE_CLS_EPOCH'] and M_['MMD_WEIGHT'] > 0
        self._seg_check(data, data_t, mmd_on)
        model, world, rank = self.model, self.world, dist.get_rank()
        geo, sem = M_['GEO_MMD'][0], M_['SEM_MMD'][0]
        mloc = label.shape[0]
        gloo_gpu = dist.get_backend() == 'gloo'

        def seg_a(S):
            from .model import Ptran_transformer as _PT
            ops.CTX.w16_cache = ops.w16_prefill(getattr(self, '_w16_plan', None) or []) if _PT.GEMM_DTYPE is not None else None
            fused_before, ops.CTX.fused_heads = ops.CTX.fused_heads, (self.fused_heads or ops.CTX.fused_heads)
            par_before, ops.CTX.parallel_branches = ops.CTX.parallel_branches, (self.parallel_branches or ops.CTX.parallel_branches)
            try:
                model._cuts = S['cuts'] = []
                pair = torch.cat((data, data_t), dim=0)
                dual = self.single_pass and mmd_on
                if mmd_on and not dual and hasattr(model, 'plan_pair_geometry') and os.environ.get('SUG_PLAN_GEOMETRY', '1') != '0':
                    model.plan_pair_geometry(pair, passes=2)
                if dual:
                    (p_s1, p_s2, f_s1, f_s2), (p_t1, p_t2, f_t1, f_t2), (node_s, node_t) = model.forward_pair(pair, dual=True)
                else:
                    (p_s1, p_s2, f_s1, f_s2), (p_t1, p_t2, f_t1, f_t2) = model.forward_pair(pair)
                S['loss_cls'] = (0.5 * M_['SRC_LOSS_WEIGHT'] * M_['CLS_WEIGHT']) * (self.criterion(p_s1, label) +
                                                                                   self.criterion(p_s2, label))
                if mmd_on:
                    if not dual:
                        node_s, node_t = model.forward_pair(pair, node_adaptation=True)
                    S['loc'] = (node_s, node_t, f_s1, f_t1, f_s2, f_t2)
                    vals = [label, label_t, node_s.detach(), node_t.detach(), f_s1.detach(), f_t1.detach(), f_s2.detach(),
                            f_t2.detach(), p_s1.detach(), p_t1.detach(), p_s2.detach(), p_t2.detach()]
                    if geo.get('GEO_WEIGHTS'):
                        vals.append(mmd.chamfer_distances(data, data_t).reshape(-1, 1))
                    cols, meta = [], []
                    for t in vals:
                        t2 = t.reshape(t.shape[0], -1)
                        meta.append((t2.shape[1], t.dtype, tuple(t.shape[1:])))
                        cols.append(t2 if t2.dtype == torch.float32 else t2.to(torch.float32))
                    S['meta'] = meta
                    S['packed'] = torch.cat(cols, dim=1)
            finally:
                model._cuts = None
                ops.CTX.fused_heads = fused_before
                ops.CTX.parallel_branches = par_before
                if ops.CTX.w16_cache is not None:
                    self._w16_plan = ops.w16_plan(ops.CTX.w16_cache)
                ops.CTX.w16_cache = None

        def col_gather(S):
            if not mmd_on:
                return
            G = S['static']['G']
            if gloo_gpu:                                  # rehearsal backend: gathers device tensors as a list
                dist.all_gather(list(G.chunk(world, dim=0)), S['packed'])
            else:
                dist.all_gather_into_tensor(G, S['packed'])

        def seg_b(S):
            if not mmd_on:
                return
            G, off, g = S['static']['G'], 0, []
            for w, dt, tail in S['meta']:
                t = G[:, off:off + w]
                off += w
                if dt != torch.float32:
                    t = t.round().to(dt)
                g.append(t.reshape((G.shape[0],) + tail))
            label_g, label_tg, fn_s, fn_t, g_s1, g_t1, g_s2, g_t2, g_p1s, g_p1t, g_p2s, g_p2t = g[:12]
            node_s, node_t, f_s1, f_t1, f_s2, f_t2 = S['loc']
            M_all, row0 = label_g.shape[0], rank * mloc
            sums = S['static']['sums']
            sums.fill_(0.0)                               # (a fill kernel: no memset node in the captured segment)
            terms = []
            w_geo = mmd.distance2weights(g[12].reshape(-1), geo['GEO_WEIGHTS']).reshape(1, -1) if geo.get('GEO_WEIGHTS') else None
            specs = [(node_s, node_t, fn_s, fn_t, float(geo['LABEL_SCALE']), w_geo)]
            for fs, ft, gs, gt, gps, gpt in ((f_s1, f_t1, g_s1, g_t1, g_p1s, g_p1t), (f_s2, f_t2, g_s2, g_t2, g_p2s, g_p2t)):
                w = mmd.sda_weights_of(sem, gps, gpt, label_g, label_tg)  # mmd_cal's rules (GEO > ENTROPY > SEM)
                specs.append((fs, ft, gs, gt, float(sem['LABEL_SCALE']), w))
            for i, (fs, ft, gs, gt, lsc, w) in enumerate(specs):
                Zloc = ops.mmd_assemble(fs, ft, label, label_t, lsc)
                Zall = ops.mmd_assemble(gs.detach(), gt.detach(), label_g, label_tg, lsc)
                Za, wt = ops.mmd_rows_local_sums(Zall, M_all, row0, mloc, w, sums[i])
                terms.append((Zloc, Za, wt))
            S['terms'], S['M_all'], S['row0'] = terms, M_all, row0

        def sum_reduce(S):
            if mmd_on:
                dist.all_reduce(S['static']['sums'])

        def _pack(ps, key, S):
            if key not in S['static']:
                S['static'][key] = torch.empty(sum(p.numel() for p in ps), dtype=torch.float32, device=data.device)
            return S['static'][key]

        def seg_c1(S):
            """MMD values, total loss, backward of everything ABOVE the encoder: heads, attention layers, losses.  Their
            gradients (38 of DGCNN's 46 MB) go into flat bucket 1, whose all-reduce then runs under segment C2."""
            loss_cls, loss_geo, loss_sem = S['loss_cls'], None, None
            loss = loss_cls
            if mmd_on:
                sums = S['static']['sums']
                v = [ops.mmd_from_reduced_sums(Zloc, sums[i], Za, wt, mloc, S['M_all'], S['row0'], world)
                     for i, (Zloc, Za, wt) in enumerate(S['terms'])]
                loss_geo = (M_['MMD_WEIGHT'] * geo['GEO_SCALE']) * v[0]
                loss_sem = (0.5 * M_['MMD_WEIGHT'] * sem['SEM_SCALE']) * (v[1] + v[2])
                loss = loss + loss_geo + loss_sem
            cuts = [t for t in S['cuts'] if t is not None and t.requires_grad]
            st_ = S['static']
            early = st_.get('early')
            if early is None:                             # (planning step) the parameters above the cut
                early = [p for m in (model.c1, model.c2, model.attention_s, model.attention_t) for p in m.parameters()
                         if p.requires_grad]
            grads = torch.autograd.grad(loss, early + cuts, allow_unused=True)
            ge, gc = grads[:len(early)], grads[len(early):]
            if 'early' not in st_:
                keep = [i for i, g in enumerate(ge) if g is not None]
                st_['early'] = [early[i] for i in keep]
                ge = [ge[i] for i in keep]
            S['cut_grads'] = [(t, g) for t, g in zip(cuts, gc) if g is not None]
            torch.cat([g.reshape(-1) for g in ge], out=_pack(st_['early'], 'flat1', S))
            S['out'] = (loss_cls.detach(), None if loss_geo is None else loss_geo.detach(),
                        None if loss_sem is None else loss_sem.detach())

        def grad_reduce1(S):
            if self.collective_events is not None:
                S['ev_b1'] = torch.cuda.Event(enable_timing=True)
                S['ev_b1'].record()
            S['work'] = dist.all_reduce(S['static']['flat1'], async_op=True)      # in flight under segment C2

        def seg_c2(S):
            """The encoder's backward from the gradients at the cut; its parameter gradients -> flat bucket 2."""
            cg = S['cut_grads']
            torch.autograd.backward([t for t, _ in cg], [g for _, g in cg])
            ops.clear_rows_cache()
            if self.share_prefix:
                model.g.clear_prefix_cache()
            for m in self._split_layers:
                m._wcat = None
            st_ = S['static']
            if 'late' not in st_:
                st_['late'] = [p for p in model.g.parameters() if p.grad is not None]
            torch.cat([p.grad.reshape(-1) for p in st_['late']], out=_pack(st_['late'], 'flat2', S))

        def grad_reduce2(S):
            w = S.pop('work', None)
            ev = self.collective_events
            if ev is not None:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()                               # segment C2 (the encoder's backward) has finished here
            if w is not None:
                w.wait()                                  # the compute stream waits for bucket 1
            if ev is not None:
                e1.record()                               # e0 -> e1: the part of bucket 1 that C2 did not hide
            dist.all_reduce(S['static']['flat2'])
            if ev is not None:
                e2.record()
                ev.setdefault('bucket1_exposed', []).append((e0, e1))
                ev.setdefault('bucket2_all_reduce', []).append((e1, e2))
                if 'ev_b1' in S:                          # launch -> done, i.e. bucket 1 AND whatever of C2 ran beside it
                    ev.setdefault('bucket1_launch_to_done', []).append((S.pop('ev_b1'), e1))

        def seg_d(S):
            st_ = S['static']
            for ps, flat in ((st_['early'], st_['flat1']), (st_['late'], st_['flat2'])):
                flat.div_(world)
                off = 0
                for p in ps:                              # the averaged gradients stay in the flat buffers
                    n = p.numel()
                    p.grad = flat[off:off + n].view_as(p)
                    off += n
            self._optimizers_step()
            self.optimizer_g.zero_grad()
            self.optimizer_c.zero_grad()
            self.optimizer_dis.zero_grad()

        return {'device': (seg_a, seg_b, seg_c1, seg_c2, seg_d), 'collective': (col_gather, sum_reduce, grad_reduce1, grad_reduce2),
                'mloc': mloc,
                'mmd_on': mmd_on, 'device_of': data.device}

    def _seg_static(self, S, segs):
        """Buffers the collectives read / write: allocated once per configuration, OUTSIDE any graph pool."""
        st = S.setdefault('static', {})
        if segs['mmd_on'] and 'sums' not in st:
            st['sums'] = torch.zeros(3, 3, dtype=torch.float64, device=segs['device_of'])
        return st

    def _run_segments(self, segs, S):
        """Uncaptured execution: segment, collective, segment, ...  (the planning step of graph mode; with
        SUG_SEGMENTED_EAGER=1 every step -- the segmented formulation without graphs)."""
        self._seg_static(S, segs)
        dev, col = segs['device'], segs['collective']
        for i, fn in enumerate(dev):
            fn(S)
            if i == 0 and segs['mmd_on'] and 'G' not in S['static']:
                S['static']['G'] = torch.empty(self.world * segs['mloc'], S['packed'].shape[1], dtype=torch.float32,
                                               device=segs['device_of'])
            if i < len(col):
                col[i](S)
        return S['out']

    def _segmented_step(self, st, key, data, label, data_t, label_t, epoch):
        """Graph mode on more than one rank: plan (uncaptured), capture the segments into hipGraphs that share one memory
        pool, then per step: replay A, all-gather, replay B, all-reduce, replay C1, all-reduce (async), replay C2, all-reduce, replay D."""
        if st is None:
            st = {'feeder': _StartFeeder(data.device), 'graphs': None, 'gens': None, 'S': {}}
            self._graphs[key] = st
            ops.CTX.start_provider = st['feeder'].record
            try:
                out = self._run_segments(self._segments(data, label, data_t, label_t, epoch), st['S'])
            finally:
                ops.CTX.start_provider = None
            st['feeder'].build()
            st['S'] = {'static': st['S']['static']}         # keep only the collective buffers
            return out
        self._graphs[key] = self._graphs.pop(key)
        if st['graphs'] is None:
            st['in'] = [t.clone() for t in (data, label, data_t, label_t)]
            for o in self._opts():
                o.zero_grad(set_to_none=True)
            st['feeder'].cursor = 0
            segs = self._segments(*st['in'], epoch)
            S = st['S']
            graphs, pool, err = [], None, None
            ops.CTX.start_provider = st['feeder'].provide
            try:
                for fn in segs['device']:
                    g = torch.cuda.CUDAGraph()
                    # thread_local: the process group's watchdog thread queries events of finished collectives while
                    # we capture; in the default (global) mode such a call from ANOTHER thread invalidates the capture
                    with ops.capture_guard(), torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):
                        fn(S)
                    pool = g.pool() if pool is None else pool
                    graphs.append(g)
            except RuntimeError as e:                      # (reported below, after every rank has been heard)
                err = e
            finally:
                ops.CTX.start_provider = None
            # the ranks must agree before the first replay: a rank whose capture failed would otherwise meet the others'
            # all-gather with a different collective.  One flag all-reduce per capture (not per step).
            ok = torch.tensor([0.0 if err is not None else 1.0], device=data.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                self._graphs.pop(key, None)
                for o in self._opts():                    # gradients of the aborted capture point into its discarded pool
                    o.zero_grad(set_to_none=True)
                if self.share_prefix:
                    self.model.g.clear_prefix_cache()
                raise RuntimeError('segmented hipGraph capture failed on %s: %s' % (
                    'this rank' if err is not None else 'another rank', str(err).splitlines()[0] if err is not None else ''))
            st['graphs'], st['segs'], st['out'] = graphs, segs, S['out']
            st['gens'] = self._plan_generations()
            # autograd objects of the capture are no longer needed: the graphs own the memory
            for k in ('loc', 'terms', 'loss_cls', 'cuts', 'cut_grads'):
                S.pop(k, None)
        for dst, src in zip(st['in'], (data, label, data_t, label_t)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        st['feeder'].refill()
        S, col = st['S'], st['segs']['collective']
        ev = self.collective_events
        for i, (g, c) in enumerate(zip(st['graphs'], col + (None,))):
            g.replay()
            if c is not None:
                if ev is not None and i < 2:              # all-gather, 9-double all-reduce: synchronous, bracketed here
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    c(S)
                    e1.record()
                    ev.setdefault(('all_gather_packed', 'sums_all_reduce')[i], []).append((e0, e1))
                else:
                    c(S)
        return st['out']

    def collective_summary(self):
        """Mean milliseconds per step of every collective recorded in `collective_events` (events on the compute stream:
        a synchronous collective makes that stream wait for the communication stream, so the pair brackets it), and the
        message sizes.  bucket1_exposed = the wait for bucket 1 AFTER the encoder backward it runs under."""
        ev = self.collective_events or {}
        torch.cuda.synchronize()
        out = {k: round(sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1), 4) for k, v in ev.items()}
        sizes = {}
        for st in (self._graphs or {}).values():
            stat = (st.get('S') or {}).get('static') or {}
            for name, key in (('all_gather_packed', 'G'), ('sums_all_reduce', 'sums'), ('bucket1', 'flat1'), ('bucket2', 'flat2')):
                if key in stat:
                    sizes[name] = stat[key].numel() * stat[key].element_size()
        return {'ms': out, 'bytes': sizes}

    def _eager_step(self, data, label, data_t, label_t, epoch=0):
        mmd_on = epoch >= self.methods['PURE_CLS_EPOCH']
        from .model import Ptran_transformer as _PT
        # 16-bit weight copies shared by this step's forwards; from the second step on they are refreshed by one
        # multi-tensor copy into the first step's buffers
        ops.CTX.w16_cache = ops.w16_prefill(getattr(self, '_w16_plan', None) or []) if _PT.GEMM_DTYPE is not None else None
        fused_before, ops.CTX.fused_heads = ops.CTX.fused_heads, (self.fused_heads or ops.CTX.fused_heads)
        par_before, ops.CTX.parallel_branches = ops.CTX.parallel_branches, (self.parallel_branches or ops.CTX.parallel_branches)
        try:
            with ops.deferred_bn_counts():            # every num_batches_tracked increment of the step's forwards: one launch
                loss_cls, loss_geo, loss_sem = self.losses(data, label, data_t, label_t, mmd_on, combine=True)
        finally:
            ops.CTX.fused_heads = fused_before
            ops.CTX.parallel_branches = par_before
            if ops.CTX.w16_cache is not None:
                self._w16_plan = ops.w16_plan(ops.CTX.w16_cache)
            ops.CTX.w16_cache = None
        loss = getattr(self, '_total', None)             # set by losses() when the tail was combined in one launch
        self._total = None
        if loss is None:
            loss = loss_cls
            if loss_geo is not None:
                loss = loss + loss_geo
            if loss_sem is not None:
                loss = loss + loss_sem
        if self.world > 1 and self.reducer is None:      # a graph-mode trainer switched to eager launches
            early = [p for m in (self.model.c1, self.model.c2, self.model.attention_s, self.model.attention_t) for p in m.parameters()]
            self.reducer = GradReducer([early, list(self.model.g.parameters())], self.world)
        if self.reducer is not None:
            self.reducer.begin(mmd_on)
        loss.backward()
        ops.clear_rows_cache()                           # the [2B,N,3] rows of this step's batch: not kept beyond the step
        if self.share_prefix:
            self.model.g.clear_prefix_cache()
        for m in self._split_layers:
            m._wcat = None
        if self.reducer is not None:
            self.reducer.finish()
        self._optimizers_step()
        self.optimizer_g.zero_grad()
        self.optimizer_c.zero_grad()
        self.optimizer_dis.zero_grad()
        return loss_cls.detach(), None if loss_geo is None else loss_geo.detach(), \
            None if loss_sem is None else loss_sem.detach()

"""Per-call hipGraph replay behind `Net_MDA.forward` -- the caller-unchanged form at graph speed.

The reference's training loop (train_dg_single_gpu.py:260-264, :309-335) calls `model(...)` four times per step -- semantic
pass on the source and on the target batch, node-adaptation pass on each -- and runs ONE backward over all of them.  Launched
kernel by kernel from Python that form is host-bound on an MI355X (~480 launches in ~8.2 ms for 6.8 ms of kernels).  This
module keeps the caller's code unchanged and removes the host from the picture:

  * the first train-mode call of a (flags, input shape) combination runs eagerly and records which farthest-point-sampling
    start draws it makes; once that call's backward has been seen (every library GEMM shape of the key has then run outside
    a capture), the next call is CAPTURED -- the forward into one hipGraph, its backward (torch.autograd.grad from static
    gradient buffers at the outputs) into a second one, both in one private memory pool -- and replayed; every later call
    is: copy the batch into the static input, refill the FPS starts (same CPU-generator draws, in call order, through a
    pinned double buffer), replay;
  * an `autograd.Function` stands in for the call in the caller's autograd graph: its backward stages the incoming
    gradients, replays the backward graph and hands the parameter gradients to `.grad` itself -- the first contribution
    since the caller's zero_grad() IS the static gradient tensor (no copy), later ones are added in one multi-tensor launch:
    no `AccumulateGrad` node, no `at::add` per parameter and call;
  * two calls on the same batch share the encoder prefix as the eager form does (`share_prefix = 'auto'`): the first call's
    graph exposes the prefix tensors as hidden outputs of its Function, the second call's graph was captured reading them in
    place, its backward hands the prefix gradients to the first call's backward through autograd;
  * as many graph instances per key as calls are in flight between two backwards (source and target semantic pass = two
    instances of one key); an instance is reused once its backward has run (or its outputs were dropped).

Same kernels, same arithmetic, same order of CPU-generator draws as the eager call-by-call form: losses and parameters are
bit-identical (tests/test_gpu_call_graphs.py).  Whenever a capture is refused the key falls back to eager launches in the
same process.  Not used: under torch.no_grad / eval mode, inside another capture, with more than one rank (gradient hooks
of DDP would not fire), or when SUG_CALL_GRAPHS=0.
"""
import os
import weakref

import torch

from . import ops

ENABLED = os.environ.get('SUG_CALL_GRAPHS', '1') != '0'
MAX_INSTANCES_PER_KEY = 4


class StartFeeder:
    """FPS start indices for a replayable forward / step: drawn from the CPU default generator in call
    order with the same (B, N) sequence as an eager run (so the random stream is the
    reference's, model/point_utils.py:17), but delivered through one static device buffer."""

    def __init__(self, device):
        self.device = device
        self.plan = []          # (B, N) per farthest_point_sample call
        self.host = self.dev = None
        self.cursor = 0

    def record(self, B, N):     # provider during the eager planning run
        self.plan.append((B, N))
        return torch.randint(0, N, (B,), dtype=torch.long)

    def build(self):
        total = max(sum(b for b, _ in self.plan), 1)
        # two pinned staging buffers, used alternately: the host must not overwrite one while its
        # asynchronous copy to the device may still be pending (replays are not synchronised)
        self.host = [torch.empty(total, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.done = [None, None]
        self.turn = 0
        self.dev = torch.zeros(total, dtype=torch.int32, device=self.device)

    def refill(self):           # before every replay
        if not self.plan:
            return
        h = self.host[self.turn]
        if self.done[self.turn] is not None:
            self.done[self.turn].synchronize()
        off = 0
        for B, N in self.plan:
            h[off:off + B] = torch.randint(0, N, (B,), dtype=torch.long).to(torch.int32)
            off += B
        self.dev.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.done[self.turn] = ev
        self.turn ^= 1

    def provide(self, B, N):    # provider during capture
        off = sum(b for b, _ in self.plan[:self.cursor])
        assert self.cursor < len(self.plan) and self.plan[self.cursor] == (B, N), \
            'forward structure changed between planning and capture'
        self.cursor += 1
        return self.dev[off:off + B]


class _Unsupported(RuntimeError):
    pass


class _KeyState:
    __slots__ = ('plan', 'instances', 'eager_only', 'bwd_seen', 'why')

    def __init__(self):
        self.plan, self.instances, self.eager_only, self.bwd_seen, self.why = None, [], False, [False], None


class _Lease:
    """Lives in the autograd node of a replayed call: when the node dies (backward done and graph freed, or the caller
    dropped the outputs) the instance is free again."""
    __slots__ = ('inst', 'gen')

    def __init__(self, inst):
        self.inst, self.gen = inst, inst.generation

    def __del__(self):
        if self.inst.generation == self.gen:
            self.inst.busy = False


class _Instance:
    """One captured (forward graph, backward graph) pair of a call key."""

    def __init__(self, mgr, ks, dep):
        self.mgr, self.ks, self.dep = mgr, ks, dep
        self.generation = 0
        self.busy = False
        self.graph_f = self.graph_b = None
        self.x = None                   # static input
        self.feeder = None
        self.outs = None                # static user-visible outputs (detached aliases of the captured tensors)
        self.single = False
        self.prefix_static = ()         # static prefix tensors this call exports (hidden Function outputs)
        self.prefix_extra = None
        self.gouts = None               # static gradient buffers at outs + prefix_static
        self.gdirty = None
        self.used = None                # [(parameter, its view of the manager's flat gradient buffer)]
        self.leaf_grads = ()            # static gradients w.r.t. the imported prefix tensors (None where unused)

    def free(self):
        if self.busy:
            return False
        for ks in self.mgr.keys.values():
            for i in ks.instances:
                if i.dep is self and i.busy:
                    return False        # an importer's backward still reads this instance's prefix tensors
        return True


class _Call(torch.autograd.Function):
    """The replayed call in the caller's autograd graph."""

    @staticmethod
    def forward(ctx, inst, token, *prefix_in):
        ctx.inst, ctx.gen = inst, inst.generation
        ctx.lease = _Lease(inst)
        ctx.set_materialize_grads(False)
        # the caller's tensors are copies (a few KB): they must survive the instance's next replay
        outs = [torch.empty_like(o) for o in inst.outs]
        torch._foreach_copy_(outs, inst.outs)
        return tuple(outs) + tuple(t.detach() for t in inst.prefix_static)

    @staticmethod
    def backward(ctx, *gs):
        inst = ctx.inst
        if ctx.gen != inst.generation:
            raise RuntimeError('sug_amd call graphs: backward through a Net_MDA call whose graph instance has been replayed '
                               'for a later call since (more than %d calls of one kind in flight without a backward); set '
                               'model.call_graphs = False for this training loop' % MAX_INSTANCES_PER_KEY)
        inst.mgr._backward(inst, gs)
        inst.busy = False
        return (None, None) + tuple(inst.leaf_grads)


def _none():
    return None


class CallGraphs:
    def __init__(self, model):
        self._model = weakref.ref(model)
        self.keys = {}
        self.sig = None
        self.token = None
        self.stats = {'eager': 0, 'captured': 0, 'replayed': 0, 'refused': 0}

    # a model copy (copy.deepcopy(model), train_dg_single_gpu.py:364) or a pickled model starts without graphs
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_none, ())

    # ------------------------------------------------------------------ validity of everything captured
    def _signature(self, model):
        ps = self._params
        return (ps[0].data_ptr(), ps[-1].data_ptr(), ps[len(ps) // 2].data_ptr(),
                tuple(p.requires_grad for p in ps), tuple(m.training for m in self._modules))

    def reset(self):
        """Forget every captured graph (parameters moved / frozen, sub-modules switched between train and eval)."""
        model = self._model()
        self.keys = {}
        self._params = list(model.parameters())
        self._modules = list(model.modules())
        self._slots = [(m, n, q) for m in self._modules for n, q in m._parameters.items() if q is not None]
        self.sig = self._signature(model)
        self.token = None

    def _prepare(self, dev):
        if self.token is None:
            self._req = [p for p in self._params if p.requires_grad]
            self.token = torch.zeros((), dtype=torch.float32, device=dev, requires_grad=True)

    # ------------------------------------------------------------------ the call
    def _eligible(self, model, x):
        if not (x.is_cuda and not x.requires_grad and ops.CTX.bn_groups == 1 and ops.CTX.start_queue is None and
                ops.CTX.geometry_plan is None and ops.CTX.start_provider is None and ops.CTX.profile is None and
                ops.CTX.bn_record is None):
            return False
        if torch.cuda.is_current_stream_capturing():
            return False
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            return False
        return True

    def _key(self, model, x, flags):
        from .model import Ptran_transformer as PT
        g = model.g
        return (flags, tuple(x.shape), x.device.index, ops.CTX.fused_heads, ops.CTX.parallel_branches, PT.GEMM_DTYPE,
                getattr(PT, 'PROJ_16BIT', None), getattr(g, 'share_prefix', None), model.dual_output_on_both_flags,
                model.dual_updates_bn_twice)

    def call(self, x, flags):
        model = self._model()
        if not self._eligible(model, x):
            return model._forward_impl(x, *flags)
        if self.sig is None or self._signature(model) != self.sig:
            self.reset()
        key = self._key(model, x, flags)
        ks = self.keys.get(key)
        if ks is None:
            ks = self.keys[key] = _KeyState()
        if ks.eager_only:
            self.stats['eager'] += 1
            return model._forward_impl(x, *flags)
        g = model.g
        sharing = hasattr(g, 'find_prefix') and bool(getattr(g, 'share_prefix', False))
        entry = g.find_prefix(x) if sharing else None
        dep = None
        if entry is not None:
            dep = entry.provider
            if dep is None or dep.mgr is not self:
                # a live prefix of an EAGER call on this batch: share it through autograd, eagerly
                return self._eager(model, ks, x, flags)
        if ks.plan is None or not ks.bwd_seen[0]:
            return self._eager(model, ks, x, flags)
        inst = None
        for i in ks.instances:
            if i.dep is dep and i.free():
                inst = i
                break
        if inst is None:
            if len(ks.instances) >= MAX_INSTANCES_PER_KEY:
                return self._eager(model, ks, x, flags)
            try:
                inst = self._capture(model, ks, x, flags, dep)
            except Exception as e:      # capture refused: this key stays eager, in this process
                ops.CTX.start_provider = None
                ks.eager_only, ks.why = True, '%s: %s' % (type(e).__name__, str(e).splitlines()[0] if str(e) else '')
                self.stats['refused'] += 1
                if os.environ.get('SUG_CALL_GRAPHS_STRICT') == '1':
                    raise
                return model._forward_impl(x, *flags)
            ks.instances.append(inst)
            self.stats['captured'] += 1
            if os.environ.get('SUG_CG_TRACE'):
                print('[call_graphs] captured', flags, 'dep', None if dep is None else id(dep) % 10000, 'existing',
                      [(id(i) % 10000, None if i.dep is None else id(i.dep) % 10000, i.busy, i.free()) for i in ks.instances], flush=True)
        return self._replay(model, inst, x, entry)

    def _eager(self, model, ks, x, flags):
        self.stats['eager'] += 1
        if ks.plan is not None and ks.bwd_seen[0]:
            return model._forward_impl(x, *flags)
        feeder = StartFeeder(x.device)
        ops.CTX.start_provider = feeder.record
        try:
            out = model._forward_impl(x, *flags)
        finally:
            ops.CTX.start_provider = None
        ks.plan = feeder.plan
        # a capture of this key waits until one eager backward has run (every GEMM shape of the key has then been looked
        # up outside a capture): any output's gradient hook says the backward has started, and the next forward call
        # cannot come before it has finished
        seen = ks.bwd_seen
        for t in ([out] if isinstance(out, torch.Tensor) else (out or ())):
            if isinstance(t, torch.Tensor) and t.requires_grad:
                t.register_hook(lambda g_, s=seen: s.__setitem__(0, True))
        return out

    # ------------------------------------------------------------------ capture
    def _capture(self, model, ks, x, flags, dep):
        dev = x.device
        self._prepare(dev)
        g = model.g
        inst = _Instance(self, ks, dep)
        inst.x = x.detach().clone()
        inst.feeder = StartFeeder(dev)
        inst.feeder.plan = list(ks.plan)
        inst.feeder.build()
        leaves = []
        if dep is not None:
            leaves = [t.detach().requires_grad_() for t in dep.prefix_static]
            g.install_prefix(inst.x, leaves, dep.prefix_extra)
        exported = None
        inst.graph_f = torch.cuda.CUDAGraph()
        # The captured forward runs on detached ALIASES of the parameters (same storage, fresh autograd leaves): the
        # gradients of the backward capture are then taken w.r.t. leaves whose AccumulateGrad nodes are created inside the
        # capture, on the capture stream.  The parameters' own AccumulateGrad nodes usually survive from the previous step
        # (the caller's `loss` keeps that graph alive until it is reassigned) and live on the stream of that step: routing a
        # captured gradient to them makes the engine synchronise the capture stream with the default stream, which
        # invalidates the capture (torch warns 'AccumulateGrad node's stream does not match'; hipStreamEndCapture crashed).
        alias = {id(q): q.detach().requires_grad_(q.requires_grad) for q in self._params}
        ops.CTX.start_provider = inst.feeder.provide
        # 16-bit weight copies (Point Transformer, fp16 mode): a caller's step-scoped cache holds tensors made outside this
        # capture, for the parameters rather than their aliases -- the captured call keeps its OWN cache: every weight is cast
        # once per call, inside the graph, from the live parameter storage
        from .model import Ptran_transformer as _PT
        keep_w16 = ops.CTX.w16_cache
        ops.CTX.w16_cache = {} if (keep_w16 is not None or _PT.GEMM_DTYPE is not None) else None
        try:
            for m_, n_, q in self._slots:
                m_._parameters[n_] = alias[id(q)]
            with ops.capture_guard(), torch.cuda.graph(inst.graph_f, capture_error_mode='thread_local'):
                with ops.deferred_bn_counts():
                    out = model._forward_impl(inst.x, *flags)
            if inst.feeder.cursor != len(inst.feeder.plan):
                raise _Unsupported('the captured forward drew %d FPS starts, the eager one %d' % (inst.feeder.cursor, len(inst.feeder.plan)))
            if hasattr(g, 'last_prefix'):
                e = g.last_prefix(inst.x)
                if e is not None:
                    g._prefix_cache = {k: v for k, v in g._prefix_cache.items() if v is not e}
                    if dep is None and bool(getattr(g, 'share_prefix', False)) and all(t.requires_grad for t in e.tensors):
                        exported = e
        finally:
            ops.CTX.start_provider = None
            ops.CTX.w16_cache = keep_w16
            for m_, n_, q in self._slots:
                m_._parameters[n_] = q
        inst.single = isinstance(out, torch.Tensor)
        outs = [out] if inst.single else list(out if out is not None else ())
        if not outs or not all(isinstance(t, torch.Tensor) and t.is_floating_point() and t.requires_grad for t in outs):
            raise _Unsupported('this forward mode does not return differentiable tensors only')
        pref = list(exported.tensors) if exported is not None else []
        diff = outs + pref
        inst.gouts = [torch.zeros_like(t, memory_format=torch.contiguous_format) for t in diff]
        inst.gdirty = [False] * len(diff)
        params = self._req
        inst.graph_b = torch.cuda.CUDAGraph()
        with ops.capture_guard(), torch.cuda.graph(inst.graph_b, pool=inst.graph_f.pool(), capture_error_mode='thread_local'):
            grads = torch.autograd.grad(diff, [alias[id(q)] for q in params] + leaves, inst.gouts, allow_unused=True)
        # static gradient tensors of the parameters this call reaches (kept referenced: their memory stays out of the pool)
        inst.used = [(q, gr) for q, gr in zip(params, grads[:len(params)]) if gr is not None]
        inst.leaf_grads = tuple(grads[len(params):])
        inst.outs = [t.detach() for t in outs]
        inst.prefix_static = tuple(t.detach() for t in pref)
        inst.prefix_extra = exported.extra if exported is not None else None
        return inst

    # ------------------------------------------------------------------ replay
    def _replay(self, model, inst, x, entry):
        inst.generation += 1
        inst.busy = True
        # A parameter whose .grad still IS one of this instance's static gradient tensors (the caller has not reset it since
        # this instance's previous backward: gradient accumulation over several steps) keeps its value in a copy -- the
        # static tensor shares the instance's memory pool with the forward's temporaries and is overwritten by the replays.
        for p, gr in inst.used:
            g = p.grad
            if g is not None and g.data_ptr() == gr.data_ptr():
                p.grad = g.clone()
        inst.x.copy_(x, non_blocking=True)
        inst.feeder.refill()
        inst.graph_f.replay()
        self.stats['replayed'] += 1
        prefix_in = tuple(entry.tensors) if inst.dep is not None else ()
        res = _Call.apply(inst, self.token, *prefix_in)
        n = len(inst.outs)
        if inst.prefix_static:
            gen = inst.generation
            model.g.install_prefix(x, res[n:], inst.prefix_extra,
                                   live=lambda i=inst, g_=gen: i.busy and i.generation == g_, provider=inst)
        return res[0] if inst.single else tuple(res[:n])

    def _backward(self, inst, gs):
        src, dst, zero = [], [], []
        for k, (g, buf) in enumerate(zip(gs, inst.gouts)):
            if g is None:
                if inst.gdirty[k]:
                    zero.append(buf)
                    inst.gdirty[k] = False
            else:
                src.append(g)
                dst.append(buf)
                inst.gdirty[k] = True
        if zero:
            torch._foreach_zero_(zero)
        if dst:
            torch._foreach_copy_(dst, src)
        for p, gr in inst.used:         # (as in _replay: a second backward through a retained graph)
            g = p.grad
            if g is not None and g.data_ptr() == gr.data_ptr():
                p.grad = g.clone()
        inst.graph_b.replay()
        # .grad semantics without AccumulateGrad nodes: the first contribution since the caller's zero_grad() becomes the
        # parameter's .grad as it is (the static tensor, no copy, no zero fill), later ones are added to it in one multi-tensor
        # launch -- in the order autograd runs the calls' backwards, i.e. the order eager accumulation adds them in.
        dst, src = [], []
        for p, gr in inst.used:
            g = p.grad
            if g is None:
                p.grad = gr
            else:
                dst.append(g)
                src.append(gr)
        if dst:
            torch._foreach_add_(dst, src)


def manager_for(model):
    """The model's call-graph manager (created on first use), or None when call graphs are switched off."""
    if not ENABLED:
        return None
    mgr = model.__dict__.get('_call_graph_mgr')
    if mgr is None:
        mgr = CallGraphs(model)
        model.__dict__['_call_graph_mgr'] = mgr
    return mgr


def drop(model):
    """Forget the model's captured call graphs (their memory pools are released)."""
    model.__dict__.pop('_call_graph_mgr', None)

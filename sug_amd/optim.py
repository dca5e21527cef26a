"""Adam for the three optimizers of a SUG step (train_dg_single_gpu.py:193-203, :333-335).

`Adam` is a torch.optim.Adam (same constructor arguments, same `state` layout, so
state_dict()/load_state_dict() interoperate with torch's) whose step() runs in
sug_adam_step: one launch per optimizer instead of torch's ~20 multi-tensor launches over
~150 small tensors.  HIP tensors only -- like the rest of the library there is no CPU path."""
import ctypes
import math

import torch

from ._lib import lib, check


class _Bucket:
    __slots__ = ('params', 'steps', 'table', 'first_dev', 'first_host', 'hyper', 'step_val', 'T', 'step_dev', 'scalars',
                 'groups', 'lr_dev', 'lr_written')


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, graph_capturable=False):
        """graph_capturable: the step count, the bias corrections AND the learning rate live on the device, so a
        step can be captured in a hipGraph and replayed through a learning-rate schedule (state['step'] entries are
        then not maintained; a new lr reaches the device in `refresh_device_scalars`, which every step() calls
        unless a capture is in progress -- the owner of a captured graph calls it before each replay)."""
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        self._plan = None
        self._plan_key = None
        self._graph_capturable = bool(graph_capturable)
        # bumped whenever the device-side plan (pointer table, step / scalar buffers) is dropped or rebuilt: a captured
        # graph holds raw pointers into the plan it was captured with and must not be replayed after a change
        self.plan_generation = 0

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plan = None
        self.plan_generation += 1

    def graph_key(self):
        """What a captured step bakes in by value: every group's (betas, eps, weight_decay) -- not lr (device)."""
        return tuple(self._hyper(g)[1:] for g in self.param_groups)

    def refresh_device_scalars(self):
        """graph_capturable mode: write each group's current lr to the device where it changed (eager fills; never
        called while capturing)."""
        if not (self._graph_capturable and self._plan):
            return
        for b in self._plan:
            lrs = {float(self.param_groups[gi]['lr']) for gi in b.groups}
            if len(lrs) > 1:                # groups that shared a bucket now have different rates: new plan
                self._plan = None
                self.plan_generation += 1
                return
            lr = lrs.pop()
            if b.lr_written != lr:
                b.lr_dev.fill_(lr)
                b.lr_written = lr

    def _sync_steps_from_device(self):
        """graph_capturable mode keeps the step count on the device (replays advance it without the
        host): copy it back into state['step'] -- before the plan is rebuilt, and for state_dict()."""
        if not (self._graph_capturable and self._plan):
            return
        for b in self._plan:
            n = float(int(b.step_dev.item()))
            b.step_val = int(n)
            for p in b.params:
                self.state[p]['step'] = torch.tensor(n, dtype=torch.float32)

    def state_dict(self):
        self._sync_steps_from_device()
        return super().state_dict()

    def _hyper(self, g):
        if g.get('amsgrad') or g.get('maximize') or g.get('capturable') or g.get('differentiable'):
            raise RuntimeError('sug_amd.optim.Adam: amsgrad / maximize / capturable / differentiable are not supported')
        lr = g['lr']
        return (float(lr), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']))

    def _build(self, key):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('sug_amd.optim.Adam: the update plan changed (other hyper-parameters or another set of '
                               'parameters with gradients) while a hipGraph is being captured; run one eager step first')
        self._sync_steps_from_device()              # a rebuilt plan continues the old step counts
        self.plan_generation += 1
        chunk = lib().sug_adam_chunk()
        buckets, members = {}, {}
        for gi, g in enumerate(self.param_groups):
            hyper = self._hyper(g)
            for p in g['params']:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError('sug_amd.optim.Adam: contiguous fp32 HIP parameters only (no CPU path)')
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st['step'].is_cuda:
                    st['step'] = st['step'].cpu()
                bk = (hyper, float(st['step']), p.device)
                buckets.setdefault(bk, []).append(p)
                members.setdefault(bk, set()).add(gi)
        plan = []
        for (hyper, step_val, dev), ps in buckets.items():
            b = _Bucket()
            b.params, b.hyper, b.step_val, b.T = ps, hyper, int(step_val), len(ps)
            # capturable mode: lr is a device value per bucket (groups with one rate share a bucket, as before)
            b.groups, b.lr_dev, b.lr_written = members[(hyper, step_val, dev)], None, None
            b.steps = [self.state[p]['step'] for p in ps]
            rows, first = [], [0]
            for p in ps:
                st = self.state[p]
                rows.append([p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel()])
                first.append(first[-1] + (p.numel() + chunk - 1) // chunk)
            b.table = torch.tensor(rows, dtype=torch.int64).to(dev)
            b.first_host = (ctypes.c_int32 * len(first))(*first)
            b.first_dev = torch.tensor(first, dtype=torch.int32).to(dev)
            b.step_dev = torch.full((1,), b.step_val, dtype=torch.int32, device=dev)
            b.scalars = torch.zeros(2, dtype=torch.float32, device=dev)
            if self._graph_capturable:
                b.lr_written = float(hyper[0])
                b.lr_dev = torch.full((1,), b.lr_written, dtype=torch.float64, device=dev)
            plan.append(b)
        self._plan, self._plan_key = plan, key
        return plan

    def _ensure_plan(self):
        """The update plan for the parameters that hold a gradient now (rebuilt when that set or a by-value
        hyper-parameter changed); graph_capturable mode: the current learning rates are on the device afterwards."""
        key = (tuple(p.grad is not None for g in self.param_groups for p in g['params']),
               tuple(self._hyper(g)[1 if self._graph_capturable else 0:] for g in self.param_groups))
        if self._graph_capturable and not torch.cuda.is_current_stream_capturing():
            self.refresh_device_scalars()
        return self._plan if (self._plan is not None and key == self._plan_key) else self._build(key)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError('sug_amd.optim.Adam: closures are not supported')
        plan = self._ensure_plan()
        L = lib()
        for b in plan:
            if not self._graph_capturable:
                b.step_val += 1
                torch._foreach_add_(b.steps, 1.0)
            ptrs = []
            for p in b.params:
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32:
                    raise RuntimeError('sug_amd.optim.Adam: dense fp32 gradients only')
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                ptrs.append(g.data_ptr())
            lr, b1, b2, eps, wd = b.hyper           # capturable mode: the rate in force is b.lr_dev (device)
            stream = torch._C._cuda_getCurrentRawStream(b.table.device.index)
            if self._graph_capturable:
                check(L.sug_adam_step_capturable(b.table.data_ptr(), b.first_dev.data_ptr(), b.first_host, b.T,
                                                 (ctypes.c_void_p * b.T)(*ptrs), b.lr_written, b1, b2, eps, wd,
                                                 b.step_dev.data_ptr(), b.scalars.data_ptr(), b.lr_dev.data_ptr(),
                                                 ctypes.c_void_p(stream)),
                      'sug_adam_step_capturable')
                torch.autograd.graph.increment_version(b.params)
                continue
            check(L.sug_adam_step(b.table.data_ptr(), b.first_dev.data_ptr(), b.first_host, b.T,
                                  (ctypes.c_void_p * b.T)(*ptrs), lr, b1, b2, eps, wd,
                                  1.0 - math.pow(b1, b.step_val), 1.0 - math.pow(b2, b.step_val),
                                  ctypes.c_void_p(stream)), 'sug_adam_step')
            # the kernel wrote through raw pointers: bump the version counters so that anything keyed
            # on p._version (conv_2d's [W1;W2-W1] cache, DGCNN's prefix cache, autograd's
            # saved-tensor checks) sees the update, as after an in-place torch op
            torch.autograd.graph.increment_version(b.params)
        return None


CHAIN_SLOTS, CHAIN_BUCKETS = 2, 8          # SUG_ADAM_CHAIN_SLOTS / SUG_ADAM_CHAIN_BUCKETS of include/sug_amd.h


class AdamChain:
    """`for o in optimizers: o.step()` for sug_amd.optim.Adam optimizers in ONE update launch (sug_adam_chain_step).

    The reference steps optimizer_dis, optimizer_g and optimizer_c back to back (train_dg_single_gpu.py:333-335), and the
    first two both own the encoder's parameters (:193-203): a parameter that sits in two optimizers gets both updates,
    in the order given, on the value held in registers -- bit-identical to the sequential calls, one launch (plus one
    one-thread-per-bucket prepare launch in graph_capturable mode) instead of three (six).  Every optimizer keeps its own
    `state` (state_dict() / load_state_dict() / a later plain `o.step()` work as before); what the chain adds is a joint
    pointer table, rebuilt whenever one of the optimizers rebuilt its plan.  Falls back to the sequential calls when the
    optimizers do not fit (a parameter in more than CHAIN_SLOTS of them, more than CHAIN_BUCKETS hyper-parameter sets,
    mixed graph_capturable modes or devices)."""

    def __init__(self, optimizers):
        self.optimizers = list(optimizers)
        if not self.optimizers or not all(isinstance(o, Adam) for o in self.optimizers):
            raise RuntimeError('sug_amd.optim.AdamChain: sug_amd.optim.Adam optimizers only')
        self._joint = None
        self._joint_gens = None
        # bumped with every rebuild of the joint table: captured graphs hold raw pointers into it (see Adam.plan_generation)
        self.plan_generation = 0

    def _build(self, plans):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('sug_amd.optim.AdamChain: the joint update plan changed while a hipGraph is being captured; '
                               'run one eager step first')
        self.plan_generation += 1
        opts = self.optimizers
        cap = opts[0]._graph_capturable
        buckets = [b for plan in plans for b in plan]
        devs = {b.table.device for b in buckets}
        J = {'fallback': True}
        self._joint = J
        if any(o._graph_capturable != cap for o in opts) or len(devs) != 1 or not 0 < len(buckets) <= CHAIN_BUCKETS:
            return J
        dev = devs.pop()
        order, slots = [], {}
        for bi, b in enumerate(buckets):
            for p in b.params:
                if id(p) not in slots:
                    slots[id(p)] = []
                    order.append(p)
                slots[id(p)].append(bi)
        if any(len(v) > CHAIN_SLOTS or len(set(v)) != len(v) for v in slots.values()):
            return J
        owner = {}                                   # bucket index -> optimizer (the moments live in ITS state)
        bi = 0
        for o, plan in zip(opts, plans):
            for _ in plan:
                owner[bi] = o
                bi += 1
        chunk = lib().sug_adam_chain_chunk()
        rows, first, bmap = [], [0], []
        for t, p in enumerate(order):
            row = [p.data_ptr(), p.numel(), 0, 0, 0, 0, 0, 0]
            code = len(slots[id(p)])
            for s, bk in enumerate(slots[id(p)]):
                st = owner[bk].state[p]
                row[2 + 2 * s], row[3 + 2 * s] = st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
                code |= bk << (8 + 8 * s)
            row[6] = code
            rows.append(row)
            nb = (p.numel() + chunk - 1) // chunk
            bmap += [(t, c) for c in range(nb)]
            first.append(first[-1] + nb)
        J.update(fallback=False, params=order, buckets=buckets, capturable=cap, T=len(order),
                 table=torch.tensor(rows, dtype=torch.int64).to(dev),
                 block_map=torch.tensor(bmap, dtype=torch.int32).reshape(-1, 2).to(dev),
                 first_host=(ctypes.c_int32 * len(first))(*first),
                 hyper=(ctypes.c_double * (6 * len(buckets)))(), steps=None, scalars=None, lr=None)
        if cap:
            # the buckets' device scalars move into joint arrays and stay reachable through the optimizers' own buckets
            # (views): refresh_device_scalars / _sync_steps_from_device / a plain o.step() keep working
            J['steps'] = torch.cat([b.step_dev for b in buckets])
            J['scalars'] = torch.zeros(2 * len(buckets), dtype=torch.float32, device=dev)
            J['lr'] = torch.cat([b.lr_dev for b in buckets])
            for i, b in enumerate(buckets):
                b.step_dev, b.lr_dev, b.scalars = J['steps'][i:i + 1], J['lr'][i:i + 1], J['scalars'][2 * i:2 * i + 2]
        return J

    @torch.no_grad()
    def step(self):
        opts = self.optimizers
        plans = [o._ensure_plan() for o in opts]
        gens = tuple(o.plan_generation for o in opts)
        J = self._joint if gens == self._joint_gens else None
        if J is not None and not J['fallback'] and J['capturable'] and any(
                b.step_dev.data_ptr() != J['steps'][i:i + 1].data_ptr() for i, b in enumerate(J['buckets'])):
            J = None                    # another chain over the same optimizers has re-homed their device scalars since
        if J is None:
            J = self._build(plans)
            self._joint_gens = gens
        if J['fallback']:
            for o in opts:
                o.step()
            return None
        ptrs = []
        for p in J['params']:
            g = p.grad
            if g.is_sparse or g.dtype != torch.float32:
                raise RuntimeError('sug_amd.optim.Adam: dense fp32 gradients only')
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            ptrs.append(g.data_ptr())
        H = J['hyper']
        for i, b in enumerate(J['buckets']):
            if not J['capturable']:
                b.step_val += 1
                torch._foreach_add_(b.steps, 1.0)
            H[6 * i:6 * i + 6] = list(b.hyper) + [float(b.step_val)]
        T = J['T']
        stream = torch._C._cuda_getCurrentRawStream(J['table'].device.index)
        p_ = lambda t: None if t is None else t.data_ptr()
        check(lib().sug_adam_chain_step(J['table'].data_ptr(), J['block_map'].data_ptr(), J['first_host'], T,
                                        (ctypes.c_void_p * T)(*ptrs), len(J['buckets']), H, p_(J['steps']), p_(J['scalars']),
                                        p_(J['lr']), ctypes.c_void_p(stream)), 'sug_adam_chain_step')
        torch.autograd.graph.increment_version(J['params'])
        return None

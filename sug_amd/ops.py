"""Torch-facing wrappers over the C ABI (include/sug_amd.h).

Tensors are plumbing here: device memory + the current HIP stream.  Every op
requires fp32/int32 tensors on a HIP device and raises otherwise -- there is no
CPU path.  Feature tensors are point-major rows [B,N,C].
"""
import contextlib
import ctypes
import os as _os

import torch

from ._lib import lib, check, STATS_BLOCKS

class StepContext:
    """The mutable state a forward / training step of this package scopes -- ONE object instead of a dozen module globals (round 6,
    VERDICT r5 #9).  One process drives one GPU, so one context per process: SUGStep, Net_MDA.forward_pair, the call-graph manager
    and bench.py set fields for the duration of a pass (always restoring the previous value), the ops read them.  The old
    module-level names (`ops.BN_GROUPS`, `ops.START_PROVIDER`, `ops.PROFILE`, ...) remain as properties of the module that forward
    to this object, so existing callers and tests keep working.

      bn_groups          G equal contiguous parts of the batch (Net_MDA.forward_pair: source clouds, then target clouds) that the
                         reference sends through the network in G separate forward calls: BatchNorm statistics and running-buffer
                         updates per part, in order
      start_queue        pre-drawn FPS starts of this forward (the reference's call order), consumed by draw_start
      start_provider     None, or a callable (B, N) -> starts: a graph's start feeder (record / provide), a test's table
      geometry_plan      FPS + ball-query results of this pass computed up front (Pointnet2_g.plan_geometry), popped in call order
      profile            None, or {kernel name: [(event, event, shape)]}: bench.py's per-kernel HIP-event timings
      profile_only       None, or the set of name prefixes that get events
      parallel_branches  independent small-kernel chains on forked streams (opt-in; measured slower)
      fused_heads        fused LayerNorm + activation in the FC heads (SUGStep switches it on for its own forwards)
      w16_cache          step-scoped cache of 16-bit weight copies (Point Transformer, fp16 mode) or None
      bn_record          None, or the list collecting (BatchNorm module, batch-statistics coefficients) of a shared prefix
      pending_counts     None, or the deferred num_batches_tracked increments of this pass
      last_coef, last_coef_pair   the batch-statistics coefficients of the most recent BatchNorm op(s), for bn_record"""
    __slots__ = ('bn_groups', 'start_queue', 'start_provider', 'geometry_plan', 'profile', 'profile_only', 'parallel_branches',
                 'fused_heads', 'w16_cache', 'bn_record', 'pending_counts', 'last_coef', 'last_coef_pair')

    def __init__(self):
        self.bn_groups = 1
        self.start_queue = self.start_provider = self.geometry_plan = None
        self.profile = self.profile_only = None
        self.parallel_branches = False
        self.fused_heads = False
        self.w16_cache = self.bn_record = self.pending_counts = None
        self.last_coef, self.last_coef_pair = None, (None, None)


CTX = StepContext()

SIGMA_LIST = (0.01, 0.1, 1, 10, 100)       # model/mmd.py:23


# bench.py sets PROFILE = {} to time selected kernels with events on the launch stream.
# PROFILE_ONLY (a set of name prefixes or None) restricts which kernels get events: every
# event pair costs host time, and an eager step is host-bound.  While a hipGraph is being
# captured the events are created `external` so that they become event-record nodes.

# FPS start indices: the reference draws torch.randint(0, N, (B,)) from the CPU default
# generator once per farthest_point_sample call.  A provider (SUGStep's graph mode) may hand
# out device-resident slices instead, filled from the same draws, so a step can be replayed.


# Independent branches of a step (the two classifier heads, the two attention layers, the three MMD terms) are chains of
# small kernels -- 5-15 us each, launch-latency bound.  Inside a captured hipGraph they can run side by side: with
# PARALLEL_BRANCHES on (SUGStep switches it on for its own forwards when it replays the step from a graph), run_parallel
# forks the branches onto side streams and joins them; autograd runs each branch's backward on the stream of its
# forward, so the backward branches overlap too.  Launched eagerly from Python the extra event records only cost host
# time, hence off by default.  (Measured on ROCm 7.2 / MI355X: also under graph replay it is a loss -- 5.35 vs 5.26 ms per
# DGCNN step -- a cross-stream edge of a hipGraph costs more than the kernels it overlaps; SUG_PARALLEL_BRANCHES=1 opts in.)
_SIDE_STREAMS = {}


def run_parallel(fns):
    """[f() for f in fns], the calls after the first on side streams forked from / joined to the current stream."""
    if not CTX.parallel_branches or len(fns) < 2 or not torch.cuda.is_available():
        return [f() for f in fns]
    main = torch.cuda.current_stream()
    dev = main.device_index
    sides = []
    for i in range(1, len(fns)):
        st = _SIDE_STREAMS.get((dev, i))
        if st is None:
            st = _SIDE_STREAMS[(dev, i)] = torch.cuda.Stream(device=dev)
        st.wait_stream(main)                  # fork point: before any branch is enqueued
        sides.append(st)
    outs = [fns[0]()]
    for f, st in zip(fns[1:], sides):
        with torch.cuda.stream(st):
            outs.append(f())
    for st in sides:
        main.wait_stream(st)                  # join
    return outs


@contextlib.contextmanager
def capture_guard():
    """Around every hipGraph capture of this package.  torch.cuda.graph no longer runs the garbage collector when a capture
    starts (torch >= 2.9: `torch.compiler.config.force_cudagraph_gc` is off), so a dead Python cycle that owns HIP objects -- an
    earlier trainer's captured graphs, tensors of a destroyed graph's private pool -- could be collected IN THE MIDDLE of a
    capture: hipGraphExecDestroy / hipFree under a global-mode capture aborted the process (seen once in a full test run,
    round 6: 'Fatal Python error: Aborted ... Garbage-collecting' inside a captured forward).  The collector stays OFF for the
    duration of the capture (dead cycles wait for the next automatic collection outside it; a full collection here costs
    ~0.1 s in a large process and a step is captured in up to six pieces), and the module-level tensor caches are dropped
    first (their tensors may live in an older graph's pool and would otherwise be released -- by reference count, collector
    or not -- by the first op of the capture that replaces them)."""
    import gc
    clear_rows_cache()
    CTX.last_coef, CTX.last_coef_pair = None, (None, None)
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


# Domain groups of a batch: with BN_GROUPS = G the batch dimension holds G equal contiguous parts
# (Net_MDA.forward_pair: source clouds then target clouds) that the reference sends through the
# network in G separate forward calls.  Everything per-cloud / per-row is oblivious to that; the
# BatchNorm ops compute statistics and update the running buffers per part, in order, so the
# result is the one of G separate calls.


@contextlib.contextmanager
def bn_groups(g):
    old, CTX.bn_groups = CTX.bn_groups, int(g)
    try:
        yield
    finally:
        CTX.bn_groups = old


# Pre-drawn FPS starts (Net_MDA.forward_pair draws them in the reference's call order: every
# farthest_point_sample of the source forward, then those of the target forward).


@contextlib.contextmanager
def start_queue(q):
    old, CTX.start_queue = CTX.start_queue, (list(q) if q is not None else None)
    try:
        yield
    finally:
        CTX.start_queue = old


# Geometry of several encoder passes over ONE batch, computed up front (Pointnet2_g.plan_geometry): farthest-point sampling and
# ball query depend on the coordinates and the start draws only, so the semantic and the node pass of a step can share
# their launches (FPS is one workgroup per cloud and `npoint` dependent rounds: 128 clouds fill half of the chip for
# 0.34 ms; both passes in one launch take the same 0.34 ms).  A list of (new_xyz, idx) per sample_and_group call, in call
# order; pointnet2_utils.sample_and_group_idx pops from it.


def draw_start(B, N):
    if CTX.start_queue:                 # draws made up front for this forward (Net_MDA.forward_pair: the reference's order)
        t = CTX.start_queue.pop(0)
        if t.numel() != B:
            raise RuntimeError('FPS start plan does not match the encoder (%d starts for %d clouds)' % (t.numel(), B))
        return t
    if CTX.start_provider is not None:
        return CTX.start_provider(B, N)
    G = CTX.bn_groups
    if G > 1 and B % G == 0:        # one CPU-generator draw per reference forward call
        return torch.cat([torch.randint(0, N, (B // G,), dtype=torch.long) for _ in range(G)])
    return torch.randint(0, N, (B,), dtype=torch.long)


def draw_group_start(B, N):
    """One reference forward's draw (B clouds of one domain group), through the graph's start feeder when one is active."""
    if CTX.start_provider is not None:
        return CTX.start_provider(B, N)
    return torch.randint(0, N, (B,), dtype=torch.long)


def _timed(name, shape, call):
    if CTX.profile is None or (CTX.profile_only is not None and not name.startswith(tuple(CTX.profile_only))):
        return call()
    if torch.cuda.is_current_stream_capturing():        # timing events cannot be captured on ROCm
        return call()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    r = call()
    b.record()
    CTX.profile.setdefault(name, []).append((a, b, shape))
    return r


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _st():
    # raw handle of the current stream; torch.cuda.current_stream() builds a Stream object (~10 us)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('sug_amd ops run on a HIP device only (got a %s tensor); '
                               'there is no CPU fallback' % t.device)


def _rows3(t):
    """[B,N,C] fp32 rows with unit channel stride -> (B, N, C, ld); copies if the layout
    is not expressible as a single row stride."""
    assert t.dim() == 3 and t.dtype == torch.float32
    B, N, C = t.shape
    if t.stride(2) != 1 or t.stride(0) != N * t.stride(1) or t.stride(1) < C:
        t = t.contiguous()
    return t, B, N, C, t.stride(1)


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


# ----------------------------------------------------------------------------- index ops
def knn(x, k):
    """x [B,N,C] rows -> idx [B,N,k] int32 (sug_knn)."""
    _need_gpu(x)
    x, B, N, C, ld = _rows3(x.detach())
    idx = torch.empty(B, N, k, dtype=torch.int32, device=x.device)
    check(_timed('knn_C%d' % C, {'B': B, 'N': N, 'C': C, 'k': k},
                 lambda: lib().sug_knn(_p(x), ld, B, N, C, k, _p(idx), _st())), 'sug_knn')
    return idx


def knn_reverse(idx):
    """idx [B,N,k] int32 -> (rev_off [B,N+1], rev_ent [B,N*k])."""
    _need_gpu(idx)
    idx = _i32(idx).contiguous()
    B, N, k = idx.shape
    off = torch.empty(B, N + 1, dtype=torch.int32, device=idx.device)
    ent = torch.empty(B, N * k, dtype=torch.int32, device=idx.device)
    check(lib().sug_knn_reverse(_p(idx), B, N, k, _p(off), _p(ent), _st()), 'sug_knn_reverse')
    return off, ent


def fps(xyz, npoint, start):
    """xyz [B,N,3], start [B] (any int dtype/device) -> [B,npoint] int32."""
    _need_gpu(xyz)
    xyz = xyz.detach().contiguous()
    B, N, _ = xyz.shape
    start = start.to(device=xyz.device, dtype=torch.int32, non_blocking=True)
    out = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
    check(lib().sug_fps(_p(xyz), _p(start), B, N, npoint, _p(out), _st()), 'sug_fps')
    return out


def ball_query(xyz, query, radius, nsample):
    """xyz [B,N,3], query [B,S,3] -> [B,S,nsample] int32; r^2 rounded to fp32 the way torch
    rounds the python scalar in `sqrdists > radius ** 2` (model/point_utils.py:102)."""
    _need_gpu(xyz, query)
    xyz, query = xyz.detach().contiguous(), query.detach().contiguous()
    B, N, _ = xyz.shape
    S = query.shape[1]
    r2 = float(torch.tensor(radius ** 2, dtype=torch.float32))
    out = torch.empty(B, S, nsample, dtype=torch.int32, device=xyz.device)
    check(lib().sug_ball_query(_p(xyz), _p(query), B, N, S, r2, nsample, _p(out), _st()), 'sug_ball_query')
    return out


def knn_query(xyz, query, k, want_dist=False, direct=False):
    """k nearest of xyz [B,N,3] for each query [B,S,3] (sort semantics) -> idx [B,S,k].
    direct: distances as sum((q - p)^2) (Point Transformer path) instead of the expanded form."""
    _need_gpu(xyz, query)
    xyz, query = xyz.detach().contiguous(), query.detach().contiguous()
    B, N, _ = xyz.shape
    S = query.shape[1]
    idx = torch.empty(B, S, k, dtype=torch.int32, device=xyz.device)
    dist = torch.empty(B, S, k, dtype=torch.float32, device=xyz.device) if want_dist else None
    fn = lib().sug_knn_query_direct if direct else lib().sug_knn_query
    check(fn(_p(xyz), _p(query), B, N, S, k, _p(idx), _p(dist), _st()), 'sug_knn_query')
    return (idx, dist) if want_dist else idx


def three_nn_raw(query, cand):
    _need_gpu(query, cand)
    query, cand = query.detach().contiguous(), cand.detach().contiguous()
    B, N, _ = query.shape
    S = cand.shape[1]
    idx = torch.empty(B, N, 3, dtype=torch.int32, device=query.device)
    dist = torch.empty(B, N, 3, dtype=torch.float32, device=query.device)
    check(lib().sug_three_nn(_p(query), _p(cand), B, N, S, _p(idx), _p(dist), _st()), 'sug_three_nn')
    return idx, dist


class _ThreeNN(torch.autograd.Function):
    """dist3 is differentiable w.r.t. the candidate (node) positions:
    d/dc |q-c|^2 = 2(c-q); the query cloud is an input and gets no gradient."""

    @staticmethod
    def forward(ctx, query, cand):
        idx, dist = three_nn_raw(query, cand)
        ctx.save_for_backward(query, cand, idx)
        ctx.mark_non_differentiable(idx)
        return idx, dist

    @staticmethod
    def backward(ctx, _gidx, gdist):
        query, cand, idx = ctx.saved_tensors
        B, N, _ = query.shape
        S = cand.shape[1]
        li = idx.long().view(B, N * 3)
        sel = torch.gather(cand, 1, li.unsqueeze(-1).expand(B, N * 3, 3)).view(B, N, 3, 3)
        g = (2.0 * gdist).unsqueeze(-1) * (sel - query.unsqueeze(2))          # [B,N,3,3]
        gc = torch.zeros(B, S, 3, device=cand.device, dtype=cand.dtype)
        gc.scatter_add_(1, li.unsqueeze(-1).expand(B, N * 3, 3), g.view(B, N * 3, 3))
        return None, gc


def three_nn(query, cand):
    """-> (idx3 [B,N,3] int32, dist3 [B,N,3]); dist3 carries gradient to `cand`."""
    return _ThreeNN.apply(query, cand)


# ----------------------------------------------------------------------------- gathers
SCATTER_ORDERED = _os.environ.get('SUG_SCATTER_ORDERED', '1') != '0'      # index_points backward in a fixed order (0: float atomics)


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx):
        _need_gpu(feat, idx)
        feat_c, B, N, C, ld = _rows3(feat)
        idx2 = _i32(idx).reshape(B, -1).contiguous()
        S = idx2.shape[1]
        out = torch.empty(B, S, C, dtype=torch.float32, device=feat.device)
        check(lib().sug_gather_rows(_p(feat_c), ld, _p(idx2), B, N, S, C, _p(out), C, _st()), 'sug_gather_rows')
        ctx.save_for_backward(idx2)
        ctx.shape = (B, N, C)
        return out.view(*idx.shape, C)

    @staticmethod
    def backward(ctx, g):
        (idx2,) = ctx.saved_tensors
        B, N, C = ctx.shape
        S = idx2.shape[1]
        g = g.reshape(B, S, C).contiguous()
        L = lib()
        if SCATTER_ORDERED and L.sug_scatter_rows_ordered_supported(B, N, S):
            # fixed summation order (sorted reverse lists): index_points' backward reproducible bit for bit
            d = torch.empty(B, N, C, dtype=torch.float32, device=g.device)
            ws = torch.empty(L.sug_scatter_rows_workspace(B, N, S), dtype=torch.int32, device=g.device)
            check(L.sug_scatter_rows_ordered(_p(g), C, _p(idx2), B, N, S, C, _p(d), C, _p(ws), _st()), 'sug_scatter_rows_ordered')
            return d, None
        d = torch.zeros(B, N, C, dtype=torch.float32, device=g.device)
        check(L.sug_scatter_add_rows(_p(g), C, _p(idx2), B, N, S, C, _p(d), C, _st()), 'sug_scatter_add_rows')
        return d, None


def gather_rows(feat, idx):
    """feat [B,N,C], idx [B,S] or [B,S,K] -> [B,S(,K),C]  (index_points)."""
    return _GatherRows.apply(feat, idx)


class _GroupMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx):
        _need_gpu(feat, idx)
        feat_c, B, N, C, ld = _rows3(feat)
        idx = _i32(idx).contiguous()
        S, ns = idx.shape[1], idx.shape[2]
        out = torch.empty(B, S, C, dtype=torch.float32, device=feat.device)
        arg = torch.empty(B, S, C, dtype=torch.int32, device=feat.device)
        check(lib().sug_group_max(_p(feat_c), ld, _p(idx), B, N, S, ns, C, _p(out), _p(arg), _st()), 'sug_group_max')
        ctx.save_for_backward(arg)
        ctx.shape = (B, N, C)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        B, N, C = ctx.shape
        S = arg.shape[1]
        g = g.contiguous()
        d = torch.zeros(B, N, C, dtype=torch.float32, device=g.device)
        check(lib().sug_group_max_bwd(_p(g), _p(arg), B, N, S, C, _p(d), C, _st()), 'sug_group_max_bwd')
        return d, None


def group_max(feat, idx):
    """max_j feat[b, idx[b,s,j], :] -> [B,S,C]."""
    return _GroupMax.apply(feat, idx)


# ----------------------------------------------------------------------------- per-point linear
# (rows, M, N) of weight gradients for which the TUNED library GEMM beats sug_linear_dw (filled by
# sug_amd.tuning.enable_tuned_gemms from a measured table; empty = always the library's own kernel).
DW_LIBRARY_SHAPES = set()
DW_FORCE_LIBRARY = False        # tuning runs: every weight gradient through the library
DW_SHAPE_LOG = None             # tuning runs: list collecting the (rows, M, N) seen
# measured on MI355X (graph mode, C2 step): the library's selection is 0.1-0.2 ms per step faster for these
# forward GEMMs, tuned table or not -- the own kernel stays opt-in (SUG_OWN_ROWS_GEMM=1)
OWN_ROWS_GEMM = _os.environ.get('SUG_OWN_ROWS_GEMM', '0') == '1'            # forward y = x.W^T of skinny layers (K in {64,128}, Co % 128 == 0) by sug_rows_gemm


class _LinearRows(torch.autograd.Function):
    """y = x . W^T (+ b) over rows; the weight gradient g^T . x (K = rows = B*N, small output)
    runs in sug_linear_dw instead of a rocBLAS GEMM that does not split K."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        M, N = weight.shape
        x2 = x.reshape(-1, N)
        if OWN_ROWS_GEMM and N in (64, 128) and M % 128 == 0 and x2.shape[0] >= 4096 and x2.stride(1) == 1 \
                and x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0 and weight.is_contiguous():
            # fp32 MFMA kernel of the library (the rocBLAS selection for these skinny shapes is 2-3x slower)
            y = torch.empty(x2.shape[0], M, dtype=torch.float32, device=x.device)
            check(_timed('rows_gemm_K%d_Co%d' % (N, M), {'B': x2.shape[0], 'N': 1, 'k': N, 'Co': M},
                         lambda: lib().sug_rows_gemm(_p(x2), x2.stride(0), x2.shape[0], N, _p(weight.detach()),
                                                     _p(None if bias is None else bias.detach()), M, _p(y), M, _st())),
                  'sug_rows_gemm')
            y = y.view(*x.shape[:-1], M)
        else:
            y = torch.nn.functional.linear(x, weight, bias)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        M, N = weight.shape
        dx, dw, db = linear_rows_backward(x.reshape(-1, N), weight, g.reshape(-1, M), ctx.needs_input_grad[0],
                                          ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        return (None if dx is None else dx.view(x.shape)), dw, db


DW_BMM = _os.environ.get('SUG_DW_BMM', '1') == '1'


def _dw_bmm_chunks(R, M, N, g2, x2):
    """Row chunks for the batched-GEMM form of a weight gradient (0 = use sug_linear_dw): outputs >= 128 x 128 (or
    256 x 64), dense operands, chunks of ~4096 rows (8 .. 64 chunks)."""
    if not DW_BMM or M * N < 128 * 128 or min(M, N) < 64 or g2.stride(0) != M or x2.stride(0) != N or \
            g2.stride(1) != 1 or x2.stride(1) != 1:
        return 0
    S = 64
    while S > 8 and R // S < 4096:
        S //= 2
    return S if (R % S == 0 and R // S >= 1024) else 0


def linear_rows_backward(x2, weight, g2, need_dx, need_dw, need_db):
    """Gradients of y = x2 . weight^T (+ b) for rows x2 [R, N], g2 [R, M]: dx by the library GEMM, the weight gradient
    g^T . x (K = rows, small output) by sug_linear_dw(_bias) -- with the bias gradient from the same pass over g."""
    M, N = weight.shape
    if g2.stride(1) != 1:
        g2 = g2.contiguous()
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    R = g2.shape[0]
    dx = dw = db = None
    if need_dx:
        dx = g2 @ weight
    if need_dw:
        if DW_SHAPE_LOG is not None:
            DW_SHAPE_LOG.append((R, M, N))
        # (not while a step graph is being captured: the library's split-K solutions for these K = rows shapes
        # clear their output with a memset, and memset nodes of a replayed hipGraph were not reliably ordered in
        # front of the accumulating kernels on ROCm 7 -- see sug_mmd_rbf_value -- which showed as NaN weights
        # after tens to hundreds of replays)
        lib_ok = DW_FORCE_LIBRARY or ((R, M, N) in DW_LIBRARY_SHAPES and not torch.cuda.is_current_stream_capturing())
        S = _dw_bmm_chunks(R, M, N, g2, x2) if not need_db else 0
        if M * N > 512 * 512 or lib_ok:
            dw = g2.t() @ x2            # larger than any encoder layer, or the tuned library GEMM is faster
        elif S:
            # outputs of 128 x 128 and more: the library's GEMM kernels beat the one-wave MFMA tiles of sug_linear_dw
            # (512 x 512 at 65536 rows: 240 vs 335 us) once K = rows is split by BATCHING over row chunks -- no split-K
            # solution, hence no memset node -- and the partials are folded by sug_linear_dw's ordered reduce
            # (outside TunableOp: its look-up of a shape that is not in the recorded table ends in a hipBLASLt call that
            # is not permitted while a stream is capturing; the default heuristic's batched kernel is)
            import torch.cuda.tunable as _tn
            tuned = _tn.is_enabled()
            if tuned:
                _tn.enable(False)
            try:
                part = torch.bmm(g2.view(S, R // S, M).transpose(1, 2), x2.view(S, R // S, N))
            finally:
                if tuned:
                    _tn.enable(True)
            dw = torch.empty(M, N, dtype=torch.float32, device=g2.device)
            check(lib().sug_linear_dw_fold(_p(part), S, M * N, _p(dw), _st()), 'sug_linear_dw_fold')
        else:
            dw = torch.empty(M, N, dtype=torch.float32, device=g2.device)
            ws = torch.empty(int(lib().sug_linear_dw_workspace(R, M, N)), dtype=torch.float32, device=g2.device)
            if need_db:       # bias gradient from the same pass over g
                db = torch.empty(M, dtype=torch.float32, device=g2.device)
            check(lib().sug_linear_dw_bias(_p(g2), g2.stride(0), _p(x2), x2.stride(0), R, M, N, _p(dw), _p(db), _p(ws),
                                           _st()), 'sug_linear_dw_bias')
    if need_db and db is None:
        db = colsum(g2)
    return dx, dw, db


def linear_rows(x, weight, bias=None):
    """F.linear for [..., Cin] rows with many rows and small Cin/Cout (the encoders' 1x1 convs)."""
    _need_gpu(x, weight)
    if x.dtype != torch.float32:
        raise RuntimeError('sug_amd.ops.linear_rows: fp32 rows only (got %s)' % x.dtype)
    return _LinearRows.apply(x, weight, bias)


# ----------------------------------------------------------------------------- SA-node glue
class _NodeOffset(torch.autograd.Function):
    """(node_off, node_loc) of adapt_layer_off from the projected features (sug_node_offset_*)."""

    @staticmethod
    def forward(ctx, proj, loc, fidx, gidx):
        _need_gpu(proj, loc, fidx, gidx)
        proj, loc = proj.contiguous(), loc.detach().contiguous()
        fidx, gidx = _i32(fidx).contiguous(), _i32(gidx).contiguous()
        B, N, _ = proj.shape
        S, ns = gidx.shape[1], gidx.shape[2]
        off = torch.empty(B, S, 3, dtype=torch.float32, device=proj.device)
        nloc = torch.empty(B, S, 3, dtype=torch.float32, device=proj.device)
        check(lib().sug_node_offset_fwd(_p(proj), _p(loc), _p(fidx), _p(gidx), B, N, S, ns, _p(off), _p(nloc), _st()),
              'sug_node_offset_fwd')
        ctx.save_for_backward(proj, loc, fidx, gidx)
        ctx.set_materialize_grads(False)
        return off, nloc

    @staticmethod
    def backward(ctx, goff, gnloc):
        proj, loc, fidx, gidx = ctx.saved_tensors
        B, N, _ = proj.shape
        S, ns = gidx.shape[1], gidx.shape[2]
        if goff is None and gnloc is None:
            return None, None, None, None
        # nloc = loc[f] + off; an output that took no part in the loss arrives as None (no zero fill + add for it)
        g = (goff if gnloc is None else gnloc if goff is None else goff + gnloc).contiguous()
        dproj = torch.empty_like(proj)              # written (or zeroed) entirely by the entry point
        check(lib().sug_node_offset_bwd(_p(proj), _p(loc), _p(fidx), _p(gidx), _p(g), B, N, S, ns, _p(dproj), _st()),
              'sug_node_offset_bwd')
        return dproj, None, None, None


def node_offset(proj, loc, fidx, gidx):
    """proj [B,N,3], loc [B,N,3], fidx [B,S], gidx [B,S,ns] -> (node_off [B,S,3], node_loc [B,S,3])."""
    return _NodeOffset.apply(proj, loc, fidx, gidx)


class _Interp3Cat(torch.autograd.Function):
    """cat(fea, inverse-distance 3-NN interpolation of the node features) in one kernel; the
    backward also carries the distance gradient back to the node positions."""

    @staticmethod
    def forward(ctx, fea, node, xyz, nloc):
        _need_gpu(fea, node, xyz, nloc)
        fea, B, N, C1, ldf = _rows3(fea)
        node = node.contiguous()
        xyz, nlocd = xyz.detach().contiguous(), nloc.detach().contiguous()
        S, C2 = node.shape[1], node.shape[2]
        idx3, d3 = three_nn_raw(xyz, nlocd)
        out = torch.empty(B, N, C1 + C2, dtype=torch.float32, device=fea.device)
        check(lib().sug_interp3_cat_fwd(_p(fea), ldf, C1, _p(node), _p(idx3), _p(d3), B, N, S, C2, _p(out),
                                        C1 + C2, _st()), 'sug_interp3_cat_fwd')
        ctx.save_for_backward(node, idx3, d3, xyz, nlocd)
        ctx.meta = (B, N, S, C1, C2)
        return out

    @staticmethod
    def backward(ctx, g):
        node, idx3, d3, xyz, nloc = ctx.saved_tensors
        B, N, S, C1, C2 = ctx.meta
        g = g.contiguous()
        dnode = torch.empty_like(node)
        dnloc = torch.empty_like(nloc)
        off = torch.empty(B, S + 1, dtype=torch.int32, device=g.device)
        ent = torch.empty(B, 3 * N, dtype=torch.int32, device=g.device)
        ddw = torch.empty(B, N, 6, dtype=torch.float32, device=g.device)
        check(lib().sug_interp3_cat_bwd_lists(_p(g), C1 + C2, C1, _p(node), _p(idx3), _p(d3), _p(xyz), _p(nloc), B, N, S,
                                              C2, _p(off), _p(ent), _p(ddw), _p(dnode), _p(dnloc), _st()),
              'sug_interp3_cat_bwd_lists')
        return g[:, :, :C1], dnode, None, dnloc


def interp3_cat(fea, node, xyz, nloc):
    """fea [B,N,C1], node [B,S,C2], xyz [B,N,3], nloc [B,S,3] -> [B,N,C1+C2]."""
    return _Interp3Cat.apply(fea, node, xyz, nloc)


# ----------------------------------------------------------------------------- BN helpers
def bn_coef(stats, gamma, beta, count, eps, momentum, running_mean, running_var, out=None):
    C = gamma.numel()
    coef = torch.empty(5, C, dtype=torch.float32, device=gamma.device) if out is None else out
    check(lib().sug_bn_finalize(_p(stats), _p(gamma), _p(beta), C, float(count), eps, momentum,
                                _p(running_mean), _p(running_var), _p(coef), _st()), 'sug_bn_finalize')
    return coef


def eval_coef(gamma, beta, running_mean, running_var, eps):
    rstd = torch.rsqrt(running_var + eps)
    scale = gamma * rstd
    return torch.stack([scale, beta - running_mean * scale, running_mean, rstd, running_var]).contiguous()


def affine_act(z, coef, slope, out=None):
    """out[...,c] = leaky(scale[c]*z + shift[c]); z [..., C] rows."""
    _need_gpu(z)
    C = z.shape[-1]
    z2 = z.reshape(-1, C)
    if z2.stride(1) != 1:
        z2 = z2.contiguous()
    rows = z2.shape[0]
    if out is None:
        out = torch.empty(rows, C, dtype=torch.float32, device=z.device)
    o2 = out.view(-1, C) if out.is_contiguous() else out
    check(lib().sug_affine_act(_p(z2), z2.stride(0), _p(coef), rows, C, float(slope), _p(o2), o2.stride(0), _st()),
          'sug_affine_act')
    return out.view(*z.shape)


def col_stats(y2, ws=None):
    """y2 [rows, C] (row stride ld) -> fp64 [2C] = column sums, column sums of squares."""
    rows, C = y2.shape
    stats = torch.empty(2 * C, dtype=torch.float64, device=y2.device)
    if ws is None:
        ws = torch.empty(STATS_BLOCKS * 2 * C, dtype=torch.float32, device=y2.device)
    check(lib().sug_col_stats(_p(y2), y2.stride(0), rows, C, _p(stats), _p(ws), _st()), 'sug_col_stats')
    return stats


class _BNActRows(torch.autograd.Function):
    """act(BatchNorm(y)) over the rows of y [rows, C]; act = LeakyReLU(slope) (0: ReLU, 1: none).
    Train mode uses batch statistics and updates the running ones like nn.BatchNorm."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, slope, eps, momentum, G):
        _need_gpu(y, gamma)
        C = y.shape[-1]
        y2 = y.reshape(-1, C)
        if y2.stride(1) != 1 or y2.stride(0) != C:
            y2 = y2.contiguous()
        rows = y2.shape[0]
        if rows % G:
            raise RuntimeError('bn_act_rows: %d rows do not split into %d domain groups' % (rows, G))
        dev = y.device
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        if training:
            coef = torch.empty(G, 5, C, dtype=torch.float32, device=dev)
        else:
            coef = eval_coef(g, b, running_mean, running_var, eps).unsqueeze(0).repeat(G, 1, 1)
        out = torch.empty(rows, C, dtype=torch.float32, device=dev)
        stats = torch.empty(2 * C, dtype=torch.float64, device=dev)
        ws = torch.empty(STATS_BLOCKS * 2 * C, dtype=torch.float32, device=dev)
        check(lib().sug_bn_act_rows_fwd(_p(y2), C, rows, C, G, _p(g), _p(b), 1 if training else 0, eps, momentum,
                                        float(slope), _p(running_mean), _p(running_var), _p(coef), _p(out), C,
                                        _p(stats), _p(ws), _st()), 'sug_bn_act_rows_fwd')
        CTX.last_coef = coef
        if any(ctx.needs_input_grad[i] for i in (0, 1, 2)):
            ctx.save_for_backward(y2, coef)
            ctx.meta = (rows, C, float(slope), bool(training), tuple(y.shape), G)
        return out.view(y.shape)

    @staticmethod
    def backward(ctx, gout):
        y2, coef = ctx.saved_tensors
        rows, C, slope, training, shape, G = ctx.meta
        dy, dgamma, dbeta = _bn_act_rows_backward(gout, y2, coef, rows, C, slope, training, G)
        return dy.view(shape), dgamma, dbeta, None, None, None, None, None, None, None


def _bn_act_rows_backward(gout, y2, coef, rows, C, slope, training, G):
    """Gradient of act(BatchNorm(y)) over rows (exact train-mode statistics terms): -> (dy [rows, C], dgamma, dbeta)."""
    dev = gout.device
    g2 = gout.reshape(rows, C)
    if g2.stride(1) != 1:
        g2 = g2.contiguous()
    a = torch.empty(rows, C, dtype=torch.float32, device=dev)
    red = torch.empty(G, 2 * C, dtype=torch.float64, device=dev)
    ws = torch.empty(STATS_BLOCKS * 2 * C, dtype=torch.float32, device=dev)
    dy = torch.empty(rows, C, dtype=torch.float32, device=dev) if training else a
    rf = torch.empty(2 * C, dtype=torch.float32, device=dev)
    check(lib().sug_bn_act_rows_bwd(_p(g2), g2.stride(0), _p(y2), C, _p(coef), rows, C, G, 1 if training else 0,
                                    slope, _p(a), _p(red), _p(dy), _p(ws), _p(rf), _st()), 'sug_bn_act_rows_bwd')
    return dy, rf[C:], rf[:C]


class _SAFirstLayer(torch.autograd.Function):
    """relu(BN(P[idx] - Q)) over the ball-query lists: the first layer of a set-abstraction MLP without the
    grouped [B,S,ns,3+D] tensor (sug_sa_first_fwd / bwd)."""

    @staticmethod
    def forward(ctx, P, Q, idx, gamma, beta, running_mean, running_var, training, eps, momentum, G):
        _need_gpu(P, Q, idx)
        P, Q = P.contiguous(), Q.contiguous()
        idx = _i32(idx).contiguous()
        B, N, C = P.shape
        S, ns = idx.shape[1], idx.shape[2]
        if B % G:
            raise RuntimeError('sa_first_layer: %d clouds do not split into %d domain groups' % (B, G))
        dev = P.device
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        if training:
            coef = torch.empty(G, 5, C, dtype=torch.float32, device=dev)
        else:
            coef = eval_coef(g, b, running_mean, running_var, eps).unsqueeze(0).repeat(G, 1, 1)
        Z = torch.empty(B, S, ns, C, dtype=torch.float32, device=dev)
        ws = torch.empty(STATS_BLOCKS * 2 * C, dtype=torch.float32, device=dev)
        check(lib().sug_sa_first_fwd(_p(P.detach()), C, _p(Q.detach()), _p(idx), B, N, S, ns, C, G, _p(g), _p(b),
                                     1 if training else 0, eps, momentum, _p(running_mean), _p(running_var), _p(coef),
                                     _p(Z), _p(ws), _st()), 'sug_sa_first_fwd')
        ctx.save_for_backward(P, Q, idx, coef)
        ctx.meta = (bool(training), G)
        return Z

    @staticmethod
    def backward(ctx, gz):
        P, Q, idx, coef = ctx.saved_tensors
        training, G = ctx.meta
        B, N, C = P.shape
        S, ns = idx.shape[1], idx.shape[2]
        dev = gz.device
        gz = gz.contiguous()
        red = torch.zeros(G + 1, 2 * C, dtype=torch.float64, device=dev)
        dP = torch.empty(B, N, C, dtype=torch.float32, device=dev)
        dQ = torch.empty(B, S, C, dtype=torch.float32, device=dev)
        rf = torch.empty(2 * C, dtype=torch.float32, device=dev)
        ws = torch.empty(STATS_BLOCKS * 2 * C, dtype=torch.float32, device=dev)
        off = torch.empty(B, N + 1, dtype=torch.int32, device=dev)
        ent = torch.empty(B, S * ns, dtype=torch.int32, device=dev)
        segsum = torch.empty(2, B, S, C, dtype=torch.float32, device=dev)
        check(lib().sug_sa_first_bwd(_p(gz), _p(P), C, _p(Q), _p(idx), B, N, S, ns, C, G, 1 if training else 0, _p(coef),
                                     _p(red), _p(off), _p(ent), _p(segsum), _p(dP), _p(dQ), _p(ws), _p(rf), _st()),
              'sug_sa_first_bwd')
        return dP, dQ, None, rf[C:], rf[:C], None, None, None, None, None, None


class _SAFirstLayerGeo(torch.autograd.Function):
    """relu(BN(Pf[idx] + b + Wx . (xyz[idx] - new_xyz))) over the ball-query lists (sug_sa_first_geo_fwd / bwd): the first
    layer of a set-abstraction MLP with the coordinate part taken from the difference the reference forms.  Px (= Wx . xyz
    per point) and Q (= Wx . new_xyz - b per centroid) enter only as autograd nodes: their VALUES are not read, their
    gradients (dP, dQ) route the loss to Wx and b through the ordinary linear backward."""

    @staticmethod
    def forward(ctx, Pf, Px, Q, idx, xyz, cent, Wx, bias, gamma, beta, running_mean, running_var, training, eps, momentum, G):
        _need_gpu(Px, Q, idx, xyz)
        idx = _i32(idx).contiguous()
        B, N, C = Px.shape
        S, ns = idx.shape[1], idx.shape[2]
        if B % G:
            raise RuntimeError('sa_first_layer: %d clouds do not split into %d domain groups' % (B, G))
        dev = Px.device
        Pfc = None if Pf is None else Pf.detach().contiguous()
        xyz, cent = xyz.detach().contiguous(), cent.detach().contiguous()
        Wxc = Wx.detach().contiguous()
        bc = None if bias is None else bias.detach().contiguous()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        if training:
            coef = torch.empty(G, 5, C, dtype=torch.float32, device=dev)
        else:
            coef = eval_coef(g, b, running_mean, running_var, eps).unsqueeze(0).repeat(G, 1, 1)
        Z = torch.empty(B, S, ns, C, dtype=torch.float32, device=dev)
        ws = torch.empty(STATS_BLOCKS * 2 * C, dtype=torch.float32, device=dev)
        check(lib().sug_sa_first_geo_fwd(_p(Pfc), C, _p(xyz), _p(cent), _p(Wxc), 3, _p(bc), _p(idx), B, N, S, ns, C, G, _p(g), _p(b),
                                         1 if training else 0, eps, momentum, _p(running_mean), _p(running_var), _p(coef), _p(Z),
                                         _p(ws), _st()), 'sug_sa_first_geo_fwd')
        saved = [idx, coef, xyz, cent, Wxc] + ([Pfc] if Pfc is not None else []) + ([bc] if bc is not None else [])
        ctx.save_for_backward(*saved)
        ctx.meta = (bool(training), G, Pfc is not None, bc is not None, N, C)
        return Z

    @staticmethod
    def backward(ctx, gz):
        training, G, has_p, has_b, N, C = ctx.meta
        sv = list(ctx.saved_tensors)
        idx, coef, xyz, cent, Wxc = sv[:5]
        Pfc = sv[5] if has_p else None
        bc = sv[5 + int(has_p)] if has_b else None
        B, S, ns = idx.shape
        dev = gz.device
        gz = gz.contiguous()
        red = torch.zeros(G + 1, 2 * C, dtype=torch.float64, device=dev)
        dP = torch.empty(B, N, C, dtype=torch.float32, device=dev)
        dQ = torch.empty(B, S, C, dtype=torch.float32, device=dev)
        rf = torch.empty(2 * C, dtype=torch.float32, device=dev)
        ws = torch.empty(STATS_BLOCKS * 2 * C, dtype=torch.float32, device=dev)
        off = torch.empty(B, N + 1, dtype=torch.int32, device=dev)
        ent = torch.empty(B, S * ns, dtype=torch.int32, device=dev)
        segsum = torch.empty(2, B, S, C, dtype=torch.float32, device=dev)
        check(lib().sug_sa_first_geo_bwd(_p(gz), _p(Pfc), C, _p(xyz), _p(cent), _p(Wxc), 3, _p(bc), _p(idx), B, N, S, ns, C, G,
                                         1 if training else 0, _p(coef), _p(red), _p(off), _p(ent), _p(segsum), _p(dP), _p(dQ),
                                         _p(ws), _p(rf), _st()), 'sug_sa_first_geo_bwd')
        # Pf and Px receive the SAME gradient tensor on purpose (y depends on them only through Pf[j] + Px-routed terms with
        # identical row gradients): both are non-leaf outputs of linear_rows whose backward only reads its upstream
        # gradient, so the alias is never written through; a clone would cost a pass over [B,N,C]
        return (dP if has_p else None), dP, dQ, None, None, None, None, None, rf[C:], rf[:C], None, None, None, None, None, None


SA_FIRST_GEO = _os.environ.get('SUG_SA_FIRST_GEO', '0') == '1'  # opt-in: y = Pf[j] + b + Wx.(x_j - c_s) (measured: no accuracy gain, 1 % slower)


def sa_first_layer_geo(Pf, Px, Q, idx, xyz, cent, Wx, bias, bn):
    _count_bn_call(bn)
    return _SAFirstLayerGeo.apply(Pf, Px, Q, idx, xyz, cent, Wx, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                  bn.training, bn.eps, bn.momentum, CTX.bn_groups)


SA_FIRST = _os.environ.get('SUG_SA_FIRST', '1') != '0'          # 0: grouped tensor + GEMM (the reference's arithmetic) instead


def sa_first_layer_supported(C):
    return SA_FIRST and C in (64, 128)


def sa_first_layer(P, Q, idx, bn):
    """P [B,N,C] = W.[xyz ; feats] per point, Q [B,S,C] = Wxyz.new_xyz - b per centroid, idx [B,S,ns] ball-query
    lists -> relu(bn(P[idx] - Q)) [B,S,ns,C] (train-mode statistics over each domain group's rows)."""
    _count_bn_call(bn)
    return _SAFirstLayer.apply(P, Q, idx, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.eps,
                               bn.momentum, CTX.bn_groups)


class _LNAct(torch.autograd.Function):
    """act(LayerNorm(x)) over the rows of a small [rows, C] tensor (sug_ln_act_fwd / bwd)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, slope):
        _need_gpu(x, gamma)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        rows, C = x2.shape
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y = torch.empty_like(x2)
        stat = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
        check(lib().sug_ln_act_fwd(_p(x2), _p(g), _p(b), rows, C, eps, float(slope), _p(y), _p(stat), _st()), 'sug_ln_act_fwd')
        ctx.save_for_backward(x2, g, b, stat)
        ctx.meta = (float(slope), tuple(x.shape))
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        x2, g, b, stat = ctx.saved_tensors
        slope, shape = ctx.meta
        rows, C = x2.shape
        gy = gy.reshape(rows, C).contiguous()
        dx, ws = torch.empty_like(x2), torch.empty_like(x2)
        dgb = torch.empty(2, C, dtype=torch.float32, device=x2.device)
        check(lib().sug_ln_act_bwd(_p(gy), _p(x2), _p(g), _p(b), _p(stat), rows, C, slope, _p(dx), _p(dgb[0]), _p(dgb[1]),
                                   _p(ws), _st()), 'sug_ln_act_bwd')
        return dx.view(shape), dgb[0], dgb[1], None, None


# The fused LayerNorm + activation of the FC heads saves 3 launches per layer on the GPU but costs more host time than
# the native ops it replaces (a Python autograd.Function per call): worth it when the step is replayed from a hipGraph
# (SUGStep(use_graph=True) switches it on), not when every launch is issued from Python.


def ln_act_supported(x, ln):
    return x.is_cuda and x.dtype == torch.float32 and x.shape[-1] <= 1024 and ln.elementwise_affine and \
        len(ln.normalized_shape) == 1 and ln.weight is not None and ln.bias is not None


def ln_act(x, ln, slope):
    """leaky_relu(ln(x), slope) for an nn.LayerNorm over the last dimension (slope 0: ReLU)."""
    return _LNAct.apply(x, ln.weight, ln.bias, ln.eps, slope)


# ---- classifier heads (Pointnet_c) in six launches ------------------------------------------------------------------
HEADS_FUSED = _os.environ.get('SUG_HEADS_FUSED', '1') != '0'       # A/B knob: 0 = library GEMMs + LayerNorm / dropout ops


def _ptrs(ts):
    """host array of device pointers (None -> null) for the per-head operands of sug_head_linear_*"""
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def heads_fused_supported(heads, x):
    """Can Pointnet_c heads `heads` (1 or 2, same architecture, same input x [M, K]) run through sug_head_linear_*?"""
    if not (HEADS_FUSED and 1 <= len(heads) <= 2 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        return False
    M, K = x.shape
    L = lib()
    h0 = heads[0]
    for hd in heads:
        if hd.PTran != h0.PTran or hd.training != h0.training or hd.dropout1.p != h0.dropout1.p or hd.dropout2.p != h0.dropout2.p:
            return False
        layers = ([] if hd.PTran else [hd.mlp1]) + [hd.mlp2]
        for fc in layers:
            if len(fc.fc) != 3 or not ln_act_supported(x, fc.fc[1]) or fc.fc[1].eps != h0.mlp2.fc[1].eps:
                return False
        if type(hd.mlp2.ac) is not type(h0.mlp2.ac) or (not hd.PTran and type(hd.mlp1.ac) is not type(hd.mlp2.ac)):
            return False
        for lin in [fc.fc[0] for fc in layers] + [hd.mlp3]:
            if lin.weight.dtype != torch.float32 or not lin.weight.is_cuda:
                return False
        if hd.mlp3.weight.shape != h0.mlp3.weight.shape or hd.mlp2.fc[0].weight.shape != h0.mlp2.fc[0].weight.shape:
            return False
    dims = [K] + ([] if h0.PTran else [h0.mlp1.fc[0].out_features]) + [h0.mlp2.fc[0].out_features, h0.mlp3.out_features]
    first = (h0.mlp2 if h0.PTran else h0.mlp1).fc[0]
    if first.in_features != K or h0.mlp3.in_features != dims[-2] or (not h0.PTran and h0.mlp2.fc[0].in_features != dims[1]):
        return False
    nl = len(dims) - 1
    for i in range(nl):
        pro, epi = int(i > 0), int(i < nl - 1)
        if not L.sug_head_linear_supported(M, dims[i], dims[i + 1], pro, epi):
            return False
    return True


class _HeadsFused(torch.autograd.Function):
    """Pointnet_c heads (model/Model.py:412-449) on one input: per layer ONE launch for all heads, the LayerNorm /
    activation / dropout between two Linear layers applied to the next layer's operand as it is loaded
    (sug_head_linear_fwd); in the backward one launch per Linear layer (sug_head_linear_bwd) and one per LayerNorm
    (sug_head_ln_bwd).  Arguments after the
    scalars: per head (W1, b1, g1, be1, W2, b2, g2, be2, W3, b3) -- the first four None for the two-layer (Point
    Transformer) head.  Returns per head (logits, mid feature)."""

    @staticmethod
    def forward(ctx, x, training, p1, p2, slope, eps, nheads, *params):
        _need_gpu(x)
        assert len(params) == 10 * nheads
        P = [[None if t is None else t.detach().contiguous() for t in params[10 * h:10 * h + 10]] for h in range(nheads)]
        three = P[0][0] is not None
        x2 = x.detach().contiguous()
        M, K = x2.shape
        dev = x.device
        N1 = P[0][0].shape[0] if three else 0
        N2, NC = P[0][4].shape[0], P[0][8].shape[0]
        drop1, drop2 = training and three and p1 > 0, training and p2 > 0
        # one launch of uniform randoms for every dropout of every head: [head][M][N1 | N2]
        # (flat: [head][M][N1] blocks, then [head][M][N2] blocks -- every operand a contiguous view)
        u = torch.rand(nheads * M * (N1 + N2), dtype=torch.float32, device=dev) if (drop1 or drop2) else None
        u1 = [u[h * M * N1:(h + 1) * M * N1].view(M, N1) if drop1 else None for h in range(nheads)] if three else None
        o2 = nheads * M * N1
        u2 = [u[o2 + h * M * N2:o2 + (h + 1) * M * N2].view(M, N2) if drop2 else None for h in range(nheads)]
        H = range(nheads)
        L = lib()
        col = lambda i: [P[h][i] for h in H]
        new = lambda n: [torch.empty(M, n, dtype=torch.float32, device=dev) for _ in H]
        z1 = st1 = None
        if three:
            z1, st1 = new(N1), new(2)
            check(L.sug_head_linear_fwd(nheads, _ptrs([x2] * nheads), K, _ptrs(col(0)), _ptrs(col(1)), _ptrs(z1), None, None, None,
                                        None, None, M, K, N1, 0, slope, eps, 0.0, _st()), 'sug_head_linear_fwd')
        z2, st2, mid, logits = new(N2), new(2), new(N2), new(NC)
        if three:
            check(L.sug_head_linear_fwd(nheads, _ptrs(z1), N1, _ptrs(col(4)), _ptrs(col(5)), _ptrs(z2), _ptrs(col(2)), _ptrs(col(3)),
                                        _ptrs(u1), _ptrs(st1), None, M, N1, N2, 1, slope, eps, p1 if drop1 else 0.0, _st()),
                  'sug_head_linear_fwd')
        else:
            check(L.sug_head_linear_fwd(nheads, _ptrs([x2] * nheads), K, _ptrs(col(4)), _ptrs(col(5)), _ptrs(z2), None, None, None,
                                        None, None, M, K, N2, 0, slope, eps, 0.0, _st()), 'sug_head_linear_fwd')
        check(L.sug_head_linear_fwd(nheads, _ptrs(z2), N2, _ptrs(col(8)), _ptrs(col(9)), _ptrs(logits), _ptrs(col(6)), _ptrs(col(7)),
                                    _ptrs(u2), _ptrs(st2), _ptrs(mid), M, N2, NC, 1, slope, eps, p2 if drop2 else 0.0, _st()),
              'sug_head_linear_fwd')
        # saved through autograd (version checks, freed with the graph): parameters, then per head z1 st1 z2 st2 u1 u2, then x
        flatP = [t for h in H for t in P[h]]
        acts = []
        for h in H:
            acts += [z1[h] if three else None, st1[h] if three else None, z2[h], st2[h],
                     u1[h] if (three and u1 is not None) else None, u2[h]]
        ctx.save_for_backward(*flatP, *acts, x2)
        ctx.meta = (three, M, K, N1, N2, NC, float(slope), float(eps), float(p1 if drop1 else 0.0), float(p2 if drop2 else 0.0),
                    nheads, tuple(x.shape))
        ctx.need_dx = ctx.needs_input_grad[0]
        out = []
        for h in H:
            out += [logits[h], mid[h]]
        return tuple(out)

    @staticmethod
    def backward(ctx, *gs):
        three, M, K, N1, N2, NC, slope, eps, p1, p2, nheads, xshape = ctx.meta
        saved = ctx.saved_tensors
        H = range(nheads)
        P = [list(saved[10 * h:10 * h + 10]) for h in H]
        acts = saved[10 * nheads:10 * nheads + 6 * nheads]
        x2 = saved[-1]
        z1, st1, z2, st2, u1, u2 = ([acts[6 * h + i] for h in H] for i in range(6))
        dev = x2.device
        L = lib()
        col = lambda i: [P[h][i] for h in H]
        zeros = lambda n: torch.zeros(M, n, dtype=torch.float32, device=dev)
        gl = [gs[2 * h].contiguous() if gs[2 * h] is not None else zeros(NC) for h in H]
        gm = [None if gs[2 * h + 1] is None else gs[2 * h + 1].contiguous() for h in H]
        like = lambda i: [None if P[h][i] is None else torch.empty_like(P[h][i]) for h in H]
        new = lambda n: [torch.empty(M, n, dtype=torch.float32, device=dev) for _ in H]
        grads = [like(i) for i in range(10)]
        # layer 3: logits = Dropout(act(LN(z2))) . W3^T + b3
        da2, dz2 = new(N2), new(N2)
        check(L.sug_head_linear_bwd(nheads, 0, _ptrs(gl), NC, _ptrs(z2), N2, _ptrs(st2), _ptrs(col(6)), _ptrs(col(7)), _ptrs(u2),
                                    _ptrs(col(8)), _ptrs(grads[8]), _ptrs(grads[9]), _ptrs(da2), N2, M, N2, NC, 1, slope, p2, _st()),
              'sug_head_linear_bwd')
        check(L.sug_head_ln_bwd(nheads, _ptrs(da2), N2, _ptrs(z2), _ptrs(st2), _ptrs(col(6)), _ptrs(col(7)), _ptrs(u2), _ptrs(gm),
                                _ptrs(dz2), _ptrs(grads[6]), _ptrs(grads[7]), _ptrs(grads[5]), M, N2, slope, p2, _st()), 'sug_head_ln_bwd')
        dx = torch.empty(M, K, dtype=torch.float32, device=dev) if ctx.need_dx else None
        dxs = _ptrs([dx] + [None] * (nheads - 1))
        if three:
            da1, dz1 = new(N1), new(N1)
            check(L.sug_head_linear_bwd(nheads, 0, _ptrs(dz2), N2, _ptrs(z1), N1, _ptrs(st1), _ptrs(col(2)), _ptrs(col(3)), _ptrs(u1),
                                        _ptrs(col(4)), _ptrs(grads[4]), None, _ptrs(da1), N1, M, N1, N2, 1, slope, p1, _st()),
                  'sug_head_linear_bwd')
            check(L.sug_head_ln_bwd(nheads, _ptrs(da1), N1, _ptrs(z1), _ptrs(st1), _ptrs(col(2)), _ptrs(col(3)), _ptrs(u1), None,
                                    _ptrs(dz1), _ptrs(grads[2]), _ptrs(grads[3]), _ptrs(grads[1]), M, N1, slope, p1, _st()),
                  'sug_head_ln_bwd')
            check(L.sug_head_linear_bwd(nheads, 1, _ptrs(dz1), N1, _ptrs([x2] * nheads), K, None, None, None, None, _ptrs(col(0)),
                                        _ptrs(grads[0]), None, dxs, K, M, K, N1, 0, slope, 0.0, _st()), 'sug_head_linear_bwd')
        else:
            check(L.sug_head_linear_bwd(nheads, 1, _ptrs(dz2), N2, _ptrs([x2] * nheads), K, None, None, None, None, _ptrs(col(4)),
                                        _ptrs(grads[4]), None, dxs, K, M, K, N2, 0, slope, 0.0, _st()), 'sug_head_linear_bwd')
        flat = []
        for h in H:
            flat += [grads[i][h] for i in range(10)]
        return (None if dx is None else dx.view(xshape), None, None, None, None, None, None) + tuple(flat)


def heads_fused(heads, x):
    """[(logits, mid feature) per head] of Pointnet_c heads on the pooled feature x [M, K] (see _HeadsFused)."""
    h0 = heads[0]
    act = h0.mlp2.ac
    slope = 0.0 if isinstance(act, torch.nn.ReLU) else float(act.negative_slope)
    params = []
    for hd in heads:
        if hd.PTran:
            params += [None, None, None, None]
        else:
            params += [hd.mlp1.fc[0].weight, hd.mlp1.fc[0].bias, hd.mlp1.fc[1].weight, hd.mlp1.fc[1].bias]
        params += [hd.mlp2.fc[0].weight, hd.mlp2.fc[0].bias, hd.mlp2.fc[1].weight, hd.mlp2.fc[1].bias, hd.mlp3.weight, hd.mlp3.bias]
    out = _HeadsFused.apply(x, h0.training, float(h0.dropout1.p), float(h0.dropout2.p), slope, float(h0.mlp2.fc[1].eps),
                            len(heads), *params)
    return [(out[2 * i], out[2 * i + 1]) for i in range(len(heads))]


# num_batches_tracked increments: one tiny launch per BatchNorm call unless deferred; inside a
# `deferred_bn_counts()` block they are collected and applied with one foreach add at the end.


@contextlib.contextmanager
def deferred_bn_counts():
    if CTX.pending_counts is not None:          # nested: the outer block flushes
        yield
        return
    CTX.pending_counts = {}
    try:
        yield
    finally:
        flush_bn_counts()
        CTX.pending_counts = None


def flush_bn_counts():
    """Apply the counter increments collected so far by the enclosing deferred_bn_counts() block (one multi-tensor launch,
    a scalar per counter); whoever reads num_batches_tracked inside such a block calls this first."""
    pend = CTX.pending_counts
    if pend:
        torch._foreach_add_([t for t, _ in pend.values()], [n for _, n in pend.values()])
        pend.clear()


def _count_bn_call(bn, n=None):
    if bn.training and bn.num_batches_tracked is not None:
        n = CTX.bn_groups if n is None else n
        if CTX.pending_counts is not None:
            t = bn.num_batches_tracked
            old = CTX.pending_counts.get(id(t))
            CTX.pending_counts[id(t)] = (t, n + (old[1] if old else 0))
        else:
            bn.num_batches_tracked.add_(n)


# A shared prefix (DGCNN / Point Transformer / PointNet encoders: the part of the semantic and of the node pass of a step that
# is the same computation) is run once; the second pass only REPLAYS the running-statistics updates of its BatchNorm layers.
# `with record_bn_stats() as rec:` collects (module, batch-statistics coefficients) of every train-mode BatchNorm op inside.


@contextlib.contextmanager
def record_bn_stats():
    prev, CTX.bn_record = CTX.bn_record, []
    try:
        yield CTX.bn_record
    finally:
        CTX.bn_record = prev


def _record_bn(bn):
    if CTX.bn_record is not None and bn.training and bn.track_running_stats:
        CTX.bn_record.append((bn, CTX.last_coef))


def replay_bn_stats(rec):
    """bn_replay for every (bn, coef) of a recorded prefix, in order; runs of distinct layers share one launch."""
    rec = list(rec)
    i = 0
    while i < len(rec):
        run, seen = [], set()
        while i < len(rec) and len(run) < 16 and id(rec[i][0]) not in seen:     # a layer recorded twice: its updates stay in order
            seen.add(id(rec[i][0]))
            run.append(rec[i])
            i += 1
        if len(run) == 1:
            bn_replay(*run[0])
            continue
        c3s = []
        for _, coef in run:
            c3 = coef if coef.dim() == 3 else coef.unsqueeze(0)
            c3s.append(c3 if c3.is_contiguous() else c3.contiguous())
        n = len(run)
        I32, F32 = ctypes.c_int32 * n, ctypes.c_float * n
        check(lib().sug_bn_replay_multi(n, _ptrs(c3s), I32(*[c.shape[0] for c in c3s]), I32(*[c.shape[2] for c in c3s]),
                                        F32(*[float(bn.momentum) for bn, _ in run]), _ptrs([bn.running_mean for bn, _ in run]),
                                        _ptrs([bn.running_var for bn, _ in run]), _st()), 'sug_bn_replay_multi')
        for (bn, _), c3 in zip(run, c3s):
            _count_bn_call(bn, c3.shape[0])


def bn_replay(bn, coef):
    """Running-statistics update of len(coef) more train-mode forwards with known batch statistics
    (coef [5,C] or [G,5,C] from a BN op), in order (sug_bn_replay)."""
    c3 = coef if coef.dim() == 3 else coef.unsqueeze(0)
    c3 = c3 if c3.is_contiguous() else c3.contiguous()
    G, _, C = c3.shape
    check(lib().sug_bn_replay(_p(c3), G, C, float(bn.momentum), _p(bn.running_mean), _p(bn.running_var), _st()),
          'sug_bn_replay')
    _count_bn_call(bn, G)


def bn_act_rows(y, bn, slope):
    """bn: an nn.BatchNorm{1,2}d module whose parameters / running buffers are used."""
    _count_bn_call(bn)
    out = _BNActRows.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, slope, bn.eps,
                           bn.momentum, CTX.bn_groups)
    _record_bn(bn)
    return out


class _BNActPool(torch.autograd.Function):
    """y [B,N,C] -> (max_n, mean_n) of LeakyReLU(slope)(BatchNorm(y)); one read of y forward,
    two backward (sug_bn_act_pool_*)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, slope, eps, momentum, G):
        _need_gpu(y, gamma)
        y, B, N, C, ld = _rows3(y)
        if B % G:
            raise RuntimeError('bn_act_pool: %d clouds do not split into %d domain groups' % (B, G))
        dev = y.device
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        if training:
            coef = torch.empty(G, 5, C, dtype=torch.float32, device=dev)
        else:
            coef = eval_coef(g, b, running_mean, running_var, eps).unsqueeze(0).repeat(G, 1, 1)
        pooled = torch.empty(B, 2 * C, dtype=torch.float32, device=dev)      # [max | mean]: torch.cat((max, mean), 1) in place
        omax, omean = pooled[:, :C], pooled[:, C:]
        arg = torch.empty(B, C, dtype=torch.int32, device=dev)
        stats = torch.empty(2 * C, dtype=torch.float64, device=dev)
        wss = torch.empty(STATS_BLOCKS * 2 * C, dtype=torch.float32, device=dev)
        wsp = torch.empty(12 * (B // G) * C, dtype=torch.float32, device=dev)
        check(lib().sug_bn_act_pool_layer_fwd(_p(y), ld, B, N, C, G, _p(g), _p(b), 1 if training else 0, eps, momentum,
                                              float(slope), _p(running_mean), _p(running_var), _p(coef), _p(omax),
                                              _p(omean), 2 * C, _p(arg), _p(stats), _p(wss), _p(wsp), _st()),
              'sug_bn_act_pool_layer_fwd')
        ctx.save_for_backward(y, coef, arg)
        ctx.meta = (B, N, C, ld, float(slope), bool(training), G)
        return pooled

    @staticmethod
    def backward(ctx, g):
        y, coef, arg = ctx.saved_tensors
        B, N, C, ld, slope, training, G = ctx.meta
        dev = y.device
        if g.stride(1) != 1 or g.stride(0) < 2 * C:
            g = g.contiguous()
        gmax, gmean, ldp = g[:, :C], g[:, C:], g.stride(0)
        red = torch.empty(G, 2 * C, dtype=torch.float64, device=dev)
        ws = torch.empty(STATS_BLOCKS * 2 * C, dtype=torch.float32, device=dev)
        dy = torch.empty(B, N, C, dtype=torch.float32, device=dev)
        rf = torch.empty(2 * C, dtype=torch.float32, device=dev)
        check(lib().sug_bn_act_pool_layer_bwd(_p(y), ld, _p(coef), _p(gmax), _p(gmean), ldp, _p(arg), B, N, C, G, slope,
                                              1 if training else 0, _p(red), _p(ws), _p(dy), C, _p(rf), _st()),
              'sug_bn_act_pool_layer_bwd')
        return dy, rf[C:], rf[:C], None, None, None, None, None, None, None


def bn_update_stats(y, bn):
    """Train-mode BatchNorm side effect only: batch statistics of y [..., C] per domain group ->
    running-buffer update (and num_batches_tracked); the normalised output is not produced.  For a
    layer whose output is discarded but whose buffers the reference still updates."""
    if not bn.training:
        return
    _need_gpu(y)
    C = y.shape[-1]
    y2 = y.detach().reshape(-1, C)
    if y2.stride(1) != 1:
        y2 = y2.contiguous()
    rows, G = y2.shape[0], CTX.bn_groups
    if rows % G:
        raise RuntimeError('bn_update_stats: %d rows do not split into %d domain groups' % (rows, G))
    rg = rows // G
    g, b = bn.weight.detach().contiguous(), bn.bias.detach().contiguous()
    coef = torch.empty(5, C, dtype=torch.float32, device=y.device)
    ws = torch.empty(STATS_BLOCKS * 2 * C, dtype=torch.float32, device=y.device)
    for i in range(G):
        yi = y2[i * rg:(i + 1) * rg]
        check(lib().sug_col_stats_bn(_p(yi), yi.stride(0), rg, C, _p(g), _p(b), bn.eps, bn.momentum,
                                     _p(bn.running_mean), _p(bn.running_var), _p(coef), _p(ws), _st()),
              'sug_col_stats_bn')
    _count_bn_call(bn)


def bn_act_pool_cat(y, bn, slope):
    """cat((max over points, mean over points), 1) [B, 2C] of act(bn(y)), y [B,N,C]: the kernels write the two halves of
    the concatenated feature (Model.py:113-116) and read the two halves of its gradient in place."""
    _count_bn_call(bn)
    return _BNActPool.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, slope, bn.eps,
                            bn.momentum, CTX.bn_groups)


def bn_act_pool(y, bn, slope):
    """(max over points, mean over points) of act(bn(y)), y [B,N,C]."""
    C = y.shape[-1]
    pooled = bn_act_pool_cat(y, bn, slope)
    return pooled[:, :C], pooled[:, C:]


# ----------------------------------------------------------------------------- EdgeConv
class _EdgeConv(torch.autograd.Function):
    """BN(train or eval) + LeakyReLU + max over k of y = P[idx] + Q, see include/sug_amd.h."""

    @staticmethod
    def forward(ctx, pq, idx, gamma, beta, running_mean, running_var, training, slope, eps, momentum, G, out_holder,
                grad_on=True):
        _need_gpu(pq, idx, gamma)
        pq, B, N, C2, ld = _rows3(pq)
        Co = C2 // 2
        idx = _i32(idx).contiguous()
        k = idx.shape[2]
        if B % G:
            raise RuntimeError('edgeconv: %d clouds do not split into %d domain groups' % (B, G))
        dev = pq.device
        gamma_c, beta_c = gamma.detach().contiguous(), beta.detach().contiguous()
        # (needs_input_grad reports requires_grad of the inputs even under torch.no_grad(): the caller passes the grad
        # mode, or the no-grad passes of a step would write s1 -- 4*Co bytes per point -- for a backward that never runs)
        need_bwd = grad_on and any(ctx.needs_input_grad[i] for i in (0, 2, 3))
        z = torch.empty(B, N, Co, dtype=torch.float32, device=dev)
        arg = torch.empty(B, N, Co, dtype=torch.uint8, device=dev)
        s1 = torch.empty(B, N, Co, dtype=torch.float32, device=dev) if need_bwd else None
        stats = torch.empty(2 * Co, dtype=torch.float64, device=dev)
        ws = torch.empty(STATS_BLOCKS * 2 * Co, dtype=torch.float32, device=dev)
        coef = torch.empty(G, 5, Co, dtype=torch.float32, device=dev)
        if not training:
            coef.copy_(eval_coef(gamma_c, beta_c, running_mean, running_var, eps))
        if out_holder is None:
            out = torch.empty(B, N, Co, dtype=torch.float32, device=dev)
        else:
            # caller-provided destination (a column slice of a wider rows buffer, see assemble_rows);
            # passed in a list so that autograd does not treat it as an input of this node
            out = out_holder[0]
            if tuple(out.shape) != (B, N, Co) or out.stride(2) != 1 or out.stride(0) != N * out.stride(1) \
                    or out.dtype != torch.float32 or out.requires_grad:
                raise RuntimeError('edgeconv: bad destination slice')
        ldo = out.stride(1)
        check(_timed('edgeconv_layer_fwd_Co%d' % Co, {'B': B, 'N': N, 'k': k, 'Co': Co},
                     lambda: lib().sug_edgeconv_layer_fwd(_p(pq), ld, _p(idx), _p(gamma_c), _p(beta_c), B, N, k, Co, G,
                                                          1 if training else 0, eps, momentum, float(slope),
                                                          _p(running_mean), _p(running_var), _p(z), _p(arg), _p(s1),
                                                          _p(coef), _p(out), ldo, _p(stats), _p(ws), _st())),
              'sug_edgeconv_layer_fwd')
        if need_bwd:
            ctx.save_for_backward(pq, idx, z, arg, s1, coef)
            ctx.meta = (B, N, k, Co, ld, float(slope), bool(training), G)
        coef_out = coef[0] if G == 1 else coef
        ctx.mark_non_differentiable(coef_out)
        ctx.set_materialize_grads(False)
        return out, coef_out

    @staticmethod
    def backward(ctx, gout, _gcoef):
        pq, idx, z, arg, s1, coef = ctx.saved_tensors
        B, N, k, Co, ld, slope, training, G = ctx.meta
        if gout is None:
            return (None,) * 13
        dev = gout.device
        gout, _, _, _, ldg = _rows3(gout)                           # a column slice of a wider buffer is fine
        a = torch.empty(B, N, Co, dtype=torch.float32, device=dev)
        # per-group dbeta | dgamma, plus a zero row the scatter reads in eval mode (statistics constant)
        red = (torch.empty if training else torch.zeros)(G + 1, 2 * Co, dtype=torch.float64, device=dev)
        ws = torch.empty(STATS_BLOCKS * 2 * Co, dtype=torch.float32, device=dev)
        off = torch.empty(B, N + 1, dtype=torch.int32, device=dev)
        ent = torch.empty(B, N * k, dtype=torch.int32, device=dev)
        dpq = torch.empty(B, N, 2 * Co, dtype=torch.float32, device=dev)
        rf = torch.empty(2 * Co, dtype=torch.float32, device=dev)
        check(_timed('edgeconv_layer_bwd_Co%d' % Co, {'B': B, 'N': N, 'k': k, 'Co': Co},
                     lambda: lib().sug_edgeconv_layer_bwd(_p(gout), ldg, _p(z), _p(arg), _p(s1), _p(pq), ld, _p(idx),
                                                          _p(coef), B, N, k, Co, G, 1 if training else 0, slope, _p(a),
                                                          _p(red), _p(off), _p(ent), _p(dpq), 2 * Co, _p(ws), _p(rf), _st())),
              'sug_edgeconv_layer_bwd')
        return dpq, None, rf[Co:], rf[:Co], None, None, None, None, None, None, None, None, None


def edgeconv_bn_act_max(pq, idx, gamma, beta, running_mean, running_var, training, slope, eps=1e-5, momentum=0.1,
                        out=None):
    """pq [B,N,2*Co] = x.[W1;W2-W1]^T, idx [B,N,k] -> (out [B,N,Co], coef [5,Co] = scale, shift,
    batch mean, rstd, unbiased batch variance; [G,5,Co] under bn_groups(G > 1))."""
    return _EdgeConv.apply(pq, idx, gamma, beta, running_mean, running_var, training, slope, eps, momentum, CTX.bn_groups,
                           None if out is None else [out], torch.is_grad_enabled())


# The 1x1 convolution inside the gather kernel (edgeconv_fused.hip): default; SUG_EDGECONV_FUSED=0 selects the
# library GEMM + edgeconv_bn_act_max path (A/B measurements).
EDGECONV_FUSED = _os.environ.get('SUG_EDGECONV_FUSED', '1') == '1'


# Widest input the fused layer takes by default.  A workgroup = (cloud, 16-channel slice) streams the cloud's x rows
# through L2 once per slice; at Cin = 128 / Co = 256 (DGCNN conv4: 16 slices x 512 KB per cloud) those re-reads cost
# more than the [P|Q] round trip of the library-GEMM path (tools/bench_edgeconv_fused.py: 208 vs ~190 us per 64 clouds),
# at Cin <= 64 the fused layer is level or ahead.  SUG_EDGECONV_FUSED_MAXC=128 fuses every layer.
EDGECONV_FUSED_MAXC = int(_os.environ.get('SUG_EDGECONV_FUSED_MAXC', '64'))


def edgeconv_fused_supported(N, k, Cin, Co):
    return EDGECONV_FUSED and Cin <= EDGECONV_FUSED_MAXC and bool(lib().sug_edgeconv_fused_supported(N, k, Cin, Co))


class _EdgeConvFused(torch.autograd.Function):
    """max_k LeakyReLU(BN(W.[x_j - x_i ; x_i])) from the rows x and the split weight [W1 ; W2-W1]: the GEMM, the
    neighbour gather, the reduction over k and the BatchNorm sums in one kernel, BatchNorm + activation in a second
    (sug_edgeconv_fused_layer_fwd).  [P|Q] is written only for a backward (sug_edgeconv_layer_bwd reads it)."""

    @staticmethod
    def forward(ctx, x, wcat, bias, idx, gamma, beta, running_mean, running_var, training, slope, eps, momentum, G,
                out_holder, grad_on=True):
        _need_gpu(x, wcat, idx, gamma)
        x3, B, N, C, ld = _rows3(x)
        Co = wcat.shape[0] // 2
        idx = _i32(idx).contiguous()
        k = idx.shape[2]
        if B % G:
            raise RuntimeError('edgeconv: %d clouds do not split into %d domain groups' % (B, G))
        dev = x.device
        w = wcat.detach().contiguous()
        b1 = None if bias is None else bias.detach().contiguous()
        gamma_c, beta_c = gamma.detach().contiguous(), beta.detach().contiguous()
        # (needs_input_grad ignores torch.no_grad(): the caller passes the grad mode -- see _EdgeConv)
        need_bwd = grad_on and any(ctx.needs_input_grad[i] for i in (0, 1, 2, 4, 5))
        z = torch.empty(B, N, Co, dtype=torch.float32, device=dev)
        arg = torch.empty(B, N, Co, dtype=torch.uint8, device=dev) if need_bwd else None
        s1 = torch.empty(B, N, Co, dtype=torch.float32, device=dev) if need_bwd else None
        pq = torch.empty(B, N, 2 * Co, dtype=torch.float32, device=dev) if need_bwd else None
        ws = torch.empty(STATS_BLOCKS * 2 * Co, dtype=torch.float32, device=dev)
        coef = torch.empty(G, 5, Co, dtype=torch.float32, device=dev)
        if not training:
            coef.copy_(eval_coef(gamma_c, beta_c, running_mean, running_var, eps))
        if out_holder is None:
            out = torch.empty(B, N, Co, dtype=torch.float32, device=dev)
        else:
            out = out_holder[0]
            if tuple(out.shape) != (B, N, Co) or out.stride(2) != 1 or out.stride(0) != N * out.stride(1) \
                    or out.dtype != torch.float32 or out.requires_grad:
                raise RuntimeError('edgeconv: bad destination slice')
        check(_timed('edgeconv_fused_fwd_C%d_Co%d' % (C, Co), {'B': B, 'N': N, 'k': k, 'Co': Co, 'C': C, 'train': int(need_bwd)},
                     lambda: lib().sug_edgeconv_fused_layer_fwd(_p(x3), ld, C, _p(w), _p(b1), _p(idx), _p(gamma_c), _p(beta_c),
                                                                B, N, k, Co, G, 1 if training else 0, eps, momentum,
                                                                float(slope), _p(running_mean), _p(running_var), _p(z),
                                                                _p(arg), _p(s1), _p(pq), 2 * Co, _p(coef), _p(out),
                                                                out.stride(1), _p(ws), _st())),
              'sug_edgeconv_fused_layer_fwd')
        if need_bwd:
            ctx.save_for_backward(x3, w, idx, z, arg, s1, coef, pq)
            ctx.meta = (B, N, k, C, Co, float(slope), bool(training), G, b1 is not None, tuple(x.shape))
        coef_out = coef[0] if G == 1 else coef
        ctx.mark_non_differentiable(coef_out)
        ctx.set_materialize_grads(False)
        return out, coef_out

    @staticmethod
    def backward(ctx, gout, _gcoef):
        if gout is None:
            return (None,) * 15
        x3, w, idx, z, arg, s1, coef, pq = ctx.saved_tensors
        B, N, k, C, Co, slope, training, G, has_bias, xshape = ctx.meta
        dev = gout.device
        gout, _, _, _, ldg = _rows3(gout)                           # a column slice of a wider buffer is fine
        a = torch.empty(B, N, Co, dtype=torch.float32, device=dev)
        red = (torch.empty if training else torch.zeros)(G + 1, 2 * Co, dtype=torch.float64, device=dev)
        ws = torch.empty(STATS_BLOCKS * 2 * Co, dtype=torch.float32, device=dev)
        off = torch.empty(B, N + 1, dtype=torch.int32, device=dev)
        ent = torch.empty(B, N * k, dtype=torch.int32, device=dev)
        dpq = torch.empty(B, N, 2 * Co, dtype=torch.float32, device=dev)
        rf = torch.empty(2 * Co, dtype=torch.float32, device=dev)
        check(_timed('edgeconv_layer_bwd_Co%d' % Co, {'B': B, 'N': N, 'k': k, 'Co': Co},
                     lambda: lib().sug_edgeconv_layer_bwd(_p(gout), ldg, _p(z), _p(arg), _p(s1), _p(pq), 2 * Co, _p(idx),
                                                          _p(coef), B, N, k, Co, G, 1 if training else 0, slope, _p(a),
                                                          _p(red), _p(off), _p(ent), _p(dpq), 2 * Co, _p(ws), _p(rf), _st())),
              'sug_edgeconv_layer_bwd')
        x2 = x3.reshape(B * N, C) if x3.is_contiguous() else x3.view(B * N, C) if x3.stride(0) == N * x3.stride(1) else x3.reshape(B * N, C)
        dpq2 = dpq.view(B * N, 2 * Co)
        dx, dw, _ = linear_rows_backward(x2, w, dpq2, ctx.needs_input_grad[0], ctx.needs_input_grad[1], False)
        db = None
        if has_bias and ctx.needs_input_grad[2]:
            db = colsum(dpq2[:, Co:])                              # the bias rides on the Q half
        return (None if dx is None else dx.view(xshape)), dw, db, None, rf[Co:], rf[:Co], None, None, None, None, None, \
            None, None, None, None


def edgeconv_fused(x, wcat, bias, idx, bn_weight, bn_bias, running_mean, running_var, training, slope, eps=1e-5,
                   momentum=0.1, out=None):
    """x [B,N,C] rows, wcat [2Co, C] = [W1 ; W2-W1], idx [B,N,k] -> (out [B,N,Co], coef) as edgeconv_bn_act_max."""
    return _EdgeConvFused.apply(x, wcat, bias, idx, bn_weight, bn_bias, running_mean, running_var, training, slope, eps,
                                momentum, CTX.bn_groups, None if out is None else [out], torch.is_grad_enabled())


# ----------------------------------------------------------------------------- per-point MLP + max
POINTMLP_MAX = _os.environ.get('SUG_POINTMLP_MAX', '1') != '0'    # 0: library GEMM + BatchNorm rows kernels + torch.max instead


def pointmlp_max_supported(K, Co, seg):
    return POINTMLP_MAX and K in (64, 128) and Co % 128 == 0 and seg >= 32 and seg % 32 == 0


class _PointMLPMax(torch.autograd.Function):
    """max over `seg` consecutive rows of act(BN(x . W^T + b)) without ever storing the [rows, Co]
    product (sug_pointmlp_max_*).  Backward: the rows that won a channel get a[s,c]*W[c,:]
    (sug_pointmlp_max_bwd_sparse); the train-mode BatchNorm statistics terms, which are dense over the
    rows but of rank K, are -(x.A + v) with A = W^T diag(k2) W -- one [rows,K]x[K,K] GEMM instead of
    the [rows,Co]x[Co,K] one -- and the matching weight-gradient term comes from X^T X."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, slope, eps, momentum, seg, G):
        _need_gpu(x, weight, gamma)
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        if x2.stride(1) != 1 or x2.stride(0) % 4 or x2.data_ptr() % 16:
            x2 = x2.contiguous()
        rows, Co = x2.shape[0], weight.shape[0]
        if rows % (G * seg):
            raise RuntimeError('pointmlp_max: %d rows do not split into %d groups of %d-row segments' % (rows, G, seg))
        S = rows // seg
        dev = x.device
        w2 = weight.detach().reshape(Co, K).contiguous()
        b1 = None if bias is None else bias.detach().contiguous()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        zext = torch.empty(S, Co, dtype=torch.float32, device=dev)
        arg = torch.empty(S, Co, dtype=torch.int32, device=dev)
        out = torch.empty(S, Co, dtype=torch.float32, device=dev)
        ws = torch.empty(STATS_BLOCKS * 2 * Co, dtype=torch.float32, device=dev)
        if training:
            coef = torch.empty(G, 5, Co, dtype=torch.float32, device=dev)
        else:
            coef = eval_coef(g, b, running_mean, running_var, eps).unsqueeze(0).repeat(G, 1, 1)
        check(_timed('pointmlp_max_K%d_Co%d' % (K, Co), {'B': rows, 'N': seg, 'k': K, 'Co': Co},
                     lambda: lib().sug_pointmlp_max_layer_fwd(_p(x2), x2.stride(0), rows, K, _p(w2), _p(b1), _p(g), _p(b),
                                                              Co, seg, G, 1 if training else 0, eps, momentum,
                                                              float(slope), _p(running_mean), _p(running_var), _p(zext),
                                                              _p(arg), _p(coef), _p(out), Co, _p(ws), _st())),
              'sug_pointmlp_max_layer_fwd')
        CTX.last_coef = coef
        if any(ctx.needs_input_grad[i] for i in (0, 1, 2, 3, 4)):
            ctx.save_for_backward(x2, w2, b1, zext, arg, coef)
            ctx.meta = (rows, K, Co, seg, G, float(slope), bool(training), tuple(x.shape), tuple(weight.shape))
        return out

    @staticmethod
    def backward(ctx, gout):
        x2, w2, b1, zext, arg, coef = ctx.saved_tensors
        rows, K, Co, seg, G, slope, training, xshape, wshape = ctx.meta
        dx, dw, db, dgamma, dbeta = _pointmlp_max_backward(gout, x2, w2, b1, zext, arg, coef, rows, K, Co, seg, G, slope, training)
        return dx.view(xshape), dw.view(wshape), db, dgamma, dbeta, None, None, None, None, None, None, None, None


def _pointmlp_max_backward(gout, x2, w2, b1, zext, arg, coef, rows, K, Co, seg, G, slope, training):
    """Backward of the fused per-point MLP + max layer (see _PointMLPMax): -> (dx [rows, K], dw [Co, K], db, dgamma, dbeta)."""
    dev = gout.device
    S = rows // seg
    rg, sg = rows // G, S // G
    gout = gout.reshape(S, Co)
    if gout.stride(1) != 1 or gout.stride(0) % 4 or gout.data_ptr() % 16:
        gout = gout.contiguous()        # e.g. a column slice of a [.., 3 + C] gradient: the float4 kernels need aligned rows
    a = torch.empty(S, Co, dtype=torch.float32, device=dev)
    red = torch.empty(G, 2 * Co, dtype=torch.float64, device=dev)
    ws = torch.empty(STATS_BLOCKS * 2 * Co, dtype=torch.float32, device=dev)
    dx = torch.empty(rows, K, dtype=torch.float32, device=dev) if training else \
        torch.zeros(rows, K, dtype=torch.float32, device=dev)
    # dw (accumulated over the groups) and the identically-zero bias gradient of a train-mode layer: ONE zero fill
    zdb = Co if (b1 is not None and training) else 0
    zbuf = torch.zeros(Co * K + zdb, dtype=torch.float32, device=dev)
    dw = zbuf[:Co * K].view(Co, K)
    dws = torch.empty(Co, K, dtype=torch.float32, device=dev)
    wsp = torch.empty(int(lib().sug_pointmlp_max_bwd_workspace(rg, K, Co, seg)), dtype=torch.float32, device=dev)
    L = lib()
    f32 = dict(dtype=torch.float32, device=dev)
    kbk2, negA, negv = torch.empty(2, Co, **f32), torch.empty(K, K, **f32), torch.empty(K, **f32)
    nkb, nwk = torch.empty(Co, **f32), torch.empty(Co, K, **f32)
    xtx, sx = torch.empty(K, K, **f32), torch.empty(K, **f32)
    wsx = torch.empty(int(L.sug_linear_dw_workspace(rg, K, K)), **f32) if training else None
    for gi in range(G):
        xg = x2[gi * rg:(gi + 1) * rg]
        zg, ag, cg = zext[gi * sg:(gi + 1) * sg], a[gi * sg:(gi + 1) * sg], coef[gi]
        gg = gout[gi * sg:(gi + 1) * sg]
        check(L.sug_edgeconv_bwd_reduce(_p(gg), gg.stride(0), _p(zg), _p(cg), sg, Co, slope, _p(ag), _p(red[gi]), _p(ws),
                                        _st()), 'sug_edgeconv_bwd_reduce')
        dxg = dx[gi * rg:(gi + 1) * rg]
        if training:
            # dy = a_full - (scale/M)(dbeta + xhat*dgamma) = a_full - k1 - k2*y over ALL rows, y = x.W^T + b:
            # the coefficients kb, k2 and the operands of -A, -v from one launch (sug_pointmlp_max_bwd_coef)
            check(L.sug_pointmlp_max_bwd_coef(_p(cg), _p(red[gi]), _p(b1), _p(w2), rg, K, Co, _p(kbk2[0]), _p(kbk2[1]),
                                              _p(nkb), _p(nwk), _st()), 'sug_pointmlp_max_bwd_coef')
            torch.mm(nwk.t(), w2, out=negA)                         # -A = -(W^T diag(k2) W)   [K,K]
            torch.mv(w2.t(), nkb, out=negv)                         # -v = -(kb . W)           [K]
            check(L.sug_linear_dw_bias(_p(xg), xg.stride(0), _p(xg), xg.stride(0), rg, K, K, _p(xtx), _p(sx), _p(wsx),
                                       _st()), 'sug_linear_dw_bias')        # X^T X and the column sums of x
            # -(x.A + v): alpha = beta = 1 keeps the library's bias epilogue (any other alpha first expands
            # the bias into dx: a full extra pass)
            torch.addmm(negv, xg, negA, out=dxg)
        check(L.sug_pointmlp_max_bwd_sparse(_p(ag), _p(arg[gi * sg:(gi + 1) * sg]), _p(xg), xg.stride(0), _p(w2), rg, K,
                                            Co, seg, _p(dxg), K, _p(dws), _p(wsp), _st()),
              'sug_pointmlp_max_bwd_sparse')
        check(L.sug_pointmlp_max_bwd_dwfix(_p(dw), _p(dws), _p(kbk2[0]), _p(kbk2[1]), _p(w2), _p(xtx), _p(sx), K, Co,
                                           1 if training else 0, _st()), 'sug_pointmlp_max_bwd_dwfix')
    rf = torch.empty(2 * Co, dtype=torch.float32, device=dev)
    check(L.sug_fold_groups(_p(red), G, 2 * Co, _p(rf), _st()), 'sug_fold_groups')      # dbeta | dgamma over the groups, in group order
    db = None
    if b1 is not None:
        # a bias in front of train-mode BatchNorm has zero gradient identically
        db = zbuf[Co * K:].view_as(b1) if training else a.sum(dim=0)
    return dx, dw, db, rf[Co:], rf[:Co]


def pointmlp_max(x, weight, bias, bn, slope, seg):
    """x [..., K] rows (segments of `seg` consecutive rows), weight [Co, K(,1,1)], bn an nn.BatchNorm
    module -> [rows/seg, Co] = max over each segment of LeakyReLU_slope(bn(x.W^T + b))."""
    _count_bn_call(bn)
    out = _PointMLPMax.apply(x, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training,
                             slope, bn.eps, bn.momentum, seg, CTX.bn_groups)
    _record_bn(bn)
    return out


# The middle layer of a set-abstraction MLP folded into the fused last layer (round 5): the middle layer's BatchNorm + ReLU is
# applied by sug_pointmlp_max_layer_fwd_xf to its x operand on the way into LDS, the activated tensor z is written once as a
# side output (only when a backward or the SA-node features need it) -- instead of a separate pass that reads y and writes z
# followed by the last layer's read of z.  SUG_SA_MID_FUSED=0 restores the separate BatchNorm pass (A/B).
SA_MID_FUSED = _os.environ.get('SUG_SA_MID_FUSED', '1') != '0'


class _BNActPointMLPMax(torch.autograd.Function):
    """(max over `seg` rows of act2(BN2(z . W^T + b)), z) with z = act1(BN1(y)) formed inside the kernel from the
    pre-activation rows y [rows, K] of the previous layer (pointnet2_utils.py:193-207: mlp_bns[i] / relu of layer i, conv /
    bn / relu / max of layer i + 1).  Outputs: out [rows/seg, Co] and z [rows, K] (None unless wanted).  Backward = the two
    layers' backward kernels unchanged: _pointmlp_max_backward on (z, gout), the gradient arriving at z added, then
    _bn_act_rows_backward through BN1."""

    @staticmethod
    def forward(ctx, y, g1, be1, rm1, rv1, slope1, eps1, mom1, weight, bias, g2, be2, rm2, rv2, slope2, eps2, mom2,
                training, seg, G, want_z, last_grad, grad_on):
        _need_gpu(y, weight, g1, g2)
        K = y.shape[-1]
        y2 = y.reshape(-1, K)
        if y2.stride(1) != 1 or y2.stride(0) != K or y2.data_ptr() % 16:
            y2 = y2.contiguous()
        rows, Co = y2.shape[0], weight.shape[0]
        if rows % (G * seg):
            raise RuntimeError('bn_act_pointmlp_max: %d rows do not split into %d groups of %d-row segments' % (rows, G, seg))
        S, rg = rows // seg, rows // G
        dev = y.device
        L = lib()
        g1c, be1c = g1.detach().contiguous(), be1.detach().contiguous()
        ws = torch.empty(STATS_BLOCKS * 2 * max(K, Co), dtype=torch.float32, device=dev)
        if training:
            coef1 = torch.empty(G, 5, K, dtype=torch.float32, device=dev)
            check(L.sug_col_stats_bn_grouped(_p(y2), K, rows, K, G, _p(g1c), _p(be1c), eps1, mom1, _p(rm1), _p(rv1), _p(coef1),
                                             _p(ws), _st()), 'sug_col_stats_bn_grouped')
        else:
            coef1 = eval_coef(g1c, be1c, rm1, rv1, eps1).unsqueeze(0).repeat(G, 1, 1).contiguous()
        need1 = grad_on and any(ctx.needs_input_grad[i] for i in (0, 1, 2))
        need2 = grad_on and last_grad and (need1 or any(ctx.needs_input_grad[i] for i in (8, 9, 10, 11)))
        z = torch.empty(rows, K, dtype=torch.float32, device=dev) if (want_z or need2) else None
        w2 = weight.detach().reshape(Co, K).contiguous()
        b1 = None if bias is None else bias.detach().contiguous()
        g2c, be2c = g2.detach().contiguous(), be2.detach().contiguous()
        zext = torch.empty(S, Co, dtype=torch.float32, device=dev)
        arg = torch.empty(S, Co, dtype=torch.int32, device=dev)
        out = torch.empty(S, Co, dtype=torch.float32, device=dev)
        if training:
            coef2 = torch.empty(G, 5, Co, dtype=torch.float32, device=dev)
        else:
            coef2 = eval_coef(g2c, be2c, rm2, rv2, eps2).unsqueeze(0).repeat(G, 1, 1).contiguous()
        check(_timed('pointmlp_max_K%d_Co%d' % (K, Co), {'B': rows, 'N': seg, 'k': K, 'Co': Co},
                     lambda: L.sug_pointmlp_max_layer_fwd_xf(_p(y2), K, rows, K, _p(coef1), float(slope1), _p(z), K, _p(w2), _p(b1),
                                                             _p(g2c), _p(be2c), Co, seg, G, 1 if training else 0, eps2, mom2,
                                                             float(slope2), _p(rm2), _p(rv2), _p(zext), _p(arg), _p(coef2),
                                                             _p(out), Co, _p(ws), _st())),
              'sug_pointmlp_max_layer_fwd_xf')
        CTX.last_coef_pair = (coef1, coef2)
        ctx.need1, ctx.need2 = need1, need2
        if need1 or need2:
            ctx.save_for_backward(y2, coef1, z, w2, b1, zext, arg, coef2)
            ctx.meta = (rows, K, Co, seg, G, float(slope1), float(slope2), bool(training), tuple(y.shape), tuple(weight.shape))
        if not last_grad:
            ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        if z is None:
            return out, None
        return out, z.view(y.shape)

    @staticmethod
    def backward(ctx, gout, gz):
        if not (ctx.need1 or ctx.need2):
            # nothing was saved: neither the input nor any parameter of the two layers takes a gradient (a gradient that
            # still arrives at z -- want_z with the layers frozen -- has nowhere to go)
            return (None,) * 23
        y2, coef1, z, w2, b1, zext, arg, coef2 = ctx.saved_tensors
        rows, K, Co, seg, G, slope1, slope2, training, yshape, wshape = ctx.meta
        dz = dw = db = dg2 = db2 = None
        if ctx.need2 and gout is not None:
            dz, dw, db, dg2, db2 = _pointmlp_max_backward(gout, z, w2, b1, zext, arg, coef2, rows, K, Co, seg, G, slope2, training)
            dw = dw.view(wshape)
        if gz is not None:                                    # the SA-node features' gradient arrives at z too
            gz2 = gz.reshape(rows, K)
            dz = gz2 if dz is None else dz.add_(gz2)
        dy = dg1 = db1 = None
        if ctx.need1 and dz is not None:
            dy, dg1, db1 = _bn_act_rows_backward(dz, y2, coef1, rows, K, slope1, training, G)
            dy = dy.view(yshape)
        return (dy, dg1, db1, None, None, None, None, None, dw, db, dg2, db2) + (None,) * 11


def bn_act_pointmlp_max(y, bn1, slope1, weight, bias, bn2, slope2, seg, want_z=False, last_grad=True):
    """y [..., K] = pre-activation rows of the layer in front -> (max over `seg`-row segments of act2(bn2(z.W^T + b)), z or
    None) with z = act1(bn1(y)) never read back from memory by the last layer (see _BNActPointMLPMax)."""
    if bn1.training != bn2.training:
        # one `training` flag drives both BatchNorms of the fused pair (batch statistics + buffer update, or running
        # statistics): a frozen BatchNorm beside a training one has no fused form -- the callers
        # (conv_2d.rows_max_after, PointNetSetAbstraction) compare the modes and take the layer-by-layer path
        raise RuntimeError('bn_act_pointmlp_max: the two BatchNorm layers must be in the same mode (train / eval)')
    _count_bn_call(bn1)
    _count_bn_call(bn2)
    out, z = _BNActPointMLPMax.apply(y, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, slope1, bn1.eps, bn1.momentum,
                                     weight, bias, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, slope2, bn2.eps,
                                     bn2.momentum, bn1.training, seg, CTX.bn_groups, want_z, last_grad, torch.is_grad_enabled())
    if CTX.bn_record is not None:                      # a shared prefix replays these running-statistics updates (record_bn_stats)
        for bn, coef in zip((bn1, bn2), CTX.last_coef_pair):
            if bn.training and bn.track_running_stats:
                CTX.bn_record.append((bn, coef))
    return out, z


# Step-scoped cache of 16-bit copies of weights / biases (opt-in: SUGStep sets a dict before the forwards of a step and
# drops it after them).  Without it every call re-casts its operands: ~280 tiny launches per step at config 5.


def cast_cached(p, lo, detach=False):
    """16-bit copy of a parameter; shared by the calls of one step when the cache is on.  Attached copies
    (detach=False) stay in the autograd graph: the fp32 parameter receives the sum of their gradients."""
    if p is None:
        return None
    if CTX.w16_cache is None:
        return (p.detach() if detach else p).to(lo)
    key = (id(p), lo, bool(detach) or not torch.is_grad_enabled())
    hit = CTX.w16_cache.get(key)
    if hit is None or hit[0] != p._version:
        hit = (p._version, (p.detach() if key[2] else p).to(lo), p)
        CTX.w16_cache[key] = hit
    return hit[1]


def w16_plan(cache):
    """The detached copies a step made, as (key, parameter, buffer) records: the next step refreshes all the buffers
    with one multi-tensor copy (w16_prefill) instead of one cast launch per parameter."""
    return [(key, hit[2], hit[1]) for key, hit in cache.items() if key[2]]


def w16_prefill(plan):
    """A cache for this step whose planned entries are already current."""
    if plan:
        torch._foreach_copy_([buf for _, _, buf in plan], [p.detach() for _, p, _ in plan])
    return {key: (p._version, buf, p) for key, p, buf in plan}


def _dweight(gy, x):
    """gy^T . x over the R rows in row chunks (batched GEMM = split-K: one [M,K] product alone fills few
    workgroups), chunk results summed in fp32; 16-bit operands give fp32 products straight out of the GEMM."""
    f32 = torch.float32
    R = gy.shape[0]
    # ~32 chunks (x 4 output tiles of the library's 256 x 256 macro tile = half the chip's workgroup slots), chunks of
    # 512 .. 16384 rows: R = 2^20 -> 64 x 16384, 2^17 -> 32 x 4096, 2^15 -> 32 x 1024, 2^13 -> 16 x 512
    S = 1
    while R % (2 * S) == 0 and R // (2 * S) >= 512 and (S < 32 or R // S > 16384):
        S *= 2
    wide = {} if gy.dtype == f32 else {'out_dtype': f32}
    if S == 1:
        return torch.mm(gy.t(), x, **wide)
    return torch.bmm(gy.view(S, R // S, gy.shape[1]).transpose(1, 2), x.view(S, R // S, x.shape[1]), **wide).sum(dim=0)


class _Linear16(torch.autograd.Function):
    """nn.Linear on [..., K] rows with 16-bit operands (MFMA, fp32 accumulation) and fp32 parameters: the weight's
    16-bit copy comes from the step cache, the result leaves the GEMM as fp32 (`out32`) or stays 16-bit for a
    following 16-bit GEMM, the parameter gradients leave their GEMMs as fp32 (split-K over row chunks).  Written
    out so that no cast kernels surround the GEMMs (an F.linear on cast operands costs ~6 per call)."""

    @staticmethod
    def forward(ctx, x, w, b, lo, out32):
        _need_gpu(x, w)
        x2 = x.reshape(-1, x.shape[-1])
        x16 = x2 if x2.dtype == lo else x2.to(lo)
        w16 = cast_cached(w, lo, detach=True)
        if out32:
            y = torch.mm(x16, w16.t(), out_dtype=torch.float32)
            if b is not None:
                y += b
        elif b is None:
            y = torch.mm(x16, w16.t())
        else:
            y = torch.addmm(cast_cached(b, lo, detach=True), x16, w16.t())
        ctx.save_for_backward(x16, w16)
        ctx.meta = (x.shape, x.dtype, b is not None, lo)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, g):
        x16, w16 = ctx.saved_tensors
        xshape, xdtype, has_b, lo = ctx.meta
        g2 = g.reshape(-1, g.shape[-1])
        g16 = g2 if g2.dtype == lo else g2.to(lo)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (torch.mm(g16, w16, out_dtype=torch.float32) if xdtype == torch.float32 else torch.mm(g16, w16)).view(xshape)
        dw = _dweight(g16, x16) if ctx.needs_input_grad[1] else None
        db = colsum(g2) if (has_b and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None


def linear16(x, layer, lo, out32=True):
    return _Linear16.apply(x, layer.weight, layer.bias, lo, out32)


# ----------------------------------------------------------------------------- Point Transformer attention
# fp16 mode: the forward of a transformer block's vector attention as ONE MFMA kernel (sug_ptran_fused_fwd) instead of pos1 +
# 3 library GEMMs + qk + attn.  Same values bit for bit (tests/test_gpu_ptran.py).  Measured at config 5's block-1 shape (1 M
# k-expanded rows; tools/bench_ptran_fused.py, profiles/r06_ptran_fused_ab.txt): without a backward to feed 2.56 vs 3.21 ms --
# nothing but delta is written --, with the four tensors a backward reads 3.11 vs 3.13 ms, and 19.73 vs 19.63 ms for the
# whole config-5 step: one 130 KB workgroup per CU serialises its row passes (L2 gathers of K / V rows) with its MFMA chains,
# the composition overlaps them across kernels at full occupancy.  Hence 'auto': the one-kernel form where no backward
# follows (eval / no_grad forwards), the composition in training.  SUG_PTRAN_FUSED=1 / 0 forces one form.
PTRAN_FUSED = {'1': True, '0': False}.get(_os.environ.get('SUG_PTRAN_FUSED', 'auto'), 'auto')


class _PTranAttention(torch.autograd.Function):
    """Vector attention of one TransformerBlock (model/Ptran_transformer.py:39-44) from the projected
    q / K / V rows: pos-encoding MLP on the neighbour offsets, attention MLP on q - k + delta, softmax
    over the k neighbours per channel, weighted sum of v + delta.  The three 512 x 512 linears on the
    k-expanded rows are library GEMMs in `dtype` (fp32 = the reference's arithmetic; fp16 = MFMA with
    fp32 accumulation, BASELINE config 5); everything between them is sug_ptran_* (one pass each).
    The backward is written out by hand: nothing but the GEMM operands T0, delta, U, T1, L is saved, the
    two gradients of delta are summed inside sug_ptran_qk_bwd, dK / dV are gathered over reverse
    neighbour lists (deterministic, no atomics)."""

    @staticmethod
    def forward(ctx, xyz, nbr, q, kf, vf, w1, b1, w2, b2, wg1, bg1, wg2, bg2, dtype):
        _need_gpu(xyz, nbr, q, kf, vf)
        B, n, k = nbr.shape
        d = q.shape[-1]
        dev = q.device
        lo = torch.float32 if dtype is None else dtype
        code = 0 if lo == torch.float32 else 1
        if lo not in (torch.float32, torch.float16):
            raise RuntimeError('ptran_attention: fp32 or fp16 (got %s)' % lo)
        R = B * n * k
        xyz, nbr = xyz.detach().contiguous(), _i32(nbr).contiguous()
        q, kf, vf = q.contiguous(), kf.contiguous(), vf.contiguous()
        w1c, b1c = w1.detach().contiguous(), b1.detach().contiguous()
        wl = [cast_cached(t, lo, detach=True) for t in (w2, b2, wg1, bg1, wg2, bg2)]
        L_ = lib()
        scale = 1.0 / (d ** 0.5)
        need = any(ctx.needs_input_grad[i] for i in range(2, 13))
        if (PTRAN_FUSED is True or (PTRAN_FUSED == 'auto' and not need)) and code == 1 and L_.sug_ptran_fused_supported(B, n, k, d):
            # the whole fp16 forward as ONE kernel on the matrix cores (csrc/ptran_fused.hip): no library GEMM, no k-expanded
            # tensor read back; T0 / U / T1 / logits are written (once) only when a backward will read them
            delta = torch.empty(R, d, dtype=lo, device=dev)
            T0, U, T1, Lg = ((torch.empty(R, d, dtype=lo, device=dev) for _ in range(4)) if need else (None,) * 4)
            mixed = torch.empty(B, n, d, dtype=torch.float32, device=dev)
            mx, sm = torch.empty_like(mixed), torch.empty_like(mixed)
            shp = {'B': B, 'N': n, 'k': k, 'd': d, 'e': 2}
            check(_timed('ptran_fused_fwd_n%d' % n, shp, lambda: L_.sug_ptran_fused_fwd(
                _p(xyz), _p(nbr), _p(q), _p(kf), _p(vf), _p(w1c), _p(b1c), _p(wl[0]), _p(wl[1]), _p(wl[2]), _p(wl[3]), _p(wl[4]),
                _p(wl[5]), B, n, k, d, scale, 1 if need else 0, _p(T0), _p(delta), _p(U), _p(T1), _p(Lg), _p(mixed), _p(mx), _p(sm),
                _st())), 'sug_ptran_fused_fwd')
            if need:
                ctx.save_for_backward(xyz, nbr, vf, w1c, b1c, wl[0], wl[2], wl[4], T0, delta, U, T1, Lg, mx, sm, mixed)
            ctx.meta = (B, n, k, d, code, scale)
            return mixed
        T0 = torch.empty(R, d, dtype=lo, device=dev)
        check(L_.sug_ptran_pos1_fwd(_p(xyz), _p(nbr), _p(w1c), _p(b1c), B, n, k, d, code, _p(T0), _st()), 'sug_ptran_pos1_fwd')
        delta = torch.addmm(wl[1], T0, wl[0].t())
        U = torch.empty(R, d, dtype=lo, device=dev)
        shp = {'B': B, 'N': n, 'k': k, 'd': d, 'e': 4 if code == 0 else 2}
        check(_timed('ptran_qk_fwd_n%d' % n, shp, lambda: L_.sug_ptran_qk_fwd(_p(q), _p(kf), _p(delta), _p(nbr), B, n, k, d, code, _p(U), _st())),
              'sug_ptran_qk_fwd')
        T1 = torch._addmm_activation(wl[3], U, wl[2].t())          # bias + ReLU in the library GEMM's epilogue
        Lg = torch.addmm(wl[5], T1, wl[4].t())
        mixed = torch.empty(B, n, d, dtype=torch.float32, device=dev)
        mx, sm = torch.empty_like(mixed), torch.empty_like(mixed)
        check(_timed('ptran_attn_fwd_n%d' % n, shp, lambda: L_.sug_ptran_attn_fwd(_p(Lg), _p(delta), _p(vf), _p(nbr), B, n, k, d, code, scale,
                                                                          _p(mixed), _p(mx), _p(sm), _st())), 'sug_ptran_attn_fwd')
        ctx.save_for_backward(xyz, nbr, vf, w1c, b1c, wl[0], wl[2], wl[4], T0, delta, U, T1, Lg, mx, sm, mixed)
        ctx.meta = (B, n, k, d, code, scale)
        return mixed

    @staticmethod
    def backward(ctx, g):
        xyz, nbr, vf, w1c, b1c, w2l, wg1l, wg2l, T0, delta, U, T1, Lg, mx, sm, mixed = ctx.saved_tensors
        B, n, k, d, code, scale = ctx.meta
        dev, lo = g.device, T0.dtype
        L_ = lib()
        g = g.contiguous().float()
        off, ent = knn_reverse(nbr)
        dL, da = torch.empty_like(Lg), torch.empty_like(Lg)
        dv = torch.empty(B, n, d, dtype=torch.float32, device=dev)
        f32 = torch.float32
        # bias gradients = column sums of the k-expanded gradients, produced by the kernels that write them
        R = B * n * k
        cws = torch.empty(L_.sug_ptran_colsum_workspace(R), dtype=f32, device=dev)
        dbg2, dbg1, db2 = (torch.empty(d, dtype=f32, device=dev) for _ in range(3))
        shp = {'B': B, 'N': n, 'k': k, 'd': d, 'e': 4 if code == 0 else 2}
        check(_timed('ptran_attn_bwd_n%d' % n, shp, lambda: L_.sug_ptran_attn_bwd(_p(g), _p(mixed), _p(Lg), _p(delta), _p(vf), _p(nbr), _p(mx),
                                                                          _p(sm), _p(off), _p(ent), B, n, k, d, code, scale, _p(dL),
                                                                          _p(da), _p(dv), _p(dbg2), _p(cws), _st())),
              'sug_ptran_attn_bwd')

        dwg2 = _dweight(dL, T1)
        dT1 = dL @ wg2l
        check(L_.sug_ptran_relu_bwd_db(_p(dT1), _p(T1), R, d, code, _p(dbg1), _p(cws), _st()), 'sug_ptran_relu_bwd_db')
        dwg1 = _dweight(dT1, U)
        dU = dT1 @ wg1l
        dq, dk = torch.empty_like(dv), torch.empty_like(dv)
        check(_timed('ptran_qk_bwd_n%d' % n, shp, lambda: L_.sug_ptran_qk_bwd(_p(dU), _p(da), _p(off), _p(ent), B, n, k, d, code, _p(dq), _p(dk),
                                                                      _p(db2), _p(cws), _st())), 'sug_ptran_qk_bwd')
        ddelta = da                                            # = dU + da
        dw2 = _dweight(ddelta, T0)
        dT0 = ddelta @ w2l
        dw1 = torch.empty(d, 3, dtype=f32, device=dev)
        db1 = torch.empty(d, dtype=f32, device=dev)
        ws = torch.empty(1024 * 4 * d, dtype=f32, device=dev)
        check(L_.sug_ptran_pos1_bwd(_p(dT0), _p(xyz), _p(nbr), _p(w1c), _p(b1c), B, n, k, d, code, _p(dw1), _p(db1), _p(ws),
                                    _st()), 'sug_ptran_pos1_bwd')
        return None, None, dq, dk, dv, dw1, db1, dw2, db2, dwg1, dbg1, dwg2, dbg2, None


def ptran_attention(xyz, nbr, q, kf, vf, fc_delta, fc_gamma, dtype=None):
    """xyz [B,n,3], nbr [B,n,k], q / kf / vf [B,n,512] (w_qs / w_ks / w_vs of the lifted features),
    fc_delta = Sequential(Linear(3,512), ReLU, Linear(512,512)), fc_gamma likewise (512,512) -> [B,n,512]."""
    return _PTranAttention.apply(xyz, nbr, q, kf, vf, fc_delta[0].weight, fc_delta[0].bias, fc_delta[2].weight,
                                 fc_delta[2].bias, fc_gamma[0].weight, fc_gamma[0].bias, fc_gamma[2].weight,
                                 fc_gamma[2].bias, dtype)


# ----------------------------------------------------------------------------- channel attention (CALayer)
CALAYER_FUSED = _os.environ.get('SUG_CALAYER_FUSED', '1') != '0'    # A/B knob: 0 = library GEMMs + relu + gate_bn per layer


def calayer_supported(layers, x):
    """Can the CALayer modules `layers` (1 or 2: attention_s [, attention_t]) run through sug_calayer_* on x [len(layers)*M, C]?"""
    if not (CALAYER_FUSED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and CTX.bn_groups == 1):
        return False
    if x.shape[0] % len(layers):
        return False
    M, C = x.shape[0] // len(layers), x.shape[1]
    for m in layers:
        c0, c2, bn = m.conv_du[0], m.conv_du[2], m.bn
        if c0.weight.shape[1] != C or c2.weight.shape[0] != C or c0.bias is None or c2.bias is None or not bn.affine \
                or not (bn.training or bn.track_running_stats) or bn.momentum is None or not bn.track_running_stats:
            return False
        if any(mm.training != layers[0].training for mm in (m, bn)):
            return False
    return bool(lib().sug_calayer_supported(len(layers), M, C, layers[0].conv_du[0].weight.shape[0]))


class _CALayers(torch.autograd.Function):
    """BatchNorm1d(v * sigmoid(z) + v), z = W2 . relu(W0 . v + b0) + b2 (CALayer, model/Model.py:28-34) for one or two
    attention layers on the row blocks of x [layers*M, C]: sug_calayer_fwd / _bwd + sug_gate_bn_fwd / _bwd.
    params per layer: W0 [Hd,C], b0, W2 [C,Hd], b2, gamma, beta, running_mean, running_var."""

    @staticmethod
    def forward(ctx, x, nl, training, eps, momentum, *params):
        _need_gpu(x)
        x = x if (x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0) else x.contiguous()
        M, C = x.shape[0] // nl, x.shape[1]
        P = [params[8 * a:8 * a + 8] for a in range(nl)]
        W0 = [p[0].detach().reshape(p[0].shape[0], -1).contiguous() for p in P]
        W2 = [p[2].detach().reshape(p[2].shape[0], -1).contiguous() for p in P]
        b0, b2 = [p[1].detach().contiguous() for p in P], [p[3].detach().contiguous() for p in P]
        gam, bet = [p[4].detach().contiguous() for p in P], [p[5].detach().contiguous() for p in P]
        Hd = W0[0].shape[0]
        dev = x.device
        L = lib()
        hp = torch.empty(nl, C // 512, M, Hd, dtype=torch.float32, device=dev)
        h = torch.empty(nl, M, Hd, dtype=torch.float32, device=dev)
        z = torch.empty(nl * M, C, dtype=torch.float32, device=dev)
        check(L.sug_calayer_fwd(nl, _p(x), x.stride(0), M, C, Hd, _ptrs(W0), _ptrs(b0), _ptrs(W2), _ptrs(b2), _p(hp), _p(h), _p(z),
                                _st()), 'sug_calayer_fwd')
        out = torch.empty(nl * M, C, dtype=torch.float32, device=dev)
        stat = torch.empty(nl, 2, C, dtype=torch.float32, device=dev)
        xc = x if x.stride(0) == C else x.contiguous()            # (sug_gate_bn_* take dense rows)
        for a in range(nl):
            check(L.sug_gate_bn_fwd(_p(xc[a * M:]), _p(z[a * M:]), M, C, _p(gam[a]), _p(bet[a]), _p(P[a][6]), _p(P[a][7]),
                                    1 if training else 0, float(eps), float(momentum), _p(out[a * M:]), _p(stat[a]), _st()),
                  'sug_gate_bn_fwd')
        ctx.save_for_backward(xc, h, z, stat, *W0, *W2, *gam)
        ctx.meta = (nl, M, C, Hd, bool(training))
        return out

    @staticmethod
    def backward(ctx, g):
        nl, M, C, Hd, training = ctx.meta
        sv = ctx.saved_tensors
        x, h, z, stat = sv[:4]
        W0, W2, gam = sv[4:4 + nl], sv[4 + nl:4 + 2 * nl], sv[4 + 2 * nl:4 + 3 * nl]
        dev = x.device
        L = lib()
        g = g.contiguous()
        dxg, dz = torch.empty_like(x), torch.empty_like(x)
        dgb = torch.empty(nl, 2, C, dtype=torch.float32, device=dev)
        for a in range(nl):
            check(L.sug_gate_bn_bwd(_p(g[a * M:]), _p(x[a * M:]), _p(z[a * M:]), M, C, _p(gam[a]), _p(stat[a]), 1 if training else 0,
                                    _p(dxg[a * M:]), _p(dz[a * M:]), _p(dgb[a, 0]), _p(dgb[a, 1]), _st()), 'sug_gate_bn_bwd')
        f32 = torch.float32
        dW0 = [torch.empty(Hd, C, dtype=f32, device=dev) for _ in range(nl)]
        dW2 = [torch.empty(C, Hd, dtype=f32, device=dev) for _ in range(nl)]
        db0 = [torch.empty(Hd, dtype=f32, device=dev) for _ in range(nl)]
        db2 = [torch.empty(C, dtype=f32, device=dev) for _ in range(nl)]
        dhp = torch.empty(nl, C // 32, M, Hd, dtype=f32, device=dev)
        dh = torch.empty(nl, M, Hd, dtype=f32, device=dev)
        dx = torch.empty(nl * M, C, dtype=f32, device=dev)
        check(L.sug_calayer_bwd(nl, _p(x), x.stride(0), M, C, Hd, _ptrs(W0), _ptrs(W2), _p(h), _p(dz), _p(dxg), _ptrs(dW0), _ptrs(db0),
                                _ptrs(dW2), _ptrs(db2), _p(dhp), _p(dh), _p(dx), C, _st()), 'sug_calayer_bwd')
        grads = []
        for a in range(nl):
            grads += [dW0[a].view(Hd, C, 1, 1), db0[a], dW2[a].view(C, Hd, 1, 1), db2[a], dgb[a, 0], dgb[a, 1], None, None]
        return (dx, None, None, None, None) + tuple(grads)


def calayers(layers, x):
    """[attention(x rows of its block) for attention in layers] as one [len(layers)*M, C] tensor (rows of layer 0 first)."""
    params = []
    for m in layers:
        _count_bn_call(m.bn)
        c0, c2 = m.conv_du[0], m.conv_du[2]
        params += [c0.weight, c0.bias, c2.weight, c2.bias, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var]
    bn = layers[0].bn
    return _CALayers.apply(x, len(layers), bn.training, bn.eps, bn.momentum, *params)


# ----------------------------------------------------------------------------- clouds as rows
_rows_cache = [None, None, None]           # weakref to the [B,3,N,1] input, its version, the [B,N,3] rows


def clear_rows_cache():
    """Drop the cached rows of the last batch (SUGStep does after every step: the rows of a captured step live in the
    graph's memory pool and must not be pinned by a module global)."""
    _rows_cache[0] = _rows_cache[1] = _rows_cache[2] = None


def cloud_rows(x):
    """x [B,3(+),N,1] (the reference's cloud layout) -> [B,N,3(+)] contiguous rows.  The semantic and the node pass of a step
    (and the Chamfer weights) transpose the same batch: the last result is kept, keyed on the tensor OBJECT (weak
    reference) and its version, so the copy is made once per batch.
    CONTRACT of the key: a write into `x` that does not bump `x._version` -- `x.data` arithmetic, a numpy view of the same
    memory, a raw-pointer kernel -- is NOT seen; whoever refills a batch buffer that way calls clear_rows_cache() (or uses
    copy_ / an in-place torch op, which bump the version).  SUGStep clears the cache at the end of every step."""
    import weakref
    ref, ver, rows = _rows_cache
    if ref is not None and ref() is x and ver == x._version and not x.requires_grad:
        return rows
    rows = x.squeeze(-1).transpose(1, 2).contiguous()
    if not x.requires_grad and not torch.is_inference_mode_enabled():
        _rows_cache[0], _rows_cache[1], _rows_cache[2] = weakref.ref(x), x._version, rows
    return rows


# ----------------------------------------------------------------------------- scalar tail of a step
class _CEPair(torch.autograd.Function):
    """w * (CE(logits1[:M], label) + CE(logits2[:M], label)) of both classifier heads on the SOURCE rows of the paired
    logits in one launch (sug_ce_pair_fwd); the backward writes the whole pair's gradient (zeros in the target rows)."""

    @staticmethod
    def forward(ctx, y1, y2, label, w, ignore_index=-100):
        _need_gpu(y1, y2, label)
        M, C = label.shape[0], y1.shape[1]
        a = y1 if y1.stride(1) == 1 else y1.contiguous()
        b = y2 if (y2.stride(1) == 1 and y2.stride(0) == a.stride(0)) else y2.contiguous()
        if b.stride(0) != a.stride(0):
            a, b = a.contiguous(), b.contiguous()
        lab = label.reshape(-1).long().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=y1.device)
        lse = torch.empty(2 * M + 1, dtype=torch.float32, device=y1.device)     # + the number of counting rows
        check(lib().sug_ce_pair_fwd(_p(a), _p(b), a.stride(0), _p(lab), M, C, float(w), int(ignore_index), _p(loss), _p(lse), _st()),
              'sug_ce_pair_fwd')
        ctx.save_for_backward(a, b, lab, lse)
        ctx.meta = (M, y1.shape[0], C, float(w), int(ignore_index))
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b, lab, lse = ctx.saved_tensors
        M, Mtot, C, w, ign = ctx.meta
        gs = g.detach().to(dtype=torch.float32).reshape(1)
        d = torch.empty(2, Mtot, C, dtype=torch.float32, device=a.device)
        check(lib().sug_ce_pair_bwd(_p(a), _p(b), a.stride(0), _p(lab), M, Mtot, C, w, ign, _p(gs), _p(lse), _p(d[0]), _p(d[1]),
                                    _st()), 'sug_ce_pair_bwd')
        return d[0], d[1], None, None, None


def ce_pair_supported(y1, y2, label):
    return y1.is_cuda and y1.dtype == torch.float32 and y1.dim() == 2 and y1.shape == y2.shape and y1.shape[1] <= 32 \
        and 2 * label.shape[0] <= 512 and label.shape[0] <= y1.shape[0]


def ce_pair(y1, y2, label, w, ignore_index=-100):
    """y1, y2 [Mtot >= M, C] logits of the two heads (the first M rows are scored), label [M] -> 0-d loss.
    nn.CrossEntropyLoss label semantics: rows labelled `ignore_index` are skipped (and left out of the mean); any other
    label outside [0, C) -- torch raises there -- turns the loss into NaN."""
    return _CEPair.apply(y1, y2, label, w, ignore_index)


class _LossCombine(torch.autograd.Function):
    """(loss_cls + wg*v_geo + ws*(v_sem1 + v_sem2), wg*v_geo, ws*(v_sem1 + v_sem2)) in one launch each way; the two parts
    are returned for reporting only (non-differentiable)."""

    @staticmethod
    def forward(ctx, loss_cls, v0, v1, v2, wg, ws):
        out = torch.empty(3, dtype=torch.float32, device=loss_cls.device)
        f = lambda t: None if t is None else t.detach().float()
        lc, a, b, c = f(loss_cls), f(v0), f(v1), f(v2)
        check(lib().sug_loss_combine_fwd(_p(lc), _p(a), _p(b), _p(c), float(wg), float(ws), _p(out), _st()), 'sug_loss_combine_fwd')
        ctx.meta = (float(wg), float(ws), v0 is not None, v1 is not None, v2 is not None)
        tot, geo, sem = out[0], out[1], out[2]
        ctx.mark_non_differentiable(geo, sem)
        ctx.set_materialize_grads(False)          # no zero-filled scalars for the two reporting outputs (a launch each)
        return tot, geo, sem

    @staticmethod
    def backward(ctx, g, _g1, _g2):
        wg, ws, h0, h1, h2 = ctx.meta
        if g is None:
            return None, None, None, None, None, None
        o = torch.empty(4, dtype=torch.float32, device=g.device)
        gs = g.detach().to(dtype=torch.float32).reshape(1)
        check(lib().sug_loss_combine_bwd(_p(gs), wg, ws, _p(o), _st()), 'sug_loss_combine_bwd')
        return o[0], (o[1] if h0 else None), (o[2] if h1 else None), (o[3] if h2 else None), None, None


def loss_combine(loss_cls, v_geo, v_sem1, v_sem2, wg, ws):
    return _LossCombine.apply(loss_cls, v_geo, v_sem1, v_sem2, wg, ws)


class _SplitHalves(torch.autograd.Function):
    """The two domain halves of a paired [2B, ...] tensor as views.  Backward: when the two gradients are adjacent row
    blocks of one buffer (what mmd_assemble's and the paired kernels' backwards hand back) the pair's gradient is that
    buffer, without the torch.stack / zero fill of unbind's backward."""

    @staticmethod
    def forward(ctx, t):
        B = t.shape[0] // 2
        ctx.meta = (tuple(t.shape), t.dtype, t.device)
        return t[:B], t[B:]

    @staticmethod
    def backward(ctx, ga, gb):
        shape, dtype, dev = ctx.meta
        B = shape[0] // 2
        if ga is not None and gb is not None and ga.shape == gb.shape and ga.stride() == gb.stride() and ga.dim() >= 1 \
                and ga.untyped_storage().data_ptr() == gb.untyped_storage().data_ptr() \
                and gb.storage_offset() == ga.storage_offset() + B * ga.stride(0):
            return torch.as_strided(ga, shape, ga.stride(), ga.storage_offset())
        if ga is None and gb is None:
            return None
        z = lambda: torch.zeros((B,) + shape[1:], dtype=dtype, device=dev)
        return torch.cat((ga if ga is not None else z(), gb if gb is not None else z()), 0)


def split_halves(t):
    """t [2B, ...] -> (t[:B], t[B:]) with a copy-free backward where the producers allow it."""
    return _SplitHalves.apply(t)


# ----------------------------------------------------------------------------- concat without copies
class _AssembleRows(torch.autograd.Function):
    """torch.cat(parts, dim=-1) for parts that (mostly) already live in column slices of `buf`
    [..., sum(widths)]: producers were asked to write there (edgeconv_bn_act_max(out=...)); a part
    stored elsewhere is copied in.  Backward hands every part its column slice of the gradient (views,
    no copies).  `buf` is passed in a list: it is storage, not a differentiable input."""

    @staticmethod
    def forward(ctx, holder, *parts):
        buf = holder[0]
        off, widths = 0, []
        for t in parts:
            w = t.shape[-1]
            dst = buf[..., off:off + w]
            if not (t.data_ptr() == dst.data_ptr() and t.stride() == dst.stride() and t.shape == dst.shape):
                # copy through an alias with its own version counter: an in-place op on a view of `buf`
                # would invalidate (for autograd) the slices the producers already returned
                t2 = t.detach()
                rows = t2.numel() // max(w, 1)
                if t2.is_cuda and t2.dtype == torch.float32 and t2.is_contiguous() and w % 4 == 0 and dst.stride(-1) == 1 \
                        and all(dst.stride(d) == dst.stride(d + 1) * dst.shape[d + 1] for d in range(dst.dim() - 2)) \
                        and dst.stride(-2) % 4 == 0 and t2.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0:
                    # own float4 copy: torch's strided copy kernel takes 30 us for a 16.8 MB part
                    check(lib().sug_copy_rows2d(_p(t2), w, _p(dst), dst.stride(-2), rows, w, _st()), 'sug_copy_rows2d')
                else:
                    alias = torch.empty(0, dtype=buf.dtype, device=buf.device).set_(
                        buf.untyped_storage(), dst.storage_offset(), dst.shape, dst.stride())
                    alias.copy_(t)
            widths.append(w)
            off += w
        if off != buf.shape[-1]:
            raise RuntimeError('assemble_rows: parts cover %d of %d columns' % (off, buf.shape[-1]))
        ctx.widths = widths
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        out, off = [], 0
        for w in ctx.widths:
            out.append(g[..., off:off + w])
            off += w
        return (None,) + tuple(out)


def assemble_rows(buf, parts):
    return _AssembleRows.apply([buf], *parts)


# ----------------------------------------------------------------------------- MMD
_neg_gamma_cache = {}


def _neg_gammas(sigmas, device):
    key = (tuple(float(s) for s in sigmas), str(device))
    if key not in _neg_gamma_cache:
        vals = [-(1.0 / (2 * s ** 2)) for s in sigmas]             # python doubles, model/mmd.py:251
        _neg_gamma_cache[key] = torch.tensor(vals, dtype=torch.float32, device=device)
    return _neg_gamma_cache[key]


class _MixRbfMMD2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Z, m, w, sigmas, biased=True):
        _need_gpu(Z)
        Zc = Z if (Z.stride(1) == 1 and Z.stride(0) >= Z.shape[1]) else Z.contiguous()
        D = Zc.shape[1]
        dev = Z.device
        ng = _neg_gammas(sigmas, dev)
        need = ctx.needs_input_grad[0]
        wt = torch.empty(2 * m, 2 * m, dtype=torch.float32, device=dev) if need else None
        wc = w.detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous() if w is not None else None
        if biased:
            sums = torch.empty(3, dtype=torch.float64, device=dev)
            val = torch.empty((), dtype=torch.float32, device=dev)
            check(lib().sug_mmd_rbf_value(_p(Zc), Zc.stride(0), m, D, _p(wc), _p(ng), len(sigmas), _p(sums), _p(wt), _p(val),
                                          _st()), 'sug_mmd_rbf_value')
        else:
            # _mmd2(biased=False), model/mmd.py:304-308 (no caller in the reference; off the hot path: a few torch ops on
            # the kernel's sums): the diagonals of K_XX / K_YY -- exactly len(sigmas) each, e_ii is identically 0 there and
            # here -- leave the sums, the within-domain pairs are averaged over m (m - 1).  The kernel's gradient weights
            # carry 2 / m^2 for those pairs (their diagonal is already zero): rescaled by m / (m - 1).
            if m < 2:
                raise RuntimeError('mix_rbf_mmd2(biased=False) needs at least two samples per domain')
            sums = torch.zeros(3, dtype=torch.float64, device=dev)
            check(lib().sug_mmd_rbf(_p(Zc), Zc.stride(0), m, D, _p(wc), _p(ng), len(sigmas), _p(sums), _p(wt), _st()), 'sug_mmd_rbf')
            diag = float(len(sigmas) * m)
            val = (((sums[0] - diag) + (sums[1] - diag)) / float(m * (m - 1)) - 2.0 * sums[2] / float(m * m)).to(torch.float32)
            if need:
                r = float(m) / float(m - 1)
                wt[:m, :m] *= r
                wt[m:, m:] *= r
        if need:
            ctx.save_for_backward(Zc, wt)
        return val

    @staticmethod
    def backward(ctx, g):
        Z, wt = ctx.saved_tensors
        M2, D = Z.shape
        gs = g.detach().to(device=Z.device, dtype=torch.float32).reshape(1)
        dZ = torch.empty(M2, D, dtype=torch.float32, device=Z.device)
        check(lib().sug_mmd_rbf_bwd(_p(Z), Z.stride(0), _p(wt), M2 // 2, D, _p(gs), _p(dZ), D, _st()),
              'sug_mmd_rbf_bwd')
        return dZ, None, None, None, None


def mix_rbf_mmd2_rows(Z, m, sample_weights=None, sigmas=SIGMA_LIST, biased=True):
    """Z = cat(X, Y) [2m, D] -> MMD^2 (0-d tensor): the biased estimator (model/mmd.py:300-303) or the unbiased one (:304-308)."""
    return _MixRbfMMD2.apply(Z, m, sample_weights, tuple(sigmas), bool(biased))


class _MixRbfMMD2Sharded(torch.autograd.Function):
    """The global-batch MMD^2 of a batch-sharded step: every rank holds the gathered
    Zall = [X of all ranks ; Y of all ranks] (values only) and evaluates the row block of the kernel
    matrix that belongs to its own samples, rows [row0, row0+mloc) of each domain, against all 2M
    columns (sug_mmd_rbf_rows); the three partial sums cross the ranks as one 3-double all-reduce.
    Backward: the gradient of the own rows comes from the own row block alone
    (sug_mmd_rbf_rows_bwd) -- no collective -- scaled by the world size because the data-parallel
    gradient averaging divides by it again (every rank's loss contains the same global term)."""

    @staticmethod
    def forward(ctx, Zloc, Zall, mloc, M, row0, w, sigmas, world, group):
        import torch.distributed as dist
        _need_gpu(Zloc, Zall)
        Za = Zall if (Zall.stride(1) == 1 and Zall.stride(0) >= Zall.shape[1]) else Zall.contiguous()
        D = Za.shape[1]
        dev = Za.device
        ng = _neg_gammas(sigmas, dev)
        sums = torch.zeros(3, dtype=torch.float64, device=dev)
        need = ctx.needs_input_grad[0]
        wt = torch.empty(2 * mloc, 2 * M, dtype=torch.float32, device=dev) if need else None
        wc = w.detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous() if w is not None else None
        check(lib().sug_mmd_rbf_rows(_p(Za), Za.stride(0), M, D, _p(wc), _p(ng), len(sigmas), row0, mloc, _p(sums),
                                     _p(wt), _st()), 'sug_mmd_rbf_rows')
        if world > 1:
            dist.all_reduce(sums, group=group)
        if need:
            ctx.save_for_backward(Za, wt)
            ctx.meta = (mloc, M, row0, world)
        mm = float(M) * float(M)
        return ((sums[0] + sums[1] - 2.0 * sums[2]) / mm).float()

    @staticmethod
    def backward(ctx, g):
        Za, wt = ctx.saved_tensors
        mloc, M, row0, world = ctx.meta
        D = Za.shape[1]
        gs = g.detach().to(device=Za.device, dtype=torch.float32).reshape(1)
        dZ = torch.empty(2 * mloc, D, dtype=torch.float32, device=Za.device)
        check(lib().sug_mmd_rbf_rows_bwd(_p(Za), Za.stride(0), _p(wt), M, D, row0, mloc, _p(gs), float(world), _p(dZ), D,
                                         _st()), 'sug_mmd_rbf_rows_bwd')
        return dZ, None, None, None, None, None, None, None, None


def mix_rbf_mmd2_rows_sharded(Zloc, Zall, mloc, M, row0, sample_weights=None, sigmas=SIGMA_LIST, world=1, group=None):
    """Zloc = cat(X_own, Y_own) [2*mloc, D] (differentiable), Zall = cat(X_all, Y_all) [2M, D] (values)."""
    return _MixRbfMMD2Sharded.apply(Zloc, Zall, mloc, M, row0, sample_weights, tuple(sigmas), world, group)


def mmd_rows_local_sums(Zall, M, row0, mloc, sample_weights, sums_out, need_wt=True, sigmas=SIGMA_LIST):
    """First half of mix_rbf_mmd2_rows_sharded for a step whose collectives sit BETWEEN captured graph segments: this
    rank's row block of the kernel matrix of the gathered batch -> its three partial sums accumulated into `sums_out`
    (fp64 [3], zeroed by the caller) and the derivative block wt [2*mloc, 2M]; no collective here."""
    _need_gpu(Zall)
    Za = Zall if (Zall.stride(1) == 1 and Zall.stride(0) >= Zall.shape[1]) else Zall.contiguous()
    D = Za.shape[1]
    ng = _neg_gammas(sigmas, Za.device)
    wt = torch.empty(2 * mloc, 2 * M, dtype=torch.float32, device=Za.device) if need_wt else None
    wc = sample_weights.detach().reshape(-1).to(device=Za.device, dtype=torch.float32).contiguous() \
        if sample_weights is not None else None
    check(lib().sug_mmd_rbf_rows(_p(Za), Za.stride(0), M, D, _p(wc), _p(ng), len(sigmas), row0, mloc, _p(sums_out), _p(wt),
                                 _st()), 'sug_mmd_rbf_rows')
    return Za, wt


class _MMDFromReducedSums(torch.autograd.Function):
    """Second half: the MMD^2 value from the all-reduced sums; the gradient of this rank's rows comes from its own row
    block (sug_mmd_rbf_rows_bwd), scaled by the world size (the gradient averaging divides by it again)."""

    @staticmethod
    def forward(ctx, Zloc, sums, Za, wt, mloc, M, row0, world):
        ctx.save_for_backward(Za, wt)
        ctx.meta = (mloc, M, row0, world)
        mm = float(M) * float(M)
        return ((sums[0] + sums[1] - 2.0 * sums[2]) / mm).float()

    @staticmethod
    def backward(ctx, g):
        Za, wt = ctx.saved_tensors
        mloc, M, row0, world = ctx.meta
        D = Za.shape[1]
        gs = g.detach().to(device=Za.device, dtype=torch.float32).reshape(1)
        dZ = torch.empty(2 * mloc, D, dtype=torch.float32, device=Za.device)
        check(lib().sug_mmd_rbf_rows_bwd(_p(Za), Za.stride(0), _p(wt), M, D, row0, mloc, _p(gs), float(world), _p(dZ), D,
                                         _st()), 'sug_mmd_rbf_rows_bwd')
        return dZ, None, None, None, None, None, None, None


def mmd_from_reduced_sums(Zloc, sums, Za, wt, mloc, M, row0, world):
    return _MMDFromReducedSums.apply(Zloc, sums, Za, wt, mloc, M, row0, world)


class _AssembleZ(torch.autograd.Function):
    """[feat_s ; feat_t | one-hot(label) * scale] of soft_mmd (model/mmd.py:56-66) in one launch instead of
    scatter + three cats + a mul; the gradient of the feature block is handed back as two row-slice views."""

    @staticmethod
    def forward(ctx, feat_s, feat_t, label_s, label_t, scale, num_class):
        _need_gpu(feat_s, feat_t, label_s, label_t)
        m, D = feat_s.shape
        fs = feat_s if feat_s.stride(1) == 1 else feat_s.contiguous()
        ft = feat_t if feat_t.stride(1) == 1 else feat_t.contiguous()
        ls, lt = label_s.reshape(-1).long().contiguous(), label_t.reshape(-1).long().contiguous()
        Z = torch.empty(2 * m, D + num_class, dtype=torch.float32, device=feat_s.device)
        check(lib().sug_mmd_assemble(_p(fs), fs.stride(0), _p(ft), ft.stride(0), _p(ls), _p(lt), m, D, num_class,
                                     float(scale), _p(Z), _st()), 'sug_mmd_assemble')
        ctx.md = (m, D)
        return Z

    @staticmethod
    def backward(ctx, g):
        m, D = ctx.md
        return g[:m, :D], g[m:, :D], None, None, None, None


def mmd_assemble(feat_s, feat_t, label_s, label_t, scale, num_class=10):
    if feat_s.dtype != torch.float32 or feat_t.dtype != torch.float32 or feat_s.shape != feat_t.shape or feat_s.dim() != 2:
        raise RuntimeError('sug_amd.ops.mmd_assemble: two fp32 [m, D] feature blocks of one shape')
    return _AssembleZ.apply(feat_s, feat_t, label_s, label_t, scale, num_class)


class _SoftMMDMulti(torch.autograd.Function):
    """soft_mmd (model/mmd.py:56-66) of up to four (feat_s, feat_t) pairs of ONE batch -- same samples, same labels -- with one
    launch per stage: assemble, kernel sums (+ derivative weights), values; one backward launch (sug_soft_mmd_multi_*).
    Every term's value and gradient equal mmd_assemble + mix_rbf_mmd2_rows of that term bit for bit."""

    @staticmethod
    def forward(ctx, label_s, label_t, scales, ws, sigmas, num_class, *feats):
        n = len(feats) // 2
        _need_gpu(label_s, label_t, *feats)
        m = feats[0].shape[0]
        dev = feats[0].device
        fs = [f if f.stride(1) == 1 else f.contiguous() for f in feats[0::2]]
        ft = [f if f.stride(1) == 1 else f.contiguous() for f in feats[1::2]]
        ls, lt = label_s.reshape(-1).long().contiguous(), label_t.reshape(-1).long().contiguous()
        Ds = [f.shape[1] for f in fs]
        need = [ctx.needs_input_grad[6 + 2 * i] or ctx.needs_input_grad[7 + 2 * i] for i in range(n)]
        Z = [torch.empty(2 * m, D + num_class, dtype=torch.float32, device=dev) for D in Ds]
        wt = [torch.empty(2 * m, 2 * m, dtype=torch.float32, device=dev) if nd else None for nd in need]
        wc = [None if w is None else w.detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous() for w in ws]
        for w in wc:
            if w is not None and w.numel() != m:
                raise RuntimeError('soft_mmd_multi: %d sample weights for %d samples' % (w.numel(), m))
        sums = torch.empty(3 * n, dtype=torch.float64, device=dev)
        vals = torch.empty(n, dtype=torch.float32, device=dev)
        ng = _neg_gammas(sigmas, dev)
        I64, I32, F32 = ctypes.c_int64 * n, ctypes.c_int32 * n, ctypes.c_float * n
        check(lib().sug_soft_mmd_multi_fwd(n, _ptrs(fs), I64(*[f.stride(0) for f in fs]), _ptrs(ft), I64(*[f.stride(0) for f in ft]),
                                           I32(*Ds), F32(*[float(v) for v in scales]), _p(ls), _p(lt), m, num_class, _ptrs(wc),
                                           _p(ng), len(sigmas), _ptrs(Z), _ptrs(wt), _p(sums), _p(vals), _st()),
              'sug_soft_mmd_multi_fwd')
        ctx.meta = (n, m, num_class, Ds, need)
        ctx.save_for_backward(*(Z + wt))
        ctx.set_materialize_grads(False)
        return tuple(vals.unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        n, m, num_class, Ds, need = ctx.meta
        saved = ctx.saved_tensors
        Z, wt = saved[:n], saved[n:]
        dev = Z[0].device
        live = [need[i] and gs[i] is not None for i in range(n)]
        gsc = [gs[i].detach().to(device=dev, dtype=torch.float32).reshape(1) if live[i] else None for i in range(n)]
        dZ = [torch.empty(2 * m, Ds[i], dtype=torch.float32, device=dev) if live[i] else None for i in range(n)]
        if any(live):
            check(lib().sug_soft_mmd_multi_bwd(n, _ptrs(Z), (ctypes.c_int32 * n)(*Ds), _ptrs(wt), _ptrs(gsc), m, num_class, _ptrs(dZ),
                                               _st()), 'sug_soft_mmd_multi_bwd')
        out = [None] * 6
        for i in range(n):
            out += [dZ[i][:m] if (live[i] and ctx.needs_input_grad[6 + 2 * i]) else None,
                    dZ[i][m:] if (live[i] and ctx.needs_input_grad[7 + 2 * i]) else None]
        return tuple(out)


def soft_mmd_multi(label_s, label_t, terms, sigmas=SIGMA_LIST, num_class=10):
    """[soft MMD^2 of (feat_s, feat_t, label_weight, sample_weights) for each of `terms`] on one batch (label_s / label_t [m]
    shared by the terms), at most four terms per launch set."""
    terms = list(terms)
    for fs, ft, _, _ in terms:
        if fs.dtype != torch.float32 or ft.dtype != torch.float32 or fs.shape != ft.shape or fs.dim() != 2 \
                or fs.shape[0] != terms[0][0].shape[0]:
            raise RuntimeError('sug_amd.ops.soft_mmd_multi: fp32 [m, D] feature blocks, one m for all terms')
    out = []
    for i in range(0, len(terms), 4):
        grp = terms[i:i + 4]
        feats = [t for fs, ft, _, _ in grp for t in (fs, ft)]
        out += list(_SoftMMDMulti.apply(label_s, label_t, tuple(float(g[2]) for g in grp), tuple(g[3] for g in grp), tuple(sigmas),
                                        int(num_class), *feats))
    return out


def colsum(x2, sign=1.0):
    """fp32 column sums [C] of a [R, C] fp32 / fp16 matrix (sug_colsum: no memset, fixed order)."""
    _need_gpu(x2)
    if x2.dim() != 2 or x2.dtype not in (torch.float32, torch.float16):
        raise RuntimeError('sug_amd.ops.colsum: a 2-D fp32 / fp16 matrix')
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    R, C = x2.shape
    out = torch.empty(C, dtype=torch.float32, device=x2.device)
    ws = torch.empty(int(lib().sug_colsum_workspace(R, C)), dtype=torch.float32, device=x2.device)
    check(lib().sug_colsum(_p(x2), x2.stride(0), R, C, 0 if x2.dtype == torch.float32 else 1, float(sign), _p(out), _p(ws),
                           _st()), 'sug_colsum')
    return out


class _SubRowBias(torch.autograd.Function):
    """rows [..., C] - bias [C]; the bias gradient by sug_colsum instead of autograd's sum_to_size."""

    @staticmethod
    def forward(ctx, rows, bias):
        return rows - bias

    @staticmethod
    def backward(ctx, g):
        db = colsum(g.reshape(-1, g.shape[-1]), -1.0) if ctx.needs_input_grad[1] else None
        return g, db


def sub_row_bias(rows, bias):
    return _SubRowBias.apply(rows, bias)


class _Gate(torch.autograd.Function):
    """x * sigmoid(z) + x (CALayer, model/Model.py:28-34) in one launch forward and one backward instead of
    sigmoid / mul / add and their four backward launches."""

    @staticmethod
    def forward(ctx, x, z):
        _need_gpu(x, z)
        x, z = x.contiguous(), z.contiguous()
        out = torch.empty_like(x)
        check(lib().sug_gate_fwd(_p(x), _p(z), x.numel(), _p(out), _st()), 'sug_gate_fwd')
        ctx.save_for_backward(x, z)
        return out

    @staticmethod
    def backward(ctx, g):
        x, z = ctx.saved_tensors
        g = g.contiguous()
        dx, dz = torch.empty_like(x), torch.empty_like(x)
        check(lib().sug_gate_bwd(_p(g), _p(x), _p(z), x.numel(), _p(dx), _p(dz), _st()), 'sug_gate_bwd')
        return dx, dz


def gate(x, z):
    if x.dtype != torch.float32 or z.dtype != torch.float32 or x.shape != z.shape:
        raise RuntimeError('sug_amd.ops.gate: two fp32 tensors of one shape')
    return _Gate.apply(x, z)


class _GateBN(torch.autograd.Function):
    """BatchNorm1d(x * sigmoid(z) + x) over the M rows of x, z [M, C] -- the tail of CALayer (sug_gate_bn_fwd / _bwd):
    one launch each way instead of the gate + torch's three BatchNorm launches (and its 32 us backward reduce)."""

    @staticmethod
    def forward(ctx, x, z, gamma, beta, running_mean, running_var, training, eps, momentum):
        _need_gpu(x, z, gamma)
        x, z = x.contiguous(), z.contiguous()
        M, C = x.shape
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        out = torch.empty_like(x)
        stat = torch.empty(2, C, dtype=torch.float32, device=x.device)
        check(lib().sug_gate_bn_fwd(_p(x), _p(z), M, C, _p(g), _p(b), _p(running_mean), _p(running_var), 1 if training else 0,
                                    float(eps), float(momentum), _p(out), _p(stat), _st()), 'sug_gate_bn_fwd')
        ctx.save_for_backward(x, z, g, stat)
        ctx.training = bool(training)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, z, g, stat = ctx.saved_tensors
        M, C = x.shape
        gout = gout.contiguous()
        dx, dz = torch.empty_like(x), torch.empty_like(x)
        dgb = torch.empty(2, C, dtype=torch.float32, device=x.device)
        check(lib().sug_gate_bn_bwd(_p(gout), _p(x), _p(z), M, C, _p(g), _p(stat), 1 if ctx.training else 0, _p(dx), _p(dz),
                                    _p(dgb[0]), _p(dgb[1]), _st()), 'sug_gate_bn_bwd')
        return dx, dz, dgb[0], dgb[1], None, None, None, None, None


def gate_bn_supported(x, bn):
    return GATE_BN_FUSED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] <= 1024 and bn.affine and \
        (bn.training or bn.track_running_stats) and bn.momentum is not None and CTX.bn_groups == 1


def gate_bn(x, z, bn):
    """bn(x * sigmoid(z) + x) for an nn.BatchNorm1d over [M, C] rows (CALayer, model/Model.py:28-34)."""
    if x.dtype != torch.float32 or z.dtype != torch.float32 or x.shape != z.shape:
        raise RuntimeError('sug_amd.ops.gate_bn: two fp32 tensors of one shape')
    _count_bn_call(bn)
    track = bn.track_running_stats
    return _GateBN.apply(x, z, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
                         bn.training or not track, bn.eps, bn.momentum)


GATE_BN_FUSED = _os.environ.get('SUG_GATE_BN_FUSED', '1') != '0'


class _EdgeWeightSplit(torch.autograd.Function):
    """W [Co, 2C] -> [W1 ; W2 - W1] [2Co, C], the EdgeConv GEMM operand (one launch forward, one backward:
    dW = [gP - gQ | gQ])."""

    @staticmethod
    def forward(ctx, W):
        _need_gpu(W)
        W = W.contiguous()
        Co, C = W.shape[0], W.shape[1] // 2
        out = torch.empty(2 * Co, C, dtype=torch.float32, device=W.device)
        check(lib().sug_edge_weight_split(_p(W), Co, C, 0, _p(out), _st()), 'sug_edge_weight_split')
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        Co, C = g.shape[0] // 2, g.shape[1]
        dW = torch.empty(Co, 2 * C, dtype=torch.float32, device=g.device)
        check(lib().sug_edge_weight_split(_p(g), Co, C, 1, _p(dW), _st()), 'sug_edge_weight_split')
        return dW


def edge_weight_split(W):
    if W.dtype != torch.float32 or W.dim() != 2 or W.shape[1] % 2:
        raise RuntimeError('sug_amd.ops.edge_weight_split: fp32 [Co, 2C] weight')
    return _EdgeWeightSplit.apply(W)


class _EdgeWeightSplitMulti(torch.autograd.Function):
    """_EdgeWeightSplit for several weights at once: one launch forward, one backward (sug_edge_weight_split_multi); a
    split that received no gradient hands None to its weight."""

    @staticmethod
    def forward(ctx, *Ws):
        _need_gpu(*Ws)
        Ws = [W.contiguous() for W in Ws]
        n = len(Ws)
        ctx.shapes = [(W.shape[0], W.shape[1] // 2) for W in Ws]
        outs = [torch.empty(2 * Co, C, dtype=torch.float32, device=W.device) for W, (Co, C) in zip(Ws, ctx.shapes)]
        I32 = ctypes.c_int32 * n
        check(lib().sug_edge_weight_split_multi(_ptrs(Ws), I32(*[s[0] for s in ctx.shapes]), I32(*[s[1] for s in ctx.shapes]), n, 0,
                                                _ptrs(outs), _st()), 'sug_edge_weight_split_multi')
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        n = len(gs)
        gs = [None if (g is None or not ctx.needs_input_grad[i]) else g.contiguous() for i, g in enumerate(gs)]
        if all(g is None for g in gs):
            return (None,) * n
        dev = next(g for g in gs if g is not None).device
        dWs = [None if g is None else torch.empty(Co, 2 * C, dtype=torch.float32, device=dev) for g, (Co, C) in zip(gs, ctx.shapes)]
        I32 = ctypes.c_int32 * n
        check(lib().sug_edge_weight_split_multi(_ptrs(gs), I32(*[s[0] for s in ctx.shapes]), I32(*[s[1] for s in ctx.shapes]), n, 1,
                                                _ptrs(dWs), _st()), 'sug_edge_weight_split_multi')
        return tuple(dWs)


def edge_weight_split_multi(Ws):
    """[edge_weight_split(W) for W in Ws] in one launch each way (at most 8 weights per launch)."""
    Ws = list(Ws)
    for W in Ws:
        if W.dtype != torch.float32 or W.dim() != 2 or W.shape[1] % 2:
            raise RuntimeError('sug_amd.ops.edge_weight_split_multi: fp32 [Co, 2C] weights')
    if len(Ws) == 1:
        return [_EdgeWeightSplit.apply(Ws[0])]
    out = []
    for i in range(0, len(Ws), 8):
        out += list(_EdgeWeightSplitMulti.apply(*Ws[i:i + 8]))
    return out


_SDA_METHODS = {'none': 0, 'naive_inverse': 1, 'exp_inverse': 2, 'mean2one': 3}


def sda_prob_weights(pred_s, pred_t, label_s, label_t, label_weight, method):
    """SDA weights [m] from logits [m,10] and labels [m] (prob_weights_soft + distance2weights,
    model/mmd.py:134-148, :178-202) in one kernel."""
    _need_gpu(pred_s, pred_t, label_s, label_t)
    if method not in _SDA_METHODS:
        raise RuntimeError('Not supported weighting method %s' % method)
    ps = pred_s.detach().reshape(-1, 10).float()
    pt = pred_t.detach().reshape(-1, 10).float()
    ps = ps if ps.stride(1) == 1 else ps.contiguous()
    pt = pt if pt.stride(1) == 1 else pt.contiguous()
    ls, lt = label_s.reshape(-1).long().contiguous(), label_t.reshape(-1).long().contiguous()
    m = ps.shape[0]
    out = torch.empty(m, dtype=torch.float32, device=ps.device)
    check(lib().sug_sda_prob_weights(_p(ps), ps.stride(0), _p(pt), pt.stride(0), _p(ls), _p(lt), m, 10,
                                     float(label_weight), _SDA_METHODS[method], _p(out), _st()),
          'sug_sda_prob_weights')
    return out


def sda_prob_weights_multi(preds, label_s, label_t, label_weight, method):
    """[sda_prob_weights(ps, pt, ...) for (ps, pt) in preds] in one launch (at most four heads; one batch, one set of labels)."""
    preds = list(preds)
    if len(preds) == 1:
        return [sda_prob_weights(preds[0][0], preds[0][1], label_s, label_t, label_weight, method)]
    if len(preds) > 4:
        return sda_prob_weights_multi(preds[:4], label_s, label_t, label_weight, method) + \
            sda_prob_weights_multi(preds[4:], label_s, label_t, label_weight, method)
    if method not in _SDA_METHODS:
        raise RuntimeError('Not supported weighting method %s' % method)
    rows = lambda p: (lambda q: q if q.stride(1) == 1 else q.contiguous())(p.detach().reshape(-1, 10).float())
    ps, pt = [rows(a) for a, _ in preds], [rows(b) for _, b in preds]
    _need_gpu(label_s, label_t, *(ps + pt))
    ls, lt = label_s.reshape(-1).long().contiguous(), label_t.reshape(-1).long().contiguous()
    m, n = ps[0].shape[0], len(preds)
    if any(t.shape[0] != m for t in ps + pt):
        raise RuntimeError('sda_prob_weights_multi: one batch size for all heads')
    outs = [torch.empty(m, dtype=torch.float32, device=ps[0].device) for _ in range(n)]
    I64 = ctypes.c_int64 * n
    check(lib().sug_sda_prob_weights_multi(n, _ptrs(ps), I64(*[t.stride(0) for t in ps]), _ptrs(pt), I64(*[t.stride(0) for t in pt]),
                                           _p(ls), _p(lt), m, 10, float(label_weight), _SDA_METHODS[method], _ptrs(outs), _st()),
          'sug_sda_prob_weights_multi')
    return outs


def chamfer(a, b):
    """a [B,N,3], b [B,M,3] -> [B]: mean_i min_j d + mean_j min_i d."""
    _need_gpu(a, b)
    a, b = a.detach().contiguous(), b.detach().contiguous()
    B, N, _ = a.shape
    M = b.shape[1]
    out = torch.empty(B, dtype=torch.float32, device=a.device)
    ws = torch.empty(lib().sug_chamfer_workspace(B, N, M), dtype=torch.float32, device=a.device)
    check(lib().sug_chamfer(_p(a), _p(b), B, N, M, _p(out), _p(ws), _st()), 'sug_chamfer')
    return out


def chamfer_weights(a, b, method):
    """distance2weights(chamfer(a, b), method) [B] for method in naive_inverse / exp_inverse / mean2one, the weighting inside
    the fold launch of the distance (sug_chamfer_weights)."""
    _need_gpu(a, b)
    if method not in ('naive_inverse', 'exp_inverse', 'mean2one'):
        raise RuntimeError('Not supported weighting method %s' % method)
    a, b = a.detach().contiguous(), b.detach().contiguous()
    B, N, _ = a.shape
    M = b.shape[1]
    out = torch.empty(B, dtype=torch.float32, device=a.device)
    ws = torch.empty(lib().sug_chamfer_workspace(B, N, M), dtype=torch.float32, device=a.device)
    check(lib().sug_chamfer_weights(_p(a), _p(b), B, N, M, _SDA_METHODS[method], _p(out), _p(ws), _st()), 'sug_chamfer_weights')
    return out


# ----------------------------------------------------------------------------- the old module-level names of the context fields
def _ctx_property(field):
    return property(lambda self: getattr(CTX, field), lambda self, v: setattr(CTX, field, v))


class _OpsModule(type(_os)):
    """`ops.BN_GROUPS`, `ops.START_PROVIDER = f`, ... read and write CTX (see StepContext)."""
    BN_GROUPS = _ctx_property('bn_groups')
    START_QUEUE = _ctx_property('start_queue')
    START_PROVIDER = _ctx_property('start_provider')
    GEOMETRY_PLAN = _ctx_property('geometry_plan')
    PROFILE_ONLY = _ctx_property('profile_only')
    PROFILE = _ctx_property('profile')
    PARALLEL_BRANCHES = _ctx_property('parallel_branches')
    FUSED_HEADS = _ctx_property('fused_heads')
    W16_CACHE = _ctx_property('w16_cache')
    BN_RECORD = _ctx_property('bn_record')
    _PENDING_COUNTS = _ctx_property('pending_counts')
    _LAST_COEF_PAIR = _ctx_property('last_coef_pair')
    _LAST_COEF = _ctx_property('last_coef')


import sys as _sys
_sys.modules[__name__].__class__ = _OpsModule

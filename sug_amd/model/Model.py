"""Encoders, heads and the Net_MDA wrapper (mirror of the reference's model/Model.py).

The forward API (`Net_MDA.forward` flags and return tuples, model/Model.py:485-520), the
sub-module names (`g`, `c1`, `c2`, `attention_s`, `attention_t`) and every parameter name
match the reference, so `train_dg_single_gpu.py`-style drivers and reference checkpoints
work unchanged.  Inside, the encoders run on point-major rows [B,N,C] and call the HIP
kernels (sug_amd.ops); inputs and outputs keep the reference layout ([B,3,N,1] in,
[B,1024] / [B,64,64,1] out).  KPConv / PointNet++-MSG are out of scope.
"""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .model_utils import conv_2d, fc_layer, transform_net, adapt_layer_off, _bn_rows, bn_module, edge_wcats
from . import model_utils as _mu
from .pointnet2_utils import PointNetSetAbstraction
from . import PTran_utils
from .Ptran_transformer import TransformerBlock

K = 20      # model/Model.py:52


class _PrefixEntry:
    """Cached result of an encoder prefix (DGCNN: kNN + conv1, kNN + conv2; Point Transformer: fc1 + transformer1) that a
    second forward on the SAME batch with the SAME weights may reuse.  `alive` turns False as soon as a backward pass
    reaches the cached tensors: their autograd graph is consumed then, and a later forward must recompute."""

    def __init__(self, tensors, extra=None, source=None, live=None, provider=None):
        self.tensors, self.extra = tensors, extra
        # call-graph mode (sug_amd.call_graphs): the tensors are hidden outputs of a replayed forward graph; `live` says
        # whether that replay is still the current one and its backward has not run, `provider` is the graph instance
        self._live, self.provider = live, provider
        # identity of the input batch: (data_ptr, _version, shape) alone is NOT tensor identity -- the caching allocator
        # hands a freed batch's address to the next one with _version 0 again -- so a hit also requires the
        # very same tensor object; a weak reference, the entry must not keep the batch alive
        self._source = weakref.ref(source) if source is not None else None
        # the hook must not reference this entry (entry -> tensor -> grad_fn -> hook -> entry would be a cycle, and the
        # cached tensors -- graph-pool memory under hipGraph capture -- would wait for the cyclic collector)
        self._flag = flag = [True]
        last = tensors[-1]
        if last.requires_grad:
            last.register_hook(lambda grad, f=flag: f.__setitem__(0, False))

    @property
    def alive(self):
        return self._flag[0]

    # copy.deepcopy(model) (train_dg_single_gpu.py:364) and pickling meet the cache inside the encoder: the cached tensors
    # are autograd non-leaves (not copyable) and belong to the original's graph -- the copy gets a dead, empty entry
    def __deepcopy__(self, memo):
        return _dead_prefix_entry()

    def __reduce__(self):
        return (_dead_prefix_entry, ())

    def servable(self):
        """Could this entry still serve its batch?"""
        return self._flag[0] and self._source is not None and self._source() is not None and (self._live is None or self._live())

    def serves(self, x):
        """True if this entry was computed from the tensor object `x` and its autograd graph is still unconsumed."""
        return self._flag[0] and self._source is not None and self._source() is x and (self._live is None or self._live())


class _PrefixSharing:
    """Helpers shared by the encoders that cache a prefix (DGCNN, Pointnet_g, PTran_g): the cache key, a look-up by input
    tensor and the installation of an entry computed elsewhere (sug_amd.call_graphs: the prefix of a replayed forward graph)."""

    def _prefix_params(self):
        raise NotImplementedError

    def _prefix_key(self, x):
        ver = sum(p._version for p in self._prefix_params())
        return (x.data_ptr(), x._version, tuple(x.shape), torch.is_grad_enabled(), ver, ops.CTX.bn_groups)

    def find_prefix(self, x):
        """The live cache entry computed from the tensor object `x` under the current weights, or None."""
        hit = self._prefix_cache.get(self._prefix_key(x))
        return hit if (hit is not None and hit.serves(x)) else None

    def last_prefix(self, x):
        """The entry a forward on `x` has just stored (whatever its liveness)."""
        return self._prefix_cache.get(self._prefix_key(x))

    def _prune_prefix_cache(self):
        """Before an insertion: entries that can no longer serve (graph consumed, batch gone, replay superseded) go; the
        cache never grows beyond a few live batches."""
        if len(self._prefix_cache) >= 4:
            self._prefix_cache = {k: e for k, e in self._prefix_cache.items() if e.servable()}
            if len(self._prefix_cache) >= 8:
                self._prefix_cache.clear()

    def install_prefix(self, x, tensors, extra, live=None, provider=None):
        self._prune_prefix_cache()
        e = self._prefix_cache[self._prefix_key(x)] = _PrefixEntry(tuple(tensors), extra, source=x, live=live, provider=provider)
        return e

    def clear_prefix_cache(self):
        self._prefix_cache = {}


def _dead_prefix_entry():
    e = _PrefixEntry.__new__(_PrefixEntry)
    e.tensors, e.extra, e._source, e._flag, e._live, e.provider = (), None, None, [False], None, None
    return e


def _sharing_on(flag, training):
    """share_prefix: True (SUGStep: one backward over all passes of a step), False, or 'auto' (default): share while it
    is provably the same computation -- the SAME input tensor object (weak reference) at the same version, same parameter
    versions, training mode, and the
    earlier pass's graph not yet consumed by a backward (what train_dg_single_gpu.py:260-335 does: four forwards, one
    backward)."""
    return bool(flag) and training


class CALayer(nn.Module):
    """Channel attention on the flattened node features (model/Model.py:16-34)."""

    def __init__(self, channel, reduction=8):
        super(CALayer, self).__init__()
        self.conv_du = nn.Sequential(
            nn.Conv2d(channel, channel // reduction, 1, padding=0, bias=True),
            nn.ReLU(inplace=False),
            nn.Conv2d(channel // reduction, channel, 1, padding=0, bias=True),
            nn.Sigmoid())
        self.bn = nn.BatchNorm1d(4096)

    def forward(self, x):
        """x [B,4096,1,1] (or [B,4096]) -> [B,4096]; the two 1x1 convs on a 1x1 map are GEMMs."""
        v = x.reshape(x.shape[0], -1)
        if ops.calayer_supported((self,), v):
            # both GEMMs, the bias / ReLU between them and all their gradients in sug_calayer_* (one launch per stage)
            return ops.calayers((self,), v)
        c0, c2 = self.conv_du[0], self.conv_du[2]
        y = F.relu(F.linear(v, c0.weight.view(c0.weight.shape[0], -1), c0.bias))
        z = F.linear(y, c2.weight.view(c2.weight.shape[0], -1), c2.bias)
        if ops.gate_bn_supported(v, self.bn):
            # gate + BatchNorm1d over the B samples in one launch (two-pass statistics, as torch's kernel)
            return ops.gate_bn(v, z, self.bn)
        return self.bn(ops.gate(v, z))                                      # v * sigmoid(z) + v


class GradReverse(torch.autograd.Function):
    """model/Model.py:37-50: the reference calls `.forward` directly, so it is the identity in
    both directions; kept for API parity."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


def grad_reverse(x, lambd=1.0):
    return x.view_as(x)


class DGCNN(nn.Module, _PrefixSharing):
    """EdgeConv encoder with the SA-node module (model/Model.py:54-121)."""

    def __init__(self):
        super(DGCNN, self).__init__()
        self.k = K
        self.input_transform_net = transform_net(6, 3)      # unused by forward, as in the reference
        self.conv1 = conv_2d(6, 64, kernel=1, bias=False, activation='leakyrelu')
        self.conv2 = conv_2d(64 * 2, 64, kernel=1, bias=False, activation='leakyrelu')
        self.conv3 = conv_2d(64 * 2, 128, kernel=1, bias=False, activation='leakyrelu')
        self.conv4 = conv_2d(128 * 2, 256, kernel=1, bias=False, activation='leakyrelu')
        self.bn5 = nn.BatchNorm1d(512)
        self.conv5 = nn.Conv1d(64 + 64 + 128 + 256, 512, kernel_size=1, bias=False)
        self.node_fea_adapt = adapt_layer_off()
        self.conv1d = nn.Conv1d(128, 64, 1)
        self.dim_redu = nn.MaxPool1d(3, stride=16)
        # Opt-in common-subexpression sharing (set by SUGStep): the semantic and the node pass of
        # one training step run the encoder twice on the same batch with the same weights; the
        # stage in front of the SA-node module (kNN + conv1, kNN + conv2) is then bit-identical
        # in both passes (train-mode BN uses batch statistics, no dropout, deterministic
        # kernels), so the second pass reuses it and only replays the BN running-stat update.
        # Requires a single backward over both passes (the graph of the prefix is shared).
        self.share_prefix = 'auto'
        self._prefix_cache = {}

    def _prefix_params(self):
        return [p for m in (self.conv1, self.conv2) for p in m.parameters()]

    def _prefix(self, x, loc, nb, out1=None, wc=(None, None)):
        """kNN + conv1, kNN + conv2.  out1: where conv1's activations should be written (a column
        slice of the conv5 input buffer); a cache hit returns the tensors of the earlier pass instead."""
        if not _sharing_on(self.share_prefix, self.training):
            x1 = self.conv1.edge_rows(loc, nb(loc, 0), out=out1, wcat=wc[0])
            return x1, self.conv2.edge_rows(x1, nb(x1, 1), wcat=wc[1])
        key = self._prefix_key(x)
        hit = self._prefix_cache.get(key)
        if hit is not None and hit.serves(x):
            x1, x2 = hit.tensors
            st1, st2 = hit.extra
            ops.replay_bn_stats([(self.conv1.conv[1], st1), (self.conv2.conv[1], st2)])     # both layers' updates, one launch
            return x1, x2
        x1, st1 = self.conv1.edge_rows(loc, nb(loc, 0), return_stats=True, out=out1, wcat=wc[0])
        x2, st2 = self.conv2.edge_rows(x1, nb(x1, 1), return_stats=True, wcat=wc[1])
        self._prune_prefix_cache()                  # a step has two inputs; never grow unbounded
        self._prefix_cache[key] = _PrefixEntry((x1, x2), (st1, st2), source=x)
        return x1, x2

    def plan_geometry(self, x, passes, groups=1):
        """FPS + ball query of the SA-node module for `passes` forwards over the same batch, in one set of launches."""
        return self.node_fea_adapt.plan_geometry(ops.cloud_rows(x), passes, groups)

    def forward(self, x, node=False, knn_idx=None, feat_grad=True):
        """x [B,3,N,1] -> (feat [B,1024], node_fea [B,64,64,1](, None)).
        `knn_idx` (4 tensors [B,N,k]) overrides the neighbour graphs (tests: teacher forcing).
        feat_grad=False: the caller discards `feat` (node-adaptation pass): the stage behind the
        SA-node module still runs -- it updates BatchNorm running statistics -- but without autograd."""
        B, N = x.size(0), x.size(2)
        loc = ops.cloud_rows(x)                                       # [B,N,3] rows (one copy per batch)
        gi = knn_idx or [None] * 4
        if knn_idx is not None:
            self._prefix_cache = {}
        nb = lambda f, i: gi[i] if gi[i] is not None else ops.knn(f, self.k)
        # conv5 consumes cat(x1, x2, x3, x4): the EdgeConv layers write their activations straight
        # into the column slices of that [B,N,512] buffer instead of concatenating afterwards
        cat_in = torch.empty(B, N, 512, dtype=torch.float32, device=x.device)
        # the [W1 ; W2-W1] operands of the four EdgeConv layers in one launch (one more in the backward)
        wc = edge_wcats((self.conv1, self.conv2, self.conv3, self.conv4))
        x1, x2 = self._prefix(x, loc, nb, out1=cat_in[:, :, 0:64], wc=wc[:2])   # [B,N,64], [B,N,64]
        x_, node_fea, _ = self.node_fea_adapt.rows(x2, loc)           # [B,N,128], [B,64,64]
        with torch.set_grad_enabled(torch.is_grad_enabled() and feat_grad):
            x2 = ops.linear_rows(x_, self.conv1d.weight.squeeze(-1), self.conv1d.bias)
            x3 = self.conv3.edge_rows(x2, nb(x2, 2), out=cat_in[:, :, 128:256], wcat=wc[2])     # [B,N,128]
            x4 = self.conv4.edge_rows(x3, nb(x3, 3), out=cat_in[:, :, 256:512], wcat=wc[3])     # [B,N,256]
            x5 = ops.linear_rows(ops.assemble_rows(cat_in, (x1, x2, x3, x4)), self.conv5.weight.squeeze(-1))
            if feat_grad:
                # bn5 -> leaky_relu(0.2) -> adaptive max | avg pool (Model.py:113-116), fused
                feat = ops.bn_act_pool_cat(x5, self.bn5, 0.2)
            else:
                ops.bn_update_stats(x5, self.bn5)        # the pooled feature is not used: buffers only
                feat = None
        node_fea = node_fea.transpose(1, 2).unsqueeze(-1)             # [B,64(ch),64(node),1]
        if node:
            return feat, node_fea, None
        return feat, node_fea


class Pointnet2_g(nn.Module):
    """PointNet++ encoder (model/Model.py:123-161)."""

    def __init__(self, normal_channel=False):
        super(Pointnet2_g, self).__init__()
        in_channel = 6 if normal_channel else 3
        self.normal_channel = normal_channel
        self.num_class = 10
        self.sa1 = PointNetSetAbstraction(512, 0.2, 32, in_channel, [64, 64, 128], False)
        self.sa2 = PointNetSetAbstraction(128, 0.4, 64, 128 + 3, [128, 128, 256], False)
        self.sa3 = PointNetSetAbstraction(None, None, None, 256 + 3, [256, 512, 1024], True)
        self.channel_redu = nn.Conv2d(512, 64, 1)
        self.dim_redu = nn.MaxPool1d(3, stride=8)

    def fps_plan(self, N):
        """Point counts of the farthest_point_sample calls of one forward, in call order."""
        return [N, self.sa1.npoint]

    def can_plan_geometry(self):
        """Does plan_geometry cover this configuration (it returns None otherwise, without drawing a start)?"""
        return not (self.normal_channel or self.sa1.group_all or self.sa2.group_all) and \
            all(sa.takes_index_path() for sa in (self.sa1, self.sa2))

    def plan_geometry(self, xyz, passes, groups=1):
        """Sampling and grouping indices of `passes` forwards over the same batch xyz [B,3,N,1], in one set of launches
        (not in the reference; used by SUGStep: the semantic and the node pass of a step).  FPS and ball query read the
        coordinates and the start draws only; the draws are made here, in the order the `passes` forwards would make them
        (per pass and domain group: sa1's, then sa2's -- pointnet2_utils.py:72), so the random stream is unchanged.
        Returns a list of per-pass plans for ops.CTX.geometry_plan."""
        if self.normal_channel or self.sa1.group_all or self.sa2.group_all:
            return None
        # only sample_and_group_idx consumes a plan: a layer that takes the grouped-tensor path (SUG_SA_FIRST=0, an
        # unsupported width) draws its own start, so planning would draw twice and break the reference's random stream
        if not all(sa.takes_index_path() for sa in (self.sa1, self.sa2)):
            return None
        loc = ops.cloud_rows(xyz)[:, :, :3].contiguous()
        B, N = loc.shape[0], loc.shape[1]
        S1, S2 = self.sa1.npoint, self.sa2.npoint
        st1, st2 = [], []
        dev = loc.device
        for _ in range(passes):
            if groups == 1:
                st1.append(ops.draw_start(B, N))
                st2.append(ops.draw_start(B, S1))
            else:                   # one forward per domain group in the reference: group 0 draws (N, S1), then group 1
                # (the same order in eager and in graph mode -- the graph's start feeder records / replays the group draws --
                # so a captured step and its eager twin see the same starts)
                d = [[ops.draw_group_start(B // groups, n) for n in (N, S1)] for _ in range(groups)]
                st1.append(torch.cat([d[g][0].to(device=dev, dtype=torch.int32, non_blocking=True) for g in range(groups)]))
                st2.append(torch.cat([d[g][1].to(device=dev, dtype=torch.int32, non_blocking=True) for g in range(groups)]))
        cat_starts = lambda sts: torch.cat([t.to(device=dev, dtype=torch.int32, non_blocking=True) for t in sts])
        locp = loc.repeat(passes, 1, 1) if passes > 1 else loc
        f1 = ops.fps(locp, S1, cat_starts(st1))
        nx1 = ops.gather_rows(locp, f1)
        i1 = ops.ball_query(locp, nx1, self.sa1.radius, self.sa1.nsample)
        f2 = ops.fps(nx1, S2, cat_starts(st2))
        nx2 = ops.gather_rows(nx1, f2)
        i2 = ops.ball_query(nx1, nx2, self.sa2.radius, self.sa2.nsample)
        return [[(nx1[p * B:(p + 1) * B], i1[p * B:(p + 1) * B]), (nx2[p * B:(p + 1) * B], i2[p * B:(p + 1) * B])]
                for p in range(passes)]

    def forward(self, xyz, node=False, feat_grad=True, need_node=True):
        """feat_grad=False: the caller discards `feat` (node-adaptation pass): everything behind the layer
        the node features are taken from still runs -- it updates BatchNorm running statistics and draws
        its FPS start -- but without autograd.  need_node=False: the caller discards the node features (semantic pass):
        the max over the groups of sa1's middle layer (a pass over [B,512,32,64]) is not formed."""
        rows = ops.cloud_rows(xyz)                                    # [B,N,3(+3)]
        B = rows.shape[0]
        norm = rows[:, :, 3:].contiguous() if self.normal_channel else None
        loc = rows[:, :, :3].contiguous()
        if not need_node:
            l1_xyz, l1_pts = self.sa1.rows(loc, norm)                 # [B,512,3], [B,512,128]
            with torch.set_grad_enabled(torch.is_grad_enabled() and feat_grad):
                l2_xyz, l2_pts = self.sa2.rows(l1_xyz, l1_pts)
                _, l3_pts = self.sa3.rows(l2_xyz, l2_pts)
            feat = l3_pts.reshape(B, 1024)
            return (feat, None, None) if node else (feat, None)
        l1_xyz, l1_pts, node_fea = self.sa1.rows(loc, norm, adapt=True, tail_grad=feat_grad)   # [B,512,3], [B,512,128], [B,512,64]
        with torch.set_grad_enabled(torch.is_grad_enabled() and feat_grad):
            l2_xyz, l2_pts = self.sa2.rows(l1_xyz, l1_pts)
            _, l3_pts = self.sa3.rows(l2_xyz, l2_pts)
        feat = l3_pts.reshape(B, 1024)
        node_fea = self.dim_redu(node_fea.transpose(1, 2)).reshape(B, 64, 64, 1)
        if node:
            return feat, node_fea, None
        return feat, node_fea


class Pointnet_g(nn.Module, _PrefixSharing):
    """PointNet encoder with the SA-node module (model/Model.py:235-283)."""

    def __init__(self):
        super(Pointnet_g, self).__init__()
        self.trans_net1 = transform_net(3, 3)
        self.trans_net2 = transform_net(64, 64)
        self.conv1 = conv_2d(3, 64, 1)
        self.conv2 = conv_2d(64, 64, 1)
        self.conv3 = adapt_layer_off()
        self.conv4 = conv_2d(128, 128, 1)
        self.conv5 = conv_2d(128, 1024, 1)
        self.bn1 = nn.BatchNorm1d(1024)
        # Common-subexpression sharing as for DGCNN: both T-Nets, conv1 and conv2 run in front of the SA-node module
        # (the first farthest-point draw), so the semantic and the node pass of a step compute them on identical inputs
        # with identical weights; the second pass reuses the result and replays the BatchNorm running-statistics
        # updates (ops.record_bn_stats / replay_bn_stats).  One backward over both passes.
        self.share_prefix = 'auto'
        self._prefix_cache = {}

    def _prefix_modules(self):
        return (self.trans_net1, self.conv1, self.conv2, self.trans_net2)

    def _prefix_params(self):
        return [p for mod in self._prefix_modules() for p in mod.parameters()]

    def _prefix(self, x, loc):
        def run():
            y = torch.bmm(loc, self.trans_net1.rows(loc))
            y = self.conv2.rows(self.conv1.rows(y))
            return torch.bmm(y, self.trans_net2.rows(y))
        if not (_sharing_on(self.share_prefix, self.training) and x.is_cuda):
            return run()
        key = self._prefix_key(x)
        hit = self._prefix_cache.get(key)
        if hit is not None and hit.serves(x):
            ops.replay_bn_stats(hit.extra)
            return hit.tensors[0]
        with ops.record_bn_stats() as rec:
            y = run()
        self._prune_prefix_cache()
        self._prefix_cache[key] = _PrefixEntry((y,), rec, source=x)
        return y

    def plan_geometry(self, x, passes, groups=1):
        """FPS + ball query of the SA-node module (conv3) for `passes` forwards over the same batch, in one set of launches."""
        return self.conv3.plan_geometry(ops.cloud_rows(x), passes, groups)

    def forward(self, x, node=False, feat_grad=True):
        """feat_grad=False: node-adaptation pass, the stage behind the SA-node module runs without autograd."""
        loc = ops.cloud_rows(x)                                       # [B,N,3]
        y = self._prefix(x, loc)
        y, node_fea, node_off = self.conv3.rows(y, loc)
        with torch.set_grad_enabled(torch.is_grad_enabled() and feat_grad):
            y = self.conv5.rows_max_after(self.conv4, y)
            bn = self.bn1
            if y.is_cuda and y.dtype == torch.float32 and _mu.OWN_BN(bn) and bn.track_running_stats and bn.momentum is not None \
                    and bn.affine:
                # BatchNorm1d over the B pooled rows, per domain group, in the library's own kernels (three launches each
                # way instead of torch's three per group + cat + counter)
                y = ops.bn_act_rows(y, bn, 1.0)
            else:
                y = bn_module(bn, y)
        node_fea = node_fea.transpose(1, 2).unsqueeze(-1)
        node_off = node_off.transpose(1, 2)
        if node:
            return y, node_fea, node_off
        return y, node_fea


class TransitionDown(nn.Module):
    """model/Model.py:285-292."""

    def __init__(self, k, nneighbor, channels):
        super().__init__()
        self.sa = PTran_utils.PointNetSetAbstraction(k, 0, nneighbor, channels[0], channels[1:], group_all=False,
                                                     knn=True)

    def forward(self, xyz, points):
        return self.sa(xyz, points)


class PTran_g(nn.Module, _PrefixSharing):
    """Point Transformer encoder (model/Model.py:295-337): 5 transformer blocks, 4 transition-down
    stages (FPS to 256/64/16/4 points -- the schedule is fixed to npoints=1024 as in the reference
    even for N=2048 -- kNN-16 grouping, 2-layer MLP, max)."""

    def __init__(self):
        super(PTran_g, self).__init__()
        npoints, nblocks, nneighbor, d_points = 1024, 4, 16, 3
        transformer_dim = 512
        self.fc1 = nn.Sequential(nn.Linear(d_points, 32), nn.ReLU(), nn.Linear(32, 32))
        self.transformer1 = TransformerBlock(32, transformer_dim, nneighbor)
        self.transition_downs = nn.ModuleList()
        self.transformers = nn.ModuleList()
        for i in range(nblocks):
            channel = 32 * 2 ** (i + 1)
            self.transition_downs.append(TransitionDown(npoints // 4 ** (i + 1), nneighbor,
                                                        [channel // 2 + 3, channel, channel]))
            self.transformers.append(TransformerBlock(channel, transformer_dim, nneighbor))
        self.nblocks = nblocks
        self.npoints = npoints
        self.conv1d = nn.Conv1d(64, 64, 1, stride=2)
        # Opt-in common-subexpression sharing (set by SUGStep), as for DGCNN: fc1 + transformer1 run at full
        # resolution, before any farthest-point sampling, have no BatchNorm and no dropout -- the semantic
        # and the node pass of one step compute them on identical inputs with identical weights, so the
        # second pass reuses the first one's result (one backward over both passes).
        self.share_prefix = 'auto'
        self._prefix_cache = {}

    def _prefix_params(self):
        return [p for m in (self.fc1, self.transformer1) for p in m.parameters()]

    def _prefix_key(self, x):
        ver = sum(p._version for p in self._prefix_params())
        return (x.data_ptr(), x._version, tuple(x.shape), torch.is_grad_enabled(), ver)

    def _lift(self, x_):
        """fc1 (Linear 3->32, ReLU, Linear 32->32) on the [B,N,3] rows; on the GPU through ops.linear_rows: its
        split-K weight gradient replaces two one-workgroup library GEMMs over all B*N rows (~0.5 ms at config 5)."""
        h = F.relu(ops.linear_rows(x_, self.fc1[0].weight, self.fc1[0].bias))
        return ops.linear_rows(h, self.fc1[2].weight, self.fc1[2].bias)

    def _prefix(self, x, x_, xyz):
        if not _sharing_on(self.share_prefix, self.training):
            return self.transformer1(xyz, self._lift(x_))[0]
        key = self._prefix_key(x)
        hit = self._prefix_cache.get(key)
        if hit is None or not hit.serves(x):
            self._prune_prefix_cache()
            hit = self._prefix_cache[key] = _PrefixEntry((self.transformer1(xyz, self._lift(x_))[0],), source=x)
        return hit.tensors[0]

    def fps_plan(self, N):
        """Point counts of the farthest_point_sample calls of one forward, in call order."""
        return [N] + [self.npoints // 4 ** (i + 1) for i in range(self.nblocks - 1)]

    def forward(self, x, node=False, feat_grad=True):
        """x [B,3,N,1] -> (feat [B,512], node_fea [B,64,64](, None)).
        feat_grad=False: node-adaptation pass, the stages behind the one the node features come from run
        without autograd (they still draw their FPS starts and update BatchNorm buffers)."""
        x_ = ops.cloud_rows(x)                                        # [B,N,3]
        xyz = x_[..., :3]
        points = self._prefix(x, x_, xyz)
        xyz_and_feats = [(xyz, points)]
        for i in range(self.nblocks):
            with torch.set_grad_enabled(torch.is_grad_enabled() and (feat_grad or i < 2)):
                xyz, points = self.transition_downs[i](xyz, points)
                points = self.transformers[i](xyz, points)[0]
            xyz_and_feats.append((xyz, points))
        # Conv1d(64, 64, 1, stride=2) over [B, 64 points (as channels), 128]: a [64,64] product on every second column
        # (MIOpen runs this shape through its naive double-accumulating kernels, ~30 us per direction)
        nf = xyz_and_feats[2][1]
        node_features = torch.matmul(self.conv1d.weight.squeeze(-1), nf[:, :, ::2]) + self.conv1d.bias.view(1, -1, 1)
        points = points.mean(1)
        if node:
            return points, node_features, None
        return points, node_features


class Pointnet_c(nn.Module):
    """Classifier head (model/Model.py:412-449)."""

    def __init__(self, num_class=10, dgcnn_flag=False, PTran_flag=False):
        super(Pointnet_c, self).__init__()
        activate, bias = ('leakyrelu', True) if dgcnn_flag else ('relu', False)
        self.mlp1 = fc_layer(1024, 512, bn=True, activation=activate, bias=bias)
        self.dropout1 = nn.Dropout2d(p=0.4)
        self.mlp2 = fc_layer(512, 256, bn=True, activation=activate, bias=True)
        self.dropout2 = nn.Dropout2d(p=0.4)
        self.mlp3 = nn.Linear(256, num_class)
        self.PTran = PTran_flag

    @staticmethod
    def _drop(layer, x):
        """nn.Dropout2d on a 2-D [B,C] input (what the reference feeds it, Model.py:428-431) draws one
        Bernoulli per (sample, channel) element, i.e. it IS element-wise dropout; F.dropout does that in one
        fused launch instead of three (bernoulli_, div_, mul) and one instead of two in the backward."""
        if x.dim() == 2:
            return F.dropout(x, layer.p, layer.training)
        return layer(x)

    def forward(self, x, adapt=False):
        if not self.PTran:
            x = self._drop(self.dropout1, self.mlp1(x))
        x = self.mlp2(x)
        mid_feature = x
        x = self.mlp3(self._drop(self.dropout2, x))
        if adapt:
            return x, mid_feature
        return x


def _check_input(x):
    """The GPU-only contract at the model boundary: the encoders are HIP kernels behind a C ABI,
    there is no CPU or 16-bit-input path -- fail here with a clear message instead of deep inside an op."""
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError('sug_amd.Net_MDA runs on a HIP device with fp32 clouds [B,3,N,1] only (got %s on %s): '
                           'there is no CPU / eager fallback; move the model and the batch to the GPU (.cuda())'
                           % (x.dtype, x.device))
    if x.dim() != 4 or x.shape[1] < 3 or x.shape[3] != 1:
        raise RuntimeError('sug_amd.Net_MDA expects clouds as [B,3,N,1] (model/Model.py:485), got %s' % (tuple(x.shape),))


class Net_MDA(nn.Module):
    """model/Model.py:452-520.  model_name in {'Pointnet', 'Pointnet2', 'DGCNN', 'PTran'}."""

    def __init__(self, model_name='Pointnet'):
        super(Net_MDA, self).__init__()
        self.dgcnn_flag = False
        self.PTran_flag = False
        if model_name == 'Pointnet':
            self.g = Pointnet_g()
        elif model_name == 'Pointnet2':
            self.g = Pointnet2_g()
        elif model_name == 'DGCNN':
            self.g = DGCNN()
            self.dgcnn_flag = True
        elif model_name == 'PTran':
            self.g = PTran_g()
            self.PTran_flag = True
        else:
            raise NotImplementedError("Unsupported model name")
        self.attention_s = CALayer(64 * 64)
        self.attention_t = CALayer(64 * 64)
        self.c1 = Pointnet_c(dgcnn_flag=self.dgcnn_flag, PTran_flag=self.PTran_flag)
        self.c2 = Pointnet_c(dgcnn_flag=self.dgcnn_flag, PTran_flag=self.PTran_flag)

    # `forward(..., semantic_adaption=True, node_adaptation_s=True)` (or node_adaptation_t) is the opt-in
    # single-pass dual-output call: ONE encoder evaluation feeds the two heads AND the attention layer.  The reference
    # (model/Model.py:505-509) tests node_adaptation_* first and would return the attention branch alone for this flag
    # combination -- which none of its callers passes; set `dual_output_on_both_flags = False` for that literal behaviour.
    dual_output_on_both_flags = True
    # The dual-output call applies the encoder's BatchNorm running-statistics update of the pass TWICE (default), so that the
    # buffers end where the reference's two calls leave them when the second call draws the first one's FPS starts: a
    # train-mode pass maps every running buffer r -> a r + c (a = (1 - momentum)^groups, c from the batch statistics), the
    # second call of the reference repeats exactly that map on the same batch, and a r1 + c = r1 + a (r1 - r0) needs only
    # the buffers before (r0) and after (r1) the one pass that is run.  num_batches_tracked likewise.  False: updated once.
    dual_updates_bn_twice = True

    def _bn_twice_begin(self):
        """Before the single encoder pass of a dual-output call: copies of every BatchNorm running buffer of the encoder and of
        the batch counters (multi-tensor launches, independent of the backbone)."""
        if not (self.dual_updates_bn_twice and self.training):
            return None
        # (rebuilt on every call: cheap next to an encoder pass, and module surgery after the first call -- convert_sync_batchnorm,
        # a replaced sub-module -- is then seen; ADVICE r5.  ASSUMPTION of the extrapolation below: every BatchNorm the pass
        # runs is run exactly ops.CTX.bn_groups times, or not at all.)
        allbn = [m for m in self.g.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.track_running_stats]
        bns = [m for m in allbn if m.momentum is not None]
        if len(bns) != len(allbn) and not getattr(self, '_bn_twice_warned', False):
            import warnings
            self._bn_twice_warned = True
            warnings.warn('Net_MDA dual-output forward: BatchNorm layers with momentum=None (cumulative average) receive the '
                          "pass's running-statistics update once, not twice as from the reference's two calls")
        # .data views: the second update must not bump the version counters autograd checks -- torch's own batch_norm node
        # keeps a reference to the running buffers it was given, for its eval-mode backward; a train-mode backward never reads
        # them, and the reference's second forward call changes them under the first call's graph as well
        bufs = [b.data for m in bns for b in (m.running_mean, m.running_var)]
        moms = [float(m.momentum) for m in bns for _ in (0, 1)]
        nbt = [m.num_batches_tracked.data for m in bns if m.num_batches_tracked is not None]
        if not bufs or not bufs[0].is_cuda:
            return None
        G = ops.CTX.bn_groups
        alphas = [(1.0 - m) ** G for m in moms]
        return bufs, alphas, torch._foreach_add(bufs, 0.0), nbt, (torch._foreach_add(nbt, 0) if nbt else [])

    @staticmethod
    def _bn_twice_end(snap):
        """r2 = a r1 + c with c = r1 - a r0, written as r1 + a (r1 - r0): a BatchNorm the pass did not run (r1 == r0) keeps
        its buffers bit for bit."""
        if snap is None:
            return
        bufs, alphas, r0, nbt, n0 = snap
        d = torch._foreach_sub(bufs, r0)
        torch._foreach_mul_(d, alphas)
        torch._foreach_add_(bufs, d)
        if nbt:
            torch._foreach_add_(nbt, torch._foreach_sub(nbt, n0))

    # Per-call hipGraph replay (sug_amd.call_graphs): a training loop that calls `model(...)` four times a
    # step, as train_dg_single_gpu.py:260-310 does, is host-bound when every kernel is launched from Python (~480 launches
    # in 8 ms against 6.8 ms of kernels).  With call_graphs on, the first train-mode call of a (flags, shape) combination runs
    # eagerly, the second is captured -- forward and backward, each as one hipGraph -- and later calls replay: same
    # kernels, same arithmetic, same CPU-generator draws (bit-identical losses and parameters, tests/test_gpu_call_graphs.py),
    # two launches per call.  'auto' (default): on for train-mode calls with autograd enabled on one process; False: off.
    call_graphs = 'auto'

    def forward(self, x, constant=1, adaptation=False, node_vis=False, mid_feat=False, node_adaptation_s=False,
                node_adaptation_t=False, semantic_adaption=False):
        _check_input(x)
        flags = (constant, bool(adaptation), bool(node_vis), bool(mid_feat), bool(node_adaptation_s), bool(node_adaptation_t),
                 bool(semantic_adaption))
        if self.call_graphs and self.training and torch.is_grad_enabled():
            from .. import call_graphs as _cg
            mgr = _cg.manager_for(self)
            if mgr is not None:
                return mgr.call(x, flags)
        return self._forward_impl(x, *flags)

    def _forward_impl(self, x, constant=1, adaptation=False, node_vis=False, mid_feat=False, node_adaptation_s=False,
                      node_adaptation_t=False, semantic_adaption=False):
        dual = self.dual_output_on_both_flags and semantic_adaption and (node_adaptation_s or node_adaptation_t) \
            and not (node_vis or mid_feat)
        only_node = (node_adaptation_s or node_adaptation_t) and not (node_vis or mid_feat or dual)
        only_feat = not (node_vis or mid_feat or node_adaptation_s or node_adaptation_t)
        if only_node:
            x, feat_ori, node_idx = self.g(x, node=True, feat_grad=False)       # the pooled feature is not used
        elif only_feat and self._g_skips_node():
            x, feat_ori, node_idx = self.g(x, node=True, need_node=False)       # the node features are not used
        elif dual:
            snap = self._bn_twice_begin()
            x, feat_ori, node_idx = self.g(x, node=True)
            self._bn_twice_end(snap)
        else:
            x, feat_ori, node_idx = self.g(x, node=True)
        batch_size = (feat_ori if feat_ori is not None else x).size(0)
        if node_vis:
            return node_idx
        if mid_feat:
            return x, feat_ori
        if dual:
            # what train_dg_single_gpu.py:260-264 + :309-310 obtain from TWO calls on the same batch.  In train mode the
            # second call recomputes the first one's encoder bit for bit when its FPS start draw is the same (BatchNorm
            # uses batch statistics, dropout sits in the heads only), so the values and -- one graph instead of two equal
            # ones -- the gradients are those of the two-call form with tied draws.  Documented difference: ONE FPS
            # start draw per sampling stage instead of two (the encoder's BatchNorm running statistics receive the pass's
            # update twice, as from the two calls: dual_updates_bn_twice).
            att = self.attention_s if node_adaptation_s else self.attention_t
            node = att(feat_ori.contiguous().view(batch_size, -1))
            if adaptation:
                x = grad_reverse(x, constant)
            (y1, sem_feature1), (y2, sem_feature2) = self._heads(x)
            return y1, y2, sem_feature1, sem_feature2, node
        if node_adaptation_s:
            return self.attention_s(feat_ori.contiguous().view(batch_size, -1))
        elif node_adaptation_t:
            return self.attention_t(feat_ori.contiguous().view(batch_size, -1))
        if adaptation:
            x = grad_reverse(x, constant)
        (y1, sem_feature1), (y2, sem_feature2) = self._heads(x)
        if not semantic_adaption:
            return y1, y2
        return y1, y2, sem_feature1, sem_feature2

    def _g_skips_node(self):
        """Does the encoder take need_node=False (PointNet++: the node features are a separate reduction there)?"""
        if not hasattr(self, '_skip_node_ok'):
            import inspect
            self._skip_node_ok = 'need_node' in inspect.signature(self.g.forward).parameters
        return self._skip_node_ok

    def _heads(self, x):
        """[(logits, mid feature)] of c1 and c2 on the pooled feature x: one launch per layer for both heads
        (ops.heads_fused, sug_head_linear_*) where the shapes allow, the module path otherwise."""
        if ops.heads_fused_supported((self.c1, self.c2), x):
            return ops.heads_fused((self.c1, self.c2), x)
        return ops.run_parallel([lambda: self.c1(x, adapt=True), lambda: self.c2(x, adapt=True)])

    def plan_pair_geometry(self, x_pair, passes=2):
        """Both passes of a step over the paired batch (semantic, then node adaptation): their sampling / grouping
        indices in one set of launches where the encoder supports it (Pointnet2_g.plan_geometry).  The plans are
        consumed, in order, by the next `passes` forward_pair calls on the same tensor."""
        self._geometry = None
        if hasattr(self.g, 'plan_geometry') and self.training:
            plans = self.g.plan_geometry(x_pair, passes, groups=2)
            if plans:
                import weakref
                self._geometry = (weakref.ref(x_pair), plans)

    def forward_pair(self, x_pair, node_adaptation=False, paired_out=False, dual=False):
        """Both domains in one encoder pass (not in the reference; used by SUGStep).
        x_pair = cat(source batch, target batch) [2B,3,N,1].  Equivalent to
        forward(source, ...) followed by forward(target, ...): every BatchNorm computes its
        statistics and updates its running buffers per domain half, source first
        (ops.bn_groups), everything else is per cloud / per row.
        semantic (default): ((y1,y2,f1,f2) of the source, (y1,y2,f1,f2) of the target);
        node_adaptation:    (attention_s(source nodes), attention_t(target nodes));
        paired_out (semantic): (y1, y2, f1, f2) as [2B, ...] tensors, source rows first (the caller scores / splits them
        itself: ops.ce_pair, ops.split_halves);
        dual (the single-pass step): ONE encoder evaluation feeds heads and attention layers -- returns the
        semantic result (in the form `paired_out` selects) followed by (attention_s(source nodes), attention_t(target
        nodes)); see forward() for what that changes against the two-pass form."""
        _check_input(x_pair)
        B2 = x_pair.size(0)
        assert B2 % 2 == 0
        B = B2 // 2
        queue = None
        geom = getattr(self, '_geometry', None)
        geometry = None
        if geom is not None:
            if geom[0]() is x_pair and geom[1]:
                geometry = geom[1].pop(0)           # this pass's indices were computed (and its starts drawn) up front
            if not geom[1] or geom[0]() is not x_pair:
                self._geometry = None
        plan = self.g.fps_plan(x_pair.size(2)) if hasattr(self.g, 'fps_plan') else [x_pair.size(2)]
        if geometry is None and (ops.CTX.start_provider is None or len(plan) > 1):
            # CPU-generator draws in the reference's order: all FPS calls of the source forward, then all of the target
            # forward.  Under a graph's start feeder the same order is recorded / replayed (one device slice per draw), so a
            # captured step and its eager twin see the same starts; with ONE FPS call per forward a single draw of 2B
            # starts is the same stream and needs no concatenation.
            draws = [[ops.draw_group_start(B, n) for n in plan] for _ in range(2)]
            queue = [torch.cat((draws[0][c], draws[1][c])) for c in range(len(plan))]
        keep_plan, ops.CTX.geometry_plan = ops.CTX.geometry_plan, (list(geometry) if geometry is not None else None)
        try:
            out = self._forward_pair(x_pair, node_adaptation, paired_out, queue, B, B2, dual)
            left = len(ops.CTX.geometry_plan or ())
        finally:
            ops.CTX.geometry_plan = keep_plan
        if left:        # a planned entry nobody popped = a sampling stage that drew its start a second time
            raise RuntimeError('geometry plan of this pass was not consumed by the encoder (%d entries left)' % left)
        return out

    def _forward_pair(self, x_pair, node_adaptation, paired_out, queue, B, B2, dual=False):
        snap = None
        if dual:
            with ops.bn_groups(2):
                snap = self._bn_twice_begin()
        with ops.bn_groups(2), ops.start_queue(queue), ops.deferred_bn_counts():
            if dual:
                x, feat_ori, _ = self.g(x_pair, node=True)                      # heads AND attention layers read this pass
            elif node_adaptation:
                x, feat_ori, _ = self.g(x_pair, node=True, feat_grad=False)     # only the node features are used
            elif self._g_skips_node():
                x, feat_ori, _ = self.g(x_pair, node=True, need_node=False)     # only the pooled feature is used
            else:
                x, feat_ori, _ = self.g(x_pair, node=True)
        if snap is not None:
            ops.flush_bn_counts()       # an enclosing deferred_bn_counts() block (SUGStep) still holds this pass's increments
        self._bn_twice_end(snap)        # (after this pass's counter increments have been applied)
        cuts = getattr(self, '_cuts', None)
        if cuts is not None:            # SUGStep's two-phase backward cuts the graph at the encoder's outputs
            if dual:
                cuts.extend((x, feat_ori))
            else:
                cuts.append(feat_ori if node_adaptation else x)
        halves = lambda t: ops.split_halves(t.reshape(B2, -1))      # backward: copy-free where the gradients are adjacent
        nodes = None
        if node_adaptation or dual:
            fo = feat_ori.contiguous().reshape(B2, -1)
            if ops.calayer_supported((self.attention_s, self.attention_t), fo):
                # both attention layers in one launch per stage, on the paired rows (source rows -> attention_s)
                nodes = tuple(halves(ops.calayers((self.attention_s, self.attention_t), fo)))
            else:
                f_s, f_t = halves(fo)
                nodes = tuple(ops.run_parallel([lambda: self.attention_s(f_s), lambda: self.attention_t(f_t)]))
            if not dual:
                return nodes
        (y1, f1), (y2, f2) = self._heads(x)
        if paired_out:
            sem = (y1, y2, f1, f2)
        else:
            (y1s, y1t), (y2s, y2t), (f1s, f1t), (f2s, f2t) = halves(y1), halves(y2), halves(f1), halves(f2)
            sem = ((y1s, y2s, f1s, f2s), (y1t, y2t, f1t, f2t))
        return sem + (nodes,) if dual else sem

"""Point-set operators in the [B,N,C] layout and the PointNet++ set-abstraction module
(mirror of the reference's model/pointnet2_utils.py:19-207), over the HIP kernels."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .model_utils import _bn_rows


def square_distance(src, dst):
    """model/pointnet2_utils.py:19-38 ([B,N,C] x [B,M,C] -> [B,N,M]); API parity only."""
    B, N, _ = src.shape
    M = dst.shape[1]
    d = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    d = d + torch.sum(src ** 2, -1).view(B, N, 1)
    d = d + torch.sum(dst ** 2, -1).view(B, 1, M)
    return d


def index_points(points, idx):
    """model/pointnet2_utils.py:41-57. points [B,N,C], idx [B,S(,K)] -> [B,S(,K),C]."""
    return ops.gather_rows(points, idx)


def farthest_point_sample(xyz, npoint):
    """model/pointnet2_utils.py:60-81. xyz [B,N,3] -> [B,npoint] int64 (CPU-generator start, :72)."""
    B, N, _ = xyz.shape
    start = ops.draw_start(B, N)
    return ops.fps(xyz, npoint, start).long()


def query_ball_point(radius, nsample, xyz, new_xyz):
    """model/pointnet2_utils.py:84-104. xyz [B,N,3], new_xyz [B,S,3] -> [B,S,nsample] int64."""
    return ops.ball_query(xyz, new_xyz, radius, nsample).long()


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    """model/pointnet2_utils.py:107-135 -> new_xyz [B,S,3], new_points [B,S,nsample,3+D]."""
    B, N, C = xyz.shape
    start = ops.draw_start(B, N)
    fps_idx = ops.fps(xyz, npoint, start)
    new_xyz = ops.gather_rows(xyz, fps_idx)
    idx = ops.ball_query(xyz, new_xyz, radius, nsample)
    grouped_xyz = ops.gather_rows(xyz, idx)
    new_points = grouped_xyz - new_xyz.view(B, npoint, 1, C)
    if points is not None:
        new_points = torch.cat([new_points, ops.gather_rows(points, idx)], dim=-1)
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx.long()
    return new_xyz, new_points


def sample_and_group_idx(npoint, radius, nsample, xyz):
    """The index half of sample_and_group (model/pointnet2_utils.py:107-124): same FPS start draw and kernels,
    -> new_xyz [B,S,3], idx [B,S,nsample] int32; the grouped tensor itself is not formed."""
    B, N, _ = xyz.shape
    plan = ops.CTX.geometry_plan
    if plan:
        new_xyz, idx = plan.pop(0)              # computed up front for this pass (Pointnet2_g.plan_geometry)
        if new_xyz.shape != (B, npoint, 3) or idx.shape != (B, npoint, nsample):
            raise RuntimeError('geometry plan does not match the encoder: %s / %s for (%d, %d, %d)' % (
                tuple(new_xyz.shape), tuple(idx.shape), B, npoint, nsample))
        return new_xyz, idx
    start = ops.draw_start(B, N)
    fps_idx = ops.fps(xyz, npoint, start)
    new_xyz = ops.gather_rows(xyz, fps_idx)
    return new_xyz, ops.ball_query(xyz, new_xyz, radius, nsample)


def sample_and_group_all(xyz, points):
    """model/pointnet2_utils.py:138-155."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped = xyz.view(B, 1, N, C)
    if points is not None:
        grouped = torch.cat([grouped, points.view(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped


class PointNetSetAbstraction(nn.Module):
    """model/pointnet2_utils.py:158-207: FPS -> ball query -> group -> 1x1 conv/BN/ReLU stack ->
    max over the group.  Parameters are named as in the reference (mlp_convs.i, mlp_bns.i)."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, adapt=False):
        super(PointNetSetAbstraction, self).__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        self.adapt = adapt
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last = out_channel
        self.group_all = group_all

    def takes_index_path(self):
        """True if rows() runs its first MLP layer on the neighbour lists (sample_and_group_idx -- the only consumer of an
        ops.CTX.geometry_plan) instead of forming the grouped tensor."""
        return (not self.group_all) and len(self.mlp_convs) > 1 and \
            ops.sa_first_layer_supported(self.mlp_convs[0].out_channels)

    def rows(self, xyz, points, adapt=False, tail_grad=True):
        """xyz [B,N,3], points [B,N,D] or None -> new_xyz [B,S,3], feats [B,S,D'](, node [B,S,D1]).
        tail_grad=False (with adapt): the layers behind the one the node features come from run without
        autograd (their output is discarded by the caller; BatchNorm buffers are still updated)."""
        first = 0
        if self.group_all:
            new_xyz, g = sample_and_group_all(xyz, points)
        elif self.takes_index_path():
            # first layer on the neighbour lists: W.[x_j - c_s ; f_j] + b = P[j] - Q[s], P per point, Q per centroid
            # (the grouped [B,S,ns,3+D] tensor and the pre-activation tensor are never formed)
            new_xyz, idx = sample_and_group_idx(self.npoint, self.radius, self.nsample, xyz)
            conv = self.mlp_convs[0]
            w = conv.weight.view(conv.weight.shape[0], -1)
            if ops.SA_FIRST_GEO and xyz.shape[-1] == 3:
                # coordinate part from the difference the reference forms: y = Pf[j] + b + Wx.(x_j - c_s) (sug_sa_first_geo_*);
                # Px and Q only route the gradients of Wx and b (their values are not read by the kernel)
                wx = w[:, :3].contiguous()
                Px = ops.linear_rows(xyz, wx)
                Pf = None if points is None else ops.linear_rows(points, w[:, 3:].contiguous())
                Q = ops.sub_row_bias(ops.linear_rows(new_xyz, wx), conv.bias)
                g = ops.sa_first_layer_geo(Pf, Px, Q, idx, xyz, new_xyz, wx, conv.bias, self.mlp_bns[0])
            else:
                P = ops.linear_rows(xyz if points is None else torch.cat((xyz, points), dim=-1), w)
                Q = ops.sub_row_bias(ops.linear_rows(new_xyz, w[:, :3]), conv.bias)
                g = ops.sa_first_layer(P, Q, idx, self.mlp_bns[0])
            first = 1
        else:
            new_xyz, g = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points)
        node = None
        last = len(self.mlp_convs) - 1
        out = None
        for i, conv in enumerate(self.mlp_convs):                      # g: [B,S,ns,C] rows
            if i < first:
                continue
            w = conv.weight.view(conv.weight.shape[0], -1)
            if i == last - 1 and ops.SA_MID_FUSED and g.is_cuda and self.mlp_bns[i].training == self.mlp_bns[last].training and \
                    ops.pointmlp_max_supported(w.shape[0], self.mlp_convs[last].out_channels, g.shape[2]):
                # middle layer + last layer: the middle layer's BatchNorm + ReLU is applied inside the fused last-layer
                # kernel (ops.bn_act_pointmlp_max): no separate pass over [B,S,ns,C], z written once and only if needed
                tail_on = tail_grad or not adapt           # (node pass: the last layer runs for its BatchNorm buffers only)
                y1 = ops.linear_rows(g, w, conv.bias)
                cl = self.mlp_convs[last]
                out, z = ops.bn_act_pointmlp_max(y1, self.mlp_bns[i], 0.0, cl.weight.view(cl.weight.shape[0], -1), cl.bias,
                                                 self.mlp_bns[last], 0.0, g.shape[2], want_z=(adapt and i == 1),
                                                 last_grad=tail_on)
                out = out.view(g.shape[0], g.shape[1], -1)
                if adapt and i == 1:
                    node = z
                break
            with torch.set_grad_enabled(torch.is_grad_enabled() and (tail_grad or not adapt or i <= 1)):
                if i == last and ops.pointmlp_max_supported(g.shape[-1], w.shape[0], g.shape[2]):
                    # last layer + max over the group in one kernel: [B,S,ns,C'] is never written
                    out = ops.pointmlp_max(g, w, conv.bias, self.mlp_bns[i], 0.0, g.shape[2]).view(g.shape[0], g.shape[1], -1)
                    break
                g = ops.bn_act_rows(ops.linear_rows(g, w, conv.bias), self.mlp_bns[i], 0.0)
            if adapt and i == 1:
                node = g
        if out is None:
            out = torch.max(g, dim=2)[0]
        if adapt:
            return new_xyz, out, torch.max(node, dim=2)[0]
        return new_xyz, out

    def forward(self, xyz, points, adapt=False):
        """Reference layout: xyz [B,3,N], points [B,D,N] -> new_xyz [B,3,S], new_points [B,D',S]."""
        r = self.rows(xyz.permute(0, 2, 1).contiguous(),
                      None if points is None else points.permute(0, 2, 1).contiguous(), adapt)
        return tuple(t.permute(0, 2, 1) for t in r)

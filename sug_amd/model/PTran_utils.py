"""Point Transformer set-abstraction utilities (mirror of the reference's model/PTran_utils.py,
rows layout [B,N,C] as there).  Distances are the DIRECT form sum((a - b)^2) and neighbours are
`argsort()[:, :, :k]` (PTran_utils.py:22-36, :117-119): sug_knn_query_direct."""
import torch
import torch.nn as nn

from .. import ops


def square_distance(src, dst):
    """PTran_utils.py:22-36 (API-parity helper; the encoders never build the matrix)."""
    return torch.sum((src[:, :, None] - dst[:, None]) ** 2, dim=-1)


def index_points(points, idx):
    """PTran_utils.py:39-50. points [B,N,C], idx [B,S(,K)] -> [B,S(,K),C]."""
    return ops.gather_rows(points, idx)


def farthest_point_sample(xyz, npoint):
    """PTran_utils.py:53-73. xyz [B,N,3] -> [B,npoint] int64; start index from the CPU generator."""
    B, N, _ = xyz.shape
    return ops.fps(xyz, npoint, ops.draw_start(B, N)).long()


def knn_point(nsample, xyz, new_xyz):
    """`square_distance(new_xyz, xyz).argsort()[:, :, :nsample]` (PTran_utils.py:117-119)."""
    return ops.knn_query(xyz, new_xyz, min(nsample, xyz.shape[1]), direct=True)


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False, knn=False):
    """PTran_utils.py:99-136. -> new_xyz [B,S,3], new_points [B,S,ns,3+D]."""
    B, N, C = xyz.shape
    fps_idx = ops.fps(xyz, npoint, ops.draw_start(B, N))
    new_xyz = ops.gather_rows(xyz, fps_idx)
    if knn:
        idx = knn_point(nsample, xyz, new_xyz)
    else:
        idx = ops.ball_query(xyz, new_xyz, radius, nsample)
    grouped_xyz = ops.gather_rows(xyz, idx)
    grouped_xyz_norm = grouped_xyz - new_xyz.view(B, npoint, 1, C)
    if points is not None:
        new_points = torch.cat([grouped_xyz_norm, ops.gather_rows(points, idx)], dim=-1)
    else:
        new_points = grouped_xyz_norm
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx.long()
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """PTran_utils.py:139-156."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped_xyz = xyz.view(B, 1, N, C)
    if points is not None:
        return new_xyz, torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped_xyz


class PointNetSetAbstraction(nn.Module):
    """PTran_utils.py:158-199 (parameters named as there): group -> (1x1 conv, BN, ReLU)* -> max
    over the group; rows in, rows out."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, knn=False):
        super(PointNetSetAbstraction, self).__init__()
        self.npoint, self.radius, self.nsample, self.knn = npoint, radius, nsample, knn
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last = out_channel
        self.group_all = group_all

    def forward(self, xyz, points):
        """xyz [B,N,3], points [B,N,D] -> new_xyz [B,S,3], new_points [B,S,D']."""
        if self.group_all:
            new_xyz, g = sample_and_group_all(xyz, points)
        else:
            new_xyz, g = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points, knn=self.knn)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):             # g [B,S,ns,C] rows
            g = ops.bn_act_rows(ops.linear_rows(g, conv.weight.view(conv.weight.shape[0], -1), conv.bias), bn, 0.0)
        return new_xyz, torch.max(g, dim=2)[0]

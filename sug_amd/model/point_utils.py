"""Point-set operators in the reference's [B,C,N] layout (model/point_utils.py), backed by
the HIP kernels of libsug_amd.so.  Index results are int64 like the reference's."""
import torch

from .. import ops


def _rows(t):
    """[B,C,N] or [B,C,N,1] -> [B,N,C] rows."""
    if t.dim() == 4:
        t = t.squeeze(-1)
    return t.transpose(1, 2).contiguous()


def farthest_point_sample(xyz, npoint):
    """model/point_utils.py:5-26. xyz [B,3,N] -> [B,npoint] int64.  The first centroid is
    drawn from the CPU default generator, one draw per call, like the reference (:17)."""
    B, _, N = xyz.shape
    start = ops.draw_start(B, N)
    return ops.fps(_rows(xyz), npoint, start).long()


def index_points(points, idx):
    """model/point_utils.py:60-83. points [B,C,N](,1), idx [B,S(,K)] -> [B,C,S(,K)]."""
    g = ops.gather_rows(_rows(points), idx)
    return g.permute(0, 2, 1) if g.dim() == 3 else g.permute(0, 3, 1, 2)


def square_distance(src, dst):
    """model/point_utils.py:112-131 ([B,C,N] x [B,C,M] -> [B,N,M]); kept for API parity, the
    kernels never materialise this matrix."""
    B, _, N = src.shape
    M = dst.shape[2]
    d = -2 * torch.matmul(src.permute(0, 2, 1), dst)
    d = d + torch.sum(src ** 2, 1).view(B, N, 1)
    d = d + torch.sum(dst ** 2, 1).view(B, 1, M)
    return d


def query_ball_point(radius, nsample, xyz, new_xyz):
    """model/point_utils.py:86-109. xyz [B,3,N], new_xyz [B,3,S] -> [B,S,nsample] int64."""
    if radius is None:
        return ops.knn_query(_rows(xyz), _rows(new_xyz), nsample).long()
    return ops.ball_query(_rows(xyz), _rows(new_xyz), radius, nsample).long()


def interpolate_rows(xyz_rows, node_rows, node_fea_rows, k=3):
    """Inverse-distance k=3 interpolation of node features onto the dense cloud
    (the second half of upsample_inter, model/point_utils.py:153-160), rows layout."""
    assert k == 3, 'the 3-NN kernel implements the only k the reference uses'
    idx3, d3 = ops.three_nn(xyz_rows, node_rows)
    d3 = torch.where(d3 < 1e-10, torch.full_like(d3, 1e-10), d3)
    w = 1.0 / d3
    w = w / torch.sum(w, dim=-1, keepdim=True)
    sel = ops.gather_rows(node_fea_rows, idx3)                 # [B,N,3,D]
    return torch.sum(sel * w.unsqueeze(-1), dim=2)


def upsample_inter(xyz1, xyz2, points1, points2, k):
    """model/point_utils.py:134-165. -> [B, D1+D2, N] (or [B,D2,N] when points1 is None)."""
    interp = interpolate_rows(_rows(xyz1), _rows(xyz2), _rows(points2), k).transpose(1, 2)
    if points1 is None:
        return interp
    if points1.dim() == 4:
        points1 = points1.squeeze(-1)
    return torch.cat([points1, interp], dim=1)

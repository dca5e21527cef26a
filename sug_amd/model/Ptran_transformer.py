"""Point Transformer block (mirror of the reference's model/Ptran_transformer.py): kNN by
direct-form distance (sug_knn_query_direct), neighbour gathers (sug_gather_rows), vector
self-attention.  The 512-wide per-neighbour linears are library GEMMs on [B*n*k, d] rows; their
weight gradients go through sug_linear_dw."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def _lin(layer, x):
    return ops.linear_rows(x, layer.weight, layer.bias)


class TransformerBlock(nn.Module):
    def __init__(self, d_points, d_model, k) -> None:
        super().__init__()
        self.fc1 = nn.Linear(d_points, d_model)
        self.fc2 = nn.Linear(d_model, d_points)
        self.fc_delta = nn.Sequential(nn.Linear(3, d_model), nn.ReLU(), nn.Linear(d_model, d_model))
        self.fc_gamma = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, d_model))
        self.w_qs = nn.Linear(d_model, d_model, bias=False)
        self.w_ks = nn.Linear(d_model, d_model, bias=False)
        self.w_vs = nn.Linear(d_model, d_model, bias=False)
        self.k = k

    def forward(self, xyz, features):
        """xyz [B,n,3], features [B,n,f] -> (res [B,n,f], attn [B,n,k,d]); Ptran_transformer.py:31-45.
        `argsort()[:, :, :k]` of n < k columns yields n neighbours: k_eff = min(k, n)."""
        xyz = xyz.contiguous()
        knn_idx = ops.knn_query(xyz, xyz, min(self.k, xyz.shape[1]), direct=True)      # [B,n,k]
        knn_xyz = ops.gather_rows(xyz, knn_idx)
        pre = features
        x = _lin(self.fc1, features)
        q = _lin(self.w_qs, x)
        k = ops.gather_rows(_lin(self.w_ks, x), knn_idx)
        v = ops.gather_rows(_lin(self.w_vs, x), knn_idx)
        pos_enc = _lin(self.fc_delta[2], F.relu(_lin(self.fc_delta[0], xyz[:, :, None] - knn_xyz)))    # [B,n,k,d]
        attn = _lin(self.fc_gamma[2], F.relu(_lin(self.fc_gamma[0], q[:, :, None] - k + pos_enc)))
        attn = F.softmax(attn / np.sqrt(k.size(-1)), dim=-2)
        res = torch.einsum('bmnf,bmnf->bmf', attn, v + pos_enc)
        res = _lin(self.fc2, res) + pre
        return res, attn

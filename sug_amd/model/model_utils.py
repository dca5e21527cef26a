"""Building blocks of the encoders (mirror of the reference's model/model_utils.py).

Module / parameter names match the reference so a reference ``state_dict`` loads
unchanged.  Every block has a ``forward`` in the reference's [B,C,N,1] layout and a
``rows`` fast path on point-major [B,N,C] rows, which is what the encoders in
``Model.py`` use (data layout).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from . import point_utils

_ACT_SLOPE = {'relu': 0.0, 'leakyrelu': 0.01}      # nn.LeakyReLU() default slope, model_utils.py:27


def OWN_BN(bn):
    """Debug hook: which BatchNorm layers use the library's BN kernels (default: all)."""
    return True


def bn_module(bn, y):
    """Apply a torch BatchNorm module per domain group (ops.bn_groups), i.e. as the separate
    forward calls of the reference would."""
    if ops.CTX.bn_groups == 1:
        return bn(y)
    return torch.cat([bn(c) for c in y.chunk(ops.CTX.bn_groups, dim=0)], dim=0)


def _bn_rows(bn, y):
    """Train/eval BatchNorm of a [..., C] rows tensor with the module's parameters."""
    if ops.CTX.bn_groups > 1:
        G = ops.CTX.bn_groups
        with ops.bn_groups(1):
            return torch.cat([_bn_rows(bn, c) for c in y.chunk(G, dim=0)], dim=0)
    shp = y.shape
    y2 = y.reshape(-1, shp[-1])
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    y2 = F.batch_norm(y2, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                      bn.training or not bn.track_running_stats, bn.momentum, bn.eps)
    return y2.view(shp)


def edge_wcats(layers):
    """The EdgeConv GEMM operands [W1 ; W2-W1] of conv_2d `layers`: a layer's cached split where it keeps one for the
    step (cache_weight_split: same weights, same step -- a graph-attached copy also serves no_grad), all the others in
    ONE launch forward and one backward (ops.edge_weight_split_multi) instead of one per layer."""
    grad = torch.is_grad_enabled()
    out, miss = [None] * len(layers), []
    for i, l in enumerate(layers):
        W = l.weight2d()
        hit = l._wcat if l.cache_weight_split else None
        if hit is not None and hit[0] == W._version and (hit[1] or not grad):
            out[i] = hit[2]
        else:
            miss.append((i, l, W))
    if miss:
        for (i, l, W), wc in zip(miss, ops.edge_weight_split_multi([W for _, _, W in miss])):
            out[i] = wc
            l._wcat = (W._version, grad, wc) if l.cache_weight_split else None
    return out


class conv_2d(nn.Module):
    """1x1 Conv2d -> BatchNorm2d -> activation (model/model_utils.py:8-32)."""

    def __init__(self, in_ch, out_ch, kernel, activation='relu', bias=True):
        super(conv_2d, self).__init__()
        assert kernel == 1 or kernel == (1, 1), 'the hot path only has 1x1 convolutions'
        act = {'relu': nn.ReLU(inplace=False), 'tanh': nn.Tanh(), 'leakyrelu': nn.LeakyReLU()}[activation]
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, kernel_size=kernel, bias=bias),
                                  nn.BatchNorm2d(out_ch), act)
        self.activation = activation
        # opt-in (SUGStep): reuse the [W1 ; W2-W1] split across the forwards of one step
        # (one backward over all of them); cleared by the owner after backward
        self.cache_weight_split = False
        self._wcat = None

    def weight2d(self):
        w = self.conv[0].weight
        return w.view(w.shape[0], w.shape[1])

    def rows(self, x):
        """x [..., Cin] -> [..., Cout]: per-point GEMM + BN over all leading dims + act."""
        y = ops.linear_rows(x, self.weight2d(), self.conv[0].bias)
        if self.activation in _ACT_SLOPE and OWN_BN(self.conv[1]):
            return ops.bn_act_rows(y, self.conv[1], _ACT_SLOPE[self.activation])
        return self.conv[2](_bn_rows(self.conv[1], y))

    def forward(self, x):
        return self.rows(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)

    def rows_max_after(self, prev, x):
        """self.rows_max(prev.rows(x)) for two conv_2d layers in a row (Model.py:272-279, model_utils.py:70-79,
        model_pointnet.py:44-48): `prev`'s BatchNorm + activation is applied inside this layer's fused kernel
        (ops.bn_act_pointmlp_max) -- no separate pass over prev's [B,N,C] output."""
        B, N, _ = x.shape
        Wp, W = prev.weight2d(), self.weight2d()
        if ops.SA_MID_FUSED and x.is_cuda and self.activation in _ACT_SLOPE and prev.activation in _ACT_SLOPE and \
                OWN_BN(self.conv[1]) and OWN_BN(prev.conv[1]) and prev.conv[1].training == self.conv[1].training and \
                ops.pointmlp_max_supported(Wp.shape[0], W.shape[0], N):
            y = ops.linear_rows(x, Wp, prev.conv[0].bias)
            return ops.bn_act_pointmlp_max(y, prev.conv[1], _ACT_SLOPE[prev.activation], W, self.conv[0].bias, self.conv[1],
                                           _ACT_SLOPE[self.activation], N)[0]
        return self.rows_max(prev.rows(x))

    def rows_max(self, x):
        """x [B,N,Cin] -> [B,Cout] = max over the N points of act(bn(conv(x))): the layer and the
        reduction in one kernel, the [B,N,Cout] tensor is never written (sug_pointmlp_max_*;
        Model.py:274-279, model_utils.py:72-79)."""
        B, N, C = x.shape
        W = self.weight2d()
        if self.activation in _ACT_SLOPE and OWN_BN(self.conv[1]) and ops.pointmlp_max_supported(C, W.shape[0], N):
            return ops.pointmlp_max(x, W, self.conv[0].bias, self.conv[1], _ACT_SLOPE[self.activation], N)
        return torch.max(self.rows(x), dim=1)[0]

    def edge_rows(self, x, idx, return_stats=False, out=None, wcat=None):
        """Fused EdgeConv layer: max_k act(bn(W.[x_j - x_i ; x_i])) for x [B,N,C] rows and
        idx [B,N,k] (get_graph_feature + conv + max, model_utils.py:188-210, Model.py:88-94).
        W.[x_j-x_i; x_i] = W1.x_j + (W2-W1).x_i, so one [B*N,C]x[C,2Co] GEMM replaces the
        k-fold one and the k-expanded tensor is never built (sug_edgeconv_fwd)."""
        assert self.activation in _ACT_SLOPE
        B, N, C = x.shape
        W = self.weight2d()
        assert W.shape[1] == 2 * C
        Wcat = wcat if wcat is not None else edge_wcats((self,))[0]       # [2Co, C] = [W1 ; W2-W1]
        bias = self.conv[0].bias
        bn = self.conv[1]
        ops._count_bn_call(bn)
        if ops.edgeconv_fused_supported(N, idx.shape[2], C, W.shape[0]):
            # the GEMM inside the gather kernel: [P|Q] never round-trips HBM (sug_edgeconv_fused_layer_fwd)
            out, coef = ops.edgeconv_fused(x, Wcat, bias, idx, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                           bn.training, _ACT_SLOPE[self.activation], bn.eps, bn.momentum, out=out)
            return (out, coef) if return_stats else out
        pq = ops.linear_rows(x.reshape(B * N, C), Wcat)
        if bias is not None:                                              # bias rides on the Q half
            pq = pq + torch.cat((torch.zeros_like(bias), bias))
        out, coef = ops.edgeconv_bn_act_max(pq.view(B, N, -1), idx, bn.weight, bn.bias, bn.running_mean,
                                            bn.running_var, bn.training, _ACT_SLOPE[self.activation],
                                            bn.eps, bn.momentum, out=out)
        return (out, coef) if return_stats else out

    def replay_bn_update(self, coef):
        """Apply the running-statistics update of one more train-mode forward on the same batch
        (coef from edge_rows(..., return_stats=True)): what nn.BatchNorm2d would do again."""
        ops.bn_replay(self.conv[1], coef)


class fc_layer(nn.Module):
    """Linear -> LayerNorm -> activation (model/model_utils.py:35-57; leaky slope 0.2)."""

    def __init__(self, in_ch, out_ch, bn=True, activation='leakyrelu', bias=False):
        super(fc_layer, self).__init__()
        self.ac = nn.ReLU(inplace=False) if activation == 'relu' else nn.LeakyReLU(negative_slope=0.2, inplace=False)
        layers = [nn.Linear(in_ch, out_ch, bias=bias)]
        if bn:
            layers.append(nn.LayerNorm(out_ch))
        layers.append(self.ac)
        self.fc = nn.Sequential(*layers)

    def forward(self, x):
        if ops.CTX.fused_heads and len(self.fc) == 3 and x.is_cuda and ops.ln_act_supported(x, self.fc[1]):
            # LayerNorm + activation in one launch (two in the backward) instead of 2 + 4 (graph replay only: ops.CTX.fused_heads)
            slope = 0.0 if isinstance(self.ac, nn.ReLU) else self.ac.negative_slope
            return ops.ln_act(self.fc[0](x), self.fc[1], slope)
        return self.fc(x)


class transform_net(nn.Module):
    """T-Net (model/model_utils.py:60-89): per-point MLP, max over points, FC head, + identity."""

    def __init__(self, in_ch, K=3):
        super(transform_net, self).__init__()
        self.K = K
        self.conv2d1 = conv_2d(in_ch, 64, 1)
        self.conv2d2 = conv_2d(64, 128, 1)
        self.conv2d3 = conv_2d(128, 1024, 1)
        self.maxpool1 = nn.MaxPool2d(kernel_size=(512, 1))
        self.fc1 = fc_layer(1024, 512)
        self.fc2 = fc_layer(512, 256)
        self.fc3 = nn.Linear(256, K * K)

    def rows(self, x):
        """x [B,N,C] -> [B,K,K]."""
        y = self.conv2d3.rows_max_after(self.conv2d2, self.conv2d1.rows(x))
        y = self.fc3(self.fc2(self.fc1(y)))
        y = y + torch.eye(self.K, device=y.device, dtype=y.dtype).view(1, self.K * self.K)
        return y.view(-1, self.K, self.K)

    def forward(self, x, DGCNN_Flag=False):
        if DGCNN_Flag:      # [B,C,N,k]: max over k after conv2d2 (model_utils.py:75-77)
            y = self.conv2d2(self.conv2d1(x)).max(dim=-1, keepdim=True)[0]
            return self.rows_tail(y.squeeze(-1).transpose(1, 2))
        return self.rows(x.squeeze(-1).transpose(1, 2))

    def rows_tail(self, y):
        y = self.conv2d3.rows_max(y)
        y = self.fc3(self.fc2(self.fc1(y)))
        y = y + torch.eye(self.K, device=y.device, dtype=y.dtype).view(1, self.K * self.K)
        return y.view(-1, self.K, self.K)


class adapt_layer_off(nn.Module):
    """SA-node module (model/model_utils.py:92-128): FPS(64) -> ball(0.3,64) -> learned node
    offset -> 64-NN of the moved nodes -> residual conv + max -> 3-NN interpolation back."""

    def __init__(self, num_node=64, offset_dim=3, trans_dim_in=64, trans_dim_out=64, fc_dim=64):
        super(adapt_layer_off, self).__init__()
        self.num_node = num_node
        self.offset_dim = offset_dim
        self.trans = conv_2d(trans_dim_in, trans_dim_out, 1)          # unused in forward, as in the reference
        self.pred_offset = nn.Sequential(
            nn.Conv2d(trans_dim_out, offset_dim, kernel_size=1, bias=False),
            nn.Tanh())
        self.residual = conv_2d(trans_dim_in, fc_dim, 1)

    def rows(self, fea, loc):
        """fea [B,N,64], loc [B,N,3] -> (out [B,N,128], node_fea [B,num_node,64], node_off [B,num_node,3])."""
        B, N, _ = loc.shape
        S = self.num_node
        plan = ops.CTX.geometry_plan
        if plan:
            fidx, f_loc, gidx = plan.pop(0)    # computed up front for this pass (plan_geometry below)
            if fidx.shape != (B, S) or gidx.shape[:2] != (B, S):
                raise RuntimeError('geometry plan does not match the SA-node module')
        else:
            start = ops.draw_start(B, N)           # CPU generator, point_utils.py:17
            fidx = ops.fps(loc, S, start)                                 # [B,S]
            f_loc = ops.gather_rows(loc, fidx)                            # [B,S,3]
            gidx = ops.ball_query(loc, f_loc, 0.3, 64)                    # [B,S,64]
        # pred_offset on (fea_j - fea_c) is linear before the tanh: project once (one GEMM), then
        # gather / tanh / weight / mean in one kernel (sug_node_offset_*)
        w_off = self.pred_offset[0].weight.view(self.offset_dim, -1)
        proj = ops.linear_rows(fea, w_off)                            # [B,N,3]
        if self.offset_dim == 3 and gidx.shape[2] == S:
            node_off, n_loc = ops.node_offset(proj, loc, fidx, gidx)  # [B,S,3] each
        else:
            sem = torch.tanh(ops.gather_rows(proj, gidx) - ops.gather_rows(proj, fidx).unsqueeze(2))
            g_loc = ops.gather_rows(loc, gidx) - f_loc.unsqueeze(2)
            node_off = (sem * g_loc).mean(dim=2)
            n_loc = f_loc + node_off
        gidx2 = ops.knn_query(loc, n_loc, 64)                         # 64-NN of the moved nodes
        res = self.residual.rows(fea)
        node_fea = ops.group_max(res, gidx2)                          # [B,S,64]
        if fea.shape[2] % 4 == 0 and node_fea.shape[2] % 4 == 0:
            out = ops.interp3_cat(fea, node_fea, loc, n_loc)          # cat(fea, 3-NN interpolation)
        else:
            out = torch.cat((fea, point_utils.interpolate_rows(loc, n_loc, node_fea, 3)), dim=2)
        return out, node_fea, node_off

    def plan_geometry(self, loc, passes, groups=1):
        """The coordinate-only part of `passes` forwards over the same clouds loc [B,N,3] -- FPS start draws (in the order
        the forwards would make them), FPS, the sampled coordinates, the radius-0.3 ball query -- in one set of launches
        (SUGStep: the semantic and the node pass of a step).  -> per pass [(fidx, f_loc, gidx)] for ops.CTX.geometry_plan."""
        B, N, _ = loc.shape
        S = self.num_node
        starts = []
        for _ in range(passes):
            if ops.CTX.start_provider is not None or groups == 1:
                starts.append(ops.draw_start(B, N))
            else:
                starts.append(torch.cat([torch.randint(0, N, (B // groups,), dtype=torch.long) for _ in range(groups)]))
        dev = loc.device
        st = torch.cat([t.to(device=dev, dtype=torch.int32, non_blocking=True) for t in starts])
        locp = loc.repeat(passes, 1, 1) if passes > 1 else loc
        fidx = ops.fps(locp, S, st)
        f_loc = ops.gather_rows(locp, fidx)
        gidx = ops.ball_query(locp, f_loc, 0.3, 64)
        return [[(fidx[p * B:(p + 1) * B], f_loc[p * B:(p + 1) * B], gidx[p * B:(p + 1) * B])] for p in range(passes)]

    def forward(self, input_fea, input_loc):
        """input_fea [B,64,N,1], input_loc [B,3,N] -> (output_fea [B,128,N,1], node_fea [B,64,S,1],
        node_offset [B,3,S])."""
        out, node_fea, node_off = self.rows(input_fea.squeeze(-1).transpose(1, 2).contiguous(),
                                            input_loc.transpose(1, 2).contiguous())
        return (out.transpose(1, 2).unsqueeze(3), node_fea.transpose(1, 2).unsqueeze(3),
                node_off.transpose(1, 2))


class focal_loss(nn.Module):
    """-alpha_y (1-p_y)^gamma log p_y (model/model_utils.py:131-176)."""

    def __init__(self, alpha=None, gamma=2, num_classes=3, size_average=True):
        super(focal_loss, self).__init__()
        self.size_average = size_average
        if isinstance(alpha, list):
            assert len(alpha) == num_classes
            self.alpha = torch.Tensor(alpha)
        else:
            self.alpha = torch.Tensor([1 / num_classes] * num_classes)
        self.gamma = gamma

    def forward(self, preds, labels):
        preds = preds.view(-1, preds.size(-1))
        logp = F.log_softmax(preds, dim=1).gather(1, labels.view(-1, 1)).view(-1)
        alpha = self.alpha.to(preds.device).gather(0, labels.view(-1))
        loss = alpha * (-(1 - torch.exp(logp)) ** self.gamma * logp)
        return loss.mean() if self.size_average else loss.sum()


def knn(x, k):
    """model/model_utils.py:178-185. x [B,C,N] -> idx [B,N,k] int64; never builds [B,N,N]."""
    return ops.knn(x.transpose(1, 2), k).long()


def get_graph_feature(x, k=20, idx=None):
    """model/model_utils.py:188-210. x [B,C,N(,1)] -> [B,2C,N,k] = cat(x_j - x_i, x_i).
    API-parity helper: the encoders use conv_2d.edge_rows, which never builds this tensor."""
    B, N = x.size(0), x.size(2)
    rows = x.reshape(B, -1, N).transpose(1, 2).contiguous()
    if idx is None:
        idx = ops.knn(rows, k)
    nbr = ops.gather_rows(rows, idx)                                  # [B,N,k,C]
    ctr = rows.unsqueeze(2).expand_as(nbr)
    return torch.cat((nbr - ctr, ctr), dim=3).permute(0, 3, 1, 2)

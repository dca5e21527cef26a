"""Source-only classifiers used by train_source.py (mirror of model/model_pointnet.py:5-161; train_source.py:5,76-77
imports Pointnet_cls, Pointnet2_cls and DGCNN from here)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .Model import Pointnet_c
from .model_utils import conv_2d, fc_layer, transform_net
from .pointnet2_utils import PointNetSetAbstraction


class Pointnet_cls(nn.Module):
    """model/model_pointnet.py:5-55."""

    def __init__(self, num_class=10):
        super(Pointnet_cls, self).__init__()
        self.trans_net1 = transform_net(3, 3)
        self.trans_net2 = transform_net(64, 64)
        self.conv1 = conv_2d(3, 64, 1)
        self.conv2 = conv_2d(64, 64, 1)
        self.conv3 = conv_2d(64, 64, 1)
        self.conv4 = conv_2d(64, 128, 1)
        self.conv5 = conv_2d(128, 1024, 1)
        self.mlp1 = fc_layer(1024, 512)
        self.dropout1 = nn.Dropout2d(p=0.7)
        self.mlp2 = fc_layer(512, 256)
        self.dropout2 = nn.Dropout2d(p=0.7)
        self.mlp3 = nn.Linear(256, num_class)

    def forward(self, x, adapt=False):
        y = x.squeeze(-1).transpose(1, 2).contiguous()                # [B,N,3] rows
        y = torch.bmm(y, self.trans_net1.rows(y))
        y = self.conv2.rows(self.conv1.rows(y))
        y = torch.bmm(y, self.trans_net2.rows(y))
        y = self.conv5.rows_max_after(self.conv4, self.conv3.rows(y))
        mid_feature = y
        y = self.dropout1(self.mlp1(y))
        y = self.dropout2(self.mlp2(y))
        y = self.mlp3(y)
        if adapt:
            return y, mid_feature
        return y


class Pointnet2_cls(nn.Module):
    """model/model_pointnet.py:58-90."""

    def __init__(self, num_class=10, normal_channel=False):
        super(Pointnet2_cls, self).__init__()
        in_channel = 6 if normal_channel else 3
        self.normal_channel = normal_channel
        self.sa1 = PointNetSetAbstraction(512, 0.2, 32, in_channel, [64, 64, 128], False)
        self.sa2 = PointNetSetAbstraction(128, 0.4, 64, 128 + 3, [128, 128, 256], False)
        self.sa3 = PointNetSetAbstraction(None, None, None, 256 + 3, [256, 512, 1024], True)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.4)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.4)
        self.fc3 = nn.Linear(256, num_class)

    def forward(self, xyz):
        rows = xyz.squeeze(-1).transpose(1, 2).contiguous()
        B = rows.shape[0]
        norm = rows[:, :, 3:].contiguous() if self.normal_channel else None
        loc = rows[:, :, :3].contiguous()
        l1_xyz, l1_pts = self.sa1.rows(loc, norm)
        l2_xyz, l2_pts = self.sa2.rows(l1_xyz, l1_pts)
        _, l3_pts = self.sa3.rows(l2_xyz, l2_pts)
        x = l3_pts.reshape(B, 1024)
        x = self.drop1(F.relu(self.bn1(self.fc1(x))))
        x = self.drop2(F.relu(self.bn2(self.fc2(x))))
        return self.fc3(x)


K = 20      # model/model_pointnet.py:92


class DGCNN(nn.Module):
    """Source-only DGCNN classifier (model/model_pointnet.py:93-161): four EdgeConv layers (no SA-node module: x2 feeds
    conv3 directly), conv5 + bn5 + leaky_relu(0.2), max | avg pool, Pointnet_c(dgcnn_flag=True).  Same parameter names
    as the reference (`input_transform_net` is constructed and unused there too)."""

    def __init__(self):
        super(DGCNN, self).__init__()
        self.k = K
        self.input_transform_net = transform_net(6, 3)
        self.conv1 = conv_2d(6, 64, kernel=1, bias=False, activation='leakyrelu')
        self.conv2 = conv_2d(64 * 2, 64, kernel=1, bias=False, activation='leakyrelu')
        self.conv3 = conv_2d(64 * 2, 128, kernel=1, bias=False, activation='leakyrelu')
        self.conv4 = conv_2d(128 * 2, 256, kernel=1, bias=False, activation='leakyrelu')
        num_f_prev = 64 + 64 + 128 + 256
        self.bn5 = nn.BatchNorm1d(512)
        self.conv5 = nn.Conv1d(num_f_prev, 512, kernel_size=1, bias=False)
        self.classifier = Pointnet_c(dgcnn_flag=True)

    def forward(self, x, node=False, knn_idx=None):
        """x [B,3,N,1] -> logits [B,10].  `knn_idx` (4 tensors [B,N,k]) overrides the neighbour graphs (tests)."""
        B, N = x.size(0), x.size(2)
        loc = x.squeeze(-1).transpose(1, 2).contiguous()              # [B,N,3] rows
        gi = knn_idx or [None] * 4
        nb = lambda f, i: gi[i] if gi[i] is not None else ops.knn(f, self.k)
        # the EdgeConv layers write straight into the column slices of conv5's input (no torch.cat)
        cat_in = torch.empty(B, N, 512, dtype=torch.float32, device=x.device)
        x1 = self.conv1.edge_rows(loc, nb(loc, 0), out=cat_in[:, :, 0:64])
        x2 = self.conv2.edge_rows(x1, nb(x1, 1), out=cat_in[:, :, 64:128])
        x3 = self.conv3.edge_rows(x2, nb(x2, 2), out=cat_in[:, :, 128:256])
        x4 = self.conv4.edge_rows(x3, nb(x3, 3), out=cat_in[:, :, 256:512])
        x5 = ops.linear_rows(ops.assemble_rows(cat_in, (x1, x2, x3, x4)), self.conv5.weight.squeeze(-1))
        feat = ops.bn_act_pool_cat(x5, self.bn5, 0.2)                  # bn5 -> leaky_relu(0.2) -> max | avg pool, concatenated
        return self.classifier(feat)

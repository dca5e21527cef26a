"""Source-only classifiers used by train_source.py (mirror of model/model_pointnet.py:5-90)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

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
        y = self.conv5.rows_max(self.conv4.rows(self.conv3.rows(y)))
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

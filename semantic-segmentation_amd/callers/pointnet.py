"""Depth branch: `detseg/models/backbones/pointnet.py:12-52` -- five 1x1 conv + BN + ReLU layers on the HHA image,
strides (2, 2, 2, 1, 1), features of layers 2-5 (64, 64, 128, 256 channels at strides 4, 8, 8, 8) feed the DDCN
layers of ResNet stages 2-4."""
import math

import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm


class _ConvBNReLU(nn.Module):
    """The reference's ConvModule with a norm layer (conv without bias, attributes `conv`, `bn`, `activate`)."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.activate(self.bn(self.conv(x)))


class PointNet(nn.Module):

    def __init__(self, in_channels=3, num_stages=5, out_channels=(64, 64, 64, 128, 256), strides=(2, 2, 2, 1, 1),
                 out_indices=(1, 2, 3, 4)):
        super().__init__()
        assert num_stages == len(out_channels) == len(strides)
        self.out_indices = out_indices
        self.layers = []
        chans = (in_channels,) + tuple(out_channels)
        for i in range(num_stages):
            name = "layer{}".format(i + 1)
            self.add_module(name, _ConvBNReLU(chans[i], chans[i + 1], strides[i]))
            self.layers.append(name)

    def init_weights(self, pretrained=None):
        for m in self.modules():   # pointnet.py:37-44
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x):
        outs = []
        for i, name in enumerate(self.layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

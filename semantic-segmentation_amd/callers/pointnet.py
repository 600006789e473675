"""Depth branch of the DDCN models (`detseg/models/backbones/pointnet.py:12-52`): a stack of pointwise (1x1)
convolutions, each followed by BN and ReLU, applied to the HHA image.  With the default plan the features of layers
2-5 -- 64, 64, 128, 256 channels at strides 4, 8, 8, 8 -- feed the DDCN layers of ResNet stages 2-4.

Parameter names match the reference's ConvModule-based layers: `layerN.conv.weight`, `layerN.bn.{weight,bias,...}`.
"""
import math

import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm


class _PointwiseBlock(nn.Sequential):
    """conv (no bias: a norm layer follows) -> bn -> activate, under the reference's attribute names."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.add_module("conv", nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False))
        self.add_module("bn", nn.BatchNorm2d(cout))
        self.add_module("activate", nn.ReLU(inplace=True))


class PointNet(nn.Module):

    def __init__(self, in_channels=3, num_stages=5, out_channels=(64, 64, 64, 128, 256), strides=(2, 2, 2, 1, 1),
                 out_indices=(1, 2, 3, 4)):
        super().__init__()
        if not (num_stages == len(out_channels) == len(strides)):
            raise AssertionError("one output width and one stride per stage")
        self.out_indices = tuple(out_indices)
        widths = [in_channels, *out_channels]
        self.layers = ["layer%d" % (i + 1) for i in range(num_stages)]
        for i, name in enumerate(self.layers):
            setattr(self, name, _PointwiseBlock(widths[i], widths[i + 1], strides[i]))

    def init_weights(self, pretrained=None):
        """He-normal with fan_out for the convolutions, unit BN (pointnet.py:37-44); `pretrained` is ignored there too."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1]
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
            elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        feats = []
        for i, name in enumerate(self.layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                feats.append(x)
        return tuple(feats)

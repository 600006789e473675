"""DDCN-ResNet backbone: `detseg/models/backbones/resnet.py` (R below) restated for the configurations the DDCN
models use (Bottleneck depths 50/101/152, style 'pytorch', BN, optional deep stem), with the DDCN layer as conv2.

Module and parameter names are the reference's (conv1 / bn1 / layerN.M.{conv1,bn1,conv2,bn2,conv3,bn3,downsample}).
Blocks whose conv2 takes depth features receive and return `[x, depth]` exactly as R:221-262 do, so that
`DANet.extract_feat_hybrid` can drive them the way `segmentors/DANet.py:21-43` does.
"""
import torch.nn as nn
from torch.nn.modules.batchnorm import _BatchNorm

from ..dcn import DeformConvPack, DepthawareConv, DepthDeformConvPack, ModulatedDeformConvPack

_DCN_OPS = {"deform": DeformConvPack, "modulated": ModulatedDeformConvPack, "depthaware": DepthawareConv,
            "depthdeform": DepthDeformConvPack}


class Bottleneck(nn.Module):
    """R:85-262, style 'pytorch' (the stride sits on the 3x3 conv)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, dcn=None):
        super().__init__()
        assert dcn is None or isinstance(dcn, dict)
        self.inplanes, self.planes, self.stride, self.dilation = inplanes, planes, stride, dilation
        self.dcn_type = dcn.get("dcn_type") if dcn else None
        self.with_dcn = dcn is not None and not dcn.get("fallback_on_stride", False)
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        if not self.with_dcn:
            self.dcn_type = None
            self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=dilation, dilation=dilation,
                                   bias=False)
        else:
            if self.dcn_type not in _DCN_OPS:
                raise KeyError("Unrecognized dcn type {}".format(self.dcn_type))   # R:176
            kwargs = dict(kernel_size=3, stride=stride, padding=dilation, dilation=dilation, bias=False)
            if self.dcn_type != "depthaware":
                kwargs["deformable_groups"] = dcn.get("deformable_groups", 1)
            if self.dcn_type == "depthdeform":
                kwargs["depth_channels"] = planes // 2                               # R:174
            self.conv2 = _DCN_OPS[self.dcn_type](planes, planes, **kwargs)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    @property
    def takes_depth(self):
        return self.with_dcn and "depth" in self.dcn_type

    def forward(self, x):
        depth = None
        if self.takes_depth:
            x, depth = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.conv2(out, depth) if self.takes_depth else self.conv2(out)
        out = self.relu(self.bn2(out))
        out = self.bn3(self.conv3(out))
        out = out + (x if self.downsample is None else self.downsample(x))
        out = self.relu(out)
        # R:259-261: only the depthdeform block hands the depth features on
        return [out, depth] if self.dcn_type == "depthdeform" and self.with_dcn else out


def make_res_layer(inplanes, planes, blocks, stride=1, dilation=1, dcn=None):
    """R:265-326.  The first block runs at half the stage's dilation; a 'depthaware' stage keeps its DCN only in
    the first block (R:309-310)."""
    downsample = None
    if stride != 1 or inplanes != planes * Bottleneck.expansion:
        downsample = nn.Sequential(
            nn.Conv2d(inplanes, planes * Bottleneck.expansion, kernel_size=1, stride=stride, bias=False),
            nn.BatchNorm2d(planes * Bottleneck.expansion))
    first_dilation = dilation // 2 if dilation > 1 else dilation
    layers = [Bottleneck(inplanes, planes, stride, first_dilation, downsample, dcn)]
    if dcn and dcn["dcn_type"] == "depthaware":
        dcn = None
    layers += [Bottleneck(planes * Bottleneck.expansion, planes, 1, dilation, None, dcn) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class ResNet(nn.Module):
    """R:330-554 for Bottleneck depths."""
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, in_channels=3, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), norm_eval=True, dcn=None, stage_with_dcn=(False, False, False, False),
                 zero_init_residual=True, deep_base=False):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError("invalid depth {} for resnet".format(depth))
        assert 1 <= num_stages <= 4 and len(strides) == len(dilations) == num_stages
        assert max(out_indices) < num_stages
        assert dcn is None or len(stage_with_dcn) == num_stages
        self.depth, self.num_stages, self.out_indices = depth, num_stages, out_indices
        self.norm_eval, self.dcn, self.zero_init_residual, self.deep_base = norm_eval, dcn, zero_init_residual, deep_base
        self.stage_blocks = self.arch_settings[depth][:num_stages]

        if deep_base:   # R:469-480: three 3x3 convolutions, 128 channels out
            self.inplanes = 128
            self.conv1 = nn.Sequential(
                nn.Conv2d(3, 64, kernel_size=3, stride=2, padding=1, bias=False), nn.BatchNorm2d(64),
                nn.ReLU(inplace=True),
                nn.Conv2d(64, 64, kernel_size=3, stride=1, padding=1, bias=False), nn.BatchNorm2d(64),
                nn.ReLU(inplace=True),
                nn.Conv2d(64, 128, kernel_size=3, stride=1, padding=1, bias=False))
        else:
            self.inplanes = 64
            self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

        self.res_layers = []
        for i, num_blocks in enumerate(self.stage_blocks):
            planes = 64 * 2 ** i
            layer = make_res_layer(self.inplanes, planes, num_blocks, stride=strides[i], dilation=dilations[i],
                                   dcn=dcn if (dcn is not None and stage_with_dcn[i]) else None)
            self.inplanes = planes * Bottleneck.expansion
            name = "layer{}".format(i + 1)
            self.add_module(name, layer)
            self.res_layers.append(name)
        self.feat_dim = Bottleneck.expansion * 64 * 2 ** (len(self.stage_blocks) - 1)

    def init_weights(self, pretrained=None, dcn_offset_init="zero"):
        """R:508-529 with pretrained=None: Kaiming-normal (fan_out, relu) for every nn.Conv2d, BN to 1/0, the last
        BN of each block to 0.  dcn_offset_init: the reference then means to zero the offset convolutions
        (R:518-522) but tests for an attribute of an older API (`conv2_offset`), so its Kaiming pass -- which also
        visits the DDCN layers' conv_offset, an nn.Conv2d -- stands and a randomly initialised model samples
        ~6 px off the grid.  "zero" (default) applies the evident intent, i.e. the layers' own init_offset();
        "kaiming" reproduces the reference literally."""
        if pretrained is not None:
            raise TypeError("checkpoints are loaded by the caller (no network, no pickles here)")
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (_BatchNorm, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if dcn_offset_init == "zero":
            for m in self.modules():
                if isinstance(m, Bottleneck) and hasattr(m.conv2, "init_offset"):
                    m.conv2.init_offset()
        elif dcn_offset_init != "kaiming":
            raise ValueError("dcn_offset_init must be 'zero' or 'kaiming'")
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def stem(self, x):
        return self.maxpool(self.relu(self.bn1(self.conv1(x))))

    def forward(self, x):
        """R:531-541 (no depth features: only for backbones without depth-fed DCN stages)."""
        x = self.stem(x)
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:   # R:547-552
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self

"""DANet head and segmentor: `detseg/models/seg_heads/danet_head.py` (D below), `seg_heads/seg_head.py`,
`segmentors/DANet.py` and the train / test forwards of `segmentors/single_stage.py:49-98`."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class PAM_Module(nn.Module):
    """Position attention (D:72-108): softmax(Q^T K) over all pixel pairs, out = gamma * V A^T + x.

    The reference materialises the whole [hw, hw] attention map (354 MB per image in fp32 at 97 x 97, 2.8 GB for a batch
    of 8, several times over with softmax's input, output and their gradients).  A softmax row belongs to ONE query
    pixel, so the map can be computed in blocks of query pixels -- softmax(Q_blk^T K), then V A_blk^T -- with the same
    arithmetic per element; `max_map_elems` bounds batch x block x hw.  Measured on MI355X (profiles/r03a_pam_blocks.txt):
    blocks cost 1 % of the fp32 DDCN-R101 step (464.3 -> 468.5 ms) and 10 % of the bf16 inference step, and take 7.4 GB off
    the R101 peak -- so the default is the reference's single map (None), and blocks of at most 2^27 elements are used only
    under torch.autocast when the map is larger than anything this build has run there (the one full-size R101 autocast
    run faulted inside PyTorch's / rocBLAS's kernels on the 8 x 9409 x 9409 map)."""

    max_map_elems = int(os.environ.get("DDCN_PAM_MAX_MAP_ELEMS", 0)) or None
    autocast_map_elems = 1 << 27

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        b, c, h, w = x.shape
        hw = h * w
        q = self.query_conv(x).flatten(2).transpose(1, 2)        # [b, hw, c/8]
        k = self.key_conv(x).flatten(2)                          # [b, c/8, hw]
        v = self.value_conv(x).flatten(2)                        # [b, c, hw]
        limit = self.max_map_elems
        if limit is None and torch.is_autocast_enabled("cuda") and x.is_cuda and b * hw * hw > 2 * self.autocast_map_elems:
            limit = self.autocast_map_elems
        blk = hw if not limit else max(1, min(hw, limit // max(1, b * hw)))
        if blk >= hw:
            attention = torch.softmax(torch.bmm(q, k), dim=-1)   # [b, hw, hw]
            out = torch.bmm(v, attention.transpose(1, 2))
        else:
            outs = []
            for s in range(0, hw, blk):
                a = torch.softmax(torch.bmm(q[:, s:s + blk], k), dim=-1)        # [b, blk, hw]: rows of the full map
                outs.append(torch.bmm(v, a.transpose(1, 2)))                    # [b, c, blk]
            out = torch.cat(outs, dim=2)
        return self.gamma * out.view(b, c, h, w) + x


class CAM_Module(nn.Module):
    """Channel attention (D:111-136): softmax(max - X X^T) over channel pairs, out = gamma * A X + x."""

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        b, c, h, w = x.shape
        flat = x.flatten(2)
        energy = torch.bmm(flat, flat.transpose(1, 2))
        energy = energy.max(dim=-1, keepdim=True)[0] - energy
        out = torch.bmm(torch.softmax(energy, dim=-1), flat).view(b, c, h, w)
        return self.gamma * out + x


def _conv_bn_relu(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU())


class DANetHead(nn.Module):
    """D:10-64.  Three logits: fused, position branch, channel branch."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        inter = in_channels // 4
        self.conv5a = _conv_bn_relu(in_channels, inter)
        self.conv5c = _conv_bn_relu(in_channels, inter)
        self.sa = PAM_Module(inter)
        self.sc = CAM_Module(inter)
        self.conv51 = _conv_bn_relu(inter, inter)
        self.conv52 = _conv_bn_relu(inter, inter)
        # D:40-43 hard-codes 512 input channels (= 2048 // 4); kept general here
        self.conv6 = nn.Sequential(nn.Dropout2d(0.1, False), nn.Conv2d(inter, out_channels, 1))
        self.conv7 = nn.Sequential(nn.Dropout2d(0.1, False), nn.Conv2d(inter, out_channels, 1))
        self.conv8 = nn.Sequential(nn.Dropout2d(0.1, False), nn.Conv2d(inter, out_channels, 1))

    def forward(self, x):
        sa_conv = self.conv51(self.sa(self.conv5a(x)))
        sc_conv = self.conv52(self.sc(self.conv5c(x)))
        return self.conv8(sa_conv + sc_conv), self.conv6(sa_conv), self.conv7(sc_conv)

    def init_weights(self):
        pass   # D:60-64: PyTorch defaults

    @staticmethod
    def loss(cls_logits, gt_labels, ignore_index=-100):
        """seg_head.py:19-43 with CrossEntropyLoss (mean over the pixels that are not ignored), summed over logits."""
        return dict(loss_head=sum(F.cross_entropy(l.float(), gt_labels, ignore_index=ignore_index) for l in cls_logits))


class DANet(nn.Module):
    """segmentors/DANet.py:7-43 + single_stage.py forward_train / simple_test, for an RGB image and an HHA (or any
    3-channel depth encoding) image."""

    def __init__(self, backbone, head, backbone_depth=None, dcn_offset_init="zero"):
        super().__init__()
        self.backbone, self.head, self.backbone_depth = backbone, head, backbone_depth
        self.init_weights(dcn_offset_init)

    def init_weights(self, dcn_offset_init="zero"):
        self.backbone.init_weights(dcn_offset_init=dcn_offset_init)
        if self.backbone_depth is not None:
            self.backbone_depth.init_weights()
        self.head.init_weights()

    def extract_feat_hybrid(self, img, hha):
        """DANet.py:21-43: stage 1 plain; the first block of stage 2 runs at stride 4 and takes the stride-4 depth
        features, every later DDCN block the features of its own stage."""
        depth = self.backbone_depth(hha)
        x = self.backbone.stem(img)
        for i, name in enumerate(self.backbone.res_layers):
            layer = getattr(self.backbone, name)
            if i == 0:
                x = layer(x)
            elif i == 1:
                for j, block in enumerate(layer):
                    x, _ = block([x, depth[i - 1] if j == 0 else depth[i]])
            else:
                x, _ = layer([x, depth[i]])
        return x

    def extract_feat(self, img, hha=None):
        if self.backbone_depth is not None:
            return self.extract_feat_hybrid(img, hha)
        return self.backbone(img)[-1]

    def _logits(self, img, hha):
        size = img.shape[2:]
        return [F.interpolate(l, size=size, mode="bilinear", align_corners=True)
                for l in self.head(self.extract_feat(img, hha))]

    def forward_train(self, img, gt_label, HHA=None, ignore_index=-100):
        return self.head.loss(self._logits(img, HHA), gt_label, ignore_index=ignore_index)

    def simple_test(self, img, HHA=None):
        return self._logits(img, HHA)[0]

    def forward(self, img, gt_label=None, HHA=None, ignore_index=-100):
        if gt_label is not None:
            return self.forward_train(img, gt_label, HHA, ignore_index)
        return self.simple_test(img, HHA)

"""DDCN layer (DepthDeformConvPack, res5 shape) forward+backward, and its parts, on one GPU."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("semantic-segmentation_amd")
dcn = importlib.import_module("semantic-segmentation_amd.dcn.deform_conv")
dev = torch.device("cuda:0")
N, C, H, D = 2, 512, 97, 4


def timed(fn, warm=5, iters=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


torch.manual_seed(0)
layer = dcn.DepthDeformConvPack(C, C, 3, stride=1, padding=D, dilation=D, deformable_groups=1, bias=False,
                                depth_channels=C // 2).to(dev)
torch.nn.init.normal_(layer.conv_offset.weight, std=0.01)
torch.nn.init.normal_(layer.conv_mask.weight, std=0.01)
x = torch.randn(N, C, H, H, device=dev, requires_grad=True)
depth = torch.randn(N, C // 2, H, H, device=dev, requires_grad=True)
gout = torch.randn(N, C, H, H, device=dev)


def full():
    layer.zero_grad(set_to_none=True)
    x.grad = None; depth.grad = None
    layer(x, depth).backward(gout)


def offset_branch():
    layer.zero_grad(set_to_none=True)
    x.grad = None; depth.grad = None
    o = layer.conv_offset(torch.cat((x, depth), dim=1))
    o.backward(torch.ones_like(o))


off = layer.conv_offset(torch.cat((x, depth), dim=1)).detach().requires_grad_(True)


def mask_branch():
    layer.zero_grad(set_to_none=True)
    depth.grad = None; off.grad = None
    m = torch.sigmoid(layer.conv_mask(depth, off))
    m.backward(torch.ones_like(m))


mask = torch.sigmoid(layer.conv_mask(depth, off)).detach().requires_grad_(True)


def main_op():
    layer.zero_grad(set_to_none=True)
    x.grad = None; off.grad = None; mask.grad = None
    pkg.modulated_deform_conv(x, off, mask, layer.weight, None, 1, D, D, 1, 1).backward(gout)


print("DDCN layer res5 (N=2, C=512, depth 256, 97x97, dil 4) fwd+bwd, ms")
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, fn in (("whole layer", full), ("conv_offset branch (cat + conv2d)", offset_branch),
                 ("conv_mask branch (DeformConv 256->9 + sigmoid)", mask_branch), ("main modulated op", main_op)):
    if only in name:
        print("  %-48s %.3f" % (name, timed(fn)), flush=True)

"""GPU: the fp64 instantiation (csrc/ddcn_f64.hip) through the autograd boundary.

* every golden fixture (fp64 expectations) to 1e-11 forward / 1e-10 backward: with double arithmetic the only
  difference to the oracle is summation order;
* the reference's own GPU self-test, detseg/ops/dcn/test.py:208-267 (e),(f): torch.autograd.gradcheck on double
  tensors with its settings (eps 1e-3, atol 1e-3, rtol 1e-2, nondet_tol 1.0), here on the HIP modules themselves;
* the DDCN layer and the depth-aware op in double.
"""
import importlib

import numpy as np
import pytest
import torch

import cases
from helpers import load_golden
from oracle import ddcn_oracle as orc

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    pkg = importlib.import_module("semantic-segmentation_amd")
DEV = "cuda:0"


def _t(a, grad=False):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(DEV)
    return t.requires_grad_(True) if grad else t


@pytest.mark.parametrize("c", cases.SMALL_CASES, ids=lambda c: c["name"])
def test_fp64_matches_golden(c):
    g = load_golden(c["name"])
    x, off, w = _t(g["x"], True), _t(g["offset"], True), _t(g["weight"], True)
    m = _t(g["mask"], True) if c["modulated"] else None
    b = _t(g["bias"], True) if c["bias"] else None
    if c["modulated"]:
        out = pkg.modulated_deform_conv(x, off, m, w, b, c["stride"], c["pad"], c["dil"], c["groups"], c["dg"])
    else:
        out = pkg.deform_conv(x, off, w, b, c["stride"], c["pad"], c["dil"], c["groups"], c["dg"])
    assert out.dtype == torch.float64
    out.backward(_t(g["gout"]))
    torch.cuda.synchronize()
    assert cases.relerr(out.detach().cpu().numpy(), g["out"]) <= 1e-11
    pairs = [(x, "gx"), (off, "goff"), (w, "gw")] + ([(m, "gmask")] if c["modulated"] else []) + \
            ([(b, "gb")] if c["bias"] else [])
    for t, k in pairs:
        assert t.grad.dtype == torch.float64
        assert cases.relerr(t.grad.cpu().numpy(), g[k]) <= 1e-10, (k, cases.relerr(t.grad.cpu().numpy(), g[k]))


@pytest.mark.parametrize("c", cases.DEPTHAWARE_CASES, ids=lambda c: c["name"])
def test_fp64_depthaware_matches_golden(c):
    g = load_golden(c["name"])
    x, w = _t(g["x"], True), _t(g["weight"], True)
    b = _t(g["bias"], True) if c["bias"] else None
    out = pkg.depthaware_conv(x, _t(g["depth"]), w, b, c["stride"], c["pad"], c["dil"], c["groups"])
    out.backward(_t(g["gout"]))
    assert cases.relerr(out.detach().cpu().numpy(), g["out"]) <= 1e-11
    assert cases.relerr(x.grad.cpu().numpy(), g["gx"]) <= 1e-10
    assert cases.relerr(w.grad.cpu().numpy(), g["gw"]) <= 1e-10
    if c["bias"]:
        assert cases.relerr(b.grad.cpu().numpy(), g["gb"]) <= 1e-10


@pytest.mark.parametrize("modulated", [False, True], ids=["deform", "modulated"])
def test_ref_selftest_gradcheck_on_the_gpu_modules(modulated):
    # detseg/ops/dcn/test.py:208-233 (e) and 238-267 (f): shapes, value ranges and tolerances of the script
    batch_size, in_channels, out_channels, image_h, image_w, groups = 2, 4, 4, 4, 4, 2
    gen = torch.Generator().manual_seed(11)
    x = (torch.rand(batch_size, in_channels, image_h, image_w, generator=gen, dtype=torch.float64) * 0.01).to(DEV)
    offset = (torch.randn(batch_size, 18, image_h, image_w, generator=gen, dtype=torch.float64) * 2).to(DEV)
    mask = torch.sigmoid(torch.rand(batch_size, 9, image_h, image_w, generator=gen, dtype=torch.float64)).to(DEV)
    weight = torch.randn(out_channels, in_channels // groups, 3, 3, generator=gen, dtype=torch.float64).to(DEV)
    bias = torch.rand(out_channels, generator=gen, dtype=torch.float64).to(DEV)
    for t in (x, offset, mask, weight, bias):
        t.requires_grad_(True)
    if modulated:
        fn = lambda a, o, m, w, b: pkg.modulated_deform_conv(a, o, m, w, b, 1, 1, 1, groups, 1)
        args = (x, offset, mask, weight, bias)
    else:
        fn = lambda a, o, w, b: pkg.deform_conv(a, o, w, b, 1, 1, 1, groups, 1)
        args = (x, offset, weight, bias)
    assert torch.autograd.gradcheck(fn, args, eps=1e-3, atol=1e-3, rtol=1e-2, raise_exception=True, nondet_tol=1.0)


def test_fp64_ddcn_layer_matches_fp32_layer():
    torch.manual_seed(3)
    layer = pkg.DepthDeformConvPack(16, 24, kernel_size=3, stride=1, padding=2, dilation=2, deformable_groups=1,
                                    bias=False, depth_channels=8).to(DEV)
    with torch.no_grad():   # non-trivial offsets and masks
        layer.conv_offset.weight.normal_(0, 0.05)
        layer.conv_mask.weight.normal_(0, 0.1)
    x32, d32 = torch.randn(2, 16, 13, 17, device=DEV), torch.randn(2, 8, 13, 17, device=DEV)
    out32 = layer(x32.requires_grad_(True), d32)
    out32.square().sum().backward()
    g32 = layer.weight.grad.clone()
    layer.zero_grad()
    layer64 = layer.double()
    x64 = x32.detach().double().requires_grad_(True)
    out64 = layer64(x64, d32.double())
    assert out64.dtype == torch.float64
    out64.square().sum().backward()
    assert cases.relerr(out32.detach().double().cpu().numpy(), out64.detach().cpu().numpy()) <= 1e-5
    assert cases.relerr(g32.double().cpu().numpy(), layer64.weight.grad.cpu().numpy()) <= 1e-4
    assert cases.relerr(x32.grad.double().cpu().numpy(), x64.grad.cpu().numpy()) <= 1e-4

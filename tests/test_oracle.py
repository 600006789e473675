"""Pins the CPU oracle (oracle/) before anything is compared against it.

1. The reference's own known-answer self-tests, detseg/ops/dcn/test.py:17-181 (a)-(d): zero offsets
   == nn.Conv2d (max-abs < 1e-5), identity weights == input (< 1e-10), same seeds/shapes.
2. The reference's gradcheck settings, test.py:208-267 (e),(f): fp64, eps 1e-3, atol 1e-3, rtol 1e-2.
3. Agreement with an independently written grid_sample/autograd formulation (tests/indep_ref.py).
The reference ships no golden vectors, so these are the pins.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

import cases
import indep_ref
from oracle import ddcn_oracle as orc

torch.manual_seed(3)  # test.py:16

deformable_groups = 1
batch_size, in_channels, image_h, image_w = 2, 4, 4, 4  # test.py:10-13
out_channels = 4
kernel_h, kernel_w = 3, 3


def _conv_identify(weight, bias, groups=1):
    # test.py:97-109
    weight.zero_()
    bias.zero_()
    o, i, h, w = weight.shape
    y, x = h // 2, w // 2
    oc = o // groups
    for p in range(i):
        for q in range(o):
            if p == (q % oc):
                weight[q, p, y, x] = 1.0


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("modulated", [False, True])
def test_ref_selftest_zero_offset_equals_conv2d(dt, modulated):
    # test.py:17-49 (a), :52-94 (b): groups=2, bias, pad 1; mask = 2*sigmoid(0) = 1
    pcn = nn.Conv2d(in_channels, out_channels, (kernel_h, kernel_w), stride=1, padding=1, dilation=1, groups=2)
    tdt = torch.float32 if dt == np.float32 else torch.float64
    pcn = pcn.to(tdt)
    x = torch.randn(batch_size, in_channels, image_h, image_w, dtype=tdt)
    offset = np.zeros((batch_size, 2 * 9, image_h, image_w), dtype=dt)
    mask = np.full((batch_size, 9, image_h, image_w), 2 * 0.5, dtype=dt) if modulated else None
    out = orc.deform_forward(x.numpy(), offset, mask, pcn.weight.detach().numpy(), pcn.bias.detach().numpy(),
                             1, 1, 1, 2, 1)
    d = np.abs(out - pcn(x).detach().numpy()).max()
    assert d < 1e-5


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("modulated", [False, True])
def test_ref_selftest_identity_weights(dt, modulated):
    # test.py:112-140 (c): output == input < 1e-10; :142-181 (d): mask = sigmoid(0) = 0.5 -> 2*output == input
    groups = 2
    w = torch.zeros(out_channels, in_channels // groups, 3, 3, dtype=torch.float64)
    b = torch.zeros(out_channels, dtype=torch.float64)
    _conv_identify(w, b, groups)
    x = torch.randn(batch_size, in_channels, image_h, image_w, dtype=torch.float64).numpy().astype(dt)
    offset = np.zeros((batch_size, 18, image_h, image_w), dtype=dt)
    mask = np.full((batch_size, 9, image_h, image_w), 0.5, dtype=dt) if modulated else None
    out = orc.deform_forward(x, offset, mask, w.numpy().astype(dt), b.numpy().astype(dt), 1, 1, 1, groups, 1)
    if modulated:
        out = out * 2
    assert np.abs(x - out).max() < 1e-10


class _OracleDeform(torch.autograd.Function):
    """Test-only autograd wrapper so torch.autograd.gradcheck can drive the oracle."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, padding, dilation, groups, dg):
        ctx.save_for_backward(x, offset, mask if mask is not None else torch.empty(0), weight)
        ctx.cfg = (stride, padding, dilation, groups, dg, mask is not None, bias is not None)
        out = orc.deform_forward(x.detach().numpy(), offset.detach().numpy(),
                                 None if mask is None else mask.detach().numpy(), weight.detach().numpy(),
                                 None if bias is None else bias.detach().numpy(), stride, padding, dilation,
                                 groups, dg)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, go):
        x, offset, mask, weight = ctx.saved_tensors
        stride, padding, dilation, groups, dg, has_mask, has_bias = ctx.cfg
        gx, goff, gmask, gw, gb = orc.deform_backward(
            x.numpy(), offset.numpy(), mask.numpy() if has_mask else None, weight.numpy(),
            go.contiguous().numpy(), has_bias, stride, padding, dilation, groups, dg)
        t = torch.from_numpy
        return (t(gx), t(goff), t(gmask) if has_mask else None, t(gw), t(gb) if has_bias else None,
                None, None, None, None, None)


@pytest.mark.parametrize("modulated", [False, True])
def test_ref_selftest_gradcheck(modulated):
    # test.py:208-233 (e), :238-267 (f)
    groups = 2
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(batch_size, in_channels, image_h, image_w, generator=g, dtype=torch.float64) * 0.01)
    offset = torch.randn(batch_size, 18, image_h, image_w, generator=g, dtype=torch.float64) * 2
    mask = torch.sigmoid(torch.rand(batch_size, 9, image_h, image_w, generator=g, dtype=torch.float64)) \
        if modulated else None
    weight = torch.randn(out_channels, in_channels // groups, 3, 3, generator=g, dtype=torch.float64)
    bias = torch.rand(out_channels, generator=g, dtype=torch.float64)
    for t in (x, offset, weight, bias):
        t.requires_grad_(True)
    if mask is not None:
        mask.requires_grad_(True)
    assert torch.autograd.gradcheck(
        lambda a, o, m, w, b: _OracleDeform.apply(a, o, m, w, b, 1, 1, 1, groups, 1),
        (x, offset, mask, weight, bias), eps=1e-3, atol=1e-3, rtol=1e-2, raise_exception=True, nondet_tol=1.0)


def _indep_all(c, inp):
    t = {k: (torch.from_numpy(v).clone().requires_grad_(True) if v is not None else None)
         for k, v in inp.items() if k in ("x", "offset", "mask", "weight", "bias")}
    out = indep_ref.deform_conv(t["x"], t["offset"], t["mask"], t["weight"], t["bias"], c["stride"], c["pad"],
                                c["dil"], c["groups"], c["dg"])
    out.backward(torch.from_numpy(inp["gout"]))
    g = {k: (v.grad.numpy() if v is not None else None) for k, v in t.items()}
    return out.detach().numpy(), g


@pytest.mark.parametrize("c", cases.SMALL_CASES + cases.MEDIUM_CASES[:2], ids=lambda c: c["name"])
def test_oracle_matches_independent_formulation_fp64(c):
    inp = cases.make_inputs(c)
    kw = cases.kwargs(c)
    out = orc.deform_forward(inp["x"], inp["offset"], inp["mask"], inp["weight"], inp["bias"], **kw)
    gx, goff, gmask, gw, gb = orc.deform_backward(inp["x"], inp["offset"], inp["mask"], inp["weight"],
                                                  inp["gout"], c["bias"], **kw)
    iout, ig = _indep_all(c, inp)
    tol = 1e-10
    assert cases.relerr(out, iout) < tol
    assert cases.relerr(gx, ig["x"]) < tol
    assert cases.relerr(gw, ig["weight"]) < tol
    if c["bias"]:
        assert cases.relerr(gb, ig["bias"]) < tol
    if c["modulated"]:
        assert cases.relerr(gmask, ig["mask"]) < tol
    if not cases.has_integer_coords(c):
        assert cases.relerr(goff, ig["offset"]) < tol


@pytest.mark.parametrize("c", cases.SMALL_CASES[:6], ids=lambda c: c["name"])
def test_oracle_fp32_close_to_fp64(c):
    inp = cases.make_inputs(c)
    kw = cases.kwargs(c)
    f = lambda a: None if a is None else a.astype(np.float32)
    out64 = orc.deform_forward(inp["x"], inp["offset"], inp["mask"], inp["weight"], inp["bias"], **kw)
    out32 = orc.deform_forward(f(inp["x"]), f(inp["offset"]), f(inp["mask"]), f(inp["weight"]), f(inp["bias"]), **kw)
    assert out32.dtype == np.float32
    assert cases.relerr(out32, out64) < 1e-5


@pytest.mark.parametrize("c", cases.DEPTHAWARE_CASES, ids=lambda c: c["name"])
def test_oracle_depthaware_matches_independent_fp64(c):
    inp = cases.make_inputs(c)
    out = orc.depthaware_forward(inp["x"], inp["depth"], inp["weight"], inp["bias"], c["stride"], c["pad"],
                                 c["dil"], c["groups"])
    gx, gw, gb = orc.depthaware_backward(inp["x"], inp["depth"], inp["weight"], inp["gout"], c["bias"],
                                         c["stride"], c["pad"], c["dil"], c["groups"])
    x = torch.from_numpy(inp["x"]).requires_grad_(True)
    w = torch.from_numpy(inp["weight"]).requires_grad_(True)
    b = torch.from_numpy(inp["bias"]).requires_grad_(True) if c["bias"] else None
    iout = indep_ref.depthaware_conv(x, torch.from_numpy(inp["depth"]), w, b, c["stride"], c["pad"], c["dil"],
                                     c["groups"])
    iout.backward(torch.from_numpy(inp["gout"]))
    assert cases.relerr(out, iout.detach().numpy()) < 1e-10
    assert cases.relerr(gx, x.grad.numpy()) < 1e-10
    assert cases.relerr(gw, w.grad.numpy()) < 1e-10
    if c["bias"]:
        assert cases.relerr(gb, b.grad.numpy()) < 1e-10


def test_oracle_stage_functions_consistent_with_whole_op():
    c = cases.SMALL_CASES[3]
    inp = cases.make_inputs(c)
    Ho, Wo, K2 = cases.geometry(c)
    col = orc.im2col(inp["x"][0], inp["offset"][0], inp["mask"][0], c["kh"], c["kw"], c["stride"], c["pad"],
                     c["dil"], c["dg"])
    icol = indep_ref.sample_columns(torch.from_numpy(inp["x"][:1]), torch.from_numpy(inp["offset"][:1]),
                                    torch.from_numpy(inp["mask"][:1]), c["kh"], c["kw"], c["stride"], c["pad"],
                                    c["dil"], c["dg"]).numpy().reshape(c["Ci"] * K2, Ho * Wo)
    assert cases.relerr(col, icol) < 1e-12

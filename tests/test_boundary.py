"""CPU: host-side logic of the drop-in boundary -- no compute call is made without a GPU.

* libddcn_hip.so loads and exports every symbol include/ddcn_hip.h declares;
* geometry inference / validation through the C ABI (pure host code);
* the nn.Modules keep the reference's constructor signatures, defaults, init and state-dict keys
  (detseg/ops/dcn/deform_conv.py:228-463);
* CPU tensors raise NotImplementedError as detseg/ops/dcn/deform_conv.py:33-34 does.
"""
import ctypes
import importlib
import inspect
import math
import os
import re

import pytest
import torch

from conftest import ROOT, load_pkg

pkg = load_pkg()
_lib = importlib.import_module("semantic-segmentation_amd.dcn._lib")


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ddcn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ddcn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared_symbols()
    assert len(names) >= 20
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), n
    assert set(names) == set(_lib.PROTOTYPES), set(names) ^ set(_lib.PROTOTYPES)


def test_version_and_strerror():
    L = _lib.lib()
    assert L.ddcn_version() == 100
    assert L.ddcn_strerror(0) == b"ok"
    assert L.ddcn_strerror(-2) == b"shape mismatch"


@pytest.mark.parametrize("H,W,k,s,p,d", [(97, 97, 3, 1, 4, 4), (193, 193, 3, 2, 1, 1), (4, 4, 3, 1, 1, 1),
                                         (7, 5, 3, 2, 0, 1), (9, 9, 5, 1, 2, 1), (480, 640, 3, 2, 2, 2)])
def test_infer_output_matches_reference_formula(H, W, k, s, p, d):
    g = _lib.Geom(1, 4, H, W, 4, k, k, s, s, p, p, d, d, 1, 1, 0, 0)
    assert _lib.lib().ddcn_infer_output(g) == 0
    assert g.Ho == (H + 2 * p - (d * (k - 1) + 1)) // s + 1   # deform_conv.py:76-79
    assert g.Wo == (W + 2 * p - (d * (k - 1) + 1)) // s + 1
    assert _lib.lib().ddcn_check_geom(g) == 0
    ld = _lib.lib().ddcn_col_ld(g)
    assert ld >= g.Ho * g.Wo and ld % 32 == 0
    assert _lib.lib().ddcn_workspace_bytes(g) == 4 * 1 * 4 * k * k * ld


@pytest.mark.parametrize("k", [1, 3, 5])
def test_16bit_g2_workspace_follows_the_kernel_size(k):
    """The 16-bit G2 writes C/groups * kh * kw rows per group: its operand workspace (blocked W^T + grad_out octets)
    and the columns buffer it fills must both be sized from the same kernel size (a 1x1 layer has 9x fewer rows)."""
    L = _lib.lib()
    N, C, Co, H, W, groups = 2, 16, 40, 9, 7, 1
    g = _lib.Geom(N, C, H, W, Co, k, k, 1, 1, k // 2, k // 2, 1, 1, groups, 1, H, W)
    assert L.ddcn_gemm_gcol_16_supported(g) == 0
    kt = (Co // groups + 31) // 32
    a_bytes = groups * kt * (C // groups * k * k) * 32 * 2
    assert L.ddcn_gemm_gcol_16_workspace_bytes(g) == (a_bytes + 255) // 256 * 256 + N * Co * H * W * 2
    assert L.ddcn_workspace_bytes(g) == 4 * N * C * k * k * L.ddcn_col_ld(g)


def test_check_geom_rejects_bad_shapes():
    L = _lib.lib()
    ok = dict(N=2, C=8, H=9, W=9, Co=8, kh=3, kw=3, sh=1, sw=1, ph=1, pw=1, dh=1, dw=1, groups=2, dg=2, Ho=9, Wo=9)

    def rc(**kw):
        d = dict(ok, **kw)
        return L.ddcn_check_geom(_lib.Geom(*[d[n] for n, _ in _lib.Geom._fields_]))
    assert rc() == 0
    assert rc(groups=3) == -2          # channels not divisible by groups (deform_conv_cuda.cpp:196-198)
    assert rc(dg=3) == -2              # deform_conv_cuda.cpp:132-133
    assert rc(Ho=8) == -2              # offset/output spatial size mismatch (deform_conv_cuda.cpp:150-153)
    assert rc(sh=0) == -2
    assert rc(dh=0) == -2
    assert rc(H=1, W=1, ph=0, pw=0, Ho=0, Wo=0) == -2   # output size is too small (:135-140)
    assert b"too small" in L.ddcn_last_error()
    with pytest.raises(RuntimeError, match="shape mismatch"):
        _lib.check(rc(groups=3))


def test_cpu_tensors_raise_not_implemented():
    x = torch.zeros(1, 4, 4, 4)
    with pytest.raises(NotImplementedError):
        pkg.deform_conv(x, torch.zeros(1, 18, 4, 4), torch.zeros(4, 4, 3, 3))
    with pytest.raises(NotImplementedError):
        pkg.modulated_deform_conv(x, torch.zeros(1, 18, 4, 4), torch.zeros(1, 9, 4, 4), torch.zeros(4, 4, 3, 3))
    with pytest.raises(NotImplementedError):
        pkg.depthaware_conv(x, torch.zeros(1, 1, 4, 4), torch.zeros(4, 4, 3, 3))


def test_function_signatures_match_reference():
    # deform_conv.py:15-24, 86-96, 159-167
    assert list(inspect.signature(pkg.DeformConvFunction.forward).parameters) == \
        ["ctx", "input", "offset", "weight", "bias", "stride", "padding", "dilation", "groups", "deformable_groups"]
    assert list(inspect.signature(pkg.ModulatedDeformConvFunction.forward).parameters) == \
        ["ctx", "input", "offset", "mask", "weight", "bias", "stride", "padding", "dilation", "groups",
         "deformable_groups"]
    assert list(inspect.signature(pkg.DepthawareConvFunction.forward).parameters) == \
        ["ctx", "input", "depth", "weight", "bias", "stride", "padding", "dilation", "groups"]
    for cls in (pkg.DeformConv, pkg.ModulatedDeformConv, pkg.DepthawareConv):
        assert list(inspect.signature(cls.__init__).parameters) == \
            ["self", "in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation", "groups",
             "deformable_groups", "bias"]


def test_module_defaults_and_init():
    torch.manual_seed(0)
    assert pkg.DeformConv(8, 4, 3).bias is None                 # bias=False default (deform_conv.py:239)
    assert pkg.ModulatedDeformConv(8, 4, 3).bias is not None    # bias=True default (:311)
    assert pkg.DepthawareConv(8, 4, 3).bias is None             # (:389)
    m = pkg.ModulatedDeformConv(8, 4, 3, groups=2)
    assert m.weight.shape == (4, 4, 3, 3)
    bound = 1.0 / math.sqrt(8 * 9)                               # :261-268
    assert m.weight.abs().max() <= bound and m.weight.abs().max() > 0.5 * bound
    assert torch.all(m.bias == 0)
    assert m.kernel_size == (3, 3)


def test_state_dict_keys_match_reference():
    # SURVEY.md section 5 (checkpoint/resume): keys a drop-in must keep
    ddcn = pkg.DepthDeformConvPack(8, 8, 3, stride=1, padding=2, dilation=2, deformable_groups=1, bias=False,
                                   depth_channels=4)
    assert sorted(ddcn.state_dict()) == ["conv_mask.bias", "conv_mask.weight", "conv_offset.bias",
                                         "conv_offset.weight", "weight"]
    assert ddcn.conv_offset.weight.shape == (18, 12, 3, 3) and ddcn.conv_offset.dilation == (2, 2)
    assert ddcn.conv_mask.weight.shape == (9, 4, 3, 3) and ddcn.conv_mask.deformable_groups == 1
    for t in (ddcn.conv_offset.weight, ddcn.conv_offset.bias, ddcn.conv_mask.weight, ddcn.conv_mask.bias):
        assert torch.all(t == 0)                                 # init_offset, deform_conv.py:449-455
    assert ddcn.conv_offset.lr_mult == 0.1 and ddcn.conv_mask.lr_mult == 0.1
    mp = pkg.ModulatedDeformConvPack(8, 8, 3, padding=2, dilation=2, deformable_groups=4, bias=False)
    assert sorted(mp.state_dict()) == ["conv_offset_mask.bias", "conv_offset_mask.weight", "weight"]
    assert mp.conv_offset_mask.weight.shape == (4 * 27, 8, 3, 3)
    assert mp.conv_offset_mask.dilation == (2, 2)                # deviation C2
    dp = pkg.DeformConvPack(8, 8, 3, padding=1, deformable_groups=2)
    assert sorted(dp.state_dict()) == ["conv_offset.bias", "conv_offset.weight", "weight"]
    assert dp.conv_offset.weight.shape == (36, 8, 3, 3)
    with pytest.raises(AssertionError):
        pkg.DepthDeformConvPack(8, 8, 3)                         # depth_channels is mandatory (:425)


def test_package_exports_match_reference_ops_init():
    # detseg/ops/dcn/__init__.py:1-13 minus the out-of-scope deform_pool names
    for n in ["DeformConv", "DeformConvPack", "ModulatedDeformConv", "ModulatedDeformConvPack", "DepthawareConv",
              "DepthDeformConvPack", "deform_conv", "modulated_deform_conv", "depthaware_conv"]:
        assert hasattr(pkg, n)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "semantic-segmentation_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower(), f

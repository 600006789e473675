"""ORACLE -- numpy/ctypes front end of oracle/libddcn_ref.so.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.  The C code behind it restates the
reference's CUDA kernels and host loops (citations in ddcn_ref_impl.inc).

The argument order of the whole-op functions follows the reference's
autograd.Functions (detseg/ops/dcn/deform_conv.py:15-24, 86-96, 159-167):
scalar stride/padding/dilation apply to both axes.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libddcn_ref.so")


class Geom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("N", "C", "H", "W", "Co", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw",
                 "groups", "dg", "Ho", "Wo")]


def build():
    """Compile the oracle (gcc). Building the checker is not using it."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.ddcn_ref_max_threads.restype = ctypes.c_int
    return _lib


def max_threads():
    return int(lib().ddcn_ref_max_threads())


def set_threads(n):
    lib().ddcn_ref_set_threads(ctypes.c_int(int(n)))


def out_hw(H, W, kh, kw, stride, padding, dilation):
    # detseg/ops/dcn/deform_conv.py:71-80 (_infer_shape), floor division
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    return Ho, Wo


def make_geom(xshape, wshape, stride, padding, dilation, groups, dg):
    N, C, H, W = xshape
    Co, Cg, kh, kw = wshape
    assert C == Cg * groups, "input channels != weight.size(1) * groups"
    Ho, Wo = out_hw(H, W, kh, kw, stride, padding, dilation)
    return Geom(N, C, H, W, Co, kh, kw, stride, stride, padding, padding, dilation, dilation,
                groups, dg, Ho, Wo)


def _suf(dt):
    if dt == np.float32:
        return "_f32"
    if dt == np.float64:
        return "_f64"
    raise TypeError("oracle supports float32/float64, got %s" % dt)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _call(name, dt, *args):
    fn = getattr(lib(), name + _suf(dt))
    fn.restype = None
    fn(*args)


# ----------------------------------------------------------------------------- stages (one image)
def im2col(x, offset, mask, kh, kw, stride, padding, dilation, dg):
    """x[C,H,W], offset[dg*2*K2,Ho,Wo], mask[dg*K2,Ho,Wo]|None -> col[C*K2, Ho*Wo]."""
    dt = x.dtype.type
    C, H, W = x.shape
    g = make_geom((1, C, H, W), (1, C, kh, kw), stride, padding, dilation, 1, dg)
    x, offset, mask = _c(x, dt), _c(offset, dt), _c(mask, dt)
    col = np.empty((C * kh * kw, g.Ho * g.Wo), dtype=dt)
    _call("ddcn_ref_im2col", dt, _p(x), _p(offset), _p(mask), _p(col), ctypes.byref(g))
    return col


def col2im(col, offset, mask, C, H, W, kh, kw, stride, padding, dilation, dg):
    dt = col.dtype.type
    g = make_geom((1, C, H, W), (1, C, kh, kw), stride, padding, dilation, 1, dg)
    col, offset, mask = _c(col, dt), _c(offset, dt), _c(mask, dt)
    gx = np.zeros((C, H, W), dtype=dt)
    _call("ddcn_ref_col2im", dt, _p(col), _p(offset), _p(mask), _p(gx), ctypes.byref(g))
    return gx


def col2im_coord(col, x, offset, mask, kh, kw, stride, padding, dilation, dg):
    dt = col.dtype.type
    C, H, W = x.shape
    g = make_geom((1, C, H, W), (1, C, kh, kw), stride, padding, dilation, 1, dg)
    col, x, offset, mask = _c(col, dt), _c(x, dt), _c(offset, dt), _c(mask, dt)
    goff = np.zeros_like(offset)
    gmask = np.zeros_like(mask) if mask is not None else None
    _call("ddcn_ref_col2im_coord", dt, _p(col), _p(x), _p(offset), _p(mask), _p(goff), _p(gmask),
          ctypes.byref(g))
    return goff, gmask


def depthaware_im2col(x, depth, kh, kw, stride, padding, dilation):
    dt = x.dtype.type
    C, H, W = x.shape
    g = make_geom((1, C, H, W), (1, C, kh, kw), stride, padding, dilation, 1, 1)
    x, depth = _c(x, dt), _c(depth, dt)
    col = np.empty((C * kh * kw, g.Ho * g.Wo), dtype=dt)
    _call("ddcn_ref_depthaware_im2col", dt, _p(x), _p(depth), _p(col), ctypes.byref(g))
    return col


def depthaware_col2im(col, depth, C, H, W, kh, kw, stride, padding, dilation):
    dt = col.dtype.type
    g = make_geom((1, C, H, W), (1, C, kh, kw), stride, padding, dilation, 1, 1)
    col, depth = _c(col, dt), _c(depth, dt)
    gx = np.zeros((C, H, W), dtype=dt)
    _call("ddcn_ref_depthaware_col2im", dt, _p(col), _p(depth), _p(gx), ctypes.byref(g))
    return gx


# ----------------------------------------------------------------------------- whole ops
def deform_forward(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                   deformable_groups=1):
    """mask=None -> DeformConvFunction.forward, else ModulatedDeformConvFunction.forward."""
    dt = x.dtype.type
    g = make_geom(x.shape, weight.shape, stride, padding, dilation, groups, deformable_groups)
    x, offset, mask, weight, bias = _c(x, dt), _c(offset, dt), _c(mask, dt), _c(weight, dt), _c(bias, dt)
    K2 = g.kh * g.kw
    assert offset.shape == (g.N, g.dg * 2 * K2, g.Ho, g.Wo), (offset.shape, (g.N, g.dg * 2 * K2, g.Ho, g.Wo))
    if mask is not None:
        assert mask.shape == (g.N, g.dg * K2, g.Ho, g.Wo)
    out = np.empty((g.N, g.Co, g.Ho, g.Wo), dtype=dt)
    col = np.empty((g.C * K2, g.Ho * g.Wo), dtype=dt)
    _call("ddcn_ref_deform_forward", dt, _p(x), _p(weight), _p(bias), _p(offset), _p(mask), _p(out), _p(col),
          ctypes.byref(g))
    return out


def deform_backward(x, offset, mask, weight, grad_out, with_bias=False, stride=1, padding=0, dilation=1,
                    groups=1, deformable_groups=1):
    """Returns (gx, goff, gmask|None, gw, gb|None) as the reference's backward tuples do."""
    dt = x.dtype.type
    g = make_geom(x.shape, weight.shape, stride, padding, dilation, groups, deformable_groups)
    x, offset, mask, weight, grad_out = _c(x, dt), _c(offset, dt), _c(mask, dt), _c(weight, dt), _c(grad_out, dt)
    K2 = g.kh * g.kw
    gx = np.zeros_like(x)
    goff = np.zeros_like(offset)
    gmask = np.zeros_like(mask) if mask is not None else None
    gw = np.zeros_like(weight)
    gb = np.zeros((g.Co,), dtype=dt) if with_bias else None
    col = np.empty((g.C * K2, g.Ho * g.Wo), dtype=dt)
    _call("ddcn_ref_deform_backward", dt, _p(x), _p(weight), _p(offset), _p(mask), _p(grad_out), _p(gx), _p(goff),
          _p(gmask), _p(gw), _p(gb), _p(col), ctypes.byref(g))
    return gx, goff, gmask, gw, gb


def depthaware_forward(x, depth, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    dt = x.dtype.type
    g = make_geom(x.shape, weight.shape, stride, padding, dilation, groups, 1)
    x, depth, weight, bias = _c(x, dt), _c(depth, dt), _c(weight, dt), _c(bias, dt)
    assert depth.shape == (g.N, 1, g.H, g.W)
    out = np.empty((g.N, g.Co, g.Ho, g.Wo), dtype=dt)
    col = np.empty((g.C * g.kh * g.kw, g.Ho * g.Wo), dtype=dt)
    _call("ddcn_ref_depthaware_forward", dt, _p(x), _p(depth), _p(weight), _p(bias), _p(out), _p(col),
          ctypes.byref(g))
    return out


def depthaware_backward(x, depth, weight, grad_out, with_bias=False, stride=1, padding=0, dilation=1, groups=1):
    dt = x.dtype.type
    g = make_geom(x.shape, weight.shape, stride, padding, dilation, groups, 1)
    x, depth, weight, grad_out = _c(x, dt), _c(depth, dt), _c(weight, dt), _c(grad_out, dt)
    gx = np.zeros_like(x)
    gw = np.zeros_like(weight)
    gb = np.zeros((g.Co,), dtype=dt) if with_bias else None
    col = np.empty((g.C * g.kh * g.kw, g.Ho * g.Wo), dtype=dt)
    _call("ddcn_ref_depthaware_backward", dt, _p(x), _p(depth), _p(weight), _p(grad_out), _p(gx), _p(gw), _p(gb),
          _p(col), ctypes.byref(g))
    return gx, gw, gb


# ----------------------------------------------------------------------------- deformable PS-RoI pooling
class RoiGeom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("B", "C", "H", "W", "R", "out_channels", "pooled", "group_size", "part_size",
                 "sample_per_part", "trans_channels", "no_trans")] + \
               [("spatial_scale", ctypes.c_float), ("trans_std", ctypes.c_float)]


def _roi_geom(data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size, part_size,
              sample_per_part, trans_std):
    # argument meaning of DeformRoIPoolingFunction.forward, detseg/ops/dcn/deform_pool.py:13-38
    B, C, H, W = data.shape
    part_size = out_size if part_size is None else part_size
    tc = 2 if no_trans else offset.shape[1]
    return RoiGeom(B, C, H, W, rois.shape[0], out_channels, out_size, group_size, part_size, sample_per_part,
                   tc, int(bool(no_trans)), spatial_scale, trans_std)


def psroi_forward(data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size=1,
                  part_size=None, sample_per_part=4, trans_std=0.0):
    """Returns (output, output_count), both [R, out_channels, out_size, out_size]."""
    dt = data.dtype.type
    g = _roi_geom(data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size, part_size,
                  sample_per_part, trans_std)
    data, rois = _c(data, dt), _c(rois, dt)
    offset = None if no_trans else _c(offset, dt)
    out = np.empty((g.R, out_channels, out_size, out_size), dtype=dt)
    cnt = np.empty_like(out)
    _call("ddcn_ref_psroi_forward", dt, _p(data), _p(rois), _p(offset), _p(out), _p(cnt), ctypes.byref(g))
    return out, cnt


def psroi_backward(grad_out, data, rois, offset, out_count, spatial_scale, out_size, out_channels, no_trans,
                   group_size=1, part_size=None, sample_per_part=4, trans_std=0.0):
    """Returns (grad_data, grad_offset|None)."""
    dt = data.dtype.type
    g = _roi_geom(data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size, part_size,
                  sample_per_part, trans_std)
    grad_out, data, rois, out_count = _c(grad_out, dt), _c(data, dt), _c(rois, dt), _c(out_count, dt)
    offset = None if no_trans else _c(offset, dt)
    gdata = np.zeros_like(data)
    goff = None if no_trans else np.zeros_like(offset)
    _call("ddcn_ref_psroi_backward", dt, _p(grad_out), _p(data), _p(rois), _p(offset), _p(out_count), _p(gdata),
          _p(goff), ctypes.byref(g))
    return gdata, goff

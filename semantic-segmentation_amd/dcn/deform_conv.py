"""Python boundary of the DDCN operator: the three autograd.Functions, their functional aliases and
the six nn.Modules of the reference's `detseg/ops/dcn/deform_conv.py` (PY below), with the same
names, positional argument order, defaults, saved tensors, gradient tuples and state-dict keys.

Where the reference called its pybind CUDA extension (PY:41-45, 58-63, 113-117, 131-136, 182-186,
198-203) this file calls the C ABI of libddcn_hip.so (include/ddcn_hip.h) on torch's current HIP
stream.  CPU tensors raise NotImplementedError exactly as PY:33-34, 51-52 do; there is no fallback.

Deliberate deviations from the reference:
  C1  DeformConvPack.forward passes self.bias (PY:296-297 dropped it, shifting every argument).
  C2  ModulatedDeformConvPack.conv_offset_mask gets dilation=self.dilation (PY:352-359 omitted it, so
      dilated layers produced a wrongly sized offset map that the CUDA kernel mis-indexed silently).
  C4  offset / mask / depth shapes are checked and made contiguous (the reference read garbage).
  C5  kernels run on torch's current stream, not the legacy default stream.
  C6  the columns scratch is torch.empty (never zero-filled) and comes from the caching allocator.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import _lib


def _geom(input, weight, stride, padding, dilation, groups, deformable_groups):
    if input.dim() != 4:
        raise RuntimeError("4D input tensor expected but got: %d" % input.dim())
    if weight.dim() != 4:
        raise RuntimeError("4D weight tensor (nOutputPlane,nInputPlane,kH,kW) expected, but got: %d" % weight.dim())
    n, c, h, w = input.shape
    co, _, kh, kw = weight.shape
    if c != weight.shape[1] * groups:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (c, weight.shape[1] * groups))
    g = _lib.Geom(n, c, h, w, co, kh, kw, stride, stride, padding, padding, dilation, dilation, groups,
                  deformable_groups, 0, 0)
    _lib.check(_lib.lib().ddcn_infer_output(g))
    _lib.check(_lib.lib().ddcn_check_geom(g))
    return g


_DTYPES = (torch.float32, torch.float64, torch.float16, torch.bfloat16)


def _check_dtype_cuda(name, t, dtype):
    """The reference dispatches on FLOATING_TYPES_AND_HALF (deform_conv_cuda_kernel.cu:258).  Here float32 and
    float64 have their own kernels; half / bfloat16 tensors are computed by the fp32 kernels (converted on the way
    in and out, so the result is at least as accurate as a native half kernel), bfloat16 inference by the bf16 ones."""
    if not t.is_cuda:
        raise NotImplementedError
    if dtype not in _DTYPES:
        raise RuntimeError("ddcn: %s dtype not supported (float32, float64, float16, bfloat16)" % (dtype,))
    if t.dtype != dtype:
        raise RuntimeError("ddcn: %s is %s but input is %s; all tensors of one call must share a dtype"
                           % (name, t.dtype, dtype))


def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


def _cdtype(dtype):
    """Dtype the kernels compute a call in: float64 tensors by the fp64 kernels, everything else in fp32."""
    return torch.float64 if dtype == torch.float64 else torch.float32


def _cast(t, cd):
    return t if t.dtype == cd else t.to(cd)


def _entry(name, cd):
    """The C ABI entry point `name` for compute dtype cd (ddcn_*_f32 / ddcn_*_f64)."""
    return getattr(_lib.lib(), name + ("_f64" if cd == torch.float64 else "_f32"))


def _contig(name, t):
    # deform_conv_cuda.cpp:179-180: input and weight must be contiguous
    if not t.is_contiguous():
        raise RuntimeError("%s tensor has to be contiguous" % name)


# Keep the forward call's columns for backward (no second im2col) at the price of holding N*C*K2*ld floats per layer
# between forward and backward; DDCN_SAVE_COLUMNS=0 restores the reference's recompute (deform_conv_cuda.cpp:325-327).
SAVE_COLUMNS = os.environ.get("DDCN_SAVE_COLUMNS", "1") != "0"


# Inference on bfloat16 / float16 tensors (no tensor of the call requires grad) runs the native 16-bit kernels: 16-bit
# columns, MFMA with fp32 accumulation (BASELINE config 5).  DDCN_BF16_NATIVE=0, a geometry they do not cover, or any tensor that
# requires grad selects the fp32 kernels (convert in, compute, convert out).
BF16_NATIVE = os.environ.get("DDCN_BF16_NATIVE", "1") != "0"
# DDCN_BF16_FUSED=1: the 16-bit forward without a columns buffer (ddcn_modulated_deform_conv_forward_fused_{bf16,f16})
# where it covers the geometry.  Opt-in: equal to im2col + GEMM at C = 128, slower beyond (profiles/r03c_fused16.txt).
BF16_FUSED = os.environ.get("DDCN_BF16_FUSED", "0") == "1"


# Saved-columns fp32 backward on two streams (ddcn_deform_conv_backward_saved_2s_f32): the weight-gradient GEMM runs on
# a side stream beside the LDS-atomic-bound scatter.  Opt-in since round 3 (DDCN_TWO_STREAM_BACKWARD=1): with the packed
# fixed-point scatter (0.35 -> 0.31 ms alone) one stream is the faster schedule -- bench layer 2.449 vs 2.481 ms, DDCN-R101
# step 465.3 vs 466.6 ms, DDCN-R50 81.2 vs 81.9 ms (profiles/r03a_schedules.txt) -- and every kernel then runs alone, as
# bench.py times it.
TWO_STREAM_BACKWARD = os.environ.get("DDCN_TWO_STREAM_BACKWARD", "0") == "1"
# the mask branch's direct backward on two streams (ddcn_deform_conv_direct_backward_2s_f32): measured -0.03 ms on the
# 3.46 ms layer, so it is opt-in
TWO_STREAM_DIRECT = os.environ.get("DDCN_TWO_STREAM_DIRECT", "0") == "1"
# 16-bit backward: grad_weight's chain (16-bit im2col, grad_out gather, GEMM) on a side stream beside G2 and the scatter
# (1.17 vs 1.23 ms on the bench layer in round 2).  DDCN_TWO_STREAM_16BIT=0: one stream.
TWO_STREAM_16BIT = os.environ.get("DDCN_TWO_STREAM_16BIT", "1") != "0"
_side = {}


def _side_stream(device):
    """(side stream, fork event, join event) of a device, created once; the events are recorded once so that their
    HIP handles exist before the library is handed them."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _side:
        with torch.cuda.device(device):
            stream = torch.cuda.Stream()
            fork, join = torch.cuda.Event(), torch.cuda.Event()
            fork.record(stream)
            join.record(stream)
        _side[key] = (stream, fork, join)
    return _side[key]


def _backward_saved(cd, input, weight, offset, mask, grad_output, cols, grad_input, grad_offset, grad_mask, grad_weight,
                    grad_bias, ws, ws_bytes, g):
    """One saved-columns backward call (mask / grad_mask None: the non-modulated op) on input's device."""
    p = _lib.ptr
    stream = torch.cuda.current_stream(input.device)
    if (TWO_STREAM_BACKWARD and cols is not None and cd == torch.float32
            and not torch.cuda.is_current_stream_capturing()):
        side, fork, join = _side_stream(input.device)
        _lib.check(_lib.lib().ddcn_deform_conv_backward_saved_2s_f32(
            p(input), p(weight), p(offset), p(mask), p(grad_output), p(cols), p(grad_input), p(grad_offset),
            p(grad_mask), p(grad_weight), p(grad_bias), p(ws), ws_bytes, g, ctypes.c_void_p(stream.cuda_stream),
            ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(fork.cuda_event), ctypes.c_void_p(join.cuda_event)))
        return
    s = ctypes.c_void_p(stream.cuda_stream)
    if mask is None:
        _lib.check(_entry("ddcn_deform_conv_backward_saved", cd)(
            p(input), p(weight), p(offset), p(grad_output), p(cols), p(grad_input), p(grad_offset), p(grad_weight),
            p(grad_bias), p(ws), ws_bytes, g, s))
    else:
        _lib.check(_entry("ddcn_modulated_deform_conv_backward_saved", cd)(
            p(input), p(weight), p(offset), p(mask), p(grad_output), p(cols), p(grad_input), p(grad_offset),
            p(grad_mask), p(grad_weight), p(grad_bias), p(ws), ws_bytes, g, s))


# Training on bfloat16 / float16 tensors: forward on the native 16-bit kernels (16-bit columns, not kept), backward with
# G2 and G3 on the 16-bit operands (ddcn_gemm_gcol_{bf16,f16}; ddcn_deform_conv_gweight_{bf16,f16}, which recomputes the
# columns in 16 bits as pixel octets) and the scatter on the fp32 kernel (fp32 column gradients, fp32 atomics).
# DDCN_16BIT_TRAIN_NATIVE=0: everything on the fp32 kernels (convert in, compute, convert out), as in round 1.
TRAIN16_NATIVE = os.environ.get("DDCN_16BIT_TRAIN_NATIVE", "1") != "0"


def _backward_16bit_covers(dtype, g):
    """True when backward of a call on `dtype` tensors will take _backward_16bit (which recomputes its columns in 16 bits):
    the forward then keeps no fp32 columns for it."""
    return (dtype in (torch.bfloat16, torch.float16) and TRAIN16_NATIVE
            and _lib.lib().ddcn_gemm_gcol_16_supported(g) == 0)


def _backward_16bit(dtype, input, offset, mask, weight, grad_output, with_bias, g):
    """(grad_input, grad_offset, grad_mask | None, grad_weight, grad_bias | None) in `dtype`, or None when the 16-bit
    G2 does not cover the geometry.  All tensors are 16-bit; mask None = the non-modulated op."""
    L = _lib.lib()
    if not _backward_16bit_covers(dtype, g):
        return None
    p = _lib.ptr
    grad_output = grad_output.contiguous()
    x32, off32 = input.float(), offset.float()
    g3_16 = L.ddcn_gweight_16_supported(g) == 0
    gout32 = grad_output.float() if (with_bias or not g3_16) else None
    mask32 = None if mask is None else mask.float()
    acts = [x32, off32] + ([mask32] if mask is not None else [])
    grads, grads_flat = _zeros_flat(acts, with_flat=True)
    gx, goff = grads[0], grads[1]
    gmask = grads[2] if mask is not None else None
    gw = torch.zeros(weight.shape, dtype=torch.float32, device=input.device)
    gb = torch.zeros(g.Co, dtype=torch.float32, device=input.device) if with_bias else None
    ws_bytes = L.ddcn_workspace_bytes(g)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=input.device)
    ws16_bytes = L.ddcn_gemm_gcol_16_workspace_bytes(g)
    ws16 = torch.empty(ws16_bytes // 2, dtype=dtype, device=input.device)
    gemm = L.ddcn_gemm_gcol_bf16 if dtype == torch.bfloat16 else L.ddcn_gemm_gcol_f16
    with torch.cuda.device(input.device):
        main = torch.cuda.current_stream(input.device)
        s = _lib.current_stream(input.device)
        gw16 = L.ddcn_deform_conv_gweight_bf16 if dtype == torch.bfloat16 else L.ddcn_deform_conv_gweight_f16
        two = g3_16 and TWO_STREAM_16BIT and not torch.cuda.is_current_stream_capturing()
        if two:
            # grad_weight needs only the inputs and grad_out: its chain (16-bit im2col, grad_out gather, GEMM) runs on the
            # side stream beside G2 and the LDS-atomic-bound scatter, in a workspace of its own
            side, fork, join = _side_stream(input.device)
            wsw_bytes = L.ddcn_gweight_16_workspace_bytes(g)
            wsw = torch.empty(wsw_bytes, dtype=torch.uint8, device=input.device)
            fork.record(main)
            side.wait_event(fork)
            _lib.check(gw16(p(input), p(offset), p(mask), p(grad_output), p(gw), p(wsw), wsw_bytes, g,
                            ctypes.c_void_p(side.cuda_stream)))
            join.record(side)
        _lib.check(gemm(p(weight), p(grad_output), p(ws), p(ws16), ws16_bytes, g, s))               # G2, 16-bit operands
        _lib.check(L.ddcn_col2im_coord_f32(p(ws), p(x32), p(off32), p(mask32), p(gx), p(goff), p(gmask), g, s))
        if two:
            main.wait_event(join)
        elif g3_16:
            # G3 on 16-bit operands: 16-bit columns as pixel octets (recomputed, never fp32), fp32 accumulation
            wsw_bytes = L.ddcn_gweight_16_workspace_bytes(g)
            wsw = ws.view(torch.uint8)[:wsw_bytes] if wsw_bytes <= ws_bytes else \
                torch.empty(wsw_bytes, dtype=torch.uint8, device=input.device)
            _lib.check(gw16(p(input), p(offset), p(mask), p(grad_output), p(gw), p(wsw), wsw_bytes, g, s))
        else:
            _lib.check(L.ddcn_im2col_f32(p(x32), p(off32), p(mask32), p(ws), g, s))                   # columns for G3
            _lib.check(L.ddcn_gemm_gweight_f32(p(gout32), p(ws), p(gw), g, s))
        if with_bias:
            _lib.check(L.ddcn_bias_grad_f32(p(gout32), p(gb), g, s))
    # the activation gradients share one flat fp32 buffer: one conversion kernel for the three of them
    g16 = _flat_views(grads_flat.to(dtype), acts)
    cv = lambda t: None if t is None else t.to(dtype)
    return g16[0], g16[1], (g16[2] if mask is not None else None), cv(gw), cv(gb)


def _bf16_forward(input, offset, mask, weight, bias, g):
    """-> output, or None when the native bf16 path does not cover this call."""
    L = _lib.lib()
    if not BF16_NATIVE or L.ddcn_bf16_supported(g) != 0:
        return None
    output = input.new_empty((g.N, g.Co, g.Ho, g.Wo))
    if BF16_FUSED and L.ddcn_fused16_supported(g) == 0:
        entry = (L.ddcn_modulated_deform_conv_forward_fused_bf16 if input.dtype == torch.bfloat16
                 else L.ddcn_modulated_deform_conv_forward_fused_f16)
        nbytes = L.ddcn_fused16_workspace_bytes(g)
    else:
        entry = (L.ddcn_modulated_deform_conv_forward_bf16 if input.dtype == torch.bfloat16
                 else L.ddcn_modulated_deform_conv_forward_f16)
        nbytes = L.ddcn_workspace_bytes_bf16(g)
    ws = torch.empty(nbytes // 2, dtype=input.dtype, device=input.device)
    with torch.cuda.device(input.device):
        _lib.check(entry(
            _lib.ptr(input), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(offset), _lib.ptr(mask), _lib.ptr(output),
            _lib.ptr(ws), nbytes, g, _lib.current_stream(input.device)))
    return output


def _workspace(input, g, cd=torch.float32):
    if cd == torch.float64:
        nbytes = _lib.lib().ddcn_workspace_bytes_f64(g)
        return torch.empty(nbytes // 8, dtype=torch.float64, device=input.device), nbytes
    nbytes = _lib.lib().ddcn_workspace_bytes(g)
    return torch.empty(nbytes // 4, dtype=torch.float32, device=input.device), nbytes


def _flat_views(flat, tensors):
    """Views of `flat` shaped like `tensors`, each starting on a multiple of 8 elements (16 bytes for 16-bit elements)."""
    out, o = [], 0
    for t in tensors:
        out.append(flat[o:o + t.numel()].view(t.shape))
        o += (t.numel() + 7) // 8 * 8
    return out


def _zeros_flat(tensors, dtype=None, with_flat=False):
    total = sum((t.numel() + 7) // 8 * 8 for t in tensors)
    flat = torch.zeros(total, dtype=dtype or tensors[0].dtype, device=tensors[0].device)
    views = _flat_views(flat, tensors)
    return (views, flat) if with_flat else views


def _zeros_like_all(*tensors):
    """Zero-filled gradients for all tensors (each view 16-byte aligned).  The activation-sized gradients (every
    tensor but the last two: ..., weight, bias) share ONE flat buffer and one fill kernel; grad_weight and grad_bias
    come from a small buffer of their own, because autograd keeps them alive as `param.grad` long after the
    activation gradients are gone and a view would pin the whole activation-sized buffer with it."""
    return _zeros_flat(tensors[:-2]) + _zeros_flat(tensors[-2:])


def _expect_shape(name, t, shape):
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError("invalid shape of %s, expected %s but got %s" % (name, tuple(shape), tuple(t.shape)))


def _infer_shape(ctx, input, weight):
    """(N, Co, Ho, Wo) with floor division -- PY:71-80, 143-153, 210-220."""
    kh, kw = weight.shape[2:4]
    ho = (input.size(2) + 2 * ctx.padding - (ctx.dilation * (kh - 1) + 1)) // ctx.stride + 1
    wo = (input.size(3) + 2 * ctx.padding - (ctx.dilation * (kw - 1) + 1)) // ctx.stride + 1
    return input.size(0), weight.size(0), ho, wo


class DeformConvFunction(Function):

    @staticmethod
    def forward(ctx,
                input,
                offset,
                weight,
                bias=None,
                stride=1,
                padding=0,
                dilation=1,
                groups=1,
                deformable_groups=1):
        ctx.stride = stride
        ctx.padding = padding
        ctx.dilation = dilation
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.with_bias = bias is not None
        if not ctx.with_bias:
            bias = input.new_empty(1)  # fake tensor
        if not input.is_cuda:
            raise NotImplementedError
        for name, t in (("input", input), ("offset", offset), ("weight", weight), ("bias", bias)):
            _check_dtype_cuda(name, t, input.dtype)
        ctx.dtype = input.dtype
        cd = _cdtype(ctx.dtype)
        _contig("input", input)
        _contig("weight", weight)
        g = _geom(input, weight, stride, padding, dilation, groups, deformable_groups)
        k2 = g.kh * g.kw
        _expect_shape("offset", offset, (g.N, deformable_groups * 2 * k2, g.Ho, g.Wo))
        offset = offset.contiguous()
        needs_grad = weight.requires_grad or offset.requires_grad or input.requires_grad
        if needs_grad:
            ctx.save_for_backward(input, offset, weight, bias)
        if ctx.dtype in (torch.bfloat16, torch.float16) and (not needs_grad or TRAIN16_NATIVE):
            output = _bf16_forward(input, offset, None, weight, bias if ctx.with_bias else None, g)
            if output is not None:
                ctx.columns = None
                return output
        input, offset, weight, bias = (_cast(t, cd) for t in (input, offset, weight, bias))
        output = input.new_empty(
            DeformConvFunction._infer_shape(ctx, input, weight))
        ws, ws_bytes = _workspace(input, g, cd)
        with torch.cuda.device(input.device):
            _lib.check(_entry("ddcn_deform_conv_forward", cd)(
                _lib.ptr(input), _lib.ptr(weight), _lib.ptr(bias) if ctx.with_bias else None, _lib.ptr(offset),
                _lib.ptr(output), _lib.ptr(ws), ws_bytes, g, _lib.current_stream(input.device)))
        ctx.columns = ws if (needs_grad and SAVE_COLUMNS and not _backward_16bit_covers(ctx.dtype, g)) else None
        return output if ctx.dtype == cd else output.to(ctx.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        cd = _cdtype(ctx.dtype)
        if ctx.dtype in (torch.bfloat16, torch.float16):
            input, offset, weight, bias = ctx.saved_tensors
            g = _geom(input, weight, ctx.stride, ctx.padding, ctx.dilation, ctx.groups, ctx.deformable_groups)
            _expect_shape("grad_output", grad_output, (g.N, g.Co, g.Ho, g.Wo))
            res = _backward_16bit(ctx.dtype, input, offset, None, weight, grad_output, ctx.with_bias, g)
            if res is not None:
                ctx.columns = None
                return (res[0], res[1], res[3], res[4], None, None, None, None, None)
        input, offset, weight, bias = (_cast(t, cd) for t in ctx.saved_tensors)
        grad_output = _cast(grad_output.contiguous(), cd)
        grad_input, grad_offset, grad_weight, grad_bias = _zeros_like_all(input, offset, weight, bias)
        g = _geom(input, weight, ctx.stride, ctx.padding, ctx.dilation, ctx.groups, ctx.deformable_groups)
        _expect_shape("grad_output", grad_output, (g.N, g.Co, g.Ho, g.Wo))
        ws, ws_bytes = _workspace(input, g, cd)
        with torch.cuda.device(input.device):
            cols, ctx.columns = getattr(ctx, "columns", None), None
            _backward_saved(cd, input, weight, offset, None, grad_output, cols, grad_input, grad_offset, None,
                            grad_weight, grad_bias if ctx.with_bias else None, ws, ws_bytes, g)
        if not ctx.with_bias:
            grad_bias = None
        if ctx.dtype != cd:
            grad_input, grad_offset, grad_weight, grad_bias = (
                None if t is None else t.to(ctx.dtype) for t in (grad_input, grad_offset, grad_weight, grad_bias))

        return (grad_input, grad_offset, grad_weight, grad_bias,
                None, None, None, None, None)

    _infer_shape = staticmethod(_infer_shape)


class ModulatedDeformConvFunction(Function):

    @staticmethod
    def forward(ctx,
                input,
                offset,
                mask,
                weight,
                bias=None,
                stride=1,
                padding=0,
                dilation=1,
                groups=1,
                deformable_groups=1):
        ctx.stride = stride
        ctx.padding = padding
        ctx.dilation = dilation
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.with_bias = bias is not None
        if not ctx.with_bias:
            bias = input.new_empty(1)  # fake tensor
        if not input.is_cuda:
            raise NotImplementedError
        for name, t in (("input", input), ("offset", offset), ("mask", mask), ("weight", weight), ("bias", bias)):
            _check_dtype_cuda(name, t, input.dtype)
        ctx.dtype = input.dtype
        cd = _cdtype(ctx.dtype)
        _contig("input", input)
        _contig("weight", weight)
        g = _geom(input, weight, stride, padding, dilation, groups, deformable_groups)
        k2 = g.kh * g.kw
        _expect_shape("offset", offset, (g.N, deformable_groups * 2 * k2, g.Ho, g.Wo))
        _expect_shape("mask", mask, (g.N, deformable_groups * k2, g.Ho, g.Wo))
        offset = offset.contiguous()
        mask = mask.contiguous()
        needs_grad = weight.requires_grad or mask.requires_grad or offset.requires_grad or input.requires_grad
        if needs_grad:
            ctx.save_for_backward(input, offset, mask, weight, bias)
        if ctx.dtype in (torch.bfloat16, torch.float16) and (not needs_grad or TRAIN16_NATIVE):
            output = _bf16_forward(input, offset, mask, weight, bias if ctx.with_bias else None, g)
            if output is not None:
                ctx.columns = None
                return output
        input, offset, mask, weight, bias = (_cast(t, cd) for t in (input, offset, mask, weight, bias))
        output = input.new_empty(
            ModulatedDeformConvFunction._infer_shape(ctx, input, weight))
        ws, ws_bytes = _workspace(input, g, cd)
        with torch.cuda.device(input.device):
            _lib.check(_entry("ddcn_modulated_deform_conv_forward", cd)(
                _lib.ptr(input), _lib.ptr(weight), _lib.ptr(bias) if ctx.with_bias else None, _lib.ptr(offset),
                _lib.ptr(mask), _lib.ptr(output), _lib.ptr(ws), ws_bytes, g, _lib.current_stream(input.device)))
        ctx.columns = ws if (needs_grad and SAVE_COLUMNS and not _backward_16bit_covers(ctx.dtype, g)) else None
        return output if ctx.dtype == cd else output.to(ctx.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        cd = _cdtype(ctx.dtype)
        if ctx.dtype in (torch.bfloat16, torch.float16):
            input, offset, mask, weight, bias = ctx.saved_tensors
            g = _geom(input, weight, ctx.stride, ctx.padding, ctx.dilation, ctx.groups, ctx.deformable_groups)
            _expect_shape("grad_output", grad_output, (g.N, g.Co, g.Ho, g.Wo))
            res = _backward_16bit(ctx.dtype, input, offset, mask, weight, grad_output, ctx.with_bias, g)
            if res is not None:
                ctx.columns = None
                return res + (None, None, None, None, None)
        input, offset, mask, weight, bias = (_cast(t, cd) for t in ctx.saved_tensors)
        grad_output = _cast(grad_output.contiguous(), cd)
        grad_input, grad_offset, grad_mask, grad_weight, grad_bias = _zeros_like_all(input, offset, mask, weight, bias)
        g = _geom(input, weight, ctx.stride, ctx.padding, ctx.dilation, ctx.groups, ctx.deformable_groups)
        _expect_shape("grad_output", grad_output, (g.N, g.Co, g.Ho, g.Wo))
        ws, ws_bytes = _workspace(input, g, cd)
        with torch.cuda.device(input.device):
            cols, ctx.columns = getattr(ctx, "columns", None), None
            _backward_saved(cd, input, weight, offset, mask, grad_output, cols, grad_input, grad_offset, grad_mask,
                            grad_weight, grad_bias if ctx.with_bias else None, ws, ws_bytes, g)
        if not ctx.with_bias:
            grad_bias = None
        if ctx.dtype != cd:
            grad_input, grad_offset, grad_mask, grad_weight, grad_bias = (
                None if t is None else t.to(ctx.dtype)
                for t in (grad_input, grad_offset, grad_mask, grad_weight, grad_bias))

        return (grad_input, grad_offset, grad_mask, grad_weight, grad_bias,
                None, None, None, None, None)

    _infer_shape = staticmethod(_infer_shape)


class DepthawareConvFunction(Function):

    @staticmethod
    def forward(ctx,
                input,
                depth,
                weight,
                bias=None,
                stride=1,
                padding=0,
                dilation=1,
                groups=1):
        ctx.stride = stride
        ctx.padding = padding
        ctx.dilation = dilation
        ctx.groups = groups
        ctx.with_bias = bias is not None
        if not ctx.with_bias:
            bias = input.new_empty(1)  # fake tensor
        if not input.is_cuda:
            raise NotImplementedError
        for name, t in (("input", input), ("depth", depth), ("weight", weight), ("bias", bias)):
            _check_dtype_cuda(name, t, input.dtype)
        ctx.dtype = input.dtype
        cd = _cdtype(ctx.dtype)
        _contig("input", input)
        _contig("weight", weight)
        g = _geom(input, weight, stride, padding, dilation, groups, 1)
        _expect_shape("depth", depth, (g.N, 1, g.H, g.W))
        depth = depth.contiguous()
        if weight.requires_grad or input.requires_grad:
            ctx.save_for_backward(input, depth, weight, bias)
        input, depth, weight, bias = (_cast(t, cd) for t in (input, depth, weight, bias))
        output = input.new_empty(
            DepthawareConvFunction._infer_shape(ctx, input, weight))
        ws, ws_bytes = _workspace(input, g, cd)
        with torch.cuda.device(input.device):
            _lib.check(_entry("ddcn_depthaware_conv_forward", cd)(
                _lib.ptr(input), _lib.ptr(depth), _lib.ptr(weight), _lib.ptr(bias) if ctx.with_bias else None,
                _lib.ptr(output), _lib.ptr(ws), ws_bytes, g, _lib.current_stream(input.device)))
        return output if ctx.dtype == cd else output.to(ctx.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        cd = _cdtype(ctx.dtype)
        input, depth, weight, bias = (_cast(t, cd) for t in ctx.saved_tensors)
        grad_output = _cast(grad_output.contiguous(), cd)
        grad_input = torch.zeros_like(input)
        grad_weight = torch.zeros_like(weight)
        grad_bias = torch.zeros_like(bias)
        g = _geom(input, weight, ctx.stride, ctx.padding, ctx.dilation, ctx.groups, 1)
        _expect_shape("grad_output", grad_output, (g.N, g.Co, g.Ho, g.Wo))
        ws, ws_bytes = _workspace(input, g, cd)
        with torch.cuda.device(input.device):
            _lib.check(_entry("ddcn_depthaware_conv_backward", cd)(
                _lib.ptr(input), _lib.ptr(depth), _lib.ptr(weight), _lib.ptr(grad_output), _lib.ptr(grad_input),
                _lib.ptr(grad_weight), _lib.ptr(grad_bias) if ctx.with_bias else None, _lib.ptr(ws), ws_bytes, g,
                _lib.current_stream(input.device)))
        if not ctx.with_bias:
            grad_bias = None
        if ctx.dtype != cd:
            grad_input, grad_weight, grad_bias = (
                None if t is None else t.to(ctx.dtype) for t in (grad_input, grad_weight, grad_bias))

        # one gradient slot per forward input (the reference returned a surplus trailing None, PY:207-208)
        return (grad_input, None, grad_weight, grad_bias,
                None, None, None, None)

    _infer_shape = staticmethod(_infer_shape)


class _DeformConvSigmoidFunction(Function):
    """sigmoid(deform_conv(input, offset, weight, bias)): the mask branch of the DDCN layer (PY:438-446, 457-458).

    Forward is ONE direct kernel for the few (9) output channels (ddcn_deform_conv_direct_forward_f32: no columns,
    no 9-row GEMM, sigmoid fused); backward applies sigmoid' and calls ddcn_deform_conv_direct_backward_f32 (the
    weight gradient without a second im2col or a 9-row GEMM)."""

    @staticmethod
    def forward(ctx, input, offset, weight, bias, stride, padding, dilation, groups, deformable_groups):
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups = groups, deformable_groups
        ctx.with_bias = bias is not None
        if not ctx.with_bias:
            bias = input.new_empty(1)  # fake tensor
        if not input.is_cuda:
            raise NotImplementedError
        for name, t in (("input", input), ("offset", offset), ("weight", weight), ("bias", bias)):
            _check_dtype_cuda(name, t, input.dtype)
        ctx.dtype = input.dtype
        cd = _cdtype(ctx.dtype)
        _contig("input", input)
        _contig("weight", weight)
        g = _geom(input, weight, stride, padding, dilation, groups, deformable_groups)
        _expect_shape("offset", offset, (g.N, deformable_groups * 2 * g.kh * g.kw, g.Ho, g.Wo))
        offset = offset.contiguous()
        saved = (input, offset, weight, bias)
        input, offset, weight, bias = (_cast(t, cd) for t in saved)
        output = input.new_empty((g.N, g.Co, g.Ho, g.Wo))
        ws, ws_bytes = _workspace(input, g, cd)
        with torch.cuda.device(input.device):
            _lib.check(_entry("ddcn_deform_conv_direct_forward", cd)(
                _lib.ptr(input), _lib.ptr(weight), _lib.ptr(bias) if ctx.with_bias else None, _lib.ptr(offset),
                _lib.ptr(output), 1, _lib.ptr(ws), ws_bytes, g, _lib.current_stream(input.device)))
        ctx.save_for_backward(*saved, output)
        return output if ctx.dtype == cd else output.to(ctx.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        cd = _cdtype(ctx.dtype)
        input, offset, weight, bias, out = ctx.saved_tensors
        input, offset, weight, bias = (_cast(t, cd) for t in (input, offset, weight, bias))
        grad_pre = (_cast(grad_output, cd) * out * (1.0 - out)).contiguous()      # d sigmoid
        grad_input, grad_offset, grad_weight, grad_bias = _zeros_like_all(input, offset, weight, bias)
        g = _geom(input, weight, ctx.stride, ctx.padding, ctx.dilation, ctx.groups, ctx.deformable_groups)
        _expect_shape("grad_output", grad_pre, (g.N, g.Co, g.Ho, g.Wo))
        ws, ws_bytes = _workspace(input, g, cd)
        with torch.cuda.device(input.device):
            args = (_lib.ptr(input), _lib.ptr(weight), _lib.ptr(offset), _lib.ptr(grad_pre), _lib.ptr(grad_input),
                    _lib.ptr(grad_offset), _lib.ptr(grad_weight), _lib.ptr(grad_bias) if ctx.with_bias else None,
                    _lib.ptr(ws), ws_bytes, g, _lib.current_stream(input.device))
            if (TWO_STREAM_DIRECT and cd == torch.float32
                    and not torch.cuda.is_current_stream_capturing()):
                # the weight-gradient kernel on the side stream beside the scatter (independent halves of the backward)
                side, fork, join = _side_stream(input.device)
                _lib.check(_lib.lib().ddcn_deform_conv_direct_backward_2s_f32(
                    *args, ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(fork.cuda_event),
                    ctypes.c_void_p(join.cuda_event)))
            else:
                _lib.check(_entry("ddcn_deform_conv_direct_backward", cd)(*args))
        if not ctx.with_bias:
            grad_bias = None
        if ctx.dtype != cd:
            grad_input, grad_offset, grad_weight, grad_bias = (
                None if t is None else t.to(ctx.dtype) for t in (grad_input, grad_offset, grad_weight, grad_bias))
        return (grad_input, grad_offset, grad_weight, grad_bias, None, None, None, None, None)


def _autocast_view(tensors):
    """Under torch.autocast("cuda", dtype) a custom Function gets its inputs as they come: 16-bit activations from the
    autocast convolutions before it, fp32 parameters.  The reference's Functions predate autocast and would fail on the
    mixed dtypes; here the functional aliases cast every floating CUDA tensor of the call to the autocast dtype (a
    differentiable cast: parameter gradients come back in fp32), so that the native 16-bit kernels run."""
    if not torch.is_autocast_enabled("cuda"):
        return tensors
    dt = torch.get_autocast_dtype("cuda")
    return tuple(t.to(dt) if isinstance(t, torch.Tensor) and t.is_cuda and t.is_floating_point() and t.dtype != dt else t
                 for t in tensors)


def _autocast_fp32_view(tensors):
    """For the Functions WITHOUT native 16-bit kernels (the mask branch of the DDCN layer, the depth-aware conv): under
    torch.autocast every floating CUDA tensor of the call is taken to float32 (the `custom_fwd(cast_inputs=torch.float32)`
    pattern) -- depth and the fp32 parameters keep their mantissa, no convert-in / convert-out kernels run inside the
    Function -- and only the result is cast to the autocast dtype.  -> (tensors, autocast dtype | None)."""
    if not torch.is_autocast_enabled("cuda"):
        return tensors, None
    return tuple(t.float() if isinstance(t, torch.Tensor) and t.is_cuda and t.is_floating_point()
                 and t.dtype in (torch.bfloat16, torch.float16) else t for t in tensors), torch.get_autocast_dtype("cuda")


def _inference_view(tensors):
    """Inside Function.forward a Parameter looks the same with and without torch.no_grad() (requires_grad stays
    True), so the reference saves its tensors for a backward that can never come.  The functional aliases below hand
    the Functions detached tensors when grad mode is off: nothing is saved, no columns are kept, and bfloat16
    calls take the native bf16 kernels."""
    if torch.is_grad_enabled():
        return tensors
    return tuple(t.detach() if isinstance(t, torch.Tensor) else t for t in tensors)


def deform_conv(input, offset, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
    """`DeformConvFunction.apply` (PY:226) with the reference's positional argument order."""
    input, offset, weight, bias = _inference_view(_autocast_view((input, offset, weight, bias)))
    return DeformConvFunction.apply(input, offset, weight, bias, stride, padding, dilation, groups,
                                    deformable_groups)


def modulated_deform_conv(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                          deformable_groups=1):
    """`ModulatedDeformConvFunction.apply` (PY:227) with the reference's positional argument order."""
    input, offset, mask, weight, bias = _inference_view(_autocast_view((input, offset, mask, weight, bias)))
    return ModulatedDeformConvFunction.apply(input, offset, mask, weight, bias, stride, padding, dilation, groups,
                                             deformable_groups)


def depthaware_conv(input, depth, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """`DepthawareConvFunction.apply` (PY:228) with the reference's positional argument order."""
    (input, depth, weight, bias), dt = _autocast_fp32_view((input, depth, weight, bias))
    input, depth, weight, bias = _inference_view((input, depth, weight, bias))
    out = DepthawareConvFunction.apply(input, depth, weight, bias, stride, padding, dilation, groups)
    return out if dt is None else out.to(dt)


class _DcnConvBase(nn.Module):
    """Parameter holder shared by the three plain modules: weight ~ U(-1/sqrt(Ci*K2), +) and zero bias
    (PY:261-268, 332-339, 410-415); attribute names and state-dict keys (`weight`, `bias`) as in the reference."""

    _default_bias = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=None):
        super().__init__()
        if bias is None:
            bias = self._default_bias
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.groups = groups
        self.deformable_groups = deformable_groups
        self.with_bias = bias
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // groups, *self.kernel_size))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        fan = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1. / math.sqrt(fan)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()

    def _offset_conv(self, in_channels, out_channels):
        """3x3 (kernel-sized) plain conv producing offsets / masks at the op's own output resolution."""
        return nn.Conv2d(in_channels, out_channels, kernel_size=self.kernel_size, stride=_pair(self.stride),
                         padding=_pair(self.padding), dilation=_pair(self.dilation), bias=True)


class DeformConv(_DcnConvBase):
    """PY:228-272.  forward(x, offset); bias defaults to False."""
    _default_bias = False

    def forward(self, x, offset):
        return deform_conv(x, offset, self.weight, self.bias, self.stride, self.padding,
                           self.dilation, self.groups, self.deformable_groups)


class DeformConvPack(DeformConv):

    def __init__(self, *args, **kwargs):
        super(DeformConvPack, self).__init__(*args, **kwargs)

        k2 = self.kernel_size[0] * self.kernel_size[1]
        # C2-style fix: the offset conv follows the op's dilation (PY:280-287 omitted it)
        self.conv_offset = self._offset_conv(self.in_channels, self.deformable_groups * 2 * k2)
        self.init_offset()

    def init_offset(self):
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        offset = self.conv_offset(x)
        # C1: the reference omitted self.bias here (PY:296-297)
        return deform_conv(x, offset, self.weight, self.bias, self.stride, self.padding,
                           self.dilation, self.groups, self.deformable_groups)


class ModulatedDeformConv(_DcnConvBase):
    """PY:300-344.  forward(x, offset, mask); bias defaults to True."""
    _default_bias = True

    def forward(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias,
                                     self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)


class ModulatedDeformConvPack(ModulatedDeformConv):

    def __init__(self, *args, **kwargs):
        super(ModulatedDeformConvPack, self).__init__(*args, **kwargs)

        k2 = self.kernel_size[0] * self.kernel_size[1]
        # C2: dilation added (PY:352-359 omitted it)
        self.conv_offset_mask = self._offset_conv(self.in_channels, self.deformable_groups * 3 * k2)
        self.init_offset()

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()
        self.conv_offset_mask.lr_mult = 0.1

    def forward(self, x):
        out = self.conv_offset_mask(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(mask)
        self.offset, self.mask = offset, mask
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias,
                                     self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)


class DepthawareConv(_DcnConvBase):
    """PY:378-419.  forward(x, depth); bias defaults to False; `deformable_groups` is accepted and unused."""
    _default_bias = False

    def reset_parameters(self):
        # PY:410-415 leaves a requested bias uninitialised; zero it here
        super().reset_parameters()

    def forward(self, x, depth):
        return depthaware_conv(x, depth, self.weight, self.bias, self.stride, self.padding,
                               self.dilation, self.groups)


class DepthDeformConvPack(ModulatedDeformConv):
    """The DDCN layer (PY:422-463): offsets from cat(x, depth), mask from a deformable conv over depth."""

    def __init__(self, *args, **kwargs):
        assert 'depth_channels' in kwargs
        depth_channels = kwargs.pop('depth_channels')
        super(DepthDeformConvPack, self).__init__(*args, **kwargs)

        k2 = self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset = self._offset_conv(self.in_channels + depth_channels, self.deformable_groups * 2 * k2)
        # positional ctor in the reference -> deformable_groups stays 1 for the mask branch (C3)
        self.conv_mask = DeformConv(depth_channels, self.deformable_groups * k2, kernel_size=self.kernel_size,
                                    stride=self.stride, padding=self.padding, dilation=self.dilation, bias=True)
        self.init_offset()

    def init_offset(self):
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()
        self.conv_mask.weight.data.zero_()
        self.conv_mask.bias.data.zero_()
        self.conv_offset.lr_mult = 0.1
        self.conv_mask.lr_mult = 0.1

    def forward(self, x, depth):
        offset = self.conv_offset(torch.cat((x, depth), dim=1))
        # C3: conv_mask is built with deformable_groups=1 (PY:438-446), so the reference kernel only ever
        # read the first 2*K2 offset channels of each image; state that explicitly instead of by pointer luck
        k2x2 = 2 * self.kernel_size[0] * self.kernel_size[1]
        mask_offset = offset if self.deformable_groups == 1 else offset[:, :k2x2].contiguous()
        cm = self.conv_mask
        # no 16-bit kernels for this branch: under autocast it computes in fp32 on fp32 views, only the mask is cast
        branch, dt = _autocast_fp32_view((depth, mask_offset, cm.weight, cm.bias))
        depth_v, mask_offset, cm_weight, cm_bias = _inference_view(branch)
        mask = _DeformConvSigmoidFunction.apply(depth_v, mask_offset, cm_weight, cm_bias, cm.stride, cm.padding,
                                                cm.dilation, cm.groups, cm.deformable_groups)
        if dt is not None:
            mask = mask.to(dt)
        self.offset, self.mask = offset, mask
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias,
                                     self.stride, self.padding, self.dilation,
                                     self.groups, self.deformable_groups)

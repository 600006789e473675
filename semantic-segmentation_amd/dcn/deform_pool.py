"""Python boundary of deformable position-sensitive RoI pooling: the autograd.Function, its functional alias and
the three nn.Modules of the reference's `detseg/ops/dcn/deform_pool.py` (PP below), with the same names,
positional argument order, defaults, gradient tuple and state-dict keys (`offset_fc.{0,2,4}`, `mask_fc.{0,2}`).

Where the reference called its pybind CUDA extension `deform_pool_cuda` (PP:47-50, 70-74) this file calls
`ddcn_deform_psroi_pool_{forward,backward}_f32` of libddcn_hip.so (include/ddcn_hip.h) on torch's current HIP
stream.  CPU tensors raise NotImplementedError as PP:41-42 and PP:61-62 do; there is no fallback.

Deviations from the reference, all on inputs it would have read out of bounds with:
  - rois must be [R, 5] and offset [R, 2k, part_size, part_size] with k dividing out_channels; data must hold
    out_channels * group_size^2 score maps (deform_pool_cuda_kernel.cu:101-102, 130 index them unchecked);
  - a RoI whose batch index is outside the batch pools to zero;
  - float32 and float64 have their own kernels; float16 / bfloat16 tensors are computed in fp32 and converted back
    (as in deform_conv.py here).
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import _lib
from .deform_conv import _cast, _cdtype, _check_dtype_cuda, _entry


def _roi_geom(data, rois, offset, spatial_scale, out_size, out_channels, no_trans, group_size, part_size,
              sample_per_part, trans_std):
    if data.dim() != 4:
        raise RuntimeError("4D data tensor expected but got: %d" % data.dim())
    if rois.dim() != 2 or rois.size(1) != 5:
        raise RuntimeError("rois must be [R, 5] (batch index, x1, y1, x2, y2), got %s" % (tuple(rois.shape),))
    trans_channels = 2
    if not no_trans:
        want = (rois.size(0), part_size, part_size)
        if offset.dim() != 4 or (offset.size(0), offset.size(2), offset.size(3)) != want:
            raise RuntimeError("invalid shape of offset, expected (%d, 2k, %d, %d) but got %s"
                               % (want[0], part_size, part_size, tuple(offset.shape)))
        trans_channels = offset.size(1)
    b, c, h, w = data.shape
    return _lib.RoiGeom(b, c, h, w, rois.size(0), out_channels, out_size, group_size, part_size, sample_per_part,
                        trans_channels, 1 if no_trans else 0, spatial_scale, trans_std)


class DeformRoIPoolingFunction(Function):

    @staticmethod
    def forward(ctx,
                data,
                rois,
                offset,
                spatial_scale,
                out_size,
                out_channels,
                no_trans,
                group_size=1,
                part_size=None,
                sample_per_part=4,
                trans_std=.0):
        # PP:25-29: only square outputs
        out_h, out_w = _pair(out_size)
        assert isinstance(out_h, int) and isinstance(out_w, int)
        assert out_h == out_w
        out_size = out_h
        part_size = out_size if part_size is None else part_size
        assert 0.0 <= trans_std <= 1.0  # PP:40

        ctx.pool_args = (spatial_scale, out_size, out_channels, bool(no_trans), group_size, part_size,
                         sample_per_part, trans_std)
        _check_dtype_cuda("data", data, data.dtype)
        # rois / offset on the host (or on another GPU) would hand the kernel a pointer it cannot read: refuse them here
        for name, t in (("rois", rois),) + ((() if no_trans else (("offset", offset),))):
            if not t.is_cuda:
                raise NotImplementedError
            if t.device != data.device:
                raise RuntimeError("ddcn: %s is on %s but data is on %s" % (name, t.device, data.device))
            if not t.is_floating_point():
                raise RuntimeError("ddcn: %s must be a floating-point tensor, got %s" % (name, t.dtype))
        ctx.dtype = data.dtype
        if not data.is_contiguous():  # deform_pool_cuda.cpp:39
            raise RuntimeError("input tensor has to be contiguous")

        g = _roi_geom(data, rois, offset, *ctx.pool_args)
        cd = _cdtype(ctx.dtype)
        data32 = _cast(data, cd)
        rois32 = _cast(rois, cd).contiguous()
        offset32 = None if no_trans else _cast(offset, cd).contiguous()
        n = rois.shape[0]
        output = torch.empty((n, out_channels, out_size, out_size), dtype=cd, device=data.device)
        output_count = torch.empty_like(output)
        with torch.cuda.device(data.device):
            _lib.check(_entry("ddcn_deform_psroi_pool_forward", cd)(
                _lib.ptr(data32), _lib.ptr(rois32), _lib.ptr(offset32), _lib.ptr(output), _lib.ptr(output_count), g,
                _lib.current_stream(data.device)))

        if data.requires_grad or rois.requires_grad or offset.requires_grad:  # PP:52-53
            ctx.save_for_backward(data, rois, offset)
        ctx.output_count = output_count
        return output if ctx.dtype == cd else output.to(ctx.dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise NotImplementedError
        data, rois, offset = ctx.saved_tensors
        if grad_output.device != data.device:
            raise RuntimeError("ddcn: grad_output is on %s but data is on %s" % (grad_output.device, data.device))
        no_trans = ctx.pool_args[3]
        g = _roi_geom(data, rois, offset, *ctx.pool_args)
        cd = _cdtype(ctx.dtype)
        data32 = _cast(data, cd)
        rois32 = _cast(rois, cd).contiguous()
        offset32 = None if no_trans else _cast(offset, cd).contiguous()
        grad_out32 = _cast(grad_output, cd).contiguous()  # deform_pool_cuda.cpp:63 demanded contiguity; made so here

        # PP:66-68: both gradients start from zero, the kernel accumulates
        grad_input = torch.zeros(data.shape, dtype=cd, device=data.device)
        grad_offset = torch.zeros(offset.shape, dtype=cd, device=data.device)
        with torch.cuda.device(data.device):
            _lib.check(_entry("ddcn_deform_psroi_pool_backward", cd)(
                _lib.ptr(grad_out32), _lib.ptr(data32), _lib.ptr(rois32), _lib.ptr(offset32),
                _lib.ptr(ctx.output_count), _lib.ptr(grad_input), None if no_trans else _lib.ptr(grad_offset), g,
                _lib.current_stream(data.device)))
        if ctx.dtype != cd:
            grad_input, grad_offset = grad_input.to(ctx.dtype), grad_offset.to(ctx.dtype)
        # no gradient for rois (PP:67) nor for the eight scalars
        return (grad_input, None, grad_offset) + (None,) * 8


deform_roi_pooling = DeformRoIPoolingFunction.apply


class DeformRoIPooling(nn.Module):

    def __init__(self,
                 spatial_scale,
                 out_size,
                 out_channels,
                 no_trans,
                 group_size=1,
                 part_size=None,
                 sample_per_part=4,
                 trans_std=.0):
        super(DeformRoIPooling, self).__init__()
        self.spatial_scale = spatial_scale
        self.out_size = _pair(out_size)
        self.out_channels = out_channels
        self.no_trans = no_trans
        self.group_size = group_size
        self.part_size = out_size if part_size is None else part_size
        self.sample_per_part = sample_per_part
        self.trans_std = trans_std

    def _pool(self, data, rois, offset=None):
        """One call of the operator with this module's settings; offset=None pools without deformation."""
        rigid = offset is None
        if rigid:
            offset = data.new_empty(0)
        return deform_roi_pooling(data, rois, offset, self.spatial_scale, self.out_size, self.out_channels,
                                  rigid, self.group_size, self.part_size, self.sample_per_part, self.trans_std)

    def forward(self, data, rois, offset):
        return self._pool(data, rois, None if self.no_trans else offset)


def _fc_stack(in_features, hidden, out_features, depth, squash=False):
    """`depth` Linear layers, ReLU between them, the last one zero-initialised so that training starts from the rigid
    pooling (PP:136-148, 196-225); `squash` appends the Sigmoid of the mask branch.  Layer indices, and with them
    the state-dict keys, are the reference's."""
    layers = []
    for i in range(depth):
        last = i == depth - 1
        layers.append(nn.Linear(in_features, out_features if last else hidden))
        if not last:
            layers.append(nn.ReLU(inplace=True))
        in_features = hidden
    nn.init.zeros_(layers[-1].weight)
    nn.init.zeros_(layers[-1].bias)
    if squash:
        layers.append(nn.Sigmoid())
    return nn.Sequential(*layers)


class DeformRoIPoolingPack(DeformRoIPooling):

    def __init__(self,
                 spatial_scale,
                 out_size,
                 out_channels,
                 no_trans,
                 group_size=1,
                 part_size=None,
                 sample_per_part=4,
                 trans_std=.0,
                 num_offset_fcs=3,
                 deform_fc_channels=1024):
        super(DeformRoIPoolingPack, self).__init__(spatial_scale, out_size, out_channels, no_trans, group_size,
                                                   part_size, sample_per_part, trans_std)
        self.num_offset_fcs = num_offset_fcs
        self.deform_fc_channels = deform_fc_channels
        if not no_trans:
            bins = self.out_size[0] * self.out_size[1]
            self.offset_fc = _fc_stack(bins * self.out_channels, deform_fc_channels, bins * 2, num_offset_fcs)

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        if self.no_trans:
            return self._pool(data, rois)
        # PP:160-172: the offsets are predicted from the rigid pooling of the same RoIs
        n = rois.shape[0]
        rigid = self._pool(data, rois)
        offset = self.offset_fc(rigid.view(n, -1)).view(n, 2, self.out_size[0], self.out_size[1])
        return self._pool(data, rois, offset)


class ModulatedDeformRoIPoolingPack(DeformRoIPooling):

    def __init__(self,
                 spatial_scale,
                 out_size,
                 out_channels,
                 no_trans,
                 group_size=1,
                 part_size=None,
                 sample_per_part=4,
                 trans_std=.0,
                 num_offset_fcs=3,
                 num_mask_fcs=2,
                 deform_fc_channels=1024):
        super(ModulatedDeformRoIPoolingPack, self).__init__(spatial_scale, out_size, out_channels, no_trans,
                                                            group_size, part_size, sample_per_part, trans_std)
        self.num_offset_fcs = num_offset_fcs
        self.num_mask_fcs = num_mask_fcs
        self.deform_fc_channels = deform_fc_channels
        if not no_trans:
            bins = self.out_size[0] * self.out_size[1]
            feats = bins * self.out_channels
            self.offset_fc = _fc_stack(feats, deform_fc_channels, bins * 2, num_offset_fcs)
            self.mask_fc = _fc_stack(feats, deform_fc_channels, bins, num_mask_fcs, squash=True)

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        if self.no_trans:
            return self._pool(data, rois)
        # PP:236-252: offsets and a per-bin modulation, both predicted from the rigid pooling
        n = rois.shape[0]
        flat = self._pool(data, rois).view(n, -1)
        offset = self.offset_fc(flat).view(n, 2, self.out_size[0], self.out_size[1])
        mask = self.mask_fc(flat).view(n, 1, self.out_size[0], self.out_size[1])
        return self._pool(data, rois, offset) * mask

"""Drop-in for `detseg.ops.dcn`: the names of `detseg/ops/dcn/__init__.py:1-13`, plus the autograd.Functions."""
from . import deform_conv as _conv
from . import deform_pool as _pool

_CONV_NAMES = ("DeformConv", "DeformConvPack", "ModulatedDeformConv", "ModulatedDeformConvPack", "DepthawareConv",
               "DepthDeformConvPack", "DeformConvFunction", "ModulatedDeformConvFunction", "DepthawareConvFunction",
               "deform_conv", "modulated_deform_conv", "depthaware_conv")
_POOL_NAMES = ("DeformRoIPooling", "DeformRoIPoolingPack", "ModulatedDeformRoIPoolingPack", "DeformRoIPoolingFunction",
               "deform_roi_pooling")

globals().update({n: getattr(_conv, n) for n in _CONV_NAMES})
globals().update({n: getattr(_pool, n) for n in _POOL_NAMES})

__all__ = list(_CONV_NAMES + _POOL_NAMES)

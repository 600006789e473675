"""MI355X-native DDCN deformable-convolution operator (drop-in for `detseg.ops.dcn`).

The directory name is hyphenated, so import it with
`importlib.import_module("semantic-segmentation_amd")`; the public names mirror
`detseg/ops/__init__.py:2-14` and `detseg/ops/dcn/__init__.py:1-13` of the reference.
"""
from .dcn import (DeformConv, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvPack, DepthawareConv,
                  DepthDeformConvPack, DeformConvFunction, ModulatedDeformConvFunction, DepthawareConvFunction,
                  deform_conv, modulated_deform_conv, depthaware_conv, DeformRoIPooling, DeformRoIPoolingPack,
                  ModulatedDeformRoIPoolingPack, DeformRoIPoolingFunction, deform_roi_pooling)

__all__ = [
    'DeformConv', 'DeformConvPack', 'ModulatedDeformConv', 'ModulatedDeformConvPack', 'DepthawareConv',
    'DepthDeformConvPack', 'DeformConvFunction', 'ModulatedDeformConvFunction', 'DepthawareConvFunction',
    'deform_conv', 'modulated_deform_conv', 'depthaware_conv',
    'DeformRoIPooling', 'DeformRoIPoolingPack', 'ModulatedDeformRoIPoolingPack', 'DeformRoIPoolingFunction',
    'deform_roi_pooling'
]

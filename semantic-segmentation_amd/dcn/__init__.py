from .deform_conv import (DeformConv, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvPack, DepthawareConv,
                          DepthDeformConvPack, DeformConvFunction, ModulatedDeformConvFunction,
                          DepthawareConvFunction, deform_conv, modulated_deform_conv, depthaware_conv)
from .deform_pool import (DeformRoIPooling, DeformRoIPoolingPack, ModulatedDeformRoIPoolingPack,
                          DeformRoIPoolingFunction, deform_roi_pooling)

__all__ = [
    'DeformConv', 'DeformConvPack', 'ModulatedDeformConv', 'ModulatedDeformConvPack', 'DepthawareConv',
    'DepthDeformConvPack', 'DeformConvFunction', 'ModulatedDeformConvFunction', 'DepthawareConvFunction',
    'deform_conv', 'modulated_deform_conv', 'depthaware_conv',
    'DeformRoIPooling', 'DeformRoIPoolingPack', 'ModulatedDeformRoIPoolingPack', 'DeformRoIPoolingFunction',
    'deform_roi_pooling'
]

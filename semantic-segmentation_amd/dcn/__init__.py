from .deform_conv import (DeformConv, DeformConvPack, ModulatedDeformConv, ModulatedDeformConvPack, DepthawareConv,
                          DepthDeformConvPack, DeformConvFunction, ModulatedDeformConvFunction,
                          DepthawareConvFunction, deform_conv, modulated_deform_conv, depthaware_conv)

__all__ = [
    'DeformConv', 'DeformConvPack', 'ModulatedDeformConv', 'ModulatedDeformConvPack', 'DepthawareConv',
    'DepthDeformConvPack', 'DeformConvFunction', 'ModulatedDeformConvFunction', 'DepthawareConvFunction',
    'deform_conv', 'modulated_deform_conv', 'depthaware_conv'
]

"""ctypes binding of libddcn_hip.so (the C ABI declared in include/ddcn_hip.h).

This is the counterpart of the reference's `from . import deform_conv_cuda`
(detseg/ops/dcn/deform_conv.py:9).  There is no CPU or PyTorch fallback: if the HIP
library is missing the import fails loudly.
"""
import ctypes
import os

import torch  # noqa: F401  (imported first so that the library binds to torch's HIP runtime)

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# DDCN_LIB_PATH: another build of the same library (kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get("DDCN_LIB_PATH") or os.path.join(_PKG, "lib", "libddcn_hip.so")


class Geom(ctypes.Structure):
    """struct ddcn_geom (include/ddcn_hip.h)."""
    _fields_ = [(n, ctypes.c_int) for n in
                ("N", "C", "H", "W", "Co", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw",
                 "groups", "dg", "Ho", "Wo")]


class RoiGeom(ctypes.Structure):
    """struct ddcn_roi_geom (include/ddcn_hip.h)."""
    _fields_ = [(n, ctypes.c_int) for n in
                ("B", "C", "H", "W", "R", "out_channels", "pooled", "group_size", "part_size", "sample_per_part",
                 "trans_channels", "no_trans")] + [("spatial_scale", ctypes.c_float), ("trans_std", ctypes.c_float)]


_P = ctypes.c_void_p
_G = ctypes.POINTER(Geom)
_RG = ctypes.POINTER(RoiGeom)
_SZ = ctypes.c_size_t

# name -> argtypes; every entry returns int unless listed in _RESTYPES
PROTOTYPES = {
    "ddcn_version": [],
    "ddcn_strerror": [ctypes.c_int],
    "ddcn_last_error": [],
    "ddcn_infer_output": [_G],
    "ddcn_check_geom": [_G],
    "ddcn_col_ld": [_G],
    "ddcn_workspace_bytes": [_G],
    "ddcn_im2col_f32": [_P, _P, _P, _P, _G, _P],
    "ddcn_col2im_coord_f32": [_P, _P, _P, _P, _P, _P, _P, _G, _P],
    "ddcn_depthaware_im2col_f32": [_P, _P, _P, _G, _P],
    "ddcn_depthaware_col2im_f32": [_P, _P, _P, _G, _P],
    "ddcn_gemm_out_f32": [_P, _P, _P, _P, _G, _P],
    "ddcn_gemm_gcol_f32": [_P, _P, _P, _G, _P],
    "ddcn_gemm_gweight_f32": [_P, _P, _P, _G, _P],
    "ddcn_bias_grad_f32": [_P, _P, _G, _P],
    "ddcn_deform_conv_forward_f32": [_P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_deform_conv_direct_forward_f32": [_P, _P, _P, _P, _P, ctypes.c_int, _P, _SZ, _G, _P],
    "ddcn_deform_conv_backward_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_deform_conv_direct_backward_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_deform_conv_direct_backward_2s_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P, _P, _P, _P],
    "ddcn_modulated_deform_conv_forward_f32": [_P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_static_scratch_bytes": [],
    "ddcn_modulated_deform_conv_forward_static_f32": [_P, _P, _P, _P, _P, _P, _G, _P],
    "ddcn_modulated_deform_conv_backward_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_deform_conv_backward_saved_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_modulated_deform_conv_backward_saved_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_deform_conv_backward_saved_2s_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P, _P, _P, _P],
    "ddcn_depthaware_conv_forward_f32": [_P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_depthaware_conv_backward_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_workspace_bytes_f64": [_G],
    "ddcn_deform_conv_forward_f64": [_P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_deform_conv_direct_forward_f64": [_P, _P, _P, _P, _P, ctypes.c_int, _P, _SZ, _G, _P],
    "ddcn_deform_conv_backward_f64": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_deform_conv_direct_backward_f64": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_deform_conv_backward_saved_f64": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_modulated_deform_conv_forward_f64": [_P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_modulated_deform_conv_backward_f64": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_modulated_deform_conv_backward_saved_f64": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_depthaware_conv_forward_f64": [_P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_depthaware_conv_backward_f64": [_P, _P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_workspace_bytes_bf16": [_G],
    "ddcn_bf16_supported": [_G],
    "ddcn_modulated_deform_conv_forward_bf16": [_P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_fused16_supported": [_G],
    "ddcn_fused16_workspace_bytes": [_G],
    "ddcn_modulated_deform_conv_forward_fused_bf16": [_P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_modulated_deform_conv_forward_fused_f16": [_P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_modulated_deform_conv_forward_f16": [_P, _P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_gemm_gcol_16_workspace_bytes": [_G],
    "ddcn_gemm_gcol_16_supported": [_G],
    "ddcn_gemm_gcol_bf16": [_P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_gemm_gcol_f16": [_P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_gweight_16_workspace_bytes": [_G],
    "ddcn_gweight_16_supported": [_G],
    "ddcn_deform_conv_gweight_bf16": [_P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_deform_conv_gweight_f16": [_P, _P, _P, _P, _P, _P, _SZ, _G, _P],
    "ddcn_deform_psroi_pool_forward_f32": [_P, _P, _P, _P, _P, _RG, _P],
    "ddcn_deform_psroi_pool_backward_f32": [_P, _P, _P, _P, _P, _P, _P, _RG, _P],
    "ddcn_deform_psroi_pool_forward_f64": [_P, _P, _P, _P, _P, _RG, _P],
    "ddcn_deform_psroi_pool_backward_f64": [_P, _P, _P, _P, _P, _P, _P, _RG, _P],
}
_RESTYPES = {
    "ddcn_strerror": ctypes.c_char_p,
    "ddcn_last_error": ctypes.c_char_p,
    "ddcn_col_ld": _SZ,
    "ddcn_workspace_bytes": _SZ,
    "ddcn_static_scratch_bytes": _SZ,
    "ddcn_workspace_bytes_bf16": _SZ,
    "ddcn_fused16_workspace_bytes": _SZ,
    "ddcn_gemm_gcol_16_workspace_bytes": _SZ,
    "ddcn_gweight_16_workspace_bytes": _SZ,
    "ddcn_workspace_bytes_f64": _SZ,
}

_lib = None


def lib():
    """Load libddcn_hip.so once; raise if it was not built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libddcn_hip.so not found at %s: build it with `make -C semantic-segmentation_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        _lib = handle
    return _lib


def check(rc):
    """Python `RuntimeError` where the reference's AT_CHECK/AT_ERROR raised (deform_conv_cuda.cpp:179-198)."""
    if rc != 0:
        handle = lib()
        raise RuntimeError("ddcn: %s: %s" % (handle.ddcn_strerror(rc).decode(), handle.ddcn_last_error().decode()))


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

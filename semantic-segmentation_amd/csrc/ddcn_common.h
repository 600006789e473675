// Internal header shared by the HIP translation units of libddcn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#include "../../include/ddcn_hip.h"

namespace ddcn {

// Sets the calling thread's detail message and returns rc (so callers can `return fail(...)`).
int fail(int rc, const char *fmt, ...);
int check_launch(const char *what);

static inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }

// Raises a kernel's dynamic-LDS limit once per device of the process (function attributes are per device; `done` is
// a bit per device id, ids >= 64 set it every time).  Kept out of the per-call path so that nothing but launches runs
// while a caller captures a HIP graph.  A race only repeats the (idempotent) call.
static inline bool raise_lds_limit(const void *fn, int bytes, std::atomic<uint64_t> &done)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = dev < 64 ? (1ull << dev) : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}

struct Dims {
    int K2, P, Ktot, Kg, Cg, Cog, cpg;
    long ld;  // columns leading dimension
};

static inline Dims dims_of(const ddcn_geom &g)
{
    Dims d;
    d.K2 = g.kh * g.kw;
    d.P = g.Ho * g.Wo;
    d.Ktot = g.C * d.K2;
    d.Cg = g.C / g.groups;
    d.Cog = g.Co / g.groups;
    d.Kg = d.Cg * d.K2;
    d.cpg = g.C / g.dg;
    d.ld = (long)round_up((size_t)d.P, 32);
    return d;
}

// ---- launchers implemented in the .hip files (all asynchronous on `stream`) ----
int launch_im2col(const float *x, const float *offset, const float *mask, float *col, const ddcn_geom &g,
                  hipStream_t stream);
int launch_skinny_forward(const float *x, const float *offset, const float *weight, const float *bias, float *out,
                          int sigmoid, void *ws, size_t ws_bytes, const ddcn_geom &g, hipStream_t stream, bool *handled);
int launch_sigmoid_inplace(float *out, long count, hipStream_t stream);
int launch_skinny_gweight(const float *x, const float *offset, const float *gpre, float *gw, void *ws, size_t ws_bytes,
                          const ddcn_geom &g, hipStream_t stream, bool *handled);
int launch_im2col_tiled(const float *x, const float *offset, const float *mask, float *col, const ddcn_geom &g,
                        hipStream_t stream, bool *handled);
int launch_col2im_coord(const float *gcol, const float *x, const float *offset, const float *mask, float *gx,
                        float *goff, float *gmask, const ddcn_geom &g, hipStream_t stream, bool clear_coord = true,
                        int chunk_channels = 0);
// chunk_channels: 0 = default (4, or DDCN_SCATTER_SC); 2 = the variant small enough (<= 168 VGPRs, <= 40 KiB LDS) to
// share a CU with a resident GEMM workgroup
int launch_col2im_coord_tile9(const float *gcol, const float *x, const float *offset, const float *mask, float *gx,
                              float *goff, float *gmask, const ddcn_geom &g, hipStream_t stream, bool *handled,
                              int chunk_channels = 0, const float *sk_w = nullptr, const float *sk_g = nullptr);
int launch_skinny_scatter(const float *x, const float *offset, const float *weight, const float *gout, float *gx,
                          float *goff, void *ws, size_t ws_bytes, const ddcn_geom &g, hipStream_t stream, bool *handled);
// bytes launch_skinny_scatter takes from the front of its workspace (the repacked weights), 256-byte granular
size_t skinny_scatter_ws_bytes(const ddcn_geom &g);
int launch_col2im_coord_tiled(const float *gcol, const float *x, const float *offset, const float *mask, float *gx,
                              float *goff, float *gmask, const ddcn_geom &g, hipStream_t stream, bool *handled);
int launch_depthaware_im2col(const float *x, const float *depth, float *col, const ddcn_geom &g,
                             hipStream_t stream);
int launch_depthaware_col2im(const float *gcol, const float *depth, float *gx, const ddcn_geom &g,
                             hipStream_t stream);
int launch_bias_grad(const float *gout, float *gb, const ddcn_geom &g, hipStream_t stream);

int launch_gemm_out(const float *weight, const float *col, const float *bias, float *out, const ddcn_geom &g,
                    hipStream_t stream);
int launch_gemm_gcol(const float *weight, const float *gout, float *gcol, const ddcn_geom &g, hipStream_t stream);
int launch_gemm_gweight(const float *gout, const float *col, float *gw, const ddcn_geom &g, hipStream_t stream);

size_t f64_workspace_bytes(const ddcn_geom &g);
int f64_forward(const double *x, const double *weight, const double *bias, const double *offset, const double *mask,
                const double *depth, double *out, int sigmoid, double *col, const ddcn_geom &g, hipStream_t s);
int f64_backward(const double *x, const double *weight, const double *offset, const double *mask, const double *depth,
                 const double *gout, const double *saved_cols, double *gx, double *goff, double *gmask, double *gw,
                 double *gb, double *col, const ddcn_geom &g, hipStream_t s);

size_t bf16_workspace_bytes(const ddcn_geom &g);
int bf16_supported(const ddcn_geom &g);
int launch_forward_bf16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                        void *out, void *ws, const ddcn_geom &g, hipStream_t stream);

size_t gcol16_workspace_bytes(const ddcn_geom &g);
int gcol16_supported(const ddcn_geom &g);
int launch_gemm_gcol_bf16(const void *weight, const void *gout, float *gcol, void *ws, const ddcn_geom &g, hipStream_t stream);
int launch_gemm_gcol_f16(const void *weight, const void *gout, float *gcol, void *ws, const ddcn_geom &g, hipStream_t stream);
size_t gweight16_workspace_bytes(const ddcn_geom &g);
int gweight16_supported(const ddcn_geom &g);
int launch_gweight_bf16(const void *x, const void *offset, const void *mask, const void *gout, float *gw, void *ws,
                        const ddcn_geom &g, hipStream_t stream);
int launch_gweight_f16(const void *x, const void *offset, const void *mask, const void *gout, float *gw, void *ws,
                       const ddcn_geom &g, hipStream_t stream);

int fused16_supported(const ddcn_geom &g);
size_t fused16_workspace_bytes(const ddcn_geom &g);
int launch_fused_forward_bf16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                              void *out, void *ws, const ddcn_geom &g, hipStream_t stream);
int launch_fused_forward_f16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                             void *out, void *ws, const ddcn_geom &g, hipStream_t stream);

int launch_forward_f16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                       void *out, void *ws, const ddcn_geom &g, hipStream_t stream);

int launch_psroi_forward(const float *data, const float *rois, const float *trans, float *out, float *out_count,
                         const ddcn_roi_geom *g, hipStream_t stream);
int launch_psroi_backward(const float *grad_out, const float *data, const float *rois, const float *trans,
                          const float *out_count, float *grad_data, float *grad_trans, const ddcn_roi_geom *g,
                          hipStream_t stream);

int launch_psroi_forward_f64(const double *data, const double *rois, const double *trans, double *out,
                             double *out_count, const ddcn_roi_geom *g, hipStream_t stream);
int launch_psroi_backward_f64(const double *grad_out, const double *data, const double *rois, const double *trans,
                              const double *out_count, double *grad_data, double *grad_trans, const ddcn_roi_geom *g,
                              hipStream_t stream);

}  // namespace ddcn

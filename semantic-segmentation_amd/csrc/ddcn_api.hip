// extern "C" surface of libddcn_hip.so (declared in include/ddcn_hip.h) and the host orchestration
// that replaces detseg/ops/dcn/src/deform_conv_cuda.cpp (H below):
//   forward : im2col (all images) -> G1 (+bias)                         H:221-250 / H:419-448
//   backward: G2 -> fused col2im+coord -> im2col -> G3 (+G4 bias grad)  H:299-363 / H:497-561
// No allocation, no synchronisation, no per-image host loop; the columns scratch is caller memory.
#include <stdarg.h>
#include <stdio.h>

#include "ddcn_common.h"

namespace ddcn {

static thread_local char g_err[512] = "";

int fail(int rc, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return rc;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DDCN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return DDCN_OK;
}

static int check_geom(const ddcn_geom *g)
{
    if (!g) return fail(DDCN_ERR_BAD_ARG, "geom is NULL");
    if (g->N <= 0 || g->C <= 0 || g->H <= 0 || g->W <= 0 || g->Co <= 0)
        return fail(DDCN_ERR_SHAPE, "non-positive tensor size N=%d C=%d H=%d W=%d Co=%d", g->N, g->C, g->H, g->W, g->Co);
    if (g->kh <= 0 || g->kw <= 0)
        return fail(DDCN_ERR_SHAPE, "kernel size should be greater than zero, but got kH: %d kW: %d", g->kh, g->kw);
    if (g->sh <= 0 || g->sw <= 0)
        return fail(DDCN_ERR_SHAPE, "stride should be greater than zero, but got dH: %d dW: %d", g->sh, g->sw);
    if (g->dh <= 0 || g->dw <= 0)
        return fail(DDCN_ERR_SHAPE, "dilation should be greater than 0, but got dilationH: %d dilationW: %d", g->dh, g->dw);
    if (g->ph < 0 || g->pw < 0) return fail(DDCN_ERR_SHAPE, "negative padding");
    if (g->groups <= 0 || g->C % g->groups || g->Co % g->groups)
        return fail(DDCN_ERR_SHAPE, "Input shape and kernel channels wont match: C=%d Co=%d groups=%d", g->C, g->Co, g->groups);
    if (g->dg <= 0 || g->C % g->dg)
        return fail(DDCN_ERR_SHAPE, "input channels must divide deformable group size (C=%d dg=%d)", g->C, g->dg);
    ddcn_geom t = *g;
    ddcn_infer_output(&t);
    if (t.Ho < 1 || t.Wo < 1)
        return fail(DDCN_ERR_SHAPE, "Given input size: (%d x %d x %d). Calculated output size: (%d x %d x %d). Output size is too small",
                    g->C, g->H, g->W, g->Co, t.Ho, t.Wo);
    if (t.Ho != g->Ho || t.Wo != g->Wo)
        return fail(DDCN_ERR_SHAPE, "invalid spatial size of offset/output, expected height: %d width: %d, but got height: %d width: %d",
                    t.Ho, t.Wo, g->Ho, g->Wo);
    // 32-bit index budget of the kernels (per-image planes and column rows), formed in 64 bits: the int products
    // inside dims_of() have already wrapped for the sizes this test is there to reject
    const long P = (long)t.Ho * t.Wo, Ktot = (long)g->C * g->kh * g->kw;
    if ((long)g->H * g->W >= (1L << 31) || P >= (1L << 31) - 1024 || Ktot >= (1L << 31))
        return fail(DDCN_ERR_UNSUPPORTED, "tensor plane exceeds 2^31 elements");
    return DDCN_OK;
}

static int check_ws(const ddcn_geom *g, const void *ws, size_t bytes)
{
    if (!ws) return fail(DDCN_ERR_BAD_ARG, "workspace is NULL");
    const size_t need = ddcn_workspace_bytes(g);
    if (bytes < need) return fail(DDCN_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", bytes, need);
    if ((uintptr_t)ws & 15) return fail(DDCN_ERR_BAD_ARG, "workspace must be 16-byte aligned");
    return DDCN_OK;
}

}  // namespace ddcn

using namespace ddcn;

#define REQUIRE(p)                                                          \
    do {                                                                    \
        if (!(p)) return fail(DDCN_ERR_BAD_ARG, "%s: " #p " is NULL", __func__); \
    } while (0)
#define TRY(expr)              \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != DDCN_OK) return rc_; \
    } while (0)

extern "C" {

int ddcn_version(void) { return DDCN_VERSION; }

const char *ddcn_strerror(int rc)
{
    switch (rc) {
    case DDCN_OK: return "ok";
    case DDCN_ERR_BAD_ARG: return "bad argument";
    case DDCN_ERR_SHAPE: return "shape mismatch";
    case DDCN_ERR_WORKSPACE: return "workspace too small";
    case DDCN_ERR_LAUNCH: return "HIP launch failure";
    case DDCN_ERR_UNSUPPORTED: return "unsupported size";
    case DDCN_ERR_NO_DEVICE: return "no gfx950 device";
    default: return "unknown error";
    }
}

const char *ddcn_last_error(void) { return g_err; }

int ddcn_infer_output(ddcn_geom *g)
{
    if (!g) return fail(DDCN_ERR_BAD_ARG, "geom is NULL");
    if (g->sh <= 0 || g->sw <= 0) return fail(DDCN_ERR_SHAPE, "stride should be greater than zero");
    // floor division like Python's // (PY:76-79): numerator may be negative for too-small inputs
    const int nh = g->H + 2 * g->ph - (g->dh * (g->kh - 1) + 1);
    const int nw = g->W + 2 * g->pw - (g->dw * (g->kw - 1) + 1);
    g->Ho = (nh >= 0 ? nh / g->sh : -((-nh + g->sh - 1) / g->sh)) + 1;
    g->Wo = (nw >= 0 ? nw / g->sw : -((-nw + g->sw - 1) / g->sw)) + 1;
    return DDCN_OK;
}

int ddcn_check_geom(const ddcn_geom *g) { return check_geom(g); }

// the size queries never throw and never divide by an unvalidated group count: 0 for a geometry dims_of() cannot take
static bool sizes_defined(const ddcn_geom *g)
{
    return g && g->groups > 0 && g->dg > 0 && g->N > 0 && g->C > 0 && g->kh > 0 && g->kw > 0 && g->Ho > 0 && g->Wo > 0 &&
           (long)g->Ho * g->Wo < (1L << 31) - 1024 && (long)g->C * g->kh * g->kw < (1L << 31);
}

size_t ddcn_col_ld(const ddcn_geom *g) { return sizes_defined(g) ? (size_t)dims_of(*g).ld : 0; }

size_t ddcn_workspace_bytes(const ddcn_geom *g)
{
    if (!sizes_defined(g)) return 0;
    const Dims d = dims_of(*g);
    return sizeof(float) * (size_t)g->N * (size_t)d.Ktot * (size_t)d.ld;
}

/* ------------------------------------------------------------------ stage level */

int ddcn_im2col_f32(const float *x, const float *offset, const float *mask, float *col, const ddcn_geom *g,
                    void *stream)
{
    REQUIRE(x); REQUIRE(offset); REQUIRE(col);
    TRY(check_geom(g));
    return launch_im2col(x, offset, mask, col, *g, (hipStream_t)stream);
}

int ddcn_col2im_coord_f32(const float *gcol, const float *x, const float *offset, const float *mask, float *gx,
                          float *goff, float *gmask, const ddcn_geom *g, void *stream)
{
    REQUIRE(gcol); REQUIRE(x); REQUIRE(offset); REQUIRE(gx); REQUIRE(goff);
    if ((mask == nullptr) != (gmask == nullptr)) return fail(DDCN_ERR_BAD_ARG, "mask and gmask must both be given or both be NULL");
    TRY(check_geom(g));
    return launch_col2im_coord(gcol, x, offset, mask, gx, goff, gmask, *g, (hipStream_t)stream);
}

int ddcn_depthaware_im2col_f32(const float *x, const float *depth, float *col, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(depth); REQUIRE(col);
    TRY(check_geom(g));
    return launch_depthaware_im2col(x, depth, col, *g, (hipStream_t)stream);
}

int ddcn_depthaware_col2im_f32(const float *gcol, const float *depth, float *gx, const ddcn_geom *g, void *stream)
{
    REQUIRE(gcol); REQUIRE(depth); REQUIRE(gx);
    TRY(check_geom(g));
    return launch_depthaware_col2im(gcol, depth, gx, *g, (hipStream_t)stream);
}

int ddcn_gemm_out_f32(const float *weight, const float *col, const float *bias, float *out, const ddcn_geom *g,
                      void *stream)
{
    REQUIRE(weight); REQUIRE(col); REQUIRE(out);
    TRY(check_geom(g));
    return launch_gemm_out(weight, col, bias, out, *g, (hipStream_t)stream);
}

int ddcn_gemm_gcol_f32(const float *weight, const float *grad_out, float *gcol, const ddcn_geom *g, void *stream)
{
    REQUIRE(weight); REQUIRE(grad_out); REQUIRE(gcol);
    TRY(check_geom(g));
    return launch_gemm_gcol(weight, grad_out, gcol, *g, (hipStream_t)stream);
}

int ddcn_gemm_gweight_f32(const float *grad_out, const float *col, float *gw, const ddcn_geom *g, void *stream)
{
    REQUIRE(grad_out); REQUIRE(col); REQUIRE(gw);
    TRY(check_geom(g));
    return launch_gemm_gweight(grad_out, col, gw, *g, (hipStream_t)stream);
}

int ddcn_bias_grad_f32(const float *grad_out, float *gb, const ddcn_geom *g, void *stream)
{
    REQUIRE(grad_out); REQUIRE(gb);
    TRY(check_geom(g));
    return launch_bias_grad(grad_out, gb, *g, (hipStream_t)stream);
}

/* ------------------------------------------------------------------ whole ops */

static int deform_forward(const float *x, const float *weight, const float *bias, const float *offset,
                          const float *mask, float *out, void *ws, size_t ws_bytes, const ddcn_geom *g, void *stream)
{
    TRY(check_geom(g));
    TRY(check_ws(g, ws, ws_bytes));
    hipStream_t s = (hipStream_t)stream;
    float *col = (float *)ws;
    TRY(launch_im2col(x, offset, mask, col, *g, s));
    TRY(launch_gemm_out(weight, col, bias, out, *g, s));
    return DDCN_OK;
}

// Fork/join for the two-stream backward: `side` (a second stream of the caller) and two events of the caller.
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};

static int deform_backward(const float *x, const float *weight, const float *offset, const float *mask,
                           const float *gout, const float *saved_cols, float *gx, float *goff, float *gmask,
                           float *gw, float *gb, void *ws, size_t ws_bytes, const ddcn_geom *g, void *stream,
                           const SideStream *side = nullptr)
{
    TRY(check_geom(g));
    TRY(check_ws(g, ws, ws_bytes));
    hipStream_t s = (hipStream_t)stream;
    float *col = (float *)ws;
    if (saved_cols && ((uintptr_t)saved_cols & 15)) return fail(DDCN_ERR_BAD_ARG, "columns must be 16-byte aligned");
    if (side && saved_cols) {
        // G3 (+ G4) needs only grad_out and the saved columns: it runs on the side stream BESIDE the scatter, whose
        // 2-channel variant is small enough to share each CU with the resident GEMM workgroup (the scatter is bound
        // by LDS atomics, the GEMM by the matrix cores).  The caller's stream continues when both are done.
        // The fork is recorded AFTER G2: two GEMMs side by side only share the matrix cores (measured: no gain), the
        // point is G3 beside the scatter.
        TRY(launch_gemm_gcol(weight, gout, col, *g, s));
        hipError_t e = hipEventRecord(side->fork, s);
        if (e == hipSuccess) e = hipStreamWaitEvent(side->stream, side->fork, 0);
        if (e != hipSuccess) return fail(DDCN_ERR_LAUNCH, "two-stream backward: fork: %s", hipGetErrorString(e));
        // from here on the side stream may hold work: whatever fails, the caller's stream still joins it before this
        // call returns (the caller frees gw / gb / the workspace on ITS stream's order)
        int rc = launch_gemm_gweight(gout, saved_cols, gw, *g, side->stream);
        if (rc == DDCN_OK && gb) rc = launch_bias_grad(gout, gb, *g, side->stream);
        if (rc == DDCN_OK)
            rc = launch_col2im_coord(col, x, offset, mask, gx, goff, gmask, *g, s, /*clear_coord=*/false, /*chunk_channels=*/2);
        e = hipEventRecord(side->join, side->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, side->join, 0);
        if (rc != DDCN_OK) return rc;
        if (e != hipSuccess) return fail(DDCN_ERR_LAUNCH, "two-stream backward: join: %s", hipGetErrorString(e));
        return DDCN_OK;
    }
    TRY(launch_gemm_gcol(weight, gout, col, *g, s));
    TRY(launch_col2im_coord(col, x, offset, mask, gx, goff, gmask, *g, s, /*clear_coord=*/false));
    if (saved_cols) {   // the forward call's columns are still there: no second im2col
        TRY(launch_gemm_gweight(gout, saved_cols, gw, *g, s));
    } else {
        TRY(launch_im2col(x, offset, mask, col, *g, s));
        TRY(launch_gemm_gweight(gout, col, gw, *g, s));
    }
    if (gb) TRY(launch_bias_grad(gout, gb, *g, s));
    return DDCN_OK;
}

int ddcn_deform_conv_forward_f32(const float *x, const float *weight, const float *bias, const float *offset,
                                 float *out, void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                 void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(out);
    return deform_forward(x, weight, bias, offset, nullptr, out, workspace, workspace_bytes, g, stream);
}

int ddcn_deform_conv_direct_forward_f32(const float *x, const float *weight, const float *bias, const float *offset,
                                        float *out, int sigmoid, void *workspace, size_t workspace_bytes,
                                        const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(out);
    TRY(check_geom(g));
    TRY(check_ws(g, workspace, workspace_bytes));
    bool handled = false;
    TRY(launch_skinny_forward(x, offset, weight, bias, out, sigmoid, workspace, workspace_bytes, *g, (hipStream_t)stream,
                              &handled));
    if (handled) return DDCN_OK;
    // geometry outside the direct kernel: columns + GEMM, then the activation
    TRY(deform_forward(x, weight, bias, offset, nullptr, out, workspace, workspace_bytes, g, stream));
    if (sigmoid) TRY(launch_sigmoid_inplace(out, (long)g->N * g->Co * g->Ho * g->Wo, (hipStream_t)stream));
    return DDCN_OK;
}

int ddcn_deform_conv_direct_backward_f32(const float *x, const float *weight, const float *offset,
                                         const float *grad_out, float *gx, float *goff, float *gw, float *gb,
                                         void *workspace, size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(grad_out); REQUIRE(gx); REQUIRE(goff); REQUIRE(gw);
    TRY(check_geom(g));
    TRY(check_ws(g, workspace, workspace_bytes));
    hipStream_t s = (hipStream_t)stream;
    float *col = (float *)workspace;
    bool handled = false;
    // grad_input / grad_offset straight from grad_out: the 9-term W^T * grad_out is formed inside the scatter kernel
    TRY(launch_skinny_scatter(x, offset, weight, grad_out, gx, goff, workspace, workspace_bytes, *g, s, &handled));
    if (!handled) {
        TRY(launch_gemm_gcol(weight, grad_out, col, *g, s));
        TRY(launch_col2im_coord(col, x, offset, nullptr, gx, goff, nullptr, *g, s, /*clear_coord=*/false));
    }
    handled = false;
    // no second im2col, no 9-row G3 (the workspace is free again: it takes the sample table)
    TRY(launch_skinny_gweight(x, offset, grad_out, gw, workspace, workspace_bytes, *g, s, &handled));
    if (!handled) {
        TRY(launch_im2col(x, offset, nullptr, col, *g, s));
        TRY(launch_gemm_gweight(grad_out, col, gw, *g, s));
    }
    if (gb) TRY(launch_bias_grad(grad_out, gb, *g, s));
    return DDCN_OK;
}

// The same backward with the weight gradient on a second stream of the caller, beside the scatter: the two direct
// kernels are independent (grad_input / grad_offset from the LDS-atomic-bound scatter, grad_weight from the VALU-bound
// per-channel sampling kernel) and share nothing but read-only inputs; the workspace is split between them (repacked
// weights first, the sample table behind).  Geometries one of them does not cover run that half on `stream`, in order.
int ddcn_deform_conv_direct_backward_2s_f32(const float *x, const float *weight, const float *offset,
                                            const float *grad_out, float *gx, float *goff, float *gw, float *gb,
                                            void *workspace, size_t workspace_bytes, const ddcn_geom *g, void *stream,
                                            void *side_stream, void *fork_event, void *join_event)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(grad_out); REQUIRE(gx); REQUIRE(goff); REQUIRE(gw);
    REQUIRE(side_stream); REQUIRE(fork_event); REQUIRE(join_event);
    if (side_stream == stream) return fail(DDCN_ERR_BAD_ARG, "side_stream must differ from stream");
    TRY(check_geom(g));
    TRY(check_ws(g, workspace, workspace_bytes));
    hipStream_t s = (hipStream_t)stream, s2 = (hipStream_t)side_stream;
    hipEvent_t fork = (hipEvent_t)fork_event, join = (hipEvent_t)join_event;
    float *col = (float *)workspace;
    const size_t lo = skinny_scatter_ws_bytes(*g);     // launch_skinny_scatter's repacked weights
    if (workspace_bytes <= lo)       // tiny maps: nothing to split, nothing to gain
        return ddcn_deform_conv_direct_backward_f32(x, weight, offset, grad_out, gx, goff, gw, gb, workspace, workspace_bytes, g, stream);
    hipError_t e = hipEventRecord(fork, s);
    if (e == hipSuccess) e = hipStreamWaitEvent(s2, fork, 0);
    if (e != hipSuccess) return fail(DDCN_ERR_LAUNCH, "two-stream direct backward: fork: %s", hipGetErrorString(e));
    // after the fork every exit joins the side stream first (its kernels write gw, gb and the workspace)
    bool hw = false, hs = false;
    int rc = launch_skinny_gweight(x, offset, grad_out, gw, (char *)workspace + lo, workspace_bytes - lo, *g, s2, &hw);
    if (rc == DDCN_OK && gb) rc = launch_bias_grad(grad_out, gb, *g, s2);
    e = hipEventRecord(join, s2);
    if (rc == DDCN_OK) rc = launch_skinny_scatter(x, offset, weight, grad_out, gx, goff, workspace, lo, *g, s, &hs);
    // everything below runs on `stream` after the side stream's work (the fallbacks use the whole workspace)
    if (e == hipSuccess) e = hipStreamWaitEvent(s, join, 0);
    if (rc != DDCN_OK) return rc;
    if (e != hipSuccess) return fail(DDCN_ERR_LAUNCH, "two-stream direct backward: join: %s", hipGetErrorString(e));
    if (!hs) {
        TRY(launch_gemm_gcol(weight, grad_out, col, *g, s));
        TRY(launch_col2im_coord(col, x, offset, nullptr, gx, goff, nullptr, *g, s, /*clear_coord=*/false));
    }
    if (!hw) {
        TRY(launch_im2col(x, offset, nullptr, col, *g, s));
        TRY(launch_gemm_gweight(grad_out, col, gw, *g, s));
    }
    return DDCN_OK;
}

int ddcn_deform_conv_backward_f32(const float *x, const float *weight, const float *offset, const float *grad_out,
                                  float *gx, float *goff, float *gw, float *gb, void *workspace,
                                  size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(grad_out); REQUIRE(gx); REQUIRE(goff); REQUIRE(gw);
    return deform_backward(x, weight, offset, nullptr, grad_out, nullptr, gx, goff, nullptr, gw, gb, workspace,
                           workspace_bytes, g, stream);
}

int ddcn_deform_conv_backward_saved_f32(const float *x, const float *weight, const float *offset,
                                        const float *grad_out, const float *columns, float *gx, float *goff,
                                        float *gw, float *gb, void *workspace, size_t workspace_bytes,
                                        const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(grad_out); REQUIRE(gx); REQUIRE(goff); REQUIRE(gw);
    return deform_backward(x, weight, offset, nullptr, grad_out, columns, gx, goff, nullptr, gw, gb, workspace,
                           workspace_bytes, g, stream);
}

int ddcn_modulated_deform_conv_forward_f32(const float *x, const float *weight, const float *bias,
                                           const float *offset, const float *mask, float *out, void *workspace,
                                           size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(mask); REQUIRE(out);
    return deform_forward(x, weight, bias, offset, mask, out, workspace, workspace_bytes, g, stream);
}

int ddcn_modulated_deform_conv_backward_f32(const float *x, const float *weight, const float *offset,
                                            const float *mask, const float *grad_out, float *gx, float *goff,
                                            float *gmask, float *gw, float *gb, void *workspace,
                                            size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(mask); REQUIRE(grad_out);
    REQUIRE(gx); REQUIRE(goff); REQUIRE(gmask); REQUIRE(gw);
    return deform_backward(x, weight, offset, mask, grad_out, nullptr, gx, goff, gmask, gw, gb, workspace,
                           workspace_bytes, g, stream);
}

int ddcn_modulated_deform_conv_backward_saved_f32(const float *x, const float *weight, const float *offset,
                                                  const float *mask, const float *grad_out, const float *columns,
                                                  float *gx, float *goff, float *gmask, float *gw, float *gb,
                                                  void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                                  void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(mask); REQUIRE(grad_out);
    REQUIRE(gx); REQUIRE(goff); REQUIRE(gmask); REQUIRE(gw);
    return deform_backward(x, weight, offset, mask, grad_out, columns, gx, goff, gmask, gw, gb, workspace,
                           workspace_bytes, g, stream);
}

int ddcn_deform_conv_backward_saved_2s_f32(const float *x, const float *weight, const float *offset,
                                           const float *mask, const float *grad_out, const float *columns, float *gx,
                                           float *goff, float *gmask, float *gw, float *gb, void *workspace,
                                           size_t workspace_bytes, const ddcn_geom *g, void *stream, void *side_stream,
                                           void *fork_event, void *join_event)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(grad_out); REQUIRE(columns);
    REQUIRE(gx); REQUIRE(goff); REQUIRE(gw); REQUIRE(side_stream); REQUIRE(fork_event); REQUIRE(join_event);
    if ((mask == nullptr) != (gmask == nullptr)) return fail(DDCN_ERR_BAD_ARG, "mask and gmask must both be given or both be NULL");
    if (side_stream == stream) return fail(DDCN_ERR_BAD_ARG, "side_stream must differ from stream");
    SideStream side;
    side.stream = (hipStream_t)side_stream; side.fork = (hipEvent_t)fork_event; side.join = (hipEvent_t)join_event;
    return deform_backward(x, weight, offset, mask, grad_out, columns, gx, goff, gmask, gw, gb, workspace,
                           workspace_bytes, g, stream, &side);
}

int ddcn_depthaware_conv_forward_f32(const float *x, const float *depth, const float *weight, const float *bias,
                                     float *out, void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                     void *stream)
{
    REQUIRE(x); REQUIRE(depth); REQUIRE(weight); REQUIRE(out);
    TRY(check_geom(g));
    TRY(check_ws(g, workspace, workspace_bytes));
    hipStream_t s = (hipStream_t)stream;
    float *col = (float *)workspace;
    TRY(launch_depthaware_im2col(x, depth, col, *g, s));
    TRY(launch_gemm_out(weight, col, bias, out, *g, s));
    return DDCN_OK;
}

int ddcn_depthaware_conv_backward_f32(const float *x, const float *depth, const float *weight,
                                      const float *grad_out, float *gx, float *gw, float *gb, void *workspace,
                                      size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(depth); REQUIRE(weight); REQUIRE(grad_out); REQUIRE(gx); REQUIRE(gw);
    TRY(check_geom(g));
    TRY(check_ws(g, workspace, workspace_bytes));
    hipStream_t s = (hipStream_t)stream;
    float *col = (float *)workspace;
    TRY(launch_gemm_gcol(weight, grad_out, col, *g, s));
    TRY(launch_depthaware_col2im(col, depth, gx, *g, s));
    TRY(launch_depthaware_im2col(x, depth, col, *g, s));
    TRY(launch_gemm_gweight(grad_out, col, gw, *g, s));
    if (gb) TRY(launch_bias_grad(grad_out, gb, *g, s));
    return DDCN_OK;
}

// ---------------------------------------------------------------- fp64
size_t ddcn_workspace_bytes_f64(const ddcn_geom *g)
{
    if (check_geom(g) != DDCN_OK) return 0;
    return f64_workspace_bytes(*g);
}

static int check_ws_f64(const ddcn_geom *g, void *ws, size_t bytes)
{
    if (!ws || bytes < f64_workspace_bytes(*g))
        return fail(DDCN_ERR_WORKSPACE, "fp64 workspace of %zu bytes, need %zu", bytes, f64_workspace_bytes(*g));
    return DDCN_OK;
}

int ddcn_deform_conv_forward_f64(const double *x, const double *weight, const double *bias, const double *offset,
                                 double *out, void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                 void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(out);
    TRY(check_geom(g));
    TRY(check_ws_f64(g, workspace, workspace_bytes));
    return f64_forward(x, weight, bias, offset, nullptr, nullptr, out, 0, (double *)workspace, *g, (hipStream_t)stream);
}

int ddcn_deform_conv_direct_forward_f64(const double *x, const double *weight, const double *bias,
                                        const double *offset, double *out, int sigmoid, void *workspace,
                                        size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(out);
    TRY(check_geom(g));
    TRY(check_ws_f64(g, workspace, workspace_bytes));
    return f64_forward(x, weight, bias, offset, nullptr, nullptr, out, sigmoid, (double *)workspace, *g,
                       (hipStream_t)stream);
}

int ddcn_modulated_deform_conv_forward_f64(const double *x, const double *weight, const double *bias,
                                           const double *offset, const double *mask, double *out, void *workspace,
                                           size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(mask); REQUIRE(out);
    TRY(check_geom(g));
    TRY(check_ws_f64(g, workspace, workspace_bytes));
    return f64_forward(x, weight, bias, offset, mask, nullptr, out, 0, (double *)workspace, *g, (hipStream_t)stream);
}

int ddcn_depthaware_conv_forward_f64(const double *x, const double *depth, const double *weight, const double *bias,
                                     double *out, void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                     void *stream)
{
    REQUIRE(x); REQUIRE(depth); REQUIRE(weight); REQUIRE(out);
    TRY(check_geom(g));
    TRY(check_ws_f64(g, workspace, workspace_bytes));
    return f64_forward(x, weight, bias, nullptr, nullptr, depth, out, 0, (double *)workspace, *g, (hipStream_t)stream);
}

int ddcn_deform_conv_backward_saved_f64(const double *x, const double *weight, const double *offset,
                                        const double *grad_out, const double *columns, double *gx, double *goff,
                                        double *gw, double *gb, void *workspace, size_t workspace_bytes,
                                        const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(grad_out); REQUIRE(gx); REQUIRE(goff); REQUIRE(gw);
    TRY(check_geom(g));
    TRY(check_ws_f64(g, workspace, workspace_bytes));
    return f64_backward(x, weight, offset, nullptr, nullptr, grad_out, columns, gx, goff, nullptr, gw, gb,
                        (double *)workspace, *g, (hipStream_t)stream);
}

int ddcn_deform_conv_backward_f64(const double *x, const double *weight, const double *offset, const double *grad_out,
                                  double *gx, double *goff, double *gw, double *gb, void *workspace,
                                  size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    return ddcn_deform_conv_backward_saved_f64(x, weight, offset, grad_out, nullptr, gx, goff, gw, gb, workspace,
                                               workspace_bytes, g, stream);
}

int ddcn_deform_conv_direct_backward_f64(const double *x, const double *weight, const double *offset,
                                         const double *grad_out, double *gx, double *goff, double *gw, double *gb,
                                         void *workspace, size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    return ddcn_deform_conv_backward_saved_f64(x, weight, offset, grad_out, nullptr, gx, goff, gw, gb, workspace,
                                               workspace_bytes, g, stream);
}

int ddcn_modulated_deform_conv_backward_saved_f64(const double *x, const double *weight, const double *offset,
                                                  const double *mask, const double *grad_out, const double *columns,
                                                  double *gx, double *goff, double *gmask, double *gw, double *gb,
                                                  void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                                  void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(mask); REQUIRE(grad_out);
    REQUIRE(gx); REQUIRE(goff); REQUIRE(gmask); REQUIRE(gw);
    TRY(check_geom(g));
    TRY(check_ws_f64(g, workspace, workspace_bytes));
    return f64_backward(x, weight, offset, mask, nullptr, grad_out, columns, gx, goff, gmask, gw, gb,
                        (double *)workspace, *g, (hipStream_t)stream);
}

int ddcn_modulated_deform_conv_backward_f64(const double *x, const double *weight, const double *offset,
                                            const double *mask, const double *grad_out, double *gx, double *goff,
                                            double *gmask, double *gw, double *gb, void *workspace,
                                            size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    return ddcn_modulated_deform_conv_backward_saved_f64(x, weight, offset, mask, grad_out, nullptr, gx, goff, gmask,
                                                         gw, gb, workspace, workspace_bytes, g, stream);
}

int ddcn_depthaware_conv_backward_f64(const double *x, const double *depth, const double *weight,
                                      const double *grad_out, double *gx, double *gw, double *gb, void *workspace,
                                      size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(depth); REQUIRE(weight); REQUIRE(grad_out); REQUIRE(gx); REQUIRE(gw);
    TRY(check_geom(g));
    TRY(check_ws_f64(g, workspace, workspace_bytes));
    return f64_backward(x, weight, nullptr, nullptr, depth, grad_out, nullptr, gx, nullptr, nullptr, gw, gb,
                        (double *)workspace, *g, (hipStream_t)stream);
}

// ---------------------------------------------------------------- bf16
size_t ddcn_workspace_bytes_bf16(const ddcn_geom *g)
{
    if (check_geom(g) != DDCN_OK) return 0;
    return bf16_workspace_bytes(*g);
}

int ddcn_bf16_supported(const ddcn_geom *g)
{
    TRY(check_geom(g));
    return bf16_supported(*g);
}

int ddcn_modulated_deform_conv_forward_bf16(const void *x, const void *weight, const void *bias, const void *offset,
                                            const void *mask, void *out, void *workspace, size_t workspace_bytes,
                                            const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(out);
    TRY(check_geom(g));
    TRY(bf16_supported(*g));
    if (!workspace || workspace_bytes < bf16_workspace_bytes(*g))
        return fail(DDCN_ERR_WORKSPACE, "bf16 workspace of %zu bytes, need %zu", workspace_bytes,
                    bf16_workspace_bytes(*g));
    return launch_forward_bf16(x, weight, bias, offset, mask, out, workspace, *g, (hipStream_t)stream);
}

int ddcn_modulated_deform_conv_forward_f16(const void *x, const void *weight, const void *bias, const void *offset,
                                           const void *mask, void *out, void *workspace, size_t workspace_bytes,
                                           const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(out);
    TRY(check_geom(g));
    TRY(bf16_supported(*g));
    if (!workspace || workspace_bytes < bf16_workspace_bytes(*g))
        return fail(DDCN_ERR_WORKSPACE, "fp16 workspace of %zu bytes, need %zu", workspace_bytes,
                    bf16_workspace_bytes(*g));
    return launch_forward_f16(x, weight, bias, offset, mask, out, workspace, *g, (hipStream_t)stream);
}

// ---------------------------------------------------------------- 16-bit forward without columns
int ddcn_fused16_supported(const ddcn_geom *g)
{
    TRY(check_geom(g));
    return fused16_supported(*g);
}

size_t ddcn_fused16_workspace_bytes(const ddcn_geom *g)
{
    if (check_geom(g) != DDCN_OK) return 0;
    return fused16_workspace_bytes(*g);
}

static int fused_forward(bool bf16, const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                         void *out, void *ws, size_t ws_bytes, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(weight); REQUIRE(offset); REQUIRE(out);
    TRY(check_geom(g));
    TRY(fused16_supported(*g));
    if (!ws || ws_bytes < fused16_workspace_bytes(*g) || ((uintptr_t)ws & 15))
        return fail(DDCN_ERR_WORKSPACE, "fused 16-bit forward: workspace of %zu bytes (16-byte aligned), need %zu", ws_bytes,
                    fused16_workspace_bytes(*g));
    return bf16 ? launch_fused_forward_bf16(x, weight, bias, offset, mask, out, ws, *g, (hipStream_t)stream)
                : launch_fused_forward_f16(x, weight, bias, offset, mask, out, ws, *g, (hipStream_t)stream);
}

int ddcn_modulated_deform_conv_forward_fused_bf16(const void *x, const void *weight, const void *bias, const void *offset,
                                                  const void *mask, void *out, void *workspace, size_t workspace_bytes,
                                                  const ddcn_geom *g, void *stream)
{
    return fused_forward(true, x, weight, bias, offset, mask, out, workspace, workspace_bytes, g, stream);
}

int ddcn_modulated_deform_conv_forward_fused_f16(const void *x, const void *weight, const void *bias, const void *offset,
                                                 const void *mask, void *out, void *workspace, size_t workspace_bytes,
                                                 const ddcn_geom *g, void *stream)
{
    return fused_forward(false, x, weight, bias, offset, mask, out, workspace, workspace_bytes, g, stream);
}

// ---------------------------------------------------------------- 16-bit backward: G2 on 16-bit operands
size_t ddcn_gemm_gcol_16_workspace_bytes(const ddcn_geom *g)
{
    if (check_geom(g) != DDCN_OK || gcol16_supported(*g) != DDCN_OK) return 0;
    return gcol16_workspace_bytes(*g);
}

int ddcn_gemm_gcol_16_supported(const ddcn_geom *g)
{
    TRY(check_geom(g));
    return gcol16_supported(*g);
}

static int gemm_gcol_16(bool bf16, const void *weight, const void *grad_out, float *gcol, void *ws, size_t ws_bytes,
                        const ddcn_geom *g, void *stream)
{
    REQUIRE(weight); REQUIRE(grad_out); REQUIRE(gcol);
    TRY(check_geom(g));
    TRY(gcol16_supported(*g));
    if (!ws || ws_bytes < gcol16_workspace_bytes(*g) || ((uintptr_t)ws & 15))
        return fail(DDCN_ERR_WORKSPACE, "16-bit G2 workspace of %zu bytes (16-byte aligned), need %zu", ws_bytes,
                    gcol16_workspace_bytes(*g));
    if ((uintptr_t)gcol & 15) return fail(DDCN_ERR_BAD_ARG, "gcol must be 16-byte aligned");
    return bf16 ? launch_gemm_gcol_bf16(weight, grad_out, gcol, ws, *g, (hipStream_t)stream)
                : launch_gemm_gcol_f16(weight, grad_out, gcol, ws, *g, (hipStream_t)stream);
}

int ddcn_gemm_gcol_bf16(const void *weight, const void *grad_out, float *gcol, void *workspace, size_t workspace_bytes,
                        const ddcn_geom *g, void *stream)
{
    return gemm_gcol_16(true, weight, grad_out, gcol, workspace, workspace_bytes, g, stream);
}

int ddcn_gemm_gcol_f16(const void *weight, const void *grad_out, float *gcol, void *workspace, size_t workspace_bytes,
                       const ddcn_geom *g, void *stream)
{
    return gemm_gcol_16(false, weight, grad_out, gcol, workspace, workspace_bytes, g, stream);
}

// ---------------------------------------------------------------- 16-bit backward: grad_weight from 16-bit tensors
size_t ddcn_gweight_16_workspace_bytes(const ddcn_geom *g)
{
    if (check_geom(g) != DDCN_OK || gweight16_supported(*g) != DDCN_OK) return 0;
    return gweight16_workspace_bytes(*g);
}

int ddcn_gweight_16_supported(const ddcn_geom *g)
{
    TRY(check_geom(g));
    return gweight16_supported(*g);
}

static int gweight_16(bool bf16, const void *x, const void *offset, const void *mask, const void *grad_out,
                      float *grad_weight, void *ws, size_t ws_bytes, const ddcn_geom *g, void *stream)
{
    REQUIRE(x); REQUIRE(offset); REQUIRE(grad_out); REQUIRE(grad_weight);
    TRY(check_geom(g));
    TRY(gweight16_supported(*g));
    if (!ws || ws_bytes < gweight16_workspace_bytes(*g) || ((uintptr_t)ws & 15))
        return fail(DDCN_ERR_WORKSPACE, "16-bit G3 workspace of %zu bytes (16-byte aligned), need %zu", ws_bytes,
                    gweight16_workspace_bytes(*g));
    return bf16 ? launch_gweight_bf16(x, offset, mask, grad_out, grad_weight, ws, *g, (hipStream_t)stream)
                : launch_gweight_f16(x, offset, mask, grad_out, grad_weight, ws, *g, (hipStream_t)stream);
}

int ddcn_deform_conv_gweight_bf16(const void *x, const void *offset, const void *mask, const void *grad_out,
                                  float *grad_weight, void *workspace, size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    return gweight_16(true, x, offset, mask, grad_out, grad_weight, workspace, workspace_bytes, g, stream);
}

int ddcn_deform_conv_gweight_f16(const void *x, const void *offset, const void *mask, const void *grad_out,
                                 float *grad_weight, void *workspace, size_t workspace_bytes, const ddcn_geom *g, void *stream)
{
    return gweight_16(false, x, offset, mask, grad_out, grad_weight, workspace, workspace_bytes, g, stream);
}

int ddcn_deform_psroi_pool_forward_f32(const float *data, const float *rois, const float *offset, float *out,
                                       float *out_count, const ddcn_roi_geom *g, void *stream)
{
    return launch_psroi_forward(data, rois, offset, out, out_count, g, (hipStream_t)stream);
}

int ddcn_deform_psroi_pool_backward_f32(const float *grad_out, const float *data, const float *rois,
                                        const float *offset, const float *out_count, float *grad_data,
                                        float *grad_offset, const ddcn_roi_geom *g, void *stream)
{
    return launch_psroi_backward(grad_out, data, rois, offset, out_count, grad_data, grad_offset, g,
                                 (hipStream_t)stream);
}

int ddcn_deform_psroi_pool_forward_f64(const double *data, const double *rois, const double *offset, double *out,
                                       double *out_count, const ddcn_roi_geom *g, void *stream)
{
    return launch_psroi_forward_f64(data, rois, offset, out, out_count, g, (hipStream_t)stream);
}

int ddcn_deform_psroi_pool_backward_f64(const double *grad_out, const double *data, const double *rois,
                                        const double *offset, const double *out_count, double *grad_data,
                                        double *grad_offset, const ddcn_roi_geom *g, void *stream)
{
    return launch_psroi_backward_f64(grad_out, data, rois, offset, out_count, grad_data, grad_offset, g,
                                     (hipStream_t)stream);
}

}  // extern "C"

/*
 * ddcn_hip.h -- C ABI of libddcn_hip.so, the MI355X (gfx950) replacement for the
 * reference's CUDA extension `detseg.ops.dcn.deform_conv_cuda`.
 *
 * Reference paths below are relative to /root/reference:
 *   PY = detseg/ops/dcn/deform_conv.py
 *   H  = detseg/ops/dcn/src/deform_conv_cuda.cpp     (pybind11 module, H:758-775)
 *   K  = detseg/ops/dcn/src/deform_conv_cuda_kernel.cu
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a contiguous NCHW fp32 tensor (tensor.data_ptr()),
 *     `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - no entry point allocates, frees or synchronises; scratch (`workspace`) is brought by the
 *     caller (ddcn_workspace_bytes says how much) -- the reference allocated and zero-filled its
 *     `columns` tensor inside every call (H:214-216, H:292-293);
 *   - every function returns 0 (DDCN_OK) or a negative DDCN_ERR_* code and never throws;
 *     ddcn_strerror() names the code, ddcn_last_error() gives the detail for the calling thread;
 *   - what the library keeps between calls: a per-device flag per kernel that its dynamic-LDS limit has been raised
 *     (hipFuncSetAttribute, once per device), tuning knobs read from the environment at first use (DDCN_*), the
 *     last-error text per thread, and the per-device address of the static scratch of the *_static_f32 entry point.
 *     No call touches another call's data: concurrent calls from several threads / streams are safe (the *_static_f32
 *     entry point excepted, see there).
 *
 * Tensor shapes (K2 = kh*kw, P = Ho*Wo):
 *   x[N,C,H,W]  weight[Co,C/groups,kh,kw]  bias[Co]  offset[N,dg*2*K2,Ho,Wo]  (channel 2k = dh, 2k+1 = dw)
 *   mask[N,dg*K2,Ho,Wo]  depth[N,1,H,W]  out/grad_out[N,Co,Ho,Wo]
 *   columns workspace: [N][C*K2][ld] with ld = ddcn_col_ld(geom) >= P (rows padded to 128 B).
 */
#ifndef DDCN_HIP_H
#define DDCN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDCN_VERSION 100

typedef struct ddcn_geom {
    int N, C, H, W;     /* input  */
    int Co;             /* output channels */
    int kh, kw;         /* kernel */
    int sh, sw;         /* stride */
    int ph, pw;         /* zero padding */
    int dh, dw;         /* dilation */
    int groups;         /* weight groups */
    int dg;             /* deformable groups (1 for the depth-aware op) */
    int Ho, Wo;         /* output size; must equal ddcn_infer_output() */
} ddcn_geom;

enum {
    DDCN_OK = 0,
    DDCN_ERR_BAD_ARG = -1,     /* null pointer / non-positive size */
    DDCN_ERR_SHAPE = -2,       /* geometry inconsistent (channels vs groups, output size, ...) */
    DDCN_ERR_WORKSPACE = -3,   /* workspace smaller than ddcn_workspace_bytes() */
    DDCN_ERR_LAUNCH = -4,      /* HIP reported a launch error */
    DDCN_ERR_UNSUPPORTED = -5, /* size beyond a documented limit */
    DDCN_ERR_NO_DEVICE = -6    /* no gfx950 device / wrong architecture */
};

int ddcn_version(void);
const char *ddcn_strerror(int rc);
const char *ddcn_last_error(void);

/* Ho = (H + 2p - (d(k-1)+1)) / s + 1 -- PY:71-80 (_infer_shape), H:200-203.  Fills g->Ho, g->Wo. */
int ddcn_infer_output(ddcn_geom *g);
/* Checks of the never-called shape_check() H:76-171 plus H:193-198.  0 or DDCN_ERR_SHAPE. */
int ddcn_check_geom(const ddcn_geom *g);
/* Leading dimension (floats) of one columns row and bytes of the whole [N][C*K2][ld] workspace. */
size_t ddcn_col_ld(const ddcn_geom *g);
size_t ddcn_workspace_bytes(const ddcn_geom *g);

/* ---------------------------------------------------------------- stage level (one launch each) */

/* Bilinear gather, all N images in one launch.  mask == NULL: K:189-242 (deformable_im2col_gpu_kernel);
 * else K:569-632 (modulated_deformable_im2col_gpu_kernel).  Writes col[N][C*K2][ld]. */
int ddcn_im2col_f32(const float *x, const float *offset, const float *mask, float *col, const ddcn_geom *g,
                    void *stream);

/* Fused backward scatter: K:278-334 / K:634-692 (col2im, accumulates into gx) together with
 * K:372-435 / K:694-766 (col2im_coord, OVERWRITES goff and gmask).  gmask and mask are NULL for the
 * non-modulated op.  gcol is the [N][C*K2][ld] result of the W^T*grad_out GEMM. */
int ddcn_col2im_coord_f32(const float *gcol, const float *x, const float *offset, const float *mask, float *gx,
                          float *goff, float *gmask, const ddcn_geom *g, void *stream);

/* Depth-aware integer-tap gather / scatter: K:868-918 and K:952-996. */
int ddcn_depthaware_im2col_f32(const float *x, const float *depth, float *col, const ddcn_geom *g, void *stream);
int ddcn_depthaware_col2im_f32(const float *gcol, const float *depth, float *gx, const ddcn_geom *g,
                               void *stream);

/* The three contractions the reference hands to at::addmm_ (cuBLAS), here fp32 MFMA kernels:
 *   G1 H:232-237 / H:430-435   out[n][g]  = W[g] * col[n][g] (+ bias)       (bias may be NULL)
 *   G2 H:305-308 / H:503-506   col[n][g]  = W[g]^T * grad_out[n][g]
 *   G3 H:341-346 / H:539-544   gw[g]     += sum_n grad_out[n][g] * col[n][g]^T   (accumulates)
 *   G4 H:347-353 / H:545-551   gb        += sum_{n,p} grad_out                   (accumulates) */
int ddcn_gemm_out_f32(const float *weight, const float *col, const float *bias, float *out, const ddcn_geom *g,
                      void *stream);
int ddcn_gemm_gcol_f32(const float *weight, const float *grad_out, float *gcol, const ddcn_geom *g, void *stream);
int ddcn_gemm_gweight_f32(const float *grad_out, const float *col, float *gw, const ddcn_geom *g, void *stream);
int ddcn_bias_grad_f32(const float *grad_out, float *gb, const ddcn_geom *g, void *stream);

/* ---------------------------------------------------------------- whole-op entry points
 * One per function of the reference's pybind module (H:758-775).  The reference's `ones` and
 * `columns` tensor arguments are replaced by (workspace, workspace_bytes); `with_bias` by a NULL
 * bias / gb pointer; the eight kernel/stride/pad/dilation ints, group and deformable_group by geom.
 * Backward ACCUMULATES into gx, goff, gmask, gw, gb: zero-fill them first, as the reference's Python does for
 * all five (PY:54-57, PY:126-130) -- one fill of one flat buffer in semantic-segmentation_amd/dcn/deform_conv.py. */

/* replaces deform_conv_cuda_forward (H:173-251) */
int ddcn_deform_conv_forward_f32(const float *x, const float *weight, const float *bias, const float *offset,
                                 float *out, void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                 void *stream);
/* No counterpart in the reference's extension: deform_conv forward for FEW output channels (the 9-channel mask branch
 * of the DDCN layer, PY:438-446 and PY:457-458, which the reference computes as im2col + a 9-row GEMM + sigmoid), as
 * one direct kernel: the bilinear samples are multiplied into the output channels in registers, no columns are
 * written.  out = sigmoid ? 1/(1+exp(-v)) : v with v = bias + W * sample(x, offset).  Covered: 3x3, groups 1,
 * deformable_groups 1, Co <= 12; anything else runs columns + GEMM (+ activation) inside the same call.
 * workspace as for ddcn_deform_conv_forward_f32. */
int ddcn_deform_conv_direct_forward_f32(const float *x, const float *weight, const float *bias, const float *offset,
                                        float *out, int sigmoid, void *workspace, size_t workspace_bytes,
                                        const ddcn_geom *g, void *stream);
/* The backward of the direct forward above (same coverage; anything else behaves like ddcn_deform_conv_backward_f32):
 * NO columns at all.  gx / goff (H:299-323: a Co-row GEMM into the columns tensor, then col2im and col2im_coord) come
 * from the fused scatter kernel in a mode that forms each column gradient, a Co-term dot product of the repacked
 * weights with the pixel's Co output gradients, right before it scatters it; the weight gradient (H:325-346: a second
 * im2col, then a Co-row GEMM) from ONE kernel that samples x again and multiplies the samples into Co x 9
 * accumulators per input channel.  grad_out is the gradient w.r.t. the PRE-activation value (the
 * caller applies the sigmoid derivative).  ACCUMULATES into gx, goff, gw, gb like the other backward entry points. */
int ddcn_deform_conv_direct_backward_f32(const float *x, const float *weight, const float *offset,
                                         const float *grad_out, float *gx, float *goff, float *gw, float *gb,
                                         void *workspace, size_t workspace_bytes, const ddcn_geom *g, void *stream);

/* The same backward with the weight gradient (and bias gradient) on a second stream of the caller, beside the scatter:
 * the caller passes the side stream and two events (recorded at least once), the library creates nothing and joins the
 * side stream back into `stream` before returning.  No counterpart in the reference (one stream, H:253-367). */
int ddcn_deform_conv_direct_backward_2s_f32(const float *x, const float *weight, const float *offset,
                                            const float *grad_out, float *gx, float *goff, float *gw, float *gb,
                                            void *workspace, size_t workspace_bytes, const ddcn_geom *g, void *stream,
                                            void *side_stream, void *fork_event, void *join_event);
/* replaces deform_conv_cuda_backward (H:253-367) */
int ddcn_deform_conv_backward_f32(const float *x, const float *weight, const float *offset, const float *grad_out,
                                  float *gx, float *goff, float *gw, float *gb, void *workspace,
                                  size_t workspace_bytes, const ddcn_geom *g, void *stream);
/* replaces modulated_deform_conv_cuda_forward (H:370-449) */
int ddcn_modulated_deform_conv_forward_f32(const float *x, const float *weight, const float *bias,
                                           const float *offset, const float *mask, float *out, void *workspace,
                                           size_t workspace_bytes, const ddcn_geom *g, void *stream);
/* No counterpart in the reference: the same forward (mask == NULL: the non-modulated op, bias may be NULL) for callers
 * that CANNOT ALLOCATE -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc -- and so
 * cannot bring a workspace either.  It calls no hipMalloc / hipFree / allocating routine: the columns live in a
 * file-scope device array of DDCN_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process
 * ends.  Limit: ddcn_workspace_bytes(g) <= DDCN_STATIC_SCRATCH_BYTES, else DDCN_ERR_UNSUPPORTED (nothing is launched).
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (other streams / threads) race. */
#define DDCN_STATIC_SCRATCH_BYTES (64u << 20)
size_t ddcn_static_scratch_bytes(void);
int ddcn_modulated_deform_conv_forward_static_f32(const float *x, const float *weight, const float *bias,
                                                  const float *offset, const float *mask, float *out,
                                                  const ddcn_geom *g, void *stream);
/* replaces modulated_deform_conv_cuda_backward (H:451-565) */
int ddcn_modulated_deform_conv_backward_f32(const float *x, const float *weight, const float *offset,
                                            const float *mask, const float *grad_out, float *gx, float *goff,
                                            float *gmask, float *gw, float *gb, void *workspace,
                                            size_t workspace_bytes, const ddcn_geom *g, void *stream);
/* Backward variants that reuse the columns the forward call left in ITS workspace (`columns`, the [N][C*K2][ld]
 * buffer of the matching forward call, unchanged since) instead of recomputing them: the reference's backward
 * re-runs im2col per image (H:325-327 / H:524-527) because its `columns` tensor is overwritten in between.
 * `workspace` is still needed for the column gradients.  columns == NULL behaves like the plain entry point. */
int ddcn_deform_conv_backward_saved_f32(const float *x, const float *weight, const float *offset,
                                        const float *grad_out, const float *columns, float *gx, float *goff,
                                        float *gw, float *gb, void *workspace, size_t workspace_bytes,
                                        const ddcn_geom *g, void *stream);
int ddcn_modulated_deform_conv_backward_saved_f32(const float *x, const float *weight, const float *offset,
                                                  const float *mask, const float *grad_out, const float *columns,
                                                  float *gx, float *goff, float *gmask, float *gw, float *gb,
                                                  void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                                  void *stream);
/* The saved-columns backward on TWO streams of the caller (mask / gmask NULL: the non-modulated op).  The reference
 * runs its backward stages one after the other (H:497-561); here the weight-gradient GEMM (G3, + G4), which depends only
 * on grad_out and the saved columns, is launched on `side_stream` beside the LDS-atomic-bound scatter, which then runs
 * in a variant small enough to share each CU with the resident GEMM workgroup.  `fork_event` / `join_event` are two
 * hipEvent_t of the caller (the library creates nothing): fork is recorded on `stream` behind G2, `side_stream` waits
 * for it; join is recorded on `side_stream` after G3 and `stream` waits for it before the call's work counts as done, so
 * the caller sees single-stream semantics on `stream`.  Everything else as the *_backward_saved_f32 entry points. */
int ddcn_deform_conv_backward_saved_2s_f32(const float *x, const float *weight, const float *offset,
                                           const float *mask, const float *grad_out, const float *columns, float *gx,
                                           float *goff, float *gmask, float *gw, float *gb, void *workspace,
                                           size_t workspace_bytes, const ddcn_geom *g, void *stream, void *side_stream,
                                           void *fork_event, void *join_event);
/* replaces depthaware_conv_cuda_forward (H:567-647) */
int ddcn_depthaware_conv_forward_f32(const float *x, const float *depth, const float *weight, const float *bias,
                                     float *out, void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                     void *stream);
/* replaces depthaware_conv_cuda_backward (H:649-756); no gradient flows to depth (PY:207) */
int ddcn_depthaware_conv_backward_f32(const float *x, const float *depth, const float *weight,
                                      const float *grad_out, float *gx, float *gw, float *gb, void *workspace,
                                      size_t workspace_bytes, const ddcn_geom *g, void *stream);

/* ---------------------------------------------------------------- fp64
 * The reference instantiates every kernel for double too (AT_DISPATCH_FLOATING_TYPES_AND_HALF, K:258) and its GPU
 * self-test is an fp64 gradcheck (detseg/ops/dcn/test.py:208-267).  Same entry points as above with double tensors and
 * a workspace of ddcn_workspace_bytes_f64() bytes; same rules (backward accumulates, columns may be NULL).  These
 * kernels are for validation in double precision and are not tuned; the gw reduction is deterministic. */
size_t ddcn_workspace_bytes_f64(const ddcn_geom *g);
int ddcn_deform_conv_forward_f64(const double *x, const double *weight, const double *bias, const double *offset,
                                 double *out, void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                 void *stream);
int ddcn_deform_conv_direct_forward_f64(const double *x, const double *weight, const double *bias,
                                        const double *offset, double *out, int sigmoid, void *workspace,
                                        size_t workspace_bytes, const ddcn_geom *g, void *stream);
int ddcn_deform_conv_direct_backward_f64(const double *x, const double *weight, const double *offset,
                                         const double *grad_out, double *gx, double *goff, double *gw, double *gb,
                                         void *workspace, size_t workspace_bytes, const ddcn_geom *g, void *stream);
int ddcn_deform_conv_backward_f64(const double *x, const double *weight, const double *offset, const double *grad_out,
                                  double *gx, double *goff, double *gw, double *gb, void *workspace,
                                  size_t workspace_bytes, const ddcn_geom *g, void *stream);
int ddcn_deform_conv_backward_saved_f64(const double *x, const double *weight, const double *offset,
                                        const double *grad_out, const double *columns, double *gx, double *goff,
                                        double *gw, double *gb, void *workspace, size_t workspace_bytes,
                                        const ddcn_geom *g, void *stream);
int ddcn_modulated_deform_conv_forward_f64(const double *x, const double *weight, const double *bias,
                                           const double *offset, const double *mask, double *out, void *workspace,
                                           size_t workspace_bytes, const ddcn_geom *g, void *stream);
int ddcn_modulated_deform_conv_backward_f64(const double *x, const double *weight, const double *offset,
                                            const double *mask, const double *grad_out, double *gx, double *goff,
                                            double *gmask, double *gw, double *gb, void *workspace,
                                            size_t workspace_bytes, const ddcn_geom *g, void *stream);
int ddcn_modulated_deform_conv_backward_saved_f64(const double *x, const double *weight, const double *offset,
                                                  const double *mask, const double *grad_out, const double *columns,
                                                  double *gx, double *goff, double *gmask, double *gw, double *gb,
                                                  void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                                  void *stream);
int ddcn_depthaware_conv_forward_f64(const double *x, const double *depth, const double *weight, const double *bias,
                                     double *out, void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                     void *stream);
int ddcn_depthaware_conv_backward_f64(const double *x, const double *depth, const double *weight,
                                      const double *grad_out, double *gx, double *gw, double *gb, void *workspace,
                                      size_t workspace_bytes, const ddcn_geom *g, void *stream);

/* ---------------------------------------------------------------- native bf16 forward (inference)
 * No counterpart in the reference (it dispatches fp64/fp32/fp16 only, K:258): BASELINE config 5, "bf16 activations with
 * fp32 accumulation".  Every tensor pointer is bfloat16 (torch.bfloat16 data_ptr()): x, weight, bias, offset, mask,
 * out; coordinates and bilinear weights are computed in fp32, samples are rounded to bf16 columns, the contraction
 * accumulates in fp32 on v_mfma_f32_16x16x32_bf16 (32x32x16 in the register-staged fallback) and the result (+ bias) is rounded to bf16 once.  Arithmetic
 * restated: K:569-632 / K:189-242 (im2col) and H:430-435 / H:232-237 (GEMM).  mask == NULL gives the non-modulated
 * op, bias may be NULL.  Covered: 3x3 kernels with C, C/groups and C/deformable_groups multiples of 8
 * (ddcn_bf16_supported() == 0); workspace of ddcn_workspace_bytes_bf16() bytes (the bf16 columns of all N images). */
size_t ddcn_workspace_bytes_bf16(const ddcn_geom *g);
int ddcn_bf16_supported(const ddcn_geom *g);
int ddcn_modulated_deform_conv_forward_bf16(const void *x, const void *weight, const void *bias, const void *offset,
                                            const void *mask, void *out, void *workspace, size_t workspace_bytes,
                                            const ddcn_geom *g, void *stream);

/* The same forward for IEEE half tensors (torch.float16): the reference's half instantiation (K:258) computes in half
 * throughout; here coordinates and accumulation are fp32 and only the columns and the result are rounded to half.
 * Same coverage (ddcn_bf16_supported) and workspace (ddcn_workspace_bytes_bf16) as the bf16 entry point. */
int ddcn_modulated_deform_conv_forward_f16(const void *x, const void *weight, const void *bias, const void *offset,
                                           const void *mask, void *out, void *workspace, size_t workspace_bytes,
                                           const ddcn_geom *g, void *stream);

/* The 16-bit forward WITHOUT a columns buffer ("an implicit-GEMM main op that never materialises columns";
 * the reference's forward, PY:457-463 -> H:419-435, writes and re-reads 302 MB of columns per image at BASELINE config 5's
 * layer4).  One kernel samples the B tile of every 32-k step (one tap x 32 channels) straight into LDS, from an input patch
 * staged by DMA out of a channels-last copy of x, and multiplies it on v_mfma_f32_16x16x32 while the next step is being
 * sampled.  Same arithmetic and the same rounding points as the columns path (a sample is rounded to 16 bits once, the
 * sum accumulates in fp32).  Covered (ddcn_fused16_supported() == 0): 3x3 kernels, groups = 1, C and C/deformable_groups
 * multiples of 32, Co a multiple of 8, a tile's input patch that fits LDS beside the weight ring (strides above 1 with a
 * wide dilation do not).  workspace: ddcn_fused16_workspace_bytes() = the channels-last copy of x + the repacked weights.
 * Opt-in (the Python layer takes it with DDCN_BF16_FUSED=1): measured equal to the columns path at C = 128 and slower at
 * C >= 256 (profiles/r03c_fused16.txt) -- for callers that cannot afford the columns workspace, not for speed. */
int ddcn_fused16_supported(const ddcn_geom *g);
size_t ddcn_fused16_workspace_bytes(const ddcn_geom *g);
int ddcn_modulated_deform_conv_forward_fused_bf16(const void *x, const void *weight, const void *bias, const void *offset,
                                                  const void *mask, void *out, void *workspace, size_t workspace_bytes,
                                                  const ddcn_geom *g, void *stream);
int ddcn_modulated_deform_conv_forward_fused_f16(const void *x, const void *weight, const void *bias, const void *offset,
                                                 const void *mask, void *out, void *workspace, size_t workspace_bytes,
                                                 const ddcn_geom *g, void *stream);

/* G2 of the backward on 16-bit operands (the reference's half instantiation of H:305-308 / H:503-506): weight and grad_out
 * are bfloat16 / float16 tensors, the column gradients gcol[N][C*K2][ld] come out in fp32 (fp32 accumulation on
 * v_mfma_f32_16x16x32_{bf16,f16}), ready for ddcn_col2im_coord_f32.  `workspace` (ddcn_gemm_gcol_16_workspace_bytes)
 * takes the repacked operands.  Covered: Co/groups a multiple of 8 (ddcn_gemm_gcol_16_supported() == 0); any kernel
 * size.  Used by the autograd functions for half / bfloat16 training: forward on the native 16-bit kernels, G2 here,
 * G3 through ddcn_deform_conv_gweight_{bf16,f16} below, the scatter on the fp32 kernel. */
size_t ddcn_gemm_gcol_16_workspace_bytes(const ddcn_geom *g);
int ddcn_gemm_gcol_16_supported(const ddcn_geom *g);
int ddcn_gemm_gcol_bf16(const void *weight, const void *grad_out, float *gcol, void *workspace, size_t workspace_bytes,
                        const ddcn_geom *g, void *stream);
int ddcn_gemm_gcol_f16(const void *weight, const void *grad_out, float *gcol, void *workspace, size_t workspace_bytes,
                       const ddcn_geom *g, void *stream);

/* G3 of the backward on 16-bit tensors (the reference's half instantiation of H:311-320 / H:539-544): grad_weight
 * (fp32, [Co][C/groups][3][3], accumulated: gw += ...) from the bfloat16 / float16 input, offset, mask (NULL = the
 * non-modulated op) and grad_out.  The columns are recomputed by the 16-bit im2col straight into the layout the matrix
 * core contracts over pixels (pixel octets, never fp32), the product runs on v_mfma_f32_16x16x32_{bf16,f16} with fp32
 * accumulation, split over images and pixel ranges.  `workspace`: ddcn_gweight_16_workspace_bytes().  Covered
 * (ddcn_gweight_16_supported() == 0): what ddcn_bf16_supported() covers, with Co/groups a multiple of 8 and a patch that
 * fits the LDS tile kernel. */
size_t ddcn_gweight_16_workspace_bytes(const ddcn_geom *g);
int ddcn_gweight_16_supported(const ddcn_geom *g);
int ddcn_deform_conv_gweight_bf16(const void *x, const void *offset, const void *mask, const void *grad_out,
                                  float *grad_weight, void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                  void *stream);
int ddcn_deform_conv_gweight_f16(const void *x, const void *offset, const void *mask, const void *grad_out,
                                 float *grad_weight, void *workspace, size_t workspace_bytes, const ddcn_geom *g,
                                 void *stream);

/* ---------------------------------------------------------------- deformable PS-RoI pooling
 * Replaces the reference's second extension, `detseg.ops.dcn.deform_pool_cuda`:
 *   PP = detseg/ops/dcn/deform_pool.py
 *   PH = detseg/ops/dcn/src/deform_pool_cuda.cpp      (pybind11 module, PH:84-90)
 *   PK = detseg/ops/dcn/src/deform_pool_cuda_kernel.cu
 * data[B,C,H,W] with C = out_channels*group_size^2 score maps, rois[R,5] = (batch index, x1, y1, x2, y2) in image
 * pixels, offset[R,trans_channels,part_size,part_size] (channel 2k = x, 2k+1 = y shift of class k, in units of the
 * RoI size / trans_std; NULL and ignored when no_trans), out / out_count / grad_out[R,out_channels,pooled,pooled].
 * The struct carries the scalar arguments of PH:34-38 in that order. */
typedef struct ddcn_roi_geom {
    int B, C, H, W;        /* data */
    int R;                 /* number of RoIs (0 is allowed: nothing is launched) */
    int out_channels;      /* output_dim */
    int pooled;            /* pooled_size (out_size; square, PP:25-29) */
    int group_size;
    int part_size;
    int sample_per_part;
    int trans_channels;    /* offset.size(1); ignored when no_trans (PH:46) */
    int no_trans;
    float spatial_scale;
    float trans_std;       /* in [0, 1] (PP:40) */
} ddcn_roi_geom;

/* replaces deform_psroi_pooling_cuda_forward (PH:34-56, kernel PK:53-139).  Writes out and out_count (the number of
 * samples each bin kept, which backward divides by).  A RoI whose batch index is outside [0, B) pools nothing
 * (output 0, count 0); the reference reads out of bounds there. */
int ddcn_deform_psroi_pool_forward_f32(const float *data, const float *rois, const float *offset, float *out,
                                       float *out_count, const ddcn_roi_geom *g, void *stream);
/* replaces deform_psroi_pooling_cuda_backward (PH:58-82, kernel PK:141-263).  ACCUMULATES into grad_data and
 * grad_offset (zero-fill them first, as PP:66-68 does); grad_offset may be NULL when no_trans.  No gradient flows to
 * rois (PP:67).  Limits: pooled, part_size <= 32 (DDCN_ERR_UNSUPPORTED beyond). */
int ddcn_deform_psroi_pool_backward_f32(const float *grad_out, const float *data, const float *rois,
                                        const float *offset, const float *out_count, float *grad_data,
                                        float *grad_offset, const ddcn_roi_geom *g, void *stream);

/* The same two entry points for double tensors (the reference instantiates its pooling kernels for every floating
 * type, PK:293 / PK:342); used by the fp64 gradcheck of the pooling operator. */
int ddcn_deform_psroi_pool_forward_f64(const double *data, const double *rois, const double *offset, double *out,
                                       double *out_count, const ddcn_roi_geom *g, void *stream);
int ddcn_deform_psroi_pool_backward_f64(const double *grad_out, const double *data, const double *rois,
                                        const double *offset, const double *out_count, double *grad_data,
                                        double *grad_offset, const ddcn_roi_geom *g, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DDCN_HIP_H */

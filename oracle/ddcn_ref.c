/*
 * ORACLE -- CPU restatement of the reference's DCN hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (semantic-segmentation_amd/) never does.
 *
 * What it restates: detseg/ops/dcn/src/deform_conv_cuda_kernel.cu (the eight
 * CUDA kernels and their device helpers) and the per-image host loops of
 * detseg/ops/dcn/src/deform_conv_cuda.cpp, in plain C, in fp32 and fp64.
 * Each function in ddcn_ref_impl.inc cites the reference file:line it follows.
 *
 * How it is pinned: the reference ships NO golden vectors and its CUDA source
 * cannot be built here (no nvcc, torch<=1.3 ATen API), so the pins are the
 * reference's own known-answer self-tests, detseg/ops/dcn/test.py:17-267
 * (zero-offset == nn.Conv2d, identity weights, fp64 gradcheck), plus an
 * independently written grid_sample/autograd formulation (tests/indep_ref.py).
 * See tests/test_oracle.py.  The deformable PS-RoI pooling restatement
 * (psroi_ref_impl.inc: deform_pool_cuda_kernel.cu) has no reference test at all:
 * "parity unpinned", see that file's header and tests/test_oracle_psroi.py.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared; no -march=native: the
 * .so is built in the authoring container and runs on the GPU box's host CPU;
 * the GEMM loops are multi-versioned with target_clones instead).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>

typedef struct {
    int N, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, Ho, Wo;
} ddcn_ref_geom;

/* deformable PS-RoI pooling (psroi_ref_impl.inc); mirrors the argument list of
 * detseg/ops/dcn/src/deform_pool_cuda.cpp:34-38 */
typedef struct {
    int B, C, H, W, R, out_channels, pooled, group_size, part_size, sample_per_part, trans_channels, no_trans;
    float spatial_scale, trans_std;
} ddcn_ref_roi_geom;

#if defined(__x86_64__)
#define GEMM_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define GEMM_CLONES
#endif

#define REAL float
#define SUF _f32
#define REXP expf
#define RFABS fabsf
#include "ddcn_ref_impl.inc"
#include "psroi_ref_impl.inc"
#undef REAL
#undef SUF
#undef REXP
#undef RFABS

#define REAL double
#define SUF _f64
#define REXP exp
#define RFABS fabs
#include "ddcn_ref_impl.inc"
#include "psroi_ref_impl.inc"
#undef REAL
#undef SUF
#undef REXP
#undef RFABS

int ddcn_ref_version(void) { return 1; }

#ifdef _OPENMP
#include <omp.h>
int ddcn_ref_max_threads(void) { return omp_get_max_threads(); }
void ddcn_ref_set_threads(int n) { omp_set_num_threads(n); }
#else
int ddcn_ref_max_threads(void) { return 1; }
void ddcn_ref_set_threads(int n) { (void)n; }
#endif

// fp32 MFMA contractions of the DDCN operator (gfx950, wave64).
//
// Replaces the three at::addmm_ (cuBLAS) call sites of the reference host code,
// detseg/ops/dcn/src/deform_conv_cuda.cpp: G1 :232-237/:430-435, G2 :305-308/:503-506,
// G3 :341-346/:539-544.  All images / weight groups of a call are one launch (grid.z).
//
// Kernel: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
// 2x2 v_mfma_f32_32x32x2_f32 accumulators), BK = 32, register-staged global->LDS double
// buffering (one barrier per K tile).  Both operands live in LDS as [k][mn] so that the MFMA
// fragment (lane l: row/col l&31, k = l>>5) is one conflict-free ds_read_b32 per operand.
// fp32-input MFMA is an exact fmaf chain (no TF32), so results differ from the reference's
// cuBLAS only by summation order.
#include "ddcn_common.h"

namespace ddcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int LD_MN = 128;   // LDS row stride for operands stored [k][mn] in memory (b128 writes)
constexpr int LD_KC = 129;   // LDS row stride for operands stored [mn][k] in memory (transposing b32 writes)

struct GemmArgs {
    const float *A, *B, *bias;
    float *C;
    int M, N, K;              // K: full reduction length
    long lda, ldb, ldc;
    long sA0, sA1, sB0, sB1, sC0, sC1;   // batch strides: z0 (image), z1 (weight group)
    int nz1, ksplit, kchunk;  // grid.z = nz0 * nz1 * ksplit ; kchunk multiple of BK
    int vecA, vecB;           // 16-byte vector loads allowed (base and ld aligned)
};

// ---- operand stored [K][MN] in memory (mn contiguous): tile [BK][128] ----
template <bool VEC>
__device__ __forceinline__ void load_mn(const float *__restrict__ p, long ld, int k0, int kend, int mn0, int mnlim,
                                        float (&r)[16], int t)
{
    if (VEC) {
        const int c4 = (t & 31) * 4, row = t >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + row + 8 * i;
            const int mn = mn0 + c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < kend) {
                const float *q = p + (long)k * ld + mn;
                if (mn + 3 < mnlim) {
                    v = *reinterpret_cast<const f32x4 *>(q);
                } else {
                    if (mn < mnlim) v.x = q[0];
                    if (mn + 1 < mnlim) v.y = q[1];
                    if (mn + 2 < mnlim) v.z = q[2];
                }
            }
            r[4 * i + 0] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
    } else {
        const int c = t & 127, row = t >> 7;
        const int mn = mn0 + c;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = k0 + row + 2 * i;
            r[i] = (k < kend && mn < mnlim) ? p[(long)k * ld + mn] : 0.f;
        }
    }
}

template <bool VEC>
__device__ __forceinline__ void store_mn(float *__restrict__ s, const float (&r)[16], int t)
{
    if (VEC) {
        const int c4 = (t & 31) * 4, row = t >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = {r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
            *reinterpret_cast<f32x4 *>(s + (row + 8 * i) * LD_MN + c4) = v;
        }
    } else {
        const int c = t & 127, row = t >> 7;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[(row + 2 * i) * LD_MN + c] = r[i];
    }
}

// ---- operand stored [MN][K] in memory (k contiguous): tile [128][BK], transposed into LDS ----
template <bool VEC>
__device__ __forceinline__ void load_kc(const float *__restrict__ p, long ld, int k0, int kend, int mn0, int mnlim,
                                        float (&r)[16], int t)
{
    if (VEC) {
        const int k4 = (t & 7) * 4, row = t >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mn = mn0 + row + 32 * i;
            const int k = k0 + k4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (mn < mnlim) {
                const float *q = p + (long)mn * ld + k;
                if (k + 3 < kend) {
                    v = *reinterpret_cast<const f32x4 *>(q);
                } else {
                    if (k < kend) v.x = q[0];
                    if (k + 1 < kend) v.y = q[1];
                    if (k + 2 < kend) v.z = q[2];
                }
            }
            r[4 * i + 0] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
    } else {
        const int kk = t & 31, row = t >> 5;
        const int k = k0 + kk;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int mn = mn0 + row + 8 * i;
            r[i] = (k < kend && mn < mnlim) ? p[(long)mn * ld + k] : 0.f;
        }
    }
}

template <bool VEC>
__device__ __forceinline__ void store_kc(float *__restrict__ s, const float (&r)[16], int t)
{
    if (VEC) {
        const int k4 = (t & 7) * 4, row = t >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[(k4 + j) * LD_KC + row + 32 * i] = r[4 * i + j];
    } else {
        const int kk = t & 31, row = t >> 5;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kk * LD_KC + row + 8 * i] = r[i];
    }
}

template <bool A_KC, bool B_KC, bool ATOMIC>
__global__ __launch_bounds__(NT) void gemm_f32_mfma(GemmArgs a)
{
    constexpr int LDA = A_KC ? LD_KC : LD_MN;
    constexpr int LDB = B_KC ? LD_KC : LD_MN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    auto As = [&](int buf) { return smem + buf * (BK * LDA); };
    auto Bs = [&](int buf) { return smem + 2 * BK * LDA + buf * (BK * LDB); };

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int z = blockIdx.z;
    const int split = z % a.ksplit;
    z /= a.ksplit;
    const int z1 = z % a.nz1, z0 = z / a.nz1;
    const float *A = a.A + z0 * a.sA0 + z1 * a.sA1;
    const float *B = a.B + z0 * a.sB0 + z1 * a.sB1;
    float *C = a.C + z0 * a.sC0 + z1 * a.sC1;

    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = split * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[16], rb[16];

    auto gload = [&](int k0) {
        if (A_KC) {
            if (a.vecA) load_kc<true>(A, a.lda, k0, kend, m0, a.M, ra, t);
            else load_kc<false>(A, a.lda, k0, kend, m0, a.M, ra, t);
        } else {
            if (a.vecA) load_mn<true>(A, a.lda, k0, kend, m0, a.M, ra, t);
            else load_mn<false>(A, a.lda, k0, kend, m0, a.M, ra, t);
        }
        if (B_KC) {
            if (a.vecB) load_kc<true>(B, a.ldb, k0, kend, n0, a.N, rb, t);
            else load_kc<false>(B, a.ldb, k0, kend, n0, a.N, rb, t);
        } else {
            if (a.vecB) load_mn<true>(B, a.ldb, k0, kend, n0, a.N, rb, t);
            else load_mn<false>(B, a.ldb, k0, kend, n0, a.N, rb, t);
        }
    };
    auto sstore = [&](int buf) {
        if (A_KC) {
            if (a.vecA) store_kc<true>(As(buf), ra, t); else store_kc<false>(As(buf), ra, t);
        } else {
            if (a.vecA) store_mn<true>(As(buf), ra, t); else store_mn<false>(As(buf), ra, t);
        }
        if (B_KC) {
            if (a.vecB) store_kc<true>(Bs(buf), rb, t); else store_kc<false>(Bs(buf), rb, t);
        } else {
            if (a.vecB) store_mn<true>(Bs(buf), rb, t); else store_mn<false>(Bs(buf), rb, t);
        }
    };

    if (ntiles > 0) {
        gload(kbeg);
        sstore(0);
    }
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    for (int tile = 0; tile < ntiles; ++tile) {
        const int cur = tile & 1;
        if (tile + 1 < ntiles) gload(kbeg + (tile + 1) * BK);
        const float *as = As(cur) + wm * 64 + li;
        const float *bs = Bs(cur) + wn * 64 + li;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + lh;
            const float a0 = as[k * LDA], a1 = as[k * LDA + 32];
            const float b0 = bs[k * LDB], b1 = bs[k * LDB + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (tile + 1 < ntiles) sstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < a.M && n < a.N) {
                    float v = acc[i][j][r];
                    float *dst = C + (long)m * a.ldc + n;
                    if (ATOMIC) {
                        atomicAdd(dst, v);
                    } else {
                        if (a.bias) v += a.bias[z1 * a.M + m];
                        *dst = v;
                    }
                }
            }
        }
}

template <bool A_KC, bool B_KC, bool ATOMIC>
static int launch(const GemmArgs &a, int nz0, hipStream_t stream, const char *what)
{
    constexpr int LDA = A_KC ? LD_KC : LD_MN;
    constexpr int LDB = B_KC ? LD_KC : LD_MN;
    const size_t smem = (size_t)(2 * BK * LDA + 2 * BK * LDB) * sizeof(float);
    // > 64 KiB of dynamic LDS needs the opt-in; set per call (cheap, and correct for every device of the process)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_f32_mfma<A_KC, B_KC, ATOMIC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return fail(DDCN_ERR_LAUNCH, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nz0 * a.nz1 * a.ksplit);
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return DDCN_OK;
    if (grid.y > 65535 || grid.z > 65535) return fail(DDCN_ERR_UNSUPPORTED, "%s: grid too large", what);
    hipLaunchKernelGGL((gemm_f32_mfma<A_KC, B_KC, ATOMIC>), grid, dim3(NT), smem, stream, a);
    return check_launch(what);
}

static inline int aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// G1: out[n][g] (Cog x P) = W[g] (Cog x Kg, k contiguous) * col[n][g] (Kg x ld) (+ bias)
int launch_gemm_out(const float *weight, const float *col, const float *bias, float *out, const ddcn_geom &g,
                    hipStream_t stream)
{
    const Dims d = dims_of(g);
    GemmArgs a{};
    a.A = weight; a.B = col; a.C = out; a.bias = bias;
    a.M = d.Cog; a.N = d.P; a.K = d.Kg;
    a.lda = d.Kg; a.ldb = d.ld; a.ldc = d.P;
    a.sA0 = 0; a.sA1 = (long)d.Cog * d.Kg;
    a.sB0 = (long)d.Ktot * d.ld; a.sB1 = (long)d.Kg * d.ld;
    a.sC0 = (long)g.Co * d.P; a.sC1 = (long)d.Cog * d.P;
    a.nz1 = g.groups; a.ksplit = 1; a.kchunk = (int)round_up(d.Kg, BK);
    a.vecA = aligned16(weight) && (d.Kg % 4 == 0);
    a.vecB = aligned16(col) && (d.ld % 4 == 0);
    return launch<true, false, false>(a, g.N, stream, "gemm_out");
}

// G2: gcol[n][g] (Kg x ld) = W[g]^T (A stored [K=Cog][M=Kg]) * gout[n][g] (Cog x P)
int launch_gemm_gcol(const float *weight, const float *gout, float *gcol, const ddcn_geom &g, hipStream_t stream)
{
    const Dims d = dims_of(g);
    GemmArgs a{};
    a.A = weight; a.B = gout; a.C = gcol; a.bias = nullptr;
    a.M = d.Kg; a.N = d.P; a.K = d.Cog;
    a.lda = d.Kg; a.ldb = d.P; a.ldc = d.ld;
    a.sA0 = 0; a.sA1 = (long)d.Cog * d.Kg;
    a.sB0 = (long)g.Co * d.P; a.sB1 = (long)d.Cog * d.P;
    a.sC0 = (long)d.Ktot * d.ld; a.sC1 = (long)d.Kg * d.ld;
    a.nz1 = g.groups; a.ksplit = 1; a.kchunk = (int)round_up(d.Cog, BK);
    a.vecA = aligned16(weight) && (d.Kg % 4 == 0);
    a.vecB = aligned16(gout) && (d.P % 4 == 0);
    return launch<false, false, false>(a, g.N, stream, "gemm_gcol");
}

// G3: gw[g] (Cog x Kg) += sum_n gout[n][g] (Cog x P, k = p contiguous) * col[n][g]^T (B stored [N=Kg][K=ld])
// split-K over images and pixel ranges, fp32 atomics into gw.
int launch_gemm_gweight(const float *gout, const float *col, float *gw, const ddcn_geom &g, hipStream_t stream)
{
    const Dims d = dims_of(g);
    GemmArgs a{};
    a.A = gout; a.B = col; a.C = gw; a.bias = nullptr;
    a.M = d.Cog; a.N = d.Kg; a.K = d.P;
    a.lda = d.P; a.ldb = d.ld; a.ldc = d.Kg;
    a.sA0 = (long)g.Co * d.P; a.sA1 = (long)d.Cog * d.P;
    a.sB0 = (long)d.Ktot * d.ld; a.sB1 = (long)d.Kg * d.ld;
    a.sC0 = 0; a.sC1 = (long)d.Cog * d.Kg;
    a.nz1 = g.groups;
    const long tiles_mn = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * g.groups * g.N;
    int ksplit = (int)((1024 + tiles_mn - 1) / tiles_mn);
    const int ktiles = (d.P + BK - 1) / BK;
    if (ksplit > ktiles) ksplit = ktiles;
    if (ksplit < 1) ksplit = 1;
    const int tiles_per = (ktiles + ksplit - 1) / ksplit;
    a.kchunk = tiles_per * BK;
    a.ksplit = (ktiles + tiles_per - 1) / tiles_per;
    a.vecA = aligned16(gout) && (d.P % 4 == 0);
    a.vecB = aligned16(col) && (d.ld % 4 == 0);
    return launch<true, true, true>(a, g.N, stream, "gemm_gweight");
}

}  // namespace ddcn

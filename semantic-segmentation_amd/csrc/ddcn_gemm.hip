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
//
// The K loop is branch-free: every global access is a buffer load/store whose bounds (tensor end,
// M/N/K edges) are enforced by the buffer descriptor or by steering the per-lane offset out of
// range, never by control flow -- the first version of this kernel guarded each load with a
// branch, which split the loop into load / MFMA / store phases that the two waves of a SIMD
// executed in lockstep (rocprofv3: matrix pipe 49-60 % busy, 5 VALU per MFMA).
#include "ddcn_common.h"

namespace ddcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int LD_MN = 128;   // LDS row stride for operands stored [k][mn] in memory (b128 writes)
constexpr int LD_KC = 129;   // LDS row stride for operands stored [mn][k] in memory (transposing b32 writes)
constexpr unsigned OOB = 0x80000000u;   // per-lane offset that every descriptor below rejects (reads 0, drops stores)

struct GemmArgs {
    const float *A, *B, *bias;
    float *C;
    int M, N, K;              // K: full reduction length
    long lda, ldb, ldc;
    long sA0, sA1, sB0, sB1, sC0, sC1;   // batch strides: z0 (image), z1 (weight group)
    long nA, nB, nC;          // elements of one (z0, z1) slice of A / B / C (descriptor bounds)
    int nz1, ksplit, kchunk;  // grid.z = nz0 * nz1 * ksplit ; kchunk multiple of BK
    int tiles_m, tiles_n;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base, long elems_left)
{
    const long bytes = elems_left * 4;
    const unsigned n = bytes <= 0 ? 0u : (bytes > 0x7fffffffL ? 0x7fffffffu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, n, 0x00020000);
}

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

// ---- operand stored [K][MN] in memory (mn contiguous): tile [BK][128]; thread: float4 column c4, rows row+8i
struct LoaderMN {
    unsigned voff[4];
    int row, c4;
    __device__ __forceinline__ void init(long ld, int mn0, int t)
    {
        c4 = (t & 31) * 4;
        row = t >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) voff[i] = (unsigned)(((long)(row + 8 * i) * ld + mn0 + c4) * 4);
    }
    // base: slice pointer advanced to row k0; kvalid = valid rows in this tile
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int kvalid, f32x4 (&v)[4]) const
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = bload4(r, (row + 8 * i < kvalid) ? voff[i] : OOB);
    }
    __device__ __forceinline__ void store(float *s, const f32x4 (&v)[4], int) const
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(s + (row + 8 * i) * LD_MN + c4) = v[i];
    }
};

// ---- operand stored [MN][K] in memory (k contiguous): tile [128][BK]; thread: float4 k4, rows row+32i; transposed into LDS
struct LoaderKC {
    unsigned voff[4];
    int k4, row;
    __device__ __forceinline__ void init(long ld, int mn0, int mnlim, int t)
    {
        k4 = (t & 7) * 4;
        row = t >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            voff[i] = (mn0 + row + 32 * i < mnlim) ? (unsigned)(((long)(row + 32 * i) * ld + k4) * 4) : OOB;
    }
    // base: slice pointer advanced to (row mn0, column k0).  Issue only: the K-edge masking (elements with
    // k4 + j >= kvalid must be zero) is applied in store(), so nothing waits on the loads before the MFMAs.
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int, f32x4 (&v)[4]) const
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = bload4(r, voff[i]);
    }
    __device__ __forceinline__ void store(float *s, const f32x4 (&v)[4], int kvalid) const
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *d = s + k4 * LD_KC + row + 32 * i;
            d[0] = (k4 + 0 < kvalid) ? v[i].x : 0.f;
            d[LD_KC] = (k4 + 1 < kvalid) ? v[i].y : 0.f;
            d[2 * LD_KC] = (k4 + 2 < kvalid) ? v[i].z : 0.f;
            d[3 * LD_KC] = (k4 + 3 < kvalid) ? v[i].w : 0.f;
        }
    }
};

template <bool A_KC, bool B_KC, bool ATOMIC>
__global__ __launch_bounds__(NT) void gemm_f32_mfma(GemmArgs a)
{
    constexpr int LDA = A_KC ? LD_KC : LD_MN;
    constexpr int LDB = B_KC ? LD_KC : LD_MN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As0 = smem, *Bs0 = smem + 2 * BK * LDA;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
    // run of the logical order (bijective for any count), with the m tile fastest: the workgroups that
    // share a B (columns) tile then run back to back on one XCD and hit its L2.
    const int nwg = a.tiles_m * a.tiles_n;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_n = logical / a.tiles_m, tile_m = logical - tile_n * a.tiles_m;

    int z = blockIdx.z;
    const int split = z % a.ksplit;
    z /= a.ksplit;
    const int z1 = z % a.nz1, z0 = z / a.nz1;
    const float *A = a.A + z0 * a.sA0 + z1 * a.sA1;
    const float *B = a.B + z0 * a.sB0 + z1 * a.sB1;
    float *C = a.C + z0 * a.sC0 + z1 * a.sC1;

    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    LoaderKC lak; LoaderMN lam; LoaderKC lbk; LoaderMN lbm;
    if (A_KC) lak.init(a.lda, m0, a.M, t); else lam.init(a.lda, m0, t);
    if (B_KC) lbk.init(a.ldb, n0, a.N, t); else lbm.init(a.ldb, n0, t);
    f32x4 ra[4], rb[4];

    auto gload = [&](int k0) {
        const int kvalid = kend - k0;   // <= 0 past the end: every lane reads zeros
        if (A_KC) {
            const long o = (long)m0 * a.lda + k0;
            lak.load(make_rsrc(A + o, a.nA - o), kvalid, ra);
        } else {
            const long o = (long)k0 * a.lda;
            lam.load(make_rsrc(A + o, a.nA - o), kvalid, ra);
        }
        if (B_KC) {
            const long o = (long)n0 * a.ldb + k0;
            lbk.load(make_rsrc(B + o, a.nB - o), kvalid, rb);
        } else {
            const long o = (long)k0 * a.ldb;
            lbm.load(make_rsrc(B + o, a.nB - o), kvalid, rb);
        }
    };
    auto sstore = [&](int buf, int k0) {
        const int kvalid = kend - k0;
        if (A_KC) lak.store(As0 + buf * (BK * LDA), ra, kvalid); else lam.store(As0 + buf * (BK * LDA), ra, kvalid);
        if (B_KC) lbk.store(Bs0 + buf * (BK * LDB), rb, kvalid); else lbm.store(Bs0 + buf * (BK * LDB), rb, kvalid);
    };

    gload(kbeg);
    sstore(0, kbeg);
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    for (int tile = 0; tile < ntiles; ++tile) {
        const int cur = tile & 1;
        gload(kbeg + (tile + 1) * BK);   // past the last tile this reads zeros (kvalid <= 0), no branch needed
        // keep the loads in flight UNDER the MFMAs: without this fence hipcc sinks them below the MFMA block and
        // waits for them at once, exposing the whole HBM latency every K tile
        __builtin_amdgcn_sched_barrier(0);
        const float *as = As0 + cur * (BK * LDA) + wm * 64 + li;
        const float *bs = Bs0 + cur * (BK * LDB) + wn * 64 + li;
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
        __builtin_amdgcn_sched_barrier(0);
        sstore(cur ^ 1, kbeg + (tile + 1) * BK);
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (ATOMIC) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (m < a.M && n < a.N) atomicAdd(C + (long)m * a.ldc + n, acc[i][j][e]);
                }
            }
    } else {
        const long o = (long)m0 * a.ldc + n0;
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(C + o, a.nC - o);
        float bv[2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) bv[i][e] = 0.f;
        if (a.bias) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ml = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    bv[i][e] = a.bias[z1 * a.M + min(m0 + ml, a.M - 1)];
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nl = wn * 64 + j * 32 + li;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ml = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float v = acc[i][j][e] + bv[i][e];
                    const bool ok = (m0 + ml < a.M) && (n0 + nl < a.N);
                    const unsigned voff = ok ? (unsigned)(((long)ml * a.ldc + nl) * 4) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff, 0, 0);
                }
            }
    }
}

template <bool A_KC, bool B_KC, bool ATOMIC>
static int launch(GemmArgs &a, int nz0, hipStream_t stream, const char *what)
{
    constexpr int LDA = A_KC ? LD_KC : LD_MN;
    constexpr int LDB = B_KC ? LD_KC : LD_MN;
    const size_t smem = (size_t)(2 * BK * LDA + 2 * BK * LDB) * sizeof(float);
    // per-lane byte offsets are 32 bit and span one tile of rows: 128 rows * ld * 4 bytes must stay below 2^30
    if (a.lda >= (1L << 21) || a.ldb >= (1L << 21) || a.ldc >= (1L << 21))
        return fail(DDCN_ERR_UNSUPPORTED, "%s: leading dimension beyond 2^21 elements", what);
    // > 64 KiB of dynamic LDS needs the opt-in; set per call (cheap, and correct for every device of the process)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_f32_mfma<A_KC, B_KC, ATOMIC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return fail(DDCN_ERR_LAUNCH, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    const long tiles = (long)a.tiles_m * a.tiles_n;
    dim3 grid((unsigned)tiles, 1, nz0 * a.nz1 * a.ksplit);
    if (tiles == 0 || grid.z == 0) return DDCN_OK;
    if (tiles > 0x7fffffffL || grid.z > 65535) return fail(DDCN_ERR_UNSUPPORTED, "%s: grid too large", what);
    hipLaunchKernelGGL((gemm_f32_mfma<A_KC, B_KC, ATOMIC>), grid, dim3(NT), smem, stream, a);
    return check_launch(what);
}

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
    a.nA = a.sA1; a.nB = a.sB1; a.nC = a.sC1;
    a.nz1 = g.groups; a.ksplit = 1; a.kchunk = (int)round_up(d.Kg, BK);
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
    a.nA = a.sA1; a.nB = a.sB1; a.nC = a.sC1;
    a.nz1 = g.groups; a.ksplit = 1; a.kchunk = (int)round_up(d.Cog, BK);
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
    a.nA = a.sA1; a.nB = a.sB1; a.nC = a.sC1;
    a.nz1 = g.groups;
    const long tiles_mn = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * g.groups * g.N;
    int ksplit = (int)((1024 + tiles_mn - 1) / tiles_mn);
    const int ktiles = (d.P + BK - 1) / BK;
    if (ksplit > ktiles) ksplit = ktiles;
    if (ksplit < 1) ksplit = 1;
    const int tiles_per = (ktiles + ksplit - 1) / ksplit;
    a.kchunk = tiles_per * BK;
    a.ksplit = (ktiles + tiles_per - 1) / tiles_per;
    return launch<true, true, true>(a, g.N, stream, "gemm_gweight");
}

}  // namespace ddcn

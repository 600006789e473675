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
#include <stdlib.h>

#include "ddcn_common.h"

namespace ddcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float;

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int BN_TAIL = 32;  // narrow column tiles that fill the last, partial round of workgroups
constexpr unsigned OOB = 0x80000000u;   // per-lane offset that every descriptor below rejects (reads 0, drops stores)

struct GemmArgs {
    const float *A, *B, *bias;
    float *C;
    int M, N, K;              // K: full reduction length
    long lda, ldb, ldc;
    long sA0, sA1, sB0, sB1, sC0, sC1;   // batch strides: z0 (image), z1 (weight group)
    long nA, nB, nC;          // elements of one (z0, z1) slice of A / B / C (descriptor bounds)
    int nz1, ksplit, kchunk;  // nz = nz0 * nz1 * ksplit slices ; kchunk multiple of BK
    int tiles_m;
    int stream_out;           // 1: C is written once and read much later (column gradients): non-temporal stores
    int nb, nn;               // column tiles per slice: nb of width BN, then nn of width BN_TAIL
    int total_big, total_nar;    // tiles_m * nb * nz wide tiles, tiles_m * nn * nz narrow tiles
    int blocks_big, blocks_nar;  // workgroups per class: two tiles (one per wave group) each
    // persistent schedule (gemm_f32_sk): G wave groups, T output tiles of S K-steps each (kimgs images folded into K)
    int sk_G, sk_T, sk_S, sk_ktiles, sk_kimgs, sk_R, sk_tiles_n, sk_utail;
};

// Descriptor over [base, base + elems_left) floats, clamped to [0, 2 GiB).  32-bit scalar compares only: a 64-bit
// ordered compare has no SALU form and would put VALU work into the memory phase of the K loop (see gemm_tile).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base, long elems_left)
{
    int hi = (int)(elems_left >> 32);
    unsigned lo = (unsigned)elems_left;
    asm("" : "+s"(hi), "+s"(lo));   // or instcombine folds the tests back into 64-bit compares (v_cmp_*_i64)
    const unsigned n = hi < 0 ? 0u : ((hi > 0 || lo >= (1u << 29)) ? 0x7fffffffu : lo * 4u);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, n, 0x00020000);
}

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

// ---- operand stored [K][MN] in memory (mn contiguous): tile [BK][EXT]; thread: float4 column c4, rows row + RSTEP*i.
// LDS image [k][mn], row stride EXT (b128 writes).
template <int EXT>
struct LoaderMN {
    static constexpr int NV = EXT / 32, TPR = EXT / 4, RSTEP = NT / TPR, LD = EXT;
    unsigned voff[NV];
    float *dst;
    // per-lane offsets do not depend on the tile: the tile's first column is part of the scalar origin
    __device__ __forceinline__ void init(float *lds, long ld, int t)
    {
        const int c4 = (t % TPR) * 4, row = t / TPR;
        dst = lds + row * LD + c4;
#pragma unroll
        for (int i = 0; i < NV; ++i) voff[i] = (unsigned)(((long)(row + RSTEP * i) * ld + c4) * 4);
    }
    __device__ __forceinline__ static long origin(long ld, int mn0, int k0) { return (long)k0 * ld + mn0; }
    // elements from the tile origin to the end of the valid rows [k0, kend): the K edge is the descriptor's bound
    // (columns of the last row that would wrap past the row end read zeros; they belong to masked outputs)
    __device__ __forceinline__ static long extent(long ld, long slice_left, int kvalid, int mn0)
    {
        const long rows = (long)kvalid * ld - mn0;
        int sign = (int)((rows - slice_left) >> 32);   // min() through a 32-bit sign test (SALU)
        asm("" : "+s"(sign));
        return sign < 0 ? rows : slice_left;
    }
    // descriptor base: slice pointer advanced to (row k0, column mn0)
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, f32x4 (&v)[NV]) const
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = bload4(r, voff[i]);
    }
    __device__ __forceinline__ void store(const f32x4 (&v)[NV], int) const
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4 *>(dst + RSTEP * i * LD) = v[i];
    }
};

// ---- operand stored [MN][K] in memory (k contiguous): tile [EXT][BK]; thread: float4 k4, rows row + 32i.
// Transposed into the same [k][mn] LDS image, row stride EXT + 1 (conflict-free b32 writes).
template <int EXT>
struct LoaderKC {
    static constexpr int NV = EXT / 32, LD = EXT + 1;
    unsigned voff[NV];
    int k4;
    lds_float *d0, *d2;   // LDS rows k4 and k4 + 2 of this thread's column: every ds_write2_b32 offset (<= 1020 B) is
                      // reachable from one of them, so the K loop needs no address arithmetic (no VALU)
    // per-lane offsets do not depend on the tile; rows past the M / N edge lie past the end of the slice (the slice IS
    // [MN][K]) and are rejected by the descriptor
    __device__ __forceinline__ void init(float *lds, long ld, int t)
    {
        k4 = (t & 7) * 4;
        const int row = t >> 3;
        d0 = (lds_float *)lds + k4 * LD + row;
        d2 = d0 + 2 * LD;
        asm volatile("" : "+v"(d2));   // a real register, not "d0 + const" rematerialised inside the loop
#pragma unroll
        for (int i = 0; i < NV; ++i) voff[i] = (unsigned)(((long)(row + 32 * i) * ld + k4) * 4);
    }
    __device__ __forceinline__ static long origin(long ld, int mn0, int k0) { return (long)mn0 * ld + k0; }
    __device__ __forceinline__ static long extent(long, long slice_left, int, int) { return slice_left; }
    // descriptor base: slice pointer advanced to (row mn0, column k0).  The K edge cuts through rows here, so it is
    // applied in store(): elements with k4 + j >= kvalid are written as zero.
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, f32x4 (&v)[NV]) const
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = bload4(r, voff[i]);
    }
    __device__ __forceinline__ void store(const f32x4 (&v)[NV], int kvalid) const
    {
        if (kvalid >= BK) {   // uniform; the full tiles carry no VALU work at all
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                d0[32 * i] = v[i].x; d0[LD + 32 * i] = v[i].y; d2[32 * i] = v[i].z; d2[LD + 32 * i] = v[i].w;
            }
            asm volatile("; full tile" ::: "memory");   // keeps hipcc from merging the two paths into one with selects
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                d0[32 * i] = (k4 + 0 < kvalid) ? v[i].x : 0.f;
                d0[LD + 32 * i] = (k4 + 1 < kvalid) ? v[i].y : 0.f;
                d2[32 * i] = (k4 + 2 < kvalid) ? v[i].z : 0.f;
                d2[LD + 32 * i] = (k4 + 3 < kvalid) ? v[i].w : 0.f;
            }
            asm volatile("; edge tile" ::: "memory");
        }
    }
};

template <bool KC, int EXT> struct LoaderSel { typedef LoaderMN<EXT> type; };
template <int EXT> struct LoaderSel<true, EXT> { typedef LoaderKC<EXT> type; };

// XCD-aware order inside a class of `count` workgroups: they are dealt round-robin over the 8 XCDs, so give each
// XCD a contiguous run of the logical order (bijective for any count).
__device__ __forceinline__ int xcd_order(int b, int count)
{
    const int q = count >> 3, r = count & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// One BM x BNT output tile of slice z, computed by one wave group (4 waves as WAVES_M x WAVES_N) of a
// two-group workgroup.  Every s_barrier below is executed by BOTH groups the same number of times.
template <bool A_KC, bool B_KC, bool ATOMIC, int BNT>
__device__ __forceinline__ void gemm_tile(const GemmArgs &a, float *smem, int grp, bool valid, int z, int tile_m, int n0)
{
    typedef typename LoaderSel<A_KC, BM>::type LA;
    typedef typename LoaderSel<B_KC, BNT>::type LB;
    constexpr int LDA = LA::LD, LDB = LB::LD;
    constexpr int WN_ = BNT >= 128 ? 64 : 32;          // columns per wave
    constexpr int WAVES_N = BNT / WN_, WAVES_M = 4 / WAVES_N;
    constexpr int WM_ = BM / WAVES_M;                  // rows per wave
    constexpr int MI = WM_ / 32, NJ = WN_ / 32;
    float *As = smem + grp * (BK * LDA + BK * LDB), *Bs = As + BK * LDA;   // one private buffer per group

    const int t = threadIdx.x & (NT - 1);
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int split = z % a.ksplit;
    z /= a.ksplit;
    const int z1 = z % a.nz1, z0 = z / a.nz1;
    const float *A = a.A + z0 * a.sA0 + z1 * a.sA1;
    const float *B = a.B + z0 * a.sB0 + z1 * a.sB1;
    float *C = a.C + z0 * a.sC0 + z1 * a.sC1;

    const int m0 = tile_m * BM;
    const int kbeg = split * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int ntiles = a.kchunk / BK;   // the same for every group of the launch (short last splits read zeros)
    // an idle group (odd tile count) runs the same loop on empty descriptors: loads return 0, stores are dropped
    const long nA = valid ? a.nA : 0, nB = valid ? a.nB : 0, nC = valid ? a.nC : 0;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    LA la; LB lb;
    la.init(As, a.lda, t);
    lb.init(Bs, a.ldb, t);
    f32x4 ra[LA::NV], rb[LB::NV];

    auto gload = [&](int k0) {
        const int kvalid = max(kend - k0, 0);   // 0 past the end: [K][MN] operands read zeros, [MN][K] ones are masked in store()
        const long oa = LA::origin(a.lda, m0, k0), ob = LB::origin(a.ldb, n0, k0);
        la.load(make_rsrc(A + oa, LA::extent(a.lda, nA - oa, kvalid, m0)), ra);
        lb.load(make_rsrc(B + ob, LB::extent(a.ldb, nB - ob, kvalid, n0)), rb);
    };
    auto sstore = [&](int k0) {
        const int kvalid = kend - k0;
        la.store(ra, kvalid);
        lb.store(rb, kvalid);
    };

    gload(kbeg);
    sstore(kbeg);
    gload(kbeg + BK);
    __syncthreads();
    // Ping-pong: the two groups alternate between a compute phase (ds_read + MFMA from the private buffer) and a
    // short memory phase (the prefetched registers -> LDS, next loads issued), one barrier apart, so that each SIMD
    // always has one wave in its MFMA stream.  Two independent 256-thread workgroups per CU ran in lockstep
    // instead: both waves of a SIMD reached their loads / LDS writes / barrier together and the matrix pipe idled.
    // The memory phase must be free of VALU instructions: measured with s_memtime, a wave's VALU work (v_cndmask
    // edge masks, 64-bit compares) does not issue while the other wave of the SIMD streams MFMAs, which stretched
    // the memory phase from ~400 to ~4500 cycles.  Hence descriptor-enforced bounds and scalar-only address math.
    if (grp == 1) __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    for (int tile = 0; tile < ntiles; ++tile) {
        __builtin_amdgcn_s_setprio(2);   // the computing wave wins every issue arbitration against the other group's memory phase
        const float *as = As + wm * WM_ + li;
        const float *bs = Bs + wn * WN_ + li;
        // fragments PF k-steps ahead of the MFMAs that consume them: the LDS latency, and the other group's burst of
        // ds_writes in its memory phase, stay hidden behind 4*PF MFMAs (left alone hipcc reuses one register set and
        // waits for every ds_read right before its MFMA group)
        constexpr int PF = 2, KS = BK / 2;
        float av[PF + 1][MI], bv[PF + 1][NJ];
#pragma unroll
        for (int s0 = 0; s0 < PF; ++s0) {
#pragma unroll
            for (int i = 0; i < MI; ++i) av[s0][i] = as[(2 * s0 + lh) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[s0][j] = bs[(2 * s0 + lh) * LDB + 32 * j];
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int c = kk % (PF + 1), nx = (kk + PF) % (PF + 1);
            if (kk + PF < KS) {
                const int k = 2 * (kk + PF) + lh;
#pragma unroll
                for (int i = 0; i < MI; ++i) av[nx][i] = as[k * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[nx][j] = bs[k * LDB + 32 * j];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i], bv[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MI + NJ, 0);   // the prefetch step's ds_reads first
            __builtin_amdgcn_sched_group_barrier(0x008, MI * NJ, 0);   // then this step's MFMAs
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();   // this group's tile is consumed; the other group starts computing
        sstore(kbeg + (tile + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);   // all LDS writes before the new loads: their vmcnt waits stay exact
        // the tile after that goes in flight now and lands during the two compute phases that follow (past the
        // last tile it reads zeros: kvalid <= 0, no branch needed)
        gload(kbeg + (tile + 2) * BK);
        __syncthreads();   // the next tile is in place; the other group starts its memory phase
    }
    if (grp == 0) __syncthreads();

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (ATOMIC) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + wn * WN_ + j * 32 + li;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * WM_ + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (valid && m < a.M && n < a.N) atomicAdd(C + (long)m * a.ldc + n, acc[i][j][e]);
                }
            }
    } else {
        const long o = (long)m0 * a.ldc + n0;
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(C + o, nC - o);
        float bias[MI][16];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) bias[i][e] = 0.f;
        if (a.bias) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ml = wm * WM_ + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    bias[i][e] = a.bias[z1 * a.M + min(m0 + ml, a.M - 1)];
                }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nl = wn * WN_ + j * 32 + li;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ml = wm * WM_ + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float v = acc[i][j][e] + bias[i][e];
                    const bool ok = (m0 + ml < a.M) && (n0 + nl < a.N);
                    const unsigned voff = ok ? (unsigned)(((long)ml * a.ldc + nl) * 4) : OOB;
                    if (a.stream_out) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff, 0, 2);
                    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff, 0, 0);
                }
            }
    }
}

// Flat grid: the wide tiles of all slices first, then the narrow tiles, so that the last (partial) round of
// workgroups is made of quarter-size tiles spread over every CU instead of a few full tiles on a few CUs.
// A workgroup is two wave groups with one tile each (adjacent in the logical order).  Inside a class the m tile is
// fastest within bands of MBAND m tiles: the tiles that share a B tile run back to back on one XCD's L2, and the A
// slices of a band (8 x 128 rows x K) stay in that L2 while the band walks along n.  With all of a tall M in one band
// (G2: 36 m tiles = 9.4 MB of W against 4 MB of L2) every n tile re-read W from the Infinity Cache.
constexpr int MBAND = 8;
__device__ __forceinline__ void band_order(int rem, int tiles_m, int tiles_n, int &tile_m, int &tile_n)
{
    const int per_band = MBAND * tiles_n;
    const int band = rem / per_band, r2 = rem - band * per_band;
    const int mcur = min(MBAND, tiles_m - band * MBAND);
    tile_n = r2 / mcur;
    tile_m = band * MBAND + (r2 - tile_n * mcur);
}
template <bool A_KC, bool B_KC, bool ATOMIC>
__global__ __launch_bounds__(2 * NT) void gemm_f32_mfma(GemmArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    const int bid = blockIdx.x;
    if (bid < a.blocks_big) {
        int idx = 2 * xcd_order(bid, a.blocks_big) + grp;
        const bool valid = idx < a.total_big;
        idx = min(idx, a.total_big - 1);
        const int per_z = a.tiles_m * a.nb;
        const int z = idx / per_z, rem = idx - z * per_z;
        int tile_m, tile_n;
        band_order(rem, a.tiles_m, a.nb, tile_m, tile_n);
        gemm_tile<A_KC, B_KC, ATOMIC, BN>(a, smem, grp, valid, z, tile_m, tile_n * BN);
    } else {
        int idx = 2 * xcd_order(bid - a.blocks_big, a.blocks_nar) + grp;
        const bool valid = idx < a.total_nar;
        idx = min(idx, a.total_nar - 1);
        const int per_z = a.tiles_m * a.nn;
        const int z = idx / per_z, rem = idx - z * per_z;
        int tile_m, tile_n;
        band_order(rem, a.tiles_m, a.nn, tile_m, tile_n);
        gemm_tile<A_KC, B_KC, ATOMIC, BN_TAIL>(a, smem, grp, valid, z, tile_m, a.nb * BN + tile_n * BN_TAIL);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Persistent variant: one workgroup per CU for the whole launch, every wave group walks a list of SEGMENTS -- (output
// tile, range of K steps) -- through ONE uninterrupted LDS pipeline: the loads of a segment's first K steps are issued
// during the last steps of the segment before it, and a segment's epilogue runs in the memory phase that follows its
// last compute phase, i.e. while the OTHER group of the workgroup streams its MFMAs.  What that removes from
// gemm_f32_mfma: the exposed prologue (first loads of a tile) and epilogue of every tile and the gap between two
// workgroups on a CU (one workgroup fits) -- together about 3 K steps per tile, measured as efficiency = steps /
// (steps + 3) on G2 (K = 512: 84 %, 256: 73 %, 128: 57 % of the MFMA peak).
// Schedule ("data-parallel rounds + stream-K tail"): with G groups and T tiles of S steps, every group takes R = T / G
// whole tiles (tile = r * G + slot: the tiles in flight at any time are neighbours in the logical order, as in the flat
// grid); the T - R * G tiles left are
//   ATOMIC (G3: the output is accumulated with atomics anyway): cut into G equal runs of steps, wherever they fall
//          inside a tile (stream-K): every group ends within one step of the others;
//   else   handed out whole (results stay bit-reproducible: plain stores, one writer per element).
// For G3 the images are folded into K (S = images x K tiles): one output tile is one (m, n) block of grad_weight.
constexpr int SK_MAXSEG = 64;          // segments per group the LDS table holds (whole tiles + stream-K pieces)
constexpr int SK_ENTRY = 12;           // dwords per table entry

__device__ __forceinline__ int sk_bound(const GemmArgs &a, int g, bool atomic)
{
    if (g <= 0) return 0;
    if (g >= a.sk_G) return a.sk_utail;
    if (!atomic) {
        const int trem = a.sk_utail / a.sk_S;
        return (int)((long)g * trem / a.sk_G) * a.sk_S;
    }
    int b = (int)((long)g * a.sk_utail / a.sk_G);
    const int r = b % a.sk_S;            // a run that would start / end one step away from a tile boundary is snapped onto it
    if (r == 1) b -= 1;
    else if (r == a.sk_S - 1) b += 1;
    return b;
}

template <bool A_KC, bool B_KC, bool ATOMIC>
__global__ __launch_bounds__(2 * NT) void gemm_f32_sk(GemmArgs a)
{
    typedef typename LoaderSel<A_KC, BM>::type LA;
    typedef typename LoaderSel<B_KC, BN>::type LB;
    constexpr int LDA = LA::LD, LDB = LB::LD;
    constexpr int WN_ = 64, WAVES_N = 2, WM_ = 64, MI = 2, NJ = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    float *As = smem + grp * (BK * LDA + BK * LDB), *Bs = As + BK * LDA;
    int *table = reinterpret_cast<int *>(smem + 2 * (BK * LDA + BK * LDB)) + grp * (SK_MAXSEG * SK_ENTRY);
    const int t = threadIdx.x & (NT - 1);
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int wg = __builtin_amdgcn_readfirstlane(xcd_order(blockIdx.x, gridDim.x));
    const int g = 2 * wg + grp;
    const bool fold = a.sk_kimgs > 1;

    // tail runs of this group and of its partner: both execute the larger step count (barriers are per workgroup)
    const int b0 = __builtin_amdgcn_readfirstlane(sk_bound(a, 2 * wg, ATOMIC));
    const int b1 = __builtin_amdgcn_readfirstlane(sk_bound(a, 2 * wg + 1, ATOMIC));
    const int b2 = __builtin_amdgcn_readfirstlane(sk_bound(a, 2 * wg + 2, ATOMIC));
    const int nsteps = a.sk_R * a.sk_S + max(b1 - b0, b2 - b1);
    const int tb = grp ? b1 : b0, te = grp ? b2 : b1;          // this group's run of tail steps
    const int tfirst = tb / a.sk_S;
    const int nseg = a.sk_R + (te > tb ? (te + a.sk_S - 1) / a.sk_S - tfirst : 0);

    // Segment table, one entry per lane (all the integer divisions of the schedule happen here, in parallel, not in the K
    // loop): {m0, n0, kt, len, offA lo/hi, offB lo/hi, offC lo/hi} -- element offsets of the segment's first step.
    if (t < nseg) {
        int tile, kb, len;
        if (t < a.sk_R) {
            tile = t * a.sk_G + g; kb = 0; len = a.sk_S;
        } else {
            const int q = t - a.sk_R;
            const int start = q == 0 ? tb : (tfirst + q) * a.sk_S;
            const int end = min((tfirst + q + 1) * a.sk_S, te);
            tile = a.sk_R * a.sk_G + tfirst + q; kb = start - (tfirst + q) * a.sk_S; len = end - start;
        }
        const int per_z = a.tiles_m * a.sk_tiles_n;
        const int z = tile / per_z, rem = tile - z * per_z;
        int tm, tn;
        band_order(rem, a.tiles_m, a.sk_tiles_n, tm, tn);
        const int z1 = z % a.nz1, z0 = z / a.nz1;              // folded images: z = weight group, the image comes from kb
        const int img = kb / a.sk_ktiles, kt = kb - img * a.sk_ktiles;
        const long zi = fold ? img : z0;
        const long offA = zi * a.sA0 + z1 * a.sA1, offB = zi * a.sB0 + z1 * a.sB1;
        const long offC = (fold ? 0 : z0 * a.sC0) + z1 * a.sC1;
        int *e = table + t * SK_ENTRY;
        e[0] = tm * BM; e[1] = tn * BN; e[2] = kt; e[3] = len;
        e[4] = (int)offA; e[5] = (int)(offA >> 32); e[6] = (int)offB; e[7] = (int)(offB >> 32);
        e[8] = (int)offC; e[9] = (int)(offC >> 32); e[10] = 0; e[11] = 0;
    }
    __syncthreads();

    // cursors: plain scalars (SGPRs) refilled from the table at a segment change
    struct Cur { int idx, left, m0, n0, kt; long offA, offB, offC; };
    auto fetch = [&](Cur &c) {
        if (c.idx < nseg) {
            const int *e = table + c.idx * SK_ENTRY;
            c.m0 = __builtin_amdgcn_readfirstlane(e[0]); c.n0 = __builtin_amdgcn_readfirstlane(e[1]);
            c.kt = __builtin_amdgcn_readfirstlane(e[2]); c.left = __builtin_amdgcn_readfirstlane(e[3]);
            const unsigned al = __builtin_amdgcn_readfirstlane(e[4]), ah = __builtin_amdgcn_readfirstlane(e[5]);
            const unsigned bl = __builtin_amdgcn_readfirstlane(e[6]), bh = __builtin_amdgcn_readfirstlane(e[7]);
            const unsigned cl = __builtin_amdgcn_readfirstlane(e[8]), ch = __builtin_amdgcn_readfirstlane(e[9]);
            c.offA = (long)(((unsigned long)ah << 32) | al);
            c.offB = (long)(((unsigned long)bh << 32) | bl);
            c.offC = (long)(((unsigned long)ch << 32) | cl);
        } else {
            c.left = 0;
        }
        ++c.idx;
    };
    Cur lc{}, cc{};                    // load cursor (two steps ahead of the compute cursor)
    fetch(lc);
    fetch(cc);

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    LA la; LB lb;
    la.init(As, a.lda, t);
    lb.init(Bs, a.ldb, t);
    f32x4 ra[LA::NV], rb[LB::NV];

    // issue the loads of the load cursor's step (an exhausted cursor reads zeros through empty descriptors), then advance it
    auto gload = [&]() -> int {
        const bool live = lc.left > 0;
        const int k0 = lc.kt * BK;
        const int kvalid = live ? max(a.K - k0, 0) : 0;
        const long oa = LA::origin(a.lda, lc.m0, k0), ob = LB::origin(a.ldb, lc.n0, k0);
        const long nA = live ? a.nA : 0, nB = live ? a.nB : 0;
        la.load(make_rsrc(a.A + lc.offA + oa, LA::extent(a.lda, nA - oa, kvalid, lc.m0)), ra);
        lb.load(make_rsrc(a.B + lc.offB + ob, LB::extent(a.ldb, nB - ob, kvalid, lc.n0)), rb);
        if (live) {
            if (--lc.left == 0) {
                fetch(lc);
            } else if (++lc.kt == a.sk_ktiles) {     // folded images: on to the next image's slice
                lc.kt = 0; lc.offA += a.sA0; lc.offB += a.sB0;
            }
        }
        return kvalid;
    };
    auto sstore = [&](int kvalid) {
        la.store(ra, kvalid);
        lb.store(rb, kvalid);
    };

    int kv = gload();
    sstore(kv);
    kv = gload();
    __syncthreads();
    if (grp == 1) __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    const unsigned lane_off = (unsigned)(((long)(wm * WM_ + 4 * lh) * a.ldc + wn * WN_ + li) * 4);
    bool clear_acc = false;
    for (int step = 0; step < nsteps; ++step) {
        __builtin_amdgcn_s_setprio(2);
        if (clear_acc) {           // first step of a new segment (uniform): the stores of the last one were issued a phase ago
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = zero16;
            clear_acc = false;
        }
        __builtin_amdgcn_sched_barrier(0);
        const float *as = As + wm * WM_ + li;
        const float *bs = Bs + wn * WN_ + li;
        constexpr int PF = 2, KS = BK / 2;
        float av[PF + 1][MI], bv[PF + 1][NJ];
#pragma unroll
        for (int s0 = 0; s0 < PF; ++s0) {
#pragma unroll
            for (int i = 0; i < MI; ++i) av[s0][i] = as[(2 * s0 + lh) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[s0][j] = bs[(2 * s0 + lh) * LDB + 32 * j];
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int c = kk % (PF + 1), nx = (kk + PF) % (PF + 1);
            if (kk + PF < KS) {
                const int k = 2 * (kk + PF) + lh;
#pragma unroll
                for (int i = 0; i < MI; ++i) av[nx][i] = as[k * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) bv[nx][j] = bs[k * LDB + 32 * j];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][i], bv[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MI + NJ, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MI * NJ, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();   // this group's step is consumed; the other group starts computing
        sstore(kv);
        __builtin_amdgcn_sched_barrier(0);
        kv = gload();
        __builtin_amdgcn_sched_barrier(0);
        // The segment's last step has just been computed: write it out under the other group's compute phase.  That
        // phase streams MFMAs at raised priority, and a wave's VALU instructions barely issue beside it (measured: x10),
        // so the epilogue of an interior tile is VALU-free: the lane's offset is tile-independent, the (i, j, e) part a
        // scalar offset, the accumulators are cleared at the start of this group's own next compute phase.  Edge tiles
        // (last row / column of tiles) mask per element.
        if (cc.left > 0 && --cc.left == 0) {
            const long o = (long)cc.m0 * a.ldc + cc.n0;
            const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.C + cc.offC + o, a.nC - o);
            const bool interior = cc.m0 + BM <= a.M && cc.n0 + BN <= a.N;
            if (interior) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int mr = i * 32 + (e & 3) + 8 * (e >> 2);
                            const unsigned soff = (unsigned)(((long)mr * a.ldc + j * 32) * 4);
                            const float v = acc[i][j][e];
                            if (ATOMIC) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rc, lane_off, soff, 0);
                            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, lane_off, soff, 0);
                        }
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int nl = wn * WN_ + j * 32 + li;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int ml = wm * WM_ + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                            const bool ok = (cc.m0 + ml < a.M) && (cc.n0 + nl < a.N);
                            const unsigned voff = ok ? (unsigned)(((long)ml * a.ldc + nl) * 4) : OOB;
                            const float v = acc[i][j][e];
                            if (ATOMIC) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rc, voff, 0, 0);
                            else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rc, voff, 0, 0);
                        }
                    }
            }
            clear_acc = true;
            fetch(cc);
        }
        __syncthreads();   // the next step is in place; the other group starts its memory phase
    }
    if (grp == 0) __syncthreads();
}

static int device_slots()
{
    // tiles in flight on the current device: one two-group workgroup per CU
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 512;
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached = 2 * cus;
        cached_dev = dev;
    }
    return cached;
}

template <bool A_KC, bool B_KC, bool ATOMIC>
static int launch(GemmArgs &a, int nz0, hipStream_t stream, const char *what)
{
    constexpr int LDA = LoaderSel<A_KC, BM>::type::LD;
    constexpr int LDB = LoaderSel<B_KC, BN>::type::LD;
    const size_t smem = (size_t)(2 * BK * LDA + 2 * BK * LDB) * sizeof(float);   // one A+B buffer per wave group
    // per-lane byte offsets are 32 bit and span one tile of rows: 128 rows * ld * 4 bytes must stay below 2^30
    if (a.lda >= (1L << 21) || a.ldb >= (1L << 21) || a.ldc >= (1L << 21))
        return fail(DDCN_ERR_UNSUPPORTED, "%s: leading dimension beyond 2^21 elements", what);
    // > 64 KiB of dynamic LDS needs the opt-in: raised once per device to this instantiation's (constant) size
    static std::atomic<uint64_t> raised{0};
    if (!raise_lds_limit(reinterpret_cast<const void *>(&gemm_f32_mfma<A_KC, B_KC, ATOMIC>), (int)smem, raised))
        return fail(DDCN_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit to %zu bytes", what, smem);
    a.tiles_m = (a.M + BM - 1) / BM;
    const long tiles_n = (a.N + BN - 1) / BN;
    const long nz = (long)nz0 * a.nz1 * a.ksplit;
    if (a.tiles_m == 0 || tiles_n == 0 || nz == 0) return DDCN_OK;
    // wide tiles for the whole rounds, narrow tiles for the remainder
    const long slots = device_slots();
    const long all_big = a.tiles_m * tiles_n * nz;
    long nb = tiles_n;
    static const int env_tail = getenv("DDCN_GEMM_TAIL") ? atoi(getenv("DDCN_GEMM_TAIL")) : 1;
    if (env_tail && all_big > slots && all_big % slots != 0) nb = (all_big / slots) * slots / (a.tiles_m * nz);
    // less than one round of wide tiles (skinny problems such as the 9-row mask convolution of the DDCN layer): the
    // launch lasts as long as ONE tile's K loop, and a narrow tile's loop is ~3x shorter
    if (env_tail && all_big < slots && !ATOMIC) nb = 0;
    if (nb > tiles_n) nb = tiles_n;
    const long rest = (long)a.N - nb * BN;
    const long nn = rest > 0 ? (rest + BN_TAIL - 1) / BN_TAIL : 0;
    const long total_big = a.tiles_m * nb * nz, total_nar = a.tiles_m * nn * nz;
    if (total_big + total_nar > 0x7fffffffL) return fail(DDCN_ERR_UNSUPPORTED, "%s: grid too large", what);
    if (a.kchunk <= 0 || a.kchunk % BK) return fail(DDCN_ERR_BAD_ARG, "%s: reduction chunk not a multiple of %d", what, BK);
    a.nb = (int)nb; a.nn = (int)nn; a.total_big = (int)total_big; a.total_nar = (int)total_nar;
    a.blocks_big = (int)((total_big + 1) / 2); a.blocks_nar = (int)((total_nar + 1) / 2);
    hipLaunchKernelGGL((gemm_f32_mfma<A_KC, B_KC, ATOMIC>), dim3((unsigned)(a.blocks_big + a.blocks_nar)), dim3(2 * NT),
                       smem, stream, a);
    return check_launch(what);
}


constexpr int SK_NOT_APPLICABLE = -1000;   // internal: the caller falls back to the flat grid

// Persistent launch (see gemm_f32_sk).  fold_images: the nz0 images are part of the contraction (G3).
template <bool A_KC, bool B_KC, bool ATOMIC>
static int launch_sk(GemmArgs &a, int nz0, bool fold_images, hipStream_t stream, const char *what)
{
    constexpr int LDA = LoaderSel<A_KC, BM>::type::LD;
    constexpr int LDB = LoaderSel<B_KC, BN>::type::LD;
    const size_t smem = (size_t)(2 * BK * LDA + 2 * BK * LDB) * sizeof(float) + 2 * SK_MAXSEG * SK_ENTRY * sizeof(int);
    if (a.lda >= (1L << 21) || a.ldb >= (1L << 21) || a.ldc >= (1L << 21))
        return fail(DDCN_ERR_UNSUPPORTED, "%s: leading dimension beyond 2^21 elements", what);
    static std::atomic<uint64_t> raised{0};
    if (!raise_lds_limit(reinterpret_cast<const void *>(&gemm_f32_sk<A_KC, B_KC, ATOMIC>), (int)smem, raised))
        return fail(DDCN_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit to %zu bytes", what, smem);
    a.tiles_m = (a.M + BM - 1) / BM;
    const long tiles_n = (a.N + BN - 1) / BN;
    const long nzo = (fold_images ? 1 : (long)nz0) * a.nz1;
    const long T = a.tiles_m * tiles_n * nzo;
    const long ktiles = (a.K + BK - 1) / BK;
    const long S = ktiles * (fold_images ? nz0 : 1);
    if (T == 0 || S == 0) return DDCN_OK;
    const long G = device_slots();
    const long R = T / G;
    const long utail = (T - R * G) * S;
    if (T > 0x3fffffffL || utail > 0x3fffffffL || S < 4) return fail(DDCN_ERR_UNSUPPORTED, "%s: schedule out of range", what);
    // segments per group: R whole tiles + the pieces of its tail run; the LDS table holds SK_MAXSEG
    if (R + (utail / G + S - 1) / S + 3 > SK_MAXSEG) return SK_NOT_APPLICABLE;
    a.sk_G = (int)G; a.sk_T = (int)T; a.sk_S = (int)S; a.sk_ktiles = (int)ktiles; a.sk_kimgs = fold_images ? nz0 : 1;
    a.sk_R = (int)R; a.sk_tiles_n = (int)tiles_n; a.sk_utail = (int)utail;
    hipLaunchKernelGGL((gemm_f32_sk<A_KC, B_KC, ATOMIC>), dim3((unsigned)(G / 2)), dim3(2 * NT), smem, stream, a);
    return check_launch(what);
}

// DDCN_GEMM_SK: 0 = flat grid everywhere (gemm_f32_mfma), 1 = persistent kernel wherever it is built (G2, G3), unset = where it
// was measured faster (profiles/r03b_gemm_persistent.txt): G2 with a short contraction (Co / groups <= 256: the flat grid
// pays ~3 K steps of prologue / epilogue / dispatch per tile, 8 steps long there: 73 % -> 80 % of the MFMA peak at the R101
// layer3 shape, 55 % -> 61 % at layer2's).  Not for long contractions: the statically partitioned persistent grid ends with its
// slowest CU and its stream-K runs share no operand tiles in L2 (G3 at config 1: 0.76 vs 0.70 ms).
static bool use_sk(long tiles, long steps_per_tile, bool g3)
{
    const char *e = getenv("DDCN_GEMM_SK");          // read per call: tests switch it
    const int env = e ? atoi(e) : -1;
    if (env >= 0) return env != 0 && steps_per_tile >= 4;
    if (g3) return false;
    return steps_per_tile >= 4 && steps_per_tile <= 8 && tiles * steps_per_tile >= 8L * device_slots();
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
    static const int env_nt = getenv("DDCN_GEMM_NT") ? atoi(getenv("DDCN_GEMM_NT")) : 0;
    a.stream_out = env_nt;    // opt-in: measured slower (0.704 vs 0.692 ms), see DDCN_IM2COL_STORE_AUX
    const long tiles = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * g.groups * g.N;
    if (!env_nt && use_sk(tiles, (a.K + BK - 1) / BK, false)) {
        const int rc = launch_sk<false, false, false>(a, g.N, false, stream, "gemm_gcol");
        if (rc != SK_NOT_APPLICABLE) return rc;
    }
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
    const long tiles_out = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * g.groups;
    if (use_sk(tiles_out, (long)((a.K + BK - 1) / BK) * g.N, true)) {
        const int rc = launch_sk<true, true, true>(a, g.N, true, stream, "gemm_gweight");
        if (rc != SK_NOT_APPLICABLE) return rc;
    }
    // Split-K factor from a cost model of the flat grid (in K steps; fitted on profiles/r03b_gw_ksplit.txt, where it ranks
    // the twelve factors tried at config 1 in their measured order): whole rounds of wide tiles cost their K loop plus ~3
    // steps of prologue / epilogue / dispatch; the tiles of the last, partial round run as quarter-width tiles, four
    // times as many, whose steps cost ~0.4 of a wide one (launch() below); less than one round stays wide.
    const long tiles_mn = (long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * g.groups * g.N;
    const int ktiles = (d.P + BK - 1) / BK;
    const long slots = device_slots();
    int ksplit = 1;
    double best = 1e30;
    for (int s = 1; s <= 16 && s <= ktiles; ++s) {
        const double steps = (double)((ktiles + s - 1) / s);
        const long tiles = tiles_mn * s, full = tiles / slots, rem = tiles - full * slots;
        double cost;
        if (full == 0) cost = steps + 3;
        else cost = full * (steps + 3) + (rem ? (double)((4 * rem + slots - 1) / slots) * (0.4 * steps + 3) : 0.0);
        if (cost < best - 1e-9) { best = cost; ksplit = s; }
    }
    if (const char *e = getenv("DDCN_GEMM_GW_KSPLIT")) ksplit = atoi(e);     // experiment
    if (ksplit > ktiles) ksplit = ktiles;
    if (ksplit < 1) ksplit = 1;
    const int tiles_per = (ktiles + ksplit - 1) / ksplit;
    a.kchunk = tiles_per * BK;
    a.ksplit = (ktiles + tiles_per - 1) / tiles_per;
    return launch<true, true, true>(a, g.N, stream, "gemm_gweight");
}

}  // namespace ddcn

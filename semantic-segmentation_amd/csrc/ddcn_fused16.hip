// Fused 16-bit forward of the (modulated) deformable convolution for gfx950: the columns are never written.
// SURVEY.md 8(f1) "an implicit-GEMM main op that never materialises `columns`"; the arithmetic is the reference's
// K:569-632 (sampling) and H:430-435 (contraction), bf16 / f16 operands, fp32 coordinate math and accumulation --
// element for element what ddcn_bf16.hip computes through its columns buffer (same blend, same rounding of a sample).
//
// One workgroup (512 threads, 8 waves as 4 row groups x 2 column groups) owns a tile of 8 x 16 output pixels and up to
// 256 output channels (more: several workgroups per pixel tile, each sampling for itself).  The contraction runs in
// steps of 32 k = ONE tap x 32 input channels:
//   A  the weights of the step, repacked per call to [step][Co][32] (64 bytes per row), go global -> LDS by DMA through a
//      ring of four stages (row chunks XOR-swizzled for conflict-free 16x16x32 fragment reads, as in gemm16_dma_kernel);
//   B  is SAMPLED: every thread owns one (pixel, channel octet) of the step, reads the four corners of the tap's sample
//      position from the input patch in LDS (one ds_read_b128 per corner = eight channels), blends them in fp32, rounds
//      once and writes the 16-byte octet -- exactly one MFMA B fragment -- into the step's B tile (4 octets x 128 pixels).
//      The patch of the tile (tile + kernel extent + a margin for the learned offsets, 32 channels) comes by DMA from a
//      channels-last copy of the input (made per call: a cell's 32 channels are 64 contiguous bytes there), double
//      buffered across channel blocks, its cells outside the image zero-filled by the DMA's range check; samples whose
//      corners leave the patch gather their corners from the channels-last copy directly.
//   A step samples B(s + 1), requests a piece of the next channel block's patch and A(s + 3) (the ring has four stages, so a
//   request has two whole steps to land: the wait at the end of a step allows the two newest steps' requests to be in
//   flight), runs the MFMAs of step s and ends in ONE barrier.
//
// Measured (profiles/r03c_fused16.txt): equal to the columns path at C = 128, slower at C = 256 and 512 -- the sampling
// arithmetic that im2col hides under its HBM writes here competes with the MFMAs for issue slots.  The entry points are
// therefore opt-in; the default 16-bit forward stays im2col + GEMM.
#include "ddcn_16bit.h"

#include <stdlib.h>

#include <algorithm>

namespace ddcn {
namespace {

constexpr int FT = 512;                     // threads
constexpr int FTH = 8, FTW = 16, FPIX = 128;   // output pixels of a tile
constexpr int FCB = 32;                     // input channels per block = k of one step
constexpr int FLDS_MAX = 160 * 1024;
constexpr int FNST = 4;                     // stages of the weight ring
constexpr int FMAXPIECE = 8;                // patch DMA requests per thread and block (one per tap 0..7)

struct FusedArgs {
    const bf16_t *x, *offset, *mask, *bias, *weight;
    bf16_t *xt;            // [N][H][W][C]
    bf16_t *wp;            // [C/32 * 9][Co][32]
    bf16_t *out;
    int N, C, H, W, Co, Ho, Wo, P;
    int sh, sw, ph, pw, dh, dw, dg, cpg;
    int M, RH, RW, RS;
    int tiles_w, tiles;
    int nblocks;           // C / 32
    int npiece;
};

// x[n][c][h][w] -> xt[n][h][w][c]: block = 32 channels x 64 consecutive pixels of one image, transposed through LDS
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const bf16_t *x, bf16_t *xt, int C, int HW)
{
    __shared__ u32x4 tile[64 * 4];
    const int t = threadIdx.x;
    const int n = blockIdx.z, cb = blockIdx.y;
    const long pix0 = (long)blockIdx.x * 64;
    {
        const int pix = t & 63, oct = t >> 6;
        const long p = pix0 + pix;
        const bf16_t *src = x + ((long)n * C + cb * 32 + oct * 8) * HW + p;
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p < HW ? src[(long)e * HW] : (unsigned short)0;
        u32x4 o;
        o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
        o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
        tile[pix * 4 + oct] = o;
    }
    __syncthreads();
    {
        const int pix = t >> 2, oct = t & 3;
        const long p = pix0 + pix;
        if (p < HW) *(u32x4 *)(xt + ((long)n * HW + p) * C + cb * 32 + oct * 8) = tile[t];
    }
}

// wp[((cb * 9 + tap) * Co + o) * 32 + c] = w[(o * C + cb * 32 + c) * 9 + tap]
__global__ __launch_bounds__(256) void repack_w_steps_kernel(const bf16_t *w, bf16_t *wp, long total, int C, int Co)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i & 31);
    long rest = i >> 5;
    const int o = (int)(rest % Co); rest /= Co;
    const int tap = (int)(rest % 9);
    const int cb = (int)(rest / 9);
    wp[i] = w[((long)o * C + cb * 32 + c) * 9 + tap];
}

// LDS slot (16-byte units) of channel octet q of patch cell c: the XOR spreads the 16 lanes of a ds_read_b128 group,
// which read consecutive cells, over all banks
__device__ __forceinline__ int patch_slot(int cell, int q) { return cell * 4 + (q ^ ((cell >> 2) & 3)); }

// RT = 16-row MFMA tiles per wave along M; the workgroup tile has 4 * RT * 16 rows.
template <typename T, bool MOD, int RT>
__global__ __launch_bounds__(FT, 1) void fused16_kernel(const FusedArgs a)
{
    constexpr int BMF = 4 * RT * 16;                    // rows of the workgroup tile
    constexpr int RA = (BMF * 4 + FT - 1) / FT;         // A requests per thread and step (BMF * 4 chunks of 16 bytes)
    constexpr int RING = RA * FT, BT = 4 * FPIX;        // 16-byte units per ring stage (whole requests) / per B tile
    extern __shared__ u32x4 lds[];
    u32x4 *ring = lds;                                  // [FNST][RING]
    u32x4 *btile = lds + FNST * RING;                   // [2][4][128]
    u32x4 *patch = btile + 2 * BT;                      // [2][npiece * 512 slots]: RS cells x 4 octets, padded to whole DMA pieces
    const int PSTR = a.npiece * FT;                     // (a piece's lanes beyond RS * 4 write zeros: they must stay inside the buffer)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kc = lane >> 4;
    const int tile = blockIdx.x, n = blockIdx.z;
    const int m0 = blockIdx.y * BMF;
    const int tyi = tile / a.tiles_w, txi = tile - tyi * a.tiles_w;
    const int rh0 = tyi * FTH * a.sh - a.ph - a.M, rw0 = txi * FTW * a.sw - a.pw - a.M;

    // sampling role: pixel sp of the tile, channel octet sq of the block
    const int sp = t & (FPIX - 1), sq = t >> 7;
    const int ho = tyi * FTH + (sp >> 4), wo = txi * FTW + (sp & 15);
    const bool live = ho < a.Ho && wo < a.Wo;
    const int pidx = live ? ho * a.Wo + wo : 0;
    const int base_h = ho * a.sh - a.ph, base_w = wo * a.sw - a.pw;

    const __amdgpu_buffer_rsrc_t rA = uniform_rsrc(a.wp, (unsigned)((long)a.nblocks * 9 * a.Co * 64));
    const __amdgpu_buffer_rsrc_t rX = uniform_rsrc(a.xt + (long)n * a.H * a.W * a.C, (unsigned)((long)a.H * a.W * a.C * 2));

    // A requests: request r of wave w fills ring chunks (r * 8 + w) * 64 + lane of a stage
    unsigned offA[RA];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int q = (r * 8 + wave) * 64 + lane;
        const int row = q >> 2, c = (q & 3) ^ a_swizzle(row);
        offA[r] = (row < BMF && m0 + row < a.Co) ? (unsigned)(((m0 + row) * 32 + c * 8) * 2) : DMA_OOB;
    }
    const int nsteps = a.nblocks * 9;
    // every wave issues exactly RA requests per step, also past the last step (out of range: zeros into a free stage), so
    // that the vmcnt arithmetic of sync_step holds to the end
    auto request_a = [&](int step) {
        char *base = (char *)(ring + (step & (FNST - 1)) * RING) + wave * 1024;
        const bool real = step < nsteps;
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const unsigned va = (real && offA[r] != DMA_OOB) ? offA[r] + (unsigned)(step * a.Co * 64) : DMA_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void *)(base + r * 8192), 16, va, 0, 0, 0);
        }
    };
    // patch requests: piece j of wave w fills patch slots (j * 8 + w) * 64 + lane; slot -> (cell, position) -> channel octet
    unsigned offX[FMAXPIECE];
#pragma unroll
    for (int j = 0; j < FMAXPIECE; ++j) {
        const int L = (j * 8 + wave) * 64 + lane;
        const int cell = L >> 2, qc = (L & 3) ^ ((cell >> 2) & 3);
        const int rr = cell / a.RW, rc = cell - rr * a.RW;
        const int y = rh0 + rr, xx = rw0 + rc;
        offX[j] = (L < a.RS * 4 && y >= 0 && y < a.H && xx >= 0 && xx < a.W)
                      ? (unsigned)((((long)y * a.W + xx) * a.C + qc * 8) * 2) : DMA_OOB;
    }
    auto request_patch = [&](int cb, int j) {           // piece j of channel block cb (cells outside the image: zeros)
        if (j < a.npiece) {
            char *base = (char *)(patch + (cb & 1) * PSTR) + (j * 8 + wave) * 1024;
            unsigned vo = DMA_OOB;
#pragma unroll
            for (int jj = 0; jj < FMAXPIECE; ++jj)
                if (jj == j) vo = offX[jj];
            if (vo != DMA_OOB) vo += (unsigned)(cb * 64);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_void *)base, 16, vo, 0, 0, 0);
        }
    };

    // offsets and masks of the pixel for the current deformable group
    float offh[9], offw[9], mk[9];
    auto load_offsets = [&](int grp) {
        const bf16_t *offb = a.offset + ((long)n * a.dg + grp) * 18 * a.P + pidx;
        const bf16_t *maskb = MOD ? a.mask + ((long)n * a.dg + grp) * 9 * a.P + pidx : nullptr;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            offh[k] = live ? T::one(offb[(long)(2 * k) * a.P]) : 0.f;
            offw[k] = live ? T::one(offb[(long)(2 * k + 1) * a.P]) : 0.f;
            mk[k] = (MOD && live) ? T::one(maskb[(long)k * a.P]) : 1.f;
        }
    };

    // one (pixel, octet) sample of step (cb, tap) -> B tile of that step
    auto sample = [&](int cb, int tap, int step) {
        const int i = tap / 3, j = tap - 3 * i;
        const float h = (float)(base_h + i * a.dh) + offh[tap];
        const float w = (float)(base_w + j * a.dw) + offw[tap];
        u32x4 o = {0u, 0u, 0u, 0u};
        if (live && h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W) {      // K:602-610
            const float hf = floorf(h), wf = floorf(w);
            const float lh = h - hf, lw = w - wf, m = mk[tap];
            const float w1 = (1.f - lh) * (1.f - lw) * m, w2 = (1.f - lh) * lw * m;
            const float w3 = lh * (1.f - lw) * m, w4 = lh * lw * m;
            const int hl = (int)hf, wl = (int)wf;
            const int rr = hl - rh0, rc = wl - rw0;
            u32x4 a1, a2, a3, a4;
            if (rr >= 0 && rr + 1 < a.RH && rc >= 0 && rc + 1 < a.RW) {
                const u32x4 *pb = patch + (cb & 1) * PSTR;
                const int c = rr * a.RW + rc;
                a1 = pb[patch_slot(c, sq)]; a2 = pb[patch_slot(c + 1, sq)];
                a3 = pb[patch_slot(c + a.RW, sq)]; a4 = pb[patch_slot(c + a.RW + 1, sq)];
            } else {
                // the corners leave the patch: from the channels-last copy (a corner outside the image reads zeros
                // through the descriptor's range check, as the reference's guards K:97-108 give it value 0)
                const unsigned cbo = (unsigned)((cb * 32 + sq * 8) * 2);
                auto corner = [&](int y, int xx) -> u32x4 {
                    const unsigned vo = (y >= 0 && y < a.H && xx >= 0 && xx < a.W)
                                            ? (unsigned)((((long)y * a.W + xx) * a.C) * 2) + cbo : DMA_OOB;
                    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, vo, 0, 0));
                };
                a1 = corner(hl, wl); a2 = corner(hl, wl + 1); a3 = corner(hl + 1, wl); a4 = corner(hl + 1, wl + 1);
            }
            o.x = blend_pair<T>(a1.x, a2.x, a3.x, a4.x, w1, w2, w3, w4);
            o.y = blend_pair<T>(a1.y, a2.y, a3.y, a4.y, w1, w2, w3, w4);
            o.z = blend_pair<T>(a1.z, a2.z, a3.z, a4.z, w1, w2, w3, w4);
            o.w = blend_pair<T>(a1.w, a2.w, a3.w, a4.w, w1, w2, w3, w4);
        }
        btile[(step & 1) * BT + sq * FPIX + sp] = o;
    };

    f32x4 acc[RT][4];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    auto mfmas = [&](int step) {
        const u32x4 *As = ring + (step & (FNST - 1)) * RING, *Bs = btile + (step & 1) * BT;
        u32x4 fb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = Bs[kc * FPIX + wn * 64 + j * 16 + r16];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int row = wm * (RT * 16) + i * 16 + r16;
            const u32x4 fa = As[row * 4 + (kc ^ a_swizzle(row))];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = T::mfma16(fa, fb[j], acc[i][j]);
        }
    };
    // the end of a step: this wave's LDS writes are done and its requests have landed, except the A requests of the two
    // newest steps (requests complete in order, and a step's patch piece is issued before its A requests, so a piece is
    // complete one step after its own)
    auto sync_step = [&]() {
        __builtin_amdgcn_s_waitcnt(wait_vm_lgkm0(2 * RA));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto sync_all = [&]() {
        __builtin_amdgcn_s_waitcnt(wait_vm_lgkm0(0));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // prologue: the whole patch of block 0, the first group's offsets, A of steps 0..2; then step 0 is sampled
#pragma unroll
    for (int j = 0; j < FMAXPIECE; ++j) request_patch(0, j);
    load_offsets(0);
    sync_all();
    sample(0, 0, 0);
    if (a.nblocks > 1) request_patch(1, 0);
    request_a(0);
    request_a(1);
    request_a(2);
    sync_step();                                        // A(0) has landed

    for (int cb = 0; cb < a.nblocks; ++cb) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int step = cb * 9 + tap;
            if (tap < 8) {
                sample(cb, tap + 1, step + 1);
                if (cb + 1 < a.nblocks && tap + 1 < 8) request_patch(cb + 1, tap + 1);
            } else if (step + 1 < nsteps) {
                const int g0 = (cb * FCB) / a.cpg, g1 = ((cb + 1) * FCB) / a.cpg;
                if (g1 != g0) load_offsets(g1);
                sample(cb + 1, 0, step + 1);
                if (cb + 2 < a.nblocks) request_patch(cb + 2, 0);
            }
            request_a(step + 3);
            mfmas(step);
            sync_step();
        }
    }
    __builtin_amdgcn_s_waitcnt(wait_vm_lgkm0(0));       // the out-of-range requests of the last steps, before the LDS is given back

    // lane l holds rows 4 * (l / 16) + e, column l % 16 of each 16 x 16 tile; column tile j of wave column wn = tile row wn * 4 + j
    const bf16_t *bias = a.bias;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = m0 + wm * (RT * 16) + i * 16 + kc * 4 + e;
            if (row >= a.Co) continue;
            const float bv = bias ? T::one(bias[row]) : 0.f;
            bf16_t *C = a.out + ((long)n * a.Co + row) * a.P;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int oy = tyi * FTH + wn * 4 + j, ox = txi * FTW + r16;
                if (oy < a.Ho && ox < a.Wo) C[(long)oy * a.Wo + ox] = (bf16_t)T::pack2(acc[i][j][e] + bv, 0.f);
            }
        }
}

struct FusedPlan { int RT, M, RH, RW, RS, npiece; size_t lds, xt_bytes, wp_bytes; };

// false: the geometry is outside the fused kernel (the caller uses the columns path)
bool fused_plan(const ddcn_geom &g, FusedPlan &pl, const char **why)
{
    *why = nullptr;
    if (g.kh != 3 || g.kw != 3) { *why = "3x3 kernels only"; return false; }
    if (g.groups != 1) { *why = "groups must be 1"; return false; }
    if (g.C % FCB || (g.C / g.dg) % FCB) { *why = "C and C/deformable_groups must be multiples of 32"; return false; }
    if (g.Co % 8) { *why = "Co must be a multiple of 8"; return false; }
    pl.RT = g.Co > 128 ? 4 : (g.Co > 64 ? 2 : 1);       // more than 256 output channels: several workgroups per pixel tile
    const size_t fixed = (size_t)FNST * std::max(4 * pl.RT * 16 * 4, FT) * 16 + (size_t)2 * 4 * FPIX * 16;
    int M = 6;
    for (; M >= 1; --M) {
        pl.RH = (FTH - 1) * g.sh + 2 * g.dh + 2 * M + 2;
        pl.RW = (FTW - 1) * g.sw + 2 * g.dw + 2 * M + 2;
        pl.RS = pl.RH * pl.RW;
        if (fixed + (size_t)2 * round_up((size_t)pl.RS * 4, FT) * 16 <= (size_t)FLDS_MAX && pl.RS * 4 <= FMAXPIECE * FT) break;
    }
    if (M < 1) { *why = "the input patch of a tile does not fit LDS beside the weight ring"; return false; }
    pl.M = M;
    pl.npiece = (pl.RS * 4 + FT - 1) / FT;
    pl.lds = fixed + (size_t)2 * pl.npiece * FT * 16;
    const long nsteps = (long)(g.C / FCB) * 9;
    pl.xt_bytes = round_up((size_t)g.N * g.H * g.W * g.C * 2, 256);
    pl.wp_bytes = round_up((size_t)nsteps * g.Co * 64, 256);
    if ((long)g.H * g.W * g.C * 2 >= (1L << 31) || nsteps * g.Co * 64 >= (1L << 31) || (long)g.Ho * g.Wo >= (1L << 31) ||
        g.N > 65535 || (g.Co + 4 * pl.RT * 16 - 1) / (4 * pl.RT * 16) > 65535) { *why = "tensor too large"; return false; }
    return true;
}

}  // namespace

int fused16_supported(const ddcn_geom &g)
{
    FusedPlan pl;
    const char *why;
    if (!fused_plan(g, pl, &why)) return fail(DDCN_ERR_UNSUPPORTED, "fused 16-bit forward: %s", why);
    return DDCN_OK;
}

size_t fused16_workspace_bytes(const ddcn_geom &g)
{
    FusedPlan pl;
    const char *why;
    if (!fused_plan(g, pl, &why)) return 0;
    return pl.xt_bytes + pl.wp_bytes;
}

template <typename T>
static int fused_forward_16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                            void *out, void *ws, const ddcn_geom &g, hipStream_t stream)
{
    FusedPlan pl;
    const char *why;
    if (!fused_plan(g, pl, &why)) return fail(DDCN_ERR_UNSUPPORTED, "fused 16-bit forward: %s", why);
    FusedArgs a = {};
    a.x = (const bf16_t *)x; a.offset = (const bf16_t *)offset; a.mask = (const bf16_t *)mask;
    a.bias = (const bf16_t *)bias; a.weight = (const bf16_t *)weight;
    a.xt = (bf16_t *)ws; a.wp = (bf16_t *)((char *)ws + pl.xt_bytes); a.out = (bf16_t *)out;
    a.N = g.N; a.C = g.C; a.H = g.H; a.W = g.W; a.Co = g.Co; a.Ho = g.Ho; a.Wo = g.Wo; a.P = g.Ho * g.Wo;
    a.sh = g.sh; a.sw = g.sw; a.ph = g.ph; a.pw = g.pw; a.dh = g.dh; a.dw = g.dw; a.dg = g.dg; a.cpg = g.C / g.dg;
    a.M = pl.M; a.RH = pl.RH; a.RW = pl.RW; a.RS = pl.RS; a.npiece = pl.npiece;
    a.tiles_w = (g.Wo + FTW - 1) / FTW;
    a.tiles = a.tiles_w * ((g.Ho + FTH - 1) / FTH);
    a.nblocks = g.C / FCB;
    const long HW = (long)g.H * g.W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)(g.C / 32), (unsigned)g.N), dim3(256), 0,
                       stream, a.x, a.xt, g.C, (int)HW);
    const long wtotal = (long)a.nblocks * 9 * g.Co * 32;
    hipLaunchKernelGGL(repack_w_steps_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, stream, a.weight, a.wp, wtotal,
                       g.C, g.Co);
    if (int rc = check_launch("fused16 prepare")) return rc;
    const int bmf = 4 * pl.RT * 16;
    const dim3 grid((unsigned)a.tiles, (unsigned)((g.Co + bmf - 1) / bmf), (unsigned)g.N);
#define DDCN_F16_LAUNCH(MODV, RTV)                                                                                   \
    do {                                                                                                             \
        static std::atomic<uint64_t> raised{0};                                                                      \
        if (!raise_lds_limit((const void *)fused16_kernel<T, MODV, RTV>, FLDS_MAX, raised))                          \
            return fail(DDCN_ERR_LAUNCH, "fused16: cannot raise the LDS limit");                                     \
        hipLaunchKernelGGL((fused16_kernel<T, MODV, RTV>), grid, dim3(FT), pl.lds, stream, a);                       \
    } while (0)
#define DDCN_F16_LAUNCH_RT(RTV)                                                                                      \
    do {                                                                                                             \
        if (mask) DDCN_F16_LAUNCH(true, RTV);                                                                        \
        else DDCN_F16_LAUNCH(false, RTV);                                                                            \
    } while (0)
    if (pl.RT == 4) DDCN_F16_LAUNCH_RT(4);
    else if (pl.RT == 2) DDCN_F16_LAUNCH_RT(2);
    else DDCN_F16_LAUNCH_RT(1);
#undef DDCN_F16_LAUNCH_RT
#undef DDCN_F16_LAUNCH
    return check_launch("fused16");
}

int launch_fused_forward_bf16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                              void *out, void *ws, const ddcn_geom &g, hipStream_t stream)
{
    return fused_forward_16<BF16>(x, weight, bias, offset, mask, out, ws, g, stream);
}

int launch_fused_forward_f16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                             void *out, void *ws, const ddcn_geom &g, hipStream_t stream)
{
    return fused_forward_16<F16>(x, weight, bias, offset, mask, out, ws, g, stream);
}

}  // namespace ddcn

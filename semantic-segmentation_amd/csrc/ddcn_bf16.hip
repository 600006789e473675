// Native bf16 forward of the (modulated) deformable convolution for gfx950 -- BASELINE config 5 (inference:
// bf16 activations / weights / columns, fp32 coordinate math, fp32 MFMA accumulation).  The reference has no
// bf16 path (deform_conv_cuda_kernel.cu:258 dispatches FLOATING_TYPES_AND_HALF); the arithmetic restated is the
// same K:569-632 (im2col) and H:430-435 (GEMM) as the fp32 path, with operands rounded to bf16.
//
// Columns layout: col[n][K/8][P][8] -- "k-octets": 8 consecutive k of one pixel are contiguous (16 bytes), consecutive
// pixels follow.  That is exactly one B fragment of v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 per 16 bytes, so the GEMM needs no
// transpose anywhere, and both the im2col stores and the GEMM loads are coalesced along the pixel axis.  The k order
// inside the path is (channel octet, tap, channel % 8) instead of the reference's (channel, tap): one tap of 8
// channels is one octet, so a lane stores a finished 16-byte vector per tap.  The weights are repacked to the same
// order by a tiny kernel at the start of every call (Co*K elements).
//
// Kernels:
//   repack_w      W[Co][C/g][9] -> W'[Co][(C/g)/8][9][8]
//   im2col_bf16_tile   workgroup = 8x32 tile of output pixels x a group of channels, lane = pixel.  The nine sample
//                 positions (patch cell, bilinear weights x mask) are computed once per lane; the input patch the
//                 tile can reach (tile + kernel extent + a margin for the learned offsets) is staged through LDS
//                 eight channels at a time as [cell][8] bf16, double buffered, so one ds_read_b128 per corner
//                 brings all eight channels of the octet; cells outside the image hold 0 (= the reference's value
//                 for out-of-image corners).  Samples whose corners leave the patch are gathered from global memory
//                 in a second pass.
//   im2col_bf16   (fallback for geometries whose patch does not fit) one lane per (pixel, 8 channels), corners read
//                 from global memory.
//   gemm_bf16     out[Co][P] = W[Co][K] * col: 128x128x64 tiles, 4 waves of 64x64 (2x2 MFMA 32x32x16), operands
//                 staged global -> registers -> LDS (W rows XOR-swizzled by 16-byte chunk, columns linear),
//                 next tile's global loads in flight during the MFMAs, bias add and bf16 rounding in the epilogue.
//                 Workgroups that share a columns tile are consecutive on one XCD (its L2 serves the re-reads).
//   gemm16_dma    the default GEMM: operands global -> LDS by DMA through a ring of tiles of 32 k, weights repacked
//                 per K step, 16x16x32 MFMAs, K loop software-pipelined across one barrier per tile; tiles of
//                 128x128, 256x128 (4 waves) or 256x256 (8 waves).
#include "ddcn_16bit.h"

#include <stdlib.h>

#include <algorithm>
#include <type_traits>

namespace ddcn {
namespace {

struct Bf16Args {
    const bf16_t *x, *offset, *mask, *weight, *bias;
    bf16_t *col, *out;
    int N, C, H, W, Co, Ho, Wo, P;
    int sh, sw, ph, pw, dh, dw, groups, dg, cpg;
    int KO;          // k-octets per image: C*9/8
    int Kg, Cog;     // per weight group: K extent, output channels
    int tiles_m, tiles_n;
    // LDS-tiled im2col
    int M, RH, RW, RS, CG, tiles, tiles_w;
    int pair, rw_shift;   // pair staging: patch starts rw_shift columns further left so that it starts on an even column
    float *outf;          // GEMM result as fp32 rows of ldc floats (the column gradients of the 16-bit backward) instead of `out`
    long ldc;
    int out_atomic;       // outf is accumulated with atomics (split-K / per-image partial products of G3)
    // generic operand placement of the LDS-DMA GEMMs (set by gemm_operands(); the forward's values reproduce round 1)
    long sA_grp;          // elements between the A slabs of two groups / of two images
    long sA_img;
    long sB_img, sB_grp;  // 16-byte chunks between the B slabs of two images / groups
    int ldb;              // 16-byte chunks per k-octet row of B (>= the columns P this GEMM reads)
    int kt_per;           // K tiles (32 k) per split; blockIdx.z = split
    // im2col with pixel octets (G3 of the 16-bit backward): col[n][tiles*32][Ktot][8], pixel order = (tile, lane)
    int pixoct;
    int pp_prio;          // experiment: s_setprio level of the MFMA phase in the ping-pong schedule
};

constexpr int IT = 256;  // im2col threads

// One sample position, 8 consecutive channel planes, corners read from global memory (K:569-632 / K:466-496):
// -> the octet of bf16 samples (mask folded in).
template <typename T>
__device__ __forceinline__ u32x4 sample_octet_global(const bf16_t *xb, long HW, int H, int W, float h, float w, float m)
{
    // K:602-610: zero unless -1 < h < H and -1 < w < W; corners outside the map contribute zero
    const bool inside = h > -1.f && w > -1.f && h < (float)H && w < (float)W;
    const float hf = floorf(h), wf = floorf(w);
    const int hl = (int)hf, wl = (int)wf;
    const float lh = h - hf, lw = w - wf;
    const bool t = inside && hl >= 0, b = inside && hl + 1 <= H - 1;
    const bool l = wl >= 0, r = wl + 1 <= W - 1;
    const float w1 = (t && l) ? (1.f - lh) * (1.f - lw) * m : 0.f;
    const float w2 = (t && r) ? (1.f - lh) * lw * m : 0.f;
    const float w3 = (b && l) ? lh * (1.f - lw) * m : 0.f;
    const float w4 = (b && r) ? lh * lw * m : 0.f;
    // a corner outside gets weight 0, a clamped (valid) address, and its value replaced by 0 (silences Inf/NaN)
    const int hc0 = min(max(hl, 0), H - 1), hc1 = min(max(hl + 1, 0), H - 1);
    const int wc0 = min(max(wl, 0), W - 1), wc1 = min(max(wl + 1, 0), W - 1);
    const int i1 = hc0 * W + wc0, i2 = hc0 * W + wc1, i3 = hc1 * W + wc0, i4 = hc1 * W + wc1;
    // All 32 loads are unconditional (the addresses are clamped into the map) and issued before the first use: a
    // load under a condition becomes a branch with its own wait, 32 memory latencies in a row.
    bf16_t raw[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const bf16_t *pl = xb + c * HW;
        raw[c][0] = pl[i1]; raw[c][1] = pl[i2]; raw[c][2] = pl[i3]; raw[c][3] = pl[i4];
    }
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float x1 = (t && l) ? T::one(raw[c][0]) : 0.f, x2 = (t && r) ? T::one(raw[c][1]) : 0.f;
        const float x3 = (b && l) ? T::one(raw[c][2]) : 0.f, x4 = (b && r) ? T::one(raw[c][3]) : 0.f;
        v[c] = w1 * x1 + w2 * x2 + w3 * x3 + w4 * x4;
    }
    u32x4 o;
    o.x = T::pack2(v[0], v[1]);
    o.y = T::pack2(v[2], v[3]);
    o.z = T::pack2(v[4], v[5]);
    o.w = T::pack2(v[6], v[7]);
    return o;
}

// K:569-632 with bf16 tensors.  grid = (ceil(P/IT), C/8, N)
template <typename T, bool MOD>
__global__ __launch_bounds__(IT, 4) void im2col_bf16_kernel(const Bf16Args a)
{
    const int p = blockIdx.x * IT + threadIdx.x;
    if (p >= a.P) return;
    const int co = blockIdx.y, n = blockIdx.z;
    const int c0 = co * 8, dgi = c0 / a.cpg;
    const int ho = p / a.Wo, wo = p % a.Wo;
    const long HW = (long)a.H * a.W;
    const bf16_t *xb = a.x + ((long)n * a.C + c0) * HW;
    const bf16_t *offb = a.offset + ((long)n * a.dg + dgi) * 18 * a.P + p;
    const bf16_t *maskb = MOD ? a.mask + ((long)n * a.dg + dgi) * 9 * a.P + p : nullptr;

    u32x4 *dst = (u32x4 *)a.col + ((long)n * a.KO + (long)co * 9) * a.P + p;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int i = k / 3, j = k % 3;
        const float oh = T::one(offb[(long)(2 * k) * a.P]), ow = T::one(offb[(long)(2 * k + 1) * a.P]);
        const float m = MOD ? T::one(maskb[(long)k * a.P]) : 1.f;
        const float h = (float)(ho * a.sh - a.ph + i * a.dh) + oh;
        const float w = (float)(wo * a.sw - a.pw + j * a.dw) + ow;
        dst[(long)k * a.P] = sample_octet_global<T>(xb, HW, a.H, a.W, h, w, m);
    }
}

// One weight row element per thread: W'[row][oct][tap][e] = W[row][(oct*8+e)*9 + tap]
__global__ __launch_bounds__(256) void repack_w_kernel(const bf16_t *w, bf16_t *wp, long total, int K)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long row = i / K;
    const int kk = (int)(i - row * K);
    const int e = kk & 7, tap = (kk >> 3) % 9, oct = (kk >> 3) / 9;
    wp[i] = w[row * K + (oct * 8 + e) * 9 + tap];
}

// The same permutation into the layout of the LDS-DMA GEMMs: W''[group][K/32][M][32], zero-padded to whole 32-k
// steps.  The A tile of one K step is then one contiguous run of 64 bytes per row (whole 128-byte lines per request;
// row-major rows of K put every 64-byte piece in a line of its own, and the other half of that line has left the L1
// by the time the next K step asks for it).
__global__ __launch_bounds__(256) void repack_w_blocked_kernel(const bf16_t *w, bf16_t *wp, long total, int K, int M)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int KT = (K + 31) / 32;
    const int k32 = (int)(i & 31);
    const long rest = i >> 5;
    const int row = (int)(rest % M);
    const long gk = rest / M;
    const int kt = (int)(gk % KT);
    const long grp = gk / KT;
    const int kk = kt * 32 + k32;
    bf16_t v = 0;
    if (kk < K) {
        const int e = kk & 7, tap = (kk >> 3) % 9, oct = (kk >> 3) / 9;
        v = w[(grp * M + row) * K + (oct * 8 + e) * 9 + tap];
    }
    wp[i] = v;
}

// ---------------------------------------------------------------- LDS-tiled im2col
constexpr int TH = 8, TW = 32;

// grid = (tiles * cgroups, 1, N); dynamic LDS = 2 buffers of RS cells x 16 bytes
// PAIR: the patch is staged two cells per 4-byte load (W even, patch columns start on an even image column)
// PIXOCT: pixel-octet columns (the B operand of G3) instead of k-octets; a template parameter because the two store paths
// in one body cost 60 VGPRs (and with them the room for the second staging register set)
template <typename T, bool MOD, int NCELL, bool PAIR, bool PIXOCT = false>
__global__ __launch_bounds__(IT, (PAIR && !PIXOCT) ? 3 : 2) void im2col_bf16_tile_kernel(const Bf16Args a)
{
    extern __shared__ u32x4 patch[];
    const int t = threadIdx.x;
    const int tile = blockIdx.x % a.tiles, cgi = blockIdx.x / a.tiles, n = blockIdx.z;
    const int tyi = tile / a.tiles_w, txi = tile - tyi * a.tiles_w;
    const int ty = t / TW, tx = t - ty * TW;
    const int ho = tyi * TH + ty, wo = txi * TW + tx;
    const bool live = ho < a.Ho && wo < a.Wo;
    // image coords of patch cell 0
    const int rh0 = tyi * TH * a.sh - a.ph - a.M, rw0 = txi * TW * a.sw - a.pw - a.M - a.rw_shift;
    const int c_begin = cgi * a.CG;
    const int dgi = c_begin / a.cpg;
    const int HW = a.H * a.W;
    const int p = live ? ho * a.Wo + wo : 0;
    const bf16_t *xb = a.x + ((long)n * a.C + c_begin) * HW;

    // offsets and masks go in flight first
    float roh[9], row_[9], rm[9];
    {
        const bf16_t *offb = a.offset + ((long)n * a.dg + dgi) * 18 * a.P + p;
        const bf16_t *maskb = MOD ? a.mask + ((long)n * a.dg + dgi) * 9 * a.P + p : nullptr;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            roh[k] = live ? T::one(offb[(long)(2 * k) * a.P]) : 0.f;
            row_[k] = live ? T::one(offb[(long)(2 * k + 1) * a.P]) : 0.f;
            rm[k] = (MOD && live) ? T::one(maskb[(long)k * a.P]) : 1.f;
        }
    }
    // staging: cell i = t + 256 j of the patch -> element offset inside one channel plane, -1 outside the image
    // (PAIR: unit i is the cell pair 2i, 2i+1; xx is even and W is even, so a pair is inside or outside as a whole)
    constexpr int NUNIT = PAIR ? NCELL / 2 : NCELL;
    int soff[NUNIT];
#pragma unroll
    for (int j = 0; j < NUNIT; ++j) {
        const int i = (t + IT * j) * (PAIR ? 2 : 1);
        const int rr = i / a.RW, rc = i - rr * a.RW;
        const int y = rh0 + rr, xx = rw0 + rc;
        soff[j] = (i < a.RS && y >= 0 && y < a.H && xx >= 0 && xx < a.W) ? y * a.W + xx : -1;
    }
    // per tap: patch cell of the (hl, wl) corner (-1 escapee, -2 sample is zero), weights with the mask folded in
    int cell[9];
    float w1[9], w2[9], w3[9], w4[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int i = k / 3, j = k - 3 * i;
        const float h = (float)(ho * a.sh - a.ph + i * a.dh) + roh[k];
        const float w = (float)(wo * a.sw - a.pw + j * a.dw) + row_[k];
        const float hf = floorf(h), wf = floorf(w);
        const float lh = h - hf, lw = w - wf;
        const float m = rm[k];
        w1[k] = (1.f - lh) * (1.f - lw) * m; w2[k] = (1.f - lh) * lw * m;
        w3[k] = lh * (1.f - lw) * m;         w4[k] = lh * lw * m;
        int c = -2;
        if (live && h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W) {  // K:602-610
            const int rr = (int)hf - rh0, rc = (int)wf - rw0;
            c = (rr >= 0 && rr + 1 < a.RH && rc >= 0 && rc + 1 < a.RW) ? rr * a.RW + rc : -1;
        }
        cell[k] = c;
    }

    u32x4 *buf0 = patch, *buf1 = patch + a.RS;
    // Staging registers hold what the loads return, untouched, until stage_store: any use of them (packing included)
    // right after the loads would make the compiler wait for the loads before the sampling they are meant to overlap.
    // PAIR: 8 dwords per unit (one per channel, two cells each); else 8 halfwords per cell, packed late as well.
    constexpr int NSTG = PAIR ? NUNIT * 8 : NCELL * 8;
    // DEEP: a second staging register set, so that the loads of octet oc+2 are issued when the sampling of octet oc starts
    // and have two iterations to arrive (counters on layer 4 of config 5: 43 % of the wave cycles are waits -- an octet's
    // sampling, 1.2 us, is shorter than a load's round trip under load).  Built and measured with pair staging (+32
    // registers): the two-octet loop body needs 253-256 VGPRs and spills 70-170 of them, layer 4 took 221 us instead of
    // 130 (profiles/r02c_bf16_pingpong.txt).  Off; the structure stays for a version that keeps the per-tap state in LDS.
    constexpr bool DEEP = false;
    unsigned stgs[DEEP ? 2 : 1][NSTG];
    auto stage_load = [&](int c0, auto SEL) {   // c0 relative to c_begin; SEL: which register set
        unsigned (&stg)[NSTG] = stgs[decltype(SEL)::value];
        if constexpr (PAIR) {
#pragma unroll
            for (int j = 0; j < NUNIT; ++j)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    stg[j * 8 + c] = soff[j] >= 0 ? *(const unsigned *)(xb + (long)(c0 + c) * HW + soff[j]) : 0u;
        } else {
#pragma unroll
            for (int j = 0; j < NCELL; ++j)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    stg[j * 8 + c] = soff[j] >= 0 ? (unsigned)xb[(long)(c0 + c) * HW + soff[j]] : 0u;
        }
    };
    auto stage_store = [&](u32x4 *buf, auto SEL) {
        const unsigned (&stg)[NSTG] = stgs[decltype(SEL)::value];
        if constexpr (PAIR) {
#pragma unroll
            for (int j = 0; j < NUNIT; ++j) {
                const int i = (t + IT * j) * 2;
                const unsigned *d = stg + j * 8;
                if (i < a.RS) {
                    u32x4 lo, hi;
                    lo.x = (d[0] & 0xffffu) | (d[1] << 16);
                    lo.y = (d[2] & 0xffffu) | (d[3] << 16);
                    lo.z = (d[4] & 0xffffu) | (d[5] << 16);
                    lo.w = (d[6] & 0xffffu) | (d[7] << 16);
                    hi.x = (d[0] >> 16) | (d[1] & 0xffff0000u);
                    hi.y = (d[2] >> 16) | (d[3] & 0xffff0000u);
                    hi.z = (d[4] >> 16) | (d[5] & 0xffff0000u);
                    hi.w = (d[6] >> 16) | (d[7] & 0xffff0000u);
                    buf[i] = lo;
                    buf[i + 1] = hi;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NCELL; ++j) {
                const unsigned *v = stg + j * 8;
                u32x4 o;
                o.x = v[0] | (v[1] << 16);
                o.y = v[2] | (v[3] << 16);
                o.z = v[4] | (v[5] << 16);
                o.w = v[6] | (v[7] << 16);
                if (t + IT * j < a.RS) buf[t + IT * j] = o;
            }
        }
    };

    const int nocts = a.CG / 8;
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, DEEP ? 1 : 0> S1;
    stage_load(0, S0{});
    stage_store(buf0, S0{});
    __syncthreads();
    if constexpr (DEEP) {
        if (nocts > 1) stage_load(8, S1{});
    }
    u32x4 *colp = (u32x4 *)a.col + ((long)n * a.KO + (long)(c_begin / 8) * 9) * a.P + p;
    // Pixel-octet mode (the B operand of G3, which contracts over pixels): col[n][pixel octet][row][8 pixels] with the
    // pixels in (tile, lane) order -- lanes outside the map are zero columns -- and rows ordered (group, tap, channel), so
    // that the 8 channels of an octet are 8 consecutive rows.  Each wave transposes its 64 x 8 block through 1 KiB of LDS
    // behind the patch buffers: lane (q, i) collects channel i of the 8 pixels of lane group q and stores one 16-byte
    // vector; the 8 lanes of a group write 128 contiguous bytes.
    const int Cg = a.C / a.groups;
    u32x4 *tb = patch + 2 * a.RS + (t & ~63);                  // this wave's 64 slots
    // pixel octets of one image: the row-major grid (ho, wo / 8) -- only octets that hold a pixel of the map exist, so the
    // contraction of G3 is Ho * ceil(Wo / 8) * 8 pixels (+7 % at 97 x 97) instead of the tiles' 256 lanes each (+41 %)
    const int OW = (a.Wo + 7) >> 3;
    const long pix8 = (long)a.Ho * OW;
    auto pix_oct = [&](int tl) -> long {                       // octet of lane tl of this tile (callers: lanes inside the map)
        return (long)(tyi * TH + tl / TW) * OW + ((txi * TW + (tl % TW)) >> 3);
    };
    auto pixoct_row = [&](int oc, int k) -> long {             // row of channel c_begin + 8*oc (+ i) for tap k
        const int c = c_begin + oc * 8, grp = c / Cg;
        return (long)grp * a.Kg + (long)k * Cg + (c - grp * Cg);
    };
    auto pixoct_store = [&](int oc, int k, u32x4 o) {
        tb[t & 63] = o;
        __builtin_amdgcn_wave_barrier();
        const unsigned short *hw = (const unsigned short *)(tb + (t & 56)) + (t & 7);   // lane group's 8 pixels, channel t & 7
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = hw[e * 8];
        __builtin_amdgcn_wave_barrier();
        u32x4 r;
        r.x = v[0] | ((unsigned)v[1] << 16); r.y = v[2] | ((unsigned)v[3] << 16);
        r.z = v[4] | ((unsigned)v[5] << 16); r.w = v[6] | ((unsigned)v[7] << 16);
        // the lane group's first pixel decides whether the octet exists (its 8 lanes share ho and wo / 8)
        const int gl = t & ~7;
        if (tyi * TH + gl / TW < a.Ho && txi * TW + (gl % TW) < a.Wo)
            ((u32x4 *)a.col)[(((long)n * pix8 + pix_oct(gl)) * a.KO * 8) + pixoct_row(oc, k) + (t & 7)] = r;
    };
    // one element of a pixel-octet column: lane tl of this tile, channel c_begin + 8*oc + e, tap k
    auto pixoct_put = [&](int oc, int k, int tl, u32x4 o) {
        bf16_t *dst = (bf16_t *)a.col + ((((long)n * pix8 + pix_oct(tl)) * a.KO * 8) + pixoct_row(oc, k)) * 8 + (tl & 7);
        dst[0] = (bf16_t)o.x; dst[8] = (bf16_t)(o.x >> 16); dst[16] = (bf16_t)o.y; dst[24] = (bf16_t)(o.y >> 16);
        dst[32] = (bf16_t)o.z; dst[40] = (bf16_t)(o.z >> 16); dst[48] = (bf16_t)o.w; dst[56] = (bf16_t)(o.w >> 16);
    };
    // one octet: rn holds (or receives) the patch of octet oc+1, rf is free for octet oc+2 (DEEP)
    auto octet = [&](int oc, auto rn, auto rf) __attribute__((always_inline)) {
        const u32x4 *cur = (oc & 1) ? buf1 : buf0;
        u32x4 *nxt = (oc & 1) ? buf0 : buf1;
        if constexpr (DEEP) {
            if (oc + 2 < nocts) stage_load((oc + 2) * 8, rf);
        } else {
            if (oc + 1 < nocts) stage_load((oc + 1) * 8, rn);
        }
        if constexpr (PIXOCT) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const bool use = live && cell[k] >= 0;         // escapees are filled in by the second pass, the rest is zero
                const int ci = use ? cell[k] : 0;
                const u32x4 a1 = cur[ci], a2 = cur[ci + 1], a3 = cur[ci + a.RW], a4 = cur[ci + a.RW + 1];
                const float q1 = use ? w1[k] : 0.f, q2 = use ? w2[k] : 0.f, q3 = use ? w3[k] : 0.f, q4 = use ? w4[k] : 0.f;
                u32x4 o;
                o.x = blend_pair<T>(a1.x, a2.x, a3.x, a4.x, q1, q2, q3, q4);
                o.y = blend_pair<T>(a1.y, a2.y, a3.y, a4.y, q1, q2, q3, q4);
                o.z = blend_pair<T>(a1.z, a2.z, a3.z, a4.z, q1, q2, q3, q4);
                o.w = blend_pair<T>(a1.w, a2.w, a3.w, a4.w, q1, q2, q3, q4);
                if (!use) o = u32x4{0u, 0u, 0u, 0u};           // 0 * Inf of a staged cell must not leak into a zero column
                pixoct_store(oc, k, o);
            }
        } else if (live) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (cell[k] == -1) continue;   // escapee: second pass
                const int ci = cell[k] < 0 ? 0 : cell[k];
                const u32x4 a1 = cur[ci], a2 = cur[ci + 1], a3 = cur[ci + a.RW], a4 = cur[ci + a.RW + 1];
                const float q1 = cell[k] < 0 ? 0.f : w1[k], q2 = cell[k] < 0 ? 0.f : w2[k];
                const float q3 = cell[k] < 0 ? 0.f : w3[k], q4 = cell[k] < 0 ? 0.f : w4[k];
                u32x4 o;
                o.x = blend_pair<T>(a1.x, a2.x, a3.x, a4.x, q1, q2, q3, q4);
                o.y = blend_pair<T>(a1.y, a2.y, a3.y, a4.y, q1, q2, q3, q4);
                o.z = blend_pair<T>(a1.z, a2.z, a3.z, a4.z, q1, q2, q3, q4);
                o.w = blend_pair<T>(a1.w, a2.w, a3.w, a4.w, q1, q2, q3, q4);
                colp[((long)oc * 9 + k) * a.P] = o;
            }
        }
        if (oc + 1 < nocts) stage_store(nxt, rn);
        __syncthreads();
    };
    if constexpr (DEEP) {
        for (int oc = 0; oc < nocts; oc += 2) {
            octet(oc, S1{}, S0{});
            if (oc + 1 < nocts) octet(oc + 1, S0{}, S1{});
        }
    } else {
        for (int oc = 0; oc < nocts; ++oc) octet(oc, S0{}, S0{});
    }
    // Second pass: samples whose corners leave the staged patch, from global memory with the reference's guards.
    // Few lanes hold one (0.5 % of the samples at sigma 2) but almost every wave does, and a wave that walks its own
    // escapees runs the 32-load gather per octet for one or two live lanes.  So the workgroup first lists its
    // escapees in LDS (the patch buffers are free now: slot 0 = counter, then {p | k << 27, h, w, mask} per escapee),
    // then deals (escapee, octet) items to all 256 lanes.  Escapees beyond the list's capacity stay with their lane.
    unsigned *cnt = (unsigned *)patch;
    const unsigned cap = (unsigned)(2 * a.RS - 1);
    if (t == 0) *cnt = 0u;
    __syncthreads();
    if (live) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (cell[k] != -1) continue;
            // rare: reload the offsets rather than keep 27 registers alive through the main loop
            const bf16_t *offb = a.offset + ((long)n * a.dg + dgi) * 18 * a.P + p;
            const float h = (float)(ho * a.sh - a.ph + (k / 3) * a.dh) + T::one(offb[(long)(2 * k) * a.P]);
            const float w = (float)(wo * a.sw - a.pw + (k % 3) * a.dw) + T::one(offb[(long)(2 * k + 1) * a.P]);
            const float m = MOD ? T::one(a.mask[(((long)n * a.dg + dgi) * 9 + k) * a.P + p]) : 1.f;
            const unsigned slot = atomicAdd(cnt, 1u);
            if (slot < cap) {
                u32x4 e;
                e.x = (unsigned)(PIXOCT ? t : p) | ((unsigned)k << 27);
                e.y = __float_as_uint(h); e.z = __float_as_uint(w); e.w = __float_as_uint(m);
                patch[1 + slot] = e;
            } else {
#pragma unroll 2
                for (int oc = 0; oc < nocts; ++oc) {
                    const u32x4 o = sample_octet_global<T>(xb + (long)oc * 8 * HW, HW, a.H, a.W, h, w, m);
                    if (PIXOCT) pixoct_put(oc, k, t, o);
                    else colp[((long)oc * 9 + k) * a.P] = o;
                }
            }
        }
    }
    __syncthreads();
    const int listed = (int)min(*cnt, cap);
    u32x4 *colg = (u32x4 *)a.col + ((long)n * a.KO + (long)(c_begin / 8) * 9) * a.P;
    for (int it = t; it < listed * nocts; it += IT) {
        const int oc = it / listed, ei = it - oc * listed;
        const u32x4 e = patch[1 + ei];
        const int ep = (int)(e.x & 0x7ffffffu), k = (int)(e.x >> 27);
        const u32x4 o = sample_octet_global<T>(xb + (long)oc * 8 * HW, HW, a.H, a.W, __uint_as_float(e.y),
                                               __uint_as_float(e.z), __uint_as_float(e.w));
        if (PIXOCT) pixoct_put(oc, k, ep, o);
        else colg[((long)oc * 9 + k) * a.P + ep] = o;
    }
}

// ---------------------------------------------------------------- GEMM
constexpr int GT = 256, BM = 128, BN = 128, BK = 64;

// Workgroups round-robin over the 8 XCDs by launch index: give each XCD a contiguous run of logical tiles
// (bijective for any count).
__device__ __forceinline__ int xcd_logical(int b, int total)
{
    const int q = total / 8, r = total % 8, xcd = b % 8, k = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// grid = (tiles_m * tiles_n, N * groups)
template <typename T>
__global__ __launch_bounds__(GT) void gemm_bf16_kernel(const Bf16Args a)
{
    __shared__ u32x4 As[BM * BK / 8];  // slot(row, chunk) = row*8 + (chunk ^ (row & 7))
    __shared__ u32x4 Bs[BK / 8 * BN];  // slot(oct, n)     = oct*BN + n

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int L = xcd_logical(blockIdx.x, a.tiles_m * a.tiles_n);
    const int m0 = (L % a.tiles_m) * BM, n0 = (L / a.tiles_m) * BN;  // tiles sharing a columns tile are neighbours
    const int img = blockIdx.y / a.groups, grp = blockIdx.y % a.groups;
    const int M = a.Cog, K = a.Kg, KOg = K / 8;

    const bf16_t *A = a.weight + (long)grp * M * K;
    const u32x4 *B = (const u32x4 *)a.col + ((long)img * a.KO + (long)grp * KOg) * a.P;

    // this thread's four A chunks and four B chunks of a tile
    int a_row[4], a_ch[4], b_oct[4], b_n[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = t + GT * r;
        a_row[r] = q >> 3;
        a_ch[r] = q & 7;
        b_oct[r] = q >> 7;
        b_n[r] = q & 127;
    }
    u32x4 ra[4], rb[4];
    const u32x4 zero = {0u, 0u, 0u, 0u};
    auto gload = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + a_row[r], kc = k0 + a_ch[r] * 8;
            ra[r] = (row < M && kc < K) ? *(const u32x4 *)(A + (long)row * K + kc) : zero;
            const int oct = k0 / 8 + b_oct[r], nn = n0 + b_n[r];
            rb[r] = (oct < KOg && nn < a.P) ? B[(long)oct * a.P + nn] : zero;
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            As[a_row[r] * 8 + (a_ch[r] ^ (a_row[r] & 7))] = ra[r];
            Bs[b_oct[r] * BN + b_n[r]] = rb[r];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int r32 = lane & 31, hi = lane >> 5;
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();  // everyone has finished reading the previous tile
        sstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);  // in flight during the MFMAs below
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            u32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + r32;
                fa[i] = As[row * 8 + ((s * 2 + hi) ^ (row & 7))];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb[j] = Bs[(s * 2 + hi) * BN + wn * 64 + j * 32 + r32];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = T::mfma(fa[i], fb[j], acc[i][j]);
        }
    }

    // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    bf16_t *C = a.out + ((long)img * a.Co + (long)grp * M) * a.P;
    const bf16_t *bias = a.bias ? a.bias + grp * M : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            if (row >= M) continue;
            const float bv = bias ? T::one(bias[row]) : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + r32;
                const float v = acc[i][j][e] + bv;
                if (col < a.P) C[(long)row * a.P + col] = (bf16_t)T::pack2(v, 0.f);
            }
        }
    }
}


// ---------------------------------------------------------------- GEMM, LDS-DMA variant
// Same tiles and fragments as gemm_bf16_kernel, but the operands go global -> LDS directly (buffer_load_dwordx4 ... lds)
// through a ring of NS stages of 32 k each: no staging registers (the kernel fits 4 workgroups on a CU), no second
// barrier, and a tile request has NS-1 K steps to arrive.  One iteration: wait for this wave's requests of tile kt
// (counted vmcnt: the younger tiles may stay in flight), barrier (everyone's part of tile kt has landed AND everyone
// is done reading tile kt-1), request tile kt+NS-1 into the stage tile kt-1 occupied, MFMAs on tile kt.
// The DMA places lane l's 16 bytes at (stage base + l*16), so the XOR swizzle of the W rows is applied to the
// global address each lane fetches from; out-of-range chunks are fetched through an out-of-range offset and land
// as zeros.
// MI = 32-row MFMA tiles per wave along M: 2 -> 128-row workgroup tile (64x64 per wave, one fragment read per MFMA),
// 4 -> 256-row tile (128x64 per wave: 6 fragment reads per 8 MFMAs, the LDS pipe stops being the co-bottleneck).
// NWN = wave columns: 2 -> 4 waves, 128 pixels per tile; 4 -> 8 waves, 256 pixels (with MI = 4 a 256x256 tile: 128
// flop per operand byte fetched through L1/L2 instead of 85, one workgroup of two waves per SIMD on a CU).
// PP (256 x 256 tiles only, opt-in): the two row groups of the workgroup -- one wave of each per SIMD -- alternate between
// an MFMA phase (all 32 MFMAs of a K step) and a memory phase (DMA requests of a later tile, all 12 fragment reads of the
// next), two barriers per K step.  The idea: in the lockstep schedule both waves of a SIMD issue their DMA requests (60-180
// cycles each) at the same time, right after the step's barrier.  Measured: 11 % SLOWER than lockstep -- one wave alone
// does not keep the matrix pipe as busy as two interleaved ones, and the second barrier costs more than the overlap gains.
template <typename T, int NS, int MI, int NWN = 2, bool PP = false>
__global__ __launch_bounds__(128 * NWN, NWN == 4 ? 1 : (MI == 4 ? 2 : (NS == 2 ? 4 : (NS == 3 ? 3 : 2))))
void gemm16_dma_kernel(const Bf16Args a)
{
    constexpr int NT = 128 * NWN;                 // threads; waves = 2 (rows) x NWN (columns)
    constexpr int NWAVES = NT / 64;
    constexpr int BNW = 64 * NWN;                 // pixels of the workgroup tile
    constexpr int BMK = 2 * MI * 32;              // rows of the workgroup tile
    constexpr int RA = BMK * 4 / NT;              // A requests per thread and stage
    constexpr int RB = BNW * 4 / NT;              // B requests per thread and stage (= 2)
    constexpr int STAGE = BMK * 4 + BNW * 4;      // 16-byte chunks per stage
    extern __shared__ u32x4 ring[];   // NS stages of [A: BMK rows x 4 chunks][B: 4 octets x BNW pixels]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // uniform: the DMA's LDS base goes through M0
    const int wm = wave / NWN, wn = wave % NWN;
    const int tiles_m = (a.Cog + BMK - 1) / BMK, tiles_n = (a.P + BNW - 1) / BNW;
    const int L = xcd_logical(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (L % tiles_m) * BMK, n0 = (L / tiles_m) * BNW;
    const int img = blockIdx.y / a.groups, grp = blockIdx.y % a.groups;
    const int M = a.Cog, K = a.Kg;

    // A as repack_w_blocked_kernel leaves it: [slab][K/32][M][32], zero-padded in k; B: k-octet rows of a.ldb chunks.
    // Split-K (blockIdx.z): this workgroup contracts K tiles kt0 .. kt0 + ntiles - 1; the descriptors start there.
    const int kt_all = (K + DK - 1) / DK;
    const int kt0 = blockIdx.z * a.kt_per;
    const int ntiles = min(a.kt_per, kt_all - kt0);
    const int KOg = K / 8 - kt0 * 4;          // k-octets of B from kt0 on (the rest of a partial last tile reads as zeros)
    const __amdgpu_buffer_rsrc_t rA = uniform_rsrc(a.weight + a.sA_img * img + a.sA_grp * grp + (long)kt0 * M * DK,
                                                   (unsigned)((long)ntiles * M * DK * 2));
    const __amdgpu_buffer_rsrc_t rB = uniform_rsrc((const u32x4 *)a.col + a.sB_img * img + a.sB_grp * grp + (long)kt0 * 4 * a.ldb,
                                                   (unsigned)((((long)KOg - 1) * a.ldb + a.P) * 16));

    // this thread's RA chunks of A and RB of B per stage; request r of wave w fills ring chunks (r*NWAVES+w)*64 + lane
    unsigned offA[RA], offB[RB];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int q = (r * NWAVES + wave) * 64 + lane;
        const int row = q >> 2, c = (q & 3) ^ a_swizzle(row);
        offA[r] = (m0 + row < M) ? (unsigned)(((m0 + row) * DK + c * 8) * 2) : DMA_OOB;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int q = (r * NWAVES + wave) * 64 + lane;
        const int oct = q / BNW, nn = n0 + (q % BNW);
        offB[r] = (nn < a.P) ? (unsigned)((oct * a.ldb + nn) * 16) : DMA_OOB;
    }
    auto request = [&](int kt) {
        const int st = kt % NS, k0 = kt * DK;
        char *base = (char *)(ring + st * STAGE) + wave * 1024;
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const unsigned va = offA[r] != DMA_OOB ? offA[r] + (unsigned)(kt * M * DK * 2) : DMA_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void *)(base + r * (NWAVES * 1024)), 16, va, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const unsigned vb = offB[r] != DMA_OOB ? offB[r] + (unsigned)((k0 / 8) * a.ldb * 16) : DMA_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void *)(base + BMK * 64 + r * (NWAVES * 1024)), 16, vb, 0,
                                                     0, 0);
        }
    };

    // v_mfma_f32_16x16x32: one instruction spans the 32 k of a tile; the wave's MI*32 x 64 block is 2*MI x 4 tiles of
    // 16 x 16 (the guide measures this form at 1.15 x the FLOP/s of 32x32x16 at the clock the chip holds under load)
    constexpr int RB16 = 2 * MI;
    f32x4 acc[RB16][4];
#pragma unroll
    for (int i = 0; i < RB16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    const int r16 = lane & 15, kc = lane >> 4;
    // A fragments of one half of the wave's rows (MI tiles of 16 rows), B fragments of its four column tiles
    auto read_a = [&](int kt, int half, u32x4 (&fa)[MI]) {
        const u32x4 *As = ring + (kt % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wm * (MI * 32) + (half * MI + i) * 16 + r16;
            fa[i] = As[row * 4 + (kc ^ a_swizzle(row))];
        }
    };
    auto read_b = [&](int kt, u32x4 (&fb)[4]) {
        const u32x4 *Bs = ring + (kt % NS) * STAGE + BMK * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = Bs[kc * BNW + wn * 64 + j * 16 + r16];
    };
    auto mfmas = [&](int half, const u32x4 (&fa)[MI], const u32x4 (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[half * MI + i][j] = T::mfma16(fa[i], fb[j], acc[half * MI + i][j]);
    };
    // one s_waitcnt: this wave's LDS reads are done (lgkmcnt 0) and all but its `tiles` youngest tile requests have
    // landed (RA + RB requests per tile).  A plain __syncthreads() would drain vmcnt to 0.
    auto wait_all_but = [&](int tiles) {
        if (tiles >= 3) __builtin_amdgcn_s_waitcnt(wait_vm_lgkm0(3 * (RA + RB)));
        else if (tiles == 2) __builtin_amdgcn_s_waitcnt(wait_vm_lgkm0(2 * (RA + RB)));
        else if (tiles == 1) __builtin_amdgcn_s_waitcnt(wait_vm_lgkm0(RA + RB));
        else __builtin_amdgcn_s_waitcnt(wait_vm_lgkm0(0));
    };
    static_assert(NS <= 4 && 3 * (RA + RB) < 64, "vmcnt is 6 bits and wait_all_but knows up to 3 tiles in flight");

    // Software pipeline over the two row halves of a tile.  Fragments are read one block of MFMAs before they are
    // used, and the barrier that publishes tile kt+1 sits between the two MFMA blocks of tile kt:
    //   read A1(kt) | MFMA A0(kt) x B(kt) | wait tile kt+1, barrier, request tile kt+NS into the stage of tile kt
    //   (every wave has read all of it), read B(kt+1), A0(kt+1) | MFMA A1(kt) x B(kt)
    // so no MFMA waits for an LDS read issued right before it, and the ring holds NS tiles.  The loop is unrolled by
    // two so that the B fragments alternate between two register sets.
    u32x4 fa0[MI], fa1[MI], fbx[4], fby[4];
    auto step = [&](int kt, u32x4 (&fbc)[4], u32x4 (&fbn)[4]) {
        read_a(kt, 1, fa1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, fa0, fbc);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < ntiles) {
            // requested so far: tiles 0 .. kt+NS-1; tile kt+1 must have landed
            wait_all_but(min(NS - 2, ntiles - 2 - kt));
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (kt + NS < ntiles) request(kt + NS);
            read_b(kt + 1, fbn);
            read_a(kt + 1, 0, fa0);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfmas(1, fa1, fbc);
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int kt = 0; kt < NS; ++kt)
        if (kt < ntiles) request(kt);
    wait_all_but(min(NS - 1, ntiles - 1));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (PP) {
        static_assert(!PP || (NWN == 4 && MI == 4 && NS == 4), "ping-pong: the 8-wave 256 x 256 tile");
        // Tile T is complete in LDS once EVERY wave has waited for its own pieces of T and a barrier has passed.
        //   phase P0(kt): group 0 MFMAs of tile kt         | group 1: request tile kt-1+NS, read fragments of tile kt
        //   phase P1(kt): group 1 MFMAs of tile kt         | group 0: request tile kt+NS,   read fragments of tile kt+1
        // Both groups wait for their pieces of tile kt+1 before the barrier that ends P0(kt) (group 0 reads it in P1(kt),
        // group 1 in P0(kt+1)).  A stage is requested again only after both groups have read it: tile kt's stage by group 0
        // in P1(kt) (its own read: P1(kt-1); group 1's: P0(kt), complete at that phase's barrier), by group 1 in P0(kt+1).
        u32x4 fa[2 * MI], fb[4];
        auto read_all = [&](int kt) {
            const u32x4 *As = ring + (kt % NS) * STAGE;
            const u32x4 *Bs = As + BMK * 4;
#pragma unroll
            for (int i = 0; i < 2 * MI; ++i) {
                const int row = wm * (MI * 32) + i * 16 + r16;
                fa[i] = As[row * 4 + (kc ^ a_swizzle(row))];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = Bs[kc * BNW + wn * 64 + j * 16 + r16];
        };
        auto mfma_all = [&]() {
            if (a.pp_prio) __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = T::mfma16(fa[i], fb[j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (a.pp_prio) __builtin_amdgcn_s_setprio(0);
        };
        // own pieces of tile `need` have landed (my youngest request is tile `last`); always drains this wave's LDS reads
        auto wait_tile = [&](int need, int last) {
            const int younger = min(last, ntiles - 1) - need;
            if (need < ntiles) wait_all_but(max(younger, 0));
            else __builtin_amdgcn_s_waitcnt(wait_vm_lgkm0(63));
        };
        if (wm == 0) read_all(0);
        for (int kt = 0; kt < ntiles; ++kt) {
            if (wm == 0) {
                mfma_all();
            } else {
                if (kt >= 1 && kt - 1 + NS < ntiles) request(kt - 1 + NS);
                read_all(kt);
            }
            __builtin_amdgcn_sched_barrier(0);
            wait_tile(kt + 1, kt - 1 + NS);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (wm == 1) {
                mfma_all();
            } else {
                if (kt + NS < ntiles) request(kt + NS);
                if (kt + 1 < ntiles) read_all(kt + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        read_b(0, fbx);
        read_a(0, 0, fa0);
        for (int kt = 0; kt < ntiles; kt += 2) {
            step(kt, fbx, fby);
            if (kt + 1 < ntiles) step(kt + 1, fby, fbx);
        }
    }

    // lane l holds rows 4*(l/16) + e, column l % 16 of each 16 x 16 tile
    bf16_t *C = a.out + ((long)img * a.Co + (long)grp * M) * a.P;
    const bf16_t *bias = a.bias ? a.bias + grp * M : nullptr;
#pragma unroll
    for (int i = 0; i < RB16; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = m0 + wm * (MI * 32) + i * 16 + kc * 4 + e;
            if (row >= M) continue;
            const float bv = bias ? T::one(bias[row]) : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wn * 64 + j * 16 + r16;
                const float v = acc[i][j][e] + bv;
                if (col < a.P) {
                    if (a.out_atomic) atomicAdd(a.outf + ((long)grp * M + row) * a.ldc + col, v);   // summed over images and splits
                    else if (a.outf) a.outf[((long)img * a.Co + (long)grp * M + row) * a.ldc + col] = v;
                    else C[(long)row * a.P + col] = (bf16_t)T::pack2(v, 0.f);
                }
            }
        }
    }
}

}  // namespace

static size_t col_bytes(const ddcn_geom &g)
{
    return round_up((size_t)g.N * g.C * 9 * g.Ho * g.Wo * 2, 256);
}

// bf16 columns of all N images, then the repacked weights
size_t bf16_workspace_bytes(const ddcn_geom &g)
{
    const size_t kpad = ((size_t)(g.C / g.groups) * 9 + 31) / 32 * 32;   // whole 32-k steps (blocked layout)
    return col_bytes(g) + (size_t)g.Co * kpad * 2;
}

// 0 when the native bf16 path covers the geometry, else DDCN_ERR_UNSUPPORTED with the reason
int bf16_supported(const ddcn_geom &g)
{
    if (g.kh != 3 || g.kw != 3) return fail(DDCN_ERR_UNSUPPORTED, "bf16 path: 3x3 kernels only (got %dx%d)", g.kh, g.kw);
    if (g.C % 8 || (g.C / g.dg) % 8 || (g.C / g.groups) % 8)
        return fail(DDCN_ERR_UNSUPPORTED, "bf16 path: C, C/deformable_groups and C/groups must be multiples of 8 "
                    "(C=%d dg=%d groups=%d)", g.C, g.dg, g.groups);
    const long P = (long)g.Ho * g.Wo;
    if (P >= (1L << 31) / 16 || (long)g.H * g.W >= (1L << 31) || (long)g.N * g.groups > 65535 || g.C / 8 > 65535 ||
        g.N > 65535 || (long)g.Co * (g.C / g.groups) * 9 >= (1L << 31))
        return fail(DDCN_ERR_UNSUPPORTED, "bf16 path: tensor too large");
    return DDCN_OK;
}

// Chooses margin, patch and channel group of the LDS-tiled im2col; false: use the direct kernel.
static bool plan_bf16_tile(Bf16Args &a, const ddcn_geom &g)
{
    const char *e = getenv("DDCN_BF16_TILE");
    if (e && atoi(e) == 0) return false;
    const char *em = getenv("DDCN_BF16_MARGIN");
    int M = em ? atoi(em) : 6;
    if (M < 0) M = 0;
    // 4-byte staging loads need even rows, even planes and an aligned base
    const char *ep = getenv("DDCN_BF16_PAIR");
    a.pair = (g.W % 2 == 0 && ((uintptr_t)a.x & 3) == 0 && !(ep && atoi(ep) == 0)) ? 1 : 0;
    for (;; --M) {
        a.rw_shift = (a.pair && ((g.pw + M) & 1)) ? 1 : 0;   // TW * sw is even: the parity is the same for every tile
        a.RH = (TH - 1) * g.sh + 2 * g.dh + 2 * M + 2;
        a.RW = (TW - 1) * g.sw + 2 * g.dw + 2 * M + 2 + a.rw_shift;
        if (a.pair) a.RW += a.RW & 1;
        a.RS = a.RH * a.RW;
        // 2 buffers x 2048 cells x 16 B = 64 KB of LDS; the pixel-octet mode keeps 4 KB of them for its transposition slots
        if (a.RS <= (a.pixoct ? 8 * IT - IT / 2 : 8 * IT)) break;
        if (M <= 2) return false;
    }
    a.M = M;
    a.tiles_w = (g.Wo + TW - 1) / TW;
    a.tiles = a.tiles_w * ((g.Ho + TH - 1) / TH);
    // channel group: inside one deformable group; split further while the launch is short of workgroups
    int CG = g.C / g.dg;
    while (CG % 16 == 0 && (long)a.tiles * g.N * (g.C / CG) < 1024) CG /= 2;
    a.CG = CG;
    return (long)a.tiles * (g.C / CG) < (1L << 31);
}

template <typename T, bool MOD>
static void run_bf16_tile(const Bf16Args &a, const dim3 &grid, hipStream_t stream)
{
    const size_t lds = (size_t)2 * a.RS * 16 + (a.pixoct ? IT * 16 : 0);   // + the waves' transposition slots
    const int ncell = (a.RS + IT - 1) / IT;
#define DDCN_I16_LAUNCH(NC, PR)                                                                                       \
    do {                                                                                                              \
        if (a.pixoct) hipLaunchKernelGGL((im2col_bf16_tile_kernel<T, MOD, NC, PR, true>), grid, dim3(IT), lds, stream, a);   \
        else hipLaunchKernelGGL((im2col_bf16_tile_kernel<T, MOD, NC, PR, false>), grid, dim3(IT), lds, stream, a);    \
    } while (0)
    if (a.pair) {
        if (ncell <= 4) DDCN_I16_LAUNCH(4, true);
        else if (ncell <= 6) DDCN_I16_LAUNCH(6, true);
        else DDCN_I16_LAUNCH(8, true);
    } else {
        if (ncell <= 4) DDCN_I16_LAUNCH(4, false);
        else if (ncell <= 6) DDCN_I16_LAUNCH(6, false);
        else DDCN_I16_LAUNCH(8, false);
    }
#undef DDCN_I16_LAUNCH
}

// 256 x 256 tiles.  DDCN_BF16_GEMM_PP=1 / 2 selects the phase-split (ping-pong) schedule, without / with a raised priority
// for the MFMA phase: measured 165-168 us against 149-152 us for the default lockstep schedule on layer4 of config 5
// (profiles/r02c_bf16_pingpong.txt), so it stays an experiment.
template <typename T>
static int launch_gemm16_sq(const dim3 &grid, hipStream_t stream, const Bf16Args &a, const char *what)
{
    const char *e = getenv("DDCN_BF16_GEMM_PP");
    if (e && atoi(e) > 0) {
        static std::atomic<uint64_t> raised{0};
        if (!raise_lds_limit((const void *)gemm16_dma_kernel<T, 4, 4, 4, true>, 4 * 32768, raised))
            return fail(DDCN_ERR_LAUNCH, "%s: cannot raise the LDS limit", what);
        Bf16Args b = a;
        b.pp_prio = atoi(e) - 1;              // DDCN_BF16_GEMM_PP=1: no priority change, 2: MFMA phase at priority 2
        hipLaunchKernelGGL((gemm16_dma_kernel<T, 4, 4, 4, true>), grid, dim3(512), 4 * 32768, stream, b);
    } else {
        static std::atomic<uint64_t> raised{0};
        if (!raise_lds_limit((const void *)gemm16_dma_kernel<T, 4, 4, 4>, 4 * 32768, raised))
            return fail(DDCN_ERR_LAUNCH, "%s: cannot raise the LDS limit", what);
        hipLaunchKernelGGL((gemm16_dma_kernel<T, 4, 4, 4>), grid, dim3(512), 4 * 32768, stream, a);
    }
    return DDCN_OK;
}

// Operand placement of the round-1 forward GEMM: W'' slabs per group, columns [n][K/8][P][8] with the groups splitting k
static void gemm_operands_forward(Bf16Args &a)
{
    const long kt = (a.Kg + 31) / 32;
    a.sA_grp = kt * a.Cog * 32; a.sA_img = 0;
    a.sB_img = (long)a.KO * a.P; a.sB_grp = (long)(a.Kg / 8) * a.P;
    a.ldb = a.P;
    a.kt_per = (int)kt;
    a.out_atomic = 0;
}

template <typename T>
static int forward_16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                      void *out, void *ws, const ddcn_geom &g, hipStream_t stream)
{
    if (int rc = bf16_supported(g)) return rc;
    Bf16Args a = {};
    a.x = (const bf16_t *)x; a.offset = (const bf16_t *)offset; a.mask = (const bf16_t *)mask;
    a.bias = (const bf16_t *)bias;
    a.col = (bf16_t *)ws; a.out = (bf16_t *)out;
    a.N = g.N; a.C = g.C; a.H = g.H; a.W = g.W; a.Co = g.Co; a.Ho = g.Ho; a.Wo = g.Wo; a.P = g.Ho * g.Wo;
    a.sh = g.sh; a.sw = g.sw; a.ph = g.ph; a.pw = g.pw; a.dh = g.dh; a.dw = g.dw;
    a.groups = g.groups; a.dg = g.dg; a.cpg = g.C / g.dg;
    a.KO = g.C * 9 / 8;
    a.Kg = g.C / g.groups * 9;
    a.Cog = g.Co / g.groups;
    a.tiles_m = (a.Cog + BM - 1) / BM;
    a.tiles_n = (a.P + BN - 1) / BN;
    gemm_operands_forward(a);

    // GEMM variant: LDS-DMA ring (default 4 stages; DDCN_BF16_GEMM_STAGES=0 selects the register-staged kernel).  The
    // DMA kernels address W and one (image, group) columns slab with 31-bit byte offsets.
    const char *es = getenv("DDCN_BF16_GEMM_STAGES");
    const int stages = es ? atoi(es) : 4;
    const long kpad = (long)(a.Kg + 31) / 32 * 32;
    const bool dma = stages >= 2 && (long)a.Cog * kpad * 2 < (1L << 31) && (long)(a.Kg / 8) * a.P * 16 < (1L << 31);

    // weights into the path's k order (a permutation of 16-bit words: the same kernel for both formats)
    bf16_t *wp = (bf16_t *)((char *)ws + col_bytes(g));
    if (dma) {
        const long wtotal = (long)g.Co * kpad;
        hipLaunchKernelGGL(repack_w_blocked_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, stream,
                           (const bf16_t *)weight, wp, wtotal, a.Kg, a.Cog);
    } else {
        const long wtotal = (long)g.Co * a.Kg;
        hipLaunchKernelGGL(repack_w_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, stream,
                           (const bf16_t *)weight, wp, wtotal, a.Kg);
    }
    if (int rc = check_launch("repack_w")) return rc;
    a.weight = wp;

    if (plan_bf16_tile(a, g)) {
        const dim3 grid((unsigned)(a.tiles * (g.C / a.CG)), 1, (unsigned)g.N);
        if (mask) run_bf16_tile<T, true>(a, grid, stream);
        else run_bf16_tile<T, false>(a, grid, stream);
    } else {
        const dim3 ig((unsigned)((a.P + IT - 1) / IT), (unsigned)(g.C / 8), (unsigned)g.N);
        if (mask) hipLaunchKernelGGL((im2col_bf16_kernel<T, true>), ig, dim3(IT), 0, stream, a);
        else hipLaunchKernelGGL((im2col_bf16_kernel<T, false>), ig, dim3(IT), 0, stream, a);
    }
    if (int rc = check_launch("im2col_16")) return rc;
    const long tiles = (long)a.tiles_m * a.tiles_n;
    if (tiles >= (1L << 31)) return fail(DDCN_ERR_UNSUPPORTED, "16-bit path: too many GEMM tiles");
    const dim3 ggrid((unsigned)tiles, (unsigned)(g.N * g.groups));
    const bool dma_ok = dma;
    // 256-row tiles (128x64 per wave) when Co fills them and they still give every CU two workgroups; DDCN_BF16_GEMM_MI forces 2 or 4
    const char *em = getenv("DDCN_BF16_GEMM_MI");
    const long tiles256 = (long)((a.Cog + 255) / 256) * a.tiles_n * g.N * g.groups;
    // 256x256 tiles (8 waves) when they still give every CU a workgroup; DDCN_BF16_GEMM_MI=8 forces them
    const long tiles_sq = (long)((a.Cog + 255) / 256) * ((a.P + 255) / 256) * g.N * g.groups;
    const int mi = em ? atoi(em) : ((a.Cog >= 192 && tiles_sq >= 256) ? 8 : (a.Cog >= 192 && tiles256 >= 512) ? 4 : 2);
    if (dma_ok && mi == 8 && stages >= 2) {
        const dim3 g8((unsigned)((a.Cog + 255) / 256 * ((a.P + 255) / 256)), (unsigned)(g.N * g.groups));
        if (int rc = launch_gemm16_sq<T>(g8, stream, a, "gemm_16")) return rc;   // 4 stages x 32 KiB
    } else if (dma_ok && mi == 4 && stages >= 2) {
        const dim3 g4((unsigned)((a.Cog + 255) / 256 * a.tiles_n), (unsigned)(g.N * g.groups));
        static std::atomic<uint64_t> raised{0};   // 3 stages x 24 KiB: above the 64 KiB default
        if (!raise_lds_limit((const void *)gemm16_dma_kernel<T, 3, 4>, 3 * 24576, raised))
            return fail(DDCN_ERR_LAUNCH, "gemm_16: cannot raise the LDS limit");
        hipLaunchKernelGGL((gemm16_dma_kernel<T, 3, 4>), g4, dim3(GT), 3 * 24576, stream, a);
    } else if (dma_ok && stages == 2) hipLaunchKernelGGL((gemm16_dma_kernel<T, 2, 2>), ggrid, dim3(GT), 2 * 16384, stream, a);
    else if (dma_ok && stages == 3) hipLaunchKernelGGL((gemm16_dma_kernel<T, 3, 2>), ggrid, dim3(GT), 3 * 16384, stream, a);
    else if (dma_ok && stages >= 4) hipLaunchKernelGGL((gemm16_dma_kernel<T, 4, 2>), ggrid, dim3(GT), 4 * 16384, stream, a);
    else hipLaunchKernelGGL(gemm_bf16_kernel<T>, ggrid, dim3(GT), 0, stream, a);
    return check_launch("gemm_16");
}

int launch_forward_bf16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                        void *out, void *ws, const ddcn_geom &g, hipStream_t stream)
{
    return forward_16<BF16>(x, weight, bias, offset, mask, out, ws, g, stream);
}

int launch_forward_f16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                       void *out, void *ws, const ddcn_geom &g, hipStream_t stream)
{
    return forward_16<F16>(x, weight, bias, offset, mask, out, ws, g, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// G2 of the 16-bit backward: gcol[n][g] (Kg x ld, fp32) = W[g]^T * grad_out[n][g] with 16-bit operands and fp32
// accumulation on the LDS-DMA GEMM above (the reference's half instantiation runs this product in half, H:305-308 /
// H:503-506 under AT_DISPATCH_FLOATING_TYPES_AND_HALF).  A = W^T in the blocked layout [group][Cog/32][Kg][32]
// (zero-padded in k), B = grad_out as channel octets [n][Co/8][P][8]; both are written into `ws` by two small kernels.
namespace {

// wp[((grp * KT + kt) * M + m) * 32 + k32] = W[(grp * Cog + kt*32 + k32) * M + m]   (M = Kg = C/groups * kh * kw)
__global__ __launch_bounds__(256) void repack_wT_blocked_kernel(const bf16_t *w, bf16_t *wp, long total, int Cog, int M)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int KT = (Cog + 31) / 32;
    const int k32 = (int)(i & 31);
    const long rest = i >> 5;
    const int m = (int)(rest % M);
    const long gk = rest / M;
    const int kt = (int)(gk % KT);
    const long grp = gk / KT;
    const int k = kt * 32 + k32;
    wp[i] = k < Cog ? w[(grp * Cog + k) * (long)M + m] : (bf16_t)0;
}

// go[((n * Co/8 + o8) * P + p) * 8 + e] = gout[(n * Co + o8*8 + e) * P + p]; one thread per (n, o8, p)
__global__ __launch_bounds__(256) void repack_gout_octets_kernel(const bf16_t *gout, u32x4 *go, int Co, int P)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int o8 = blockIdx.y, n = blockIdx.z;
    const bf16_t *src = gout + ((long)n * Co + o8 * 8) * P + p;
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[(long)e * P];
    u32x4 o;
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
    o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
    go[((long)n * (Co / 8) + o8) * P + p] = o;
}

}  // namespace

size_t gcol16_workspace_bytes(const ddcn_geom &g)
{
    const size_t Cog = g.Co / g.groups, M = (size_t)(g.C / g.groups) * g.kh * g.kw, KT = (Cog + 31) / 32;
    return round_up((size_t)g.groups * KT * M * 32 * 2, 256) + (size_t)g.N * g.Co * g.Ho * g.Wo * 2;
}

int gcol16_supported(const ddcn_geom &g)
{
    const long P = (long)g.Ho * g.Wo, Cog = g.Co / g.groups, M = (long)(g.C / g.groups) * g.kh * g.kw;
    if (Cog % 8) return fail(DDCN_ERR_UNSUPPORTED, "16-bit G2: Co/groups must be a multiple of 8 (Co=%d groups=%d)", g.Co, g.groups);
    const long kpad = (Cog + 31) / 32 * 32;
    if (M * kpad * 2 >= (1L << 31) || (Cog / 8) * P * 16 >= (1L << 31) || P >= (1L << 31) / 16 ||
        (long)g.N * g.groups > 65535 || g.Co / 8 > 65535 || g.N > 65535)
        return fail(DDCN_ERR_UNSUPPORTED, "16-bit G2: tensor too large");
    return DDCN_OK;
}

template <typename T>
static int gemm_gcol_16(const void *weight, const void *gout, float *gcol, void *ws, const ddcn_geom &g, hipStream_t stream)
{
    if (int rc = gcol16_supported(g)) return rc;
    const Dims d = dims_of(g);
    const int Cog = d.Cog, M = d.Kg, P = d.P, KT = (Cog + 31) / 32;   // M = C/groups * kh * kw: any kernel size
    bf16_t *wp = (bf16_t *)ws;
    u32x4 *go = (u32x4 *)((char *)ws + round_up((size_t)g.groups * KT * M * 32 * 2, 256));
    const long wtotal = (long)g.groups * KT * M * 32;
    hipLaunchKernelGGL(repack_wT_blocked_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t *)weight, wp, wtotal, Cog, M);
    hipLaunchKernelGGL(repack_gout_octets_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)(g.Co / 8), (unsigned)g.N),
                       dim3(256), 0, stream, (const bf16_t *)gout, go, g.Co, P);
    if (int rc = check_launch("repack_16")) return rc;
    Bf16Args a = {};
    a.weight = wp; a.col = (bf16_t *)go; a.bias = nullptr; a.out = nullptr;
    a.outf = gcol; a.ldc = d.ld;
    a.N = g.N; a.P = P; a.groups = g.groups;
    a.Co = g.groups * M;          // rows of one image's result (the kernel's `Co`: the C * kh * kw column rows)
    a.Cog = M;                    // rows per group
    a.Kg = Cog;                   // reduction length per group
    a.KO = g.Co / 8;              // k-octets of one image of B
    gemm_operands_forward(a);
    // tile class as in the forward: 256x256 tiles when they fill the chip, else 128x128 (4 stages)
    const long tiles_sq = (long)((M + 255) / 256) * ((P + 255) / 256) * g.N * g.groups;
    if (M >= 192 && tiles_sq >= 256) {
        const dim3 g8((unsigned)((M + 255) / 256 * ((P + 255) / 256)), (unsigned)(g.N * g.groups));
        if (int rc = launch_gemm16_sq<T>(g8, stream, a, "gemm_gcol_16")) return rc;
    } else {
        const dim3 gg((unsigned)((M + 127) / 128 * ((P + 127) / 128)), (unsigned)(g.N * g.groups));
        hipLaunchKernelGGL((gemm16_dma_kernel<T, 4, 2>), gg, dim3(GT), 4 * 16384, stream, a);
    }
    return check_launch("gemm_gcol_16");
}

int launch_gemm_gcol_bf16(const void *weight, const void *gout, float *gcol, void *ws, const ddcn_geom &g, hipStream_t stream)
{
    return gemm_gcol_16<BF16>(weight, gout, gcol, ws, g, stream);
}

int launch_gemm_gcol_f16(const void *weight, const void *gout, float *gcol, void *ws, const ddcn_geom &g, hipStream_t stream)
{
    return gemm_gcol_16<F16>(weight, gout, gcol, ws, g, stream);
}


// ---------------------------------------------------------------------------------------------------------------
// G3 of the 16-bit backward: grad_weight[g] (Cog x Kg) += sum_n grad_out[n][g] (Cog x P) * col[n][g]^T, contracted over
// the pixels (H:539-544 under the half instantiation).  The pixels are taken in the im2col kernel's own order -- (tile,
// lane), lanes outside the map being zero columns -- so k = tiles * 256 per image, always whole 32-k tiles:
//   A = grad_out gathered into [n][group][k/32][Cog][32] (zero outside the map)        gout_pixblock_kernel
//   B = columns as pixel octets [n][k/8][C*9 rows][8], rows (group, tap, channel)        im2col_bf16_tile_kernel, pixoct
//   C = fp32 partial sums added into gw'[Co][tap][Cg] (atomics: images x K splits), then gw[o][c][tap] += gw'[o][tap][c].
namespace {

// one thread per 8 consecutive k (= one pixel octet: 8 pixels of one map row): eight 2-byte loads, one 16-byte store
__global__ __launch_bounds__(256) void gout_pixblock_kernel(const bf16_t *gout, u32x4 *ap, long total8, int Co, int Cog,
                                                           int KT, int Ho, int Wo)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total8) return;
    const int k8 = (int)(i & 3);
    long rest = i >> 2;
    const int o = (int)(rest % Cog); rest /= Cog;
    const int kt = (int)(rest % KT); rest /= KT;
    const int groups = Co / Cog;
    const int grp = (int)(rest % groups);
    const long n = rest / groups;
    const int OW = (Wo + 7) >> 3;
    const int oct = kt * 4 + k8;                      // pixel octet (ho, wo / 8); beyond the map: zeros
    const int ho = oct / OW, wo0 = (oct - ho * OW) * 8;
    const bf16_t *src = gout + ((n * Co + (long)grp * Cog + o) * Ho + min(ho, Ho - 1)) * Wo;
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (ho < Ho && wo0 + e < Wo) ? src[wo0 + e] : (unsigned short)0;
    u32x4 r;
    r.x = v[0] | ((unsigned)v[1] << 16); r.y = v[2] | ((unsigned)v[3] << 16);
    r.z = v[4] | ((unsigned)v[5] << 16); r.w = v[6] | ((unsigned)v[7] << 16);
    ap[i] = r;
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float *p, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// gw[o][c][tap] += gwp[o][tap][c]   (o over all Co rows, c inside the row's group)
__global__ __launch_bounds__(256) void gweight_finish_kernel(const float *gwp, float *gw, long total, int Cg)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int tap = (int)(i % 9);
    const long oc = i / 9;
    const int c = (int)(oc % Cg);
    const long o = oc / Cg;
    gw[i] += gwp[(o * 9 + tap) * Cg + c];
}

// kpix: pixels of one image in the contraction (whole octets); kpad: the same rounded up to whole 32-k tiles (the A operand
// is zero-padded to it, the B operand's tail is cut by its descriptor)
struct Gw16Plan { size_t b_bytes, a_bytes, p_bytes; long kpix, kpad; };

Gw16Plan gw16_plan(const ddcn_geom &g)
{
    Gw16Plan pl;
    pl.kpix = (long)g.Ho * ((g.Wo + 7) / 8) * 8;
    pl.kpad = (pl.kpix + 31) / 32 * 32;
    pl.b_bytes = round_up((size_t)g.N * pl.kpix * g.C * 9 * 2, 256);
    pl.a_bytes = round_up((size_t)g.N * pl.kpad * g.Co * 2, 256);
    pl.p_bytes = round_up((size_t)g.Co * (g.C / g.groups) * 9 * 4, 256);
    return pl;
}

}  // namespace

int gweight16_supported(const ddcn_geom &g)
{
    if (int rc = bf16_supported(g)) return rc;
    const Gw16Plan pl = gw16_plan(g);
    const long Cog = g.Co / g.groups, Kg = (long)(g.C / g.groups) * 9;
    if (Cog % 8) return fail(DDCN_ERR_UNSUPPORTED, "16-bit G3: Co/groups must be a multiple of 8 (Co=%d groups=%d)", g.Co, g.groups);
    // 31-bit byte offsets inside one image's B slab and one (image, group) A slab
    if (pl.kpix / 8 * (long)g.C * 9 * 16 >= (1L << 31) || pl.kpad * Cog * 2 >= (1L << 31) || Kg >= (1L << 27))
        return fail(DDCN_ERR_UNSUPPORTED, "16-bit G3: tensor too large");
    Bf16Args a = {};
    a.pixoct = 1;
    if (!plan_bf16_tile(a, g)) return fail(DDCN_ERR_UNSUPPORTED, "16-bit G3: the im2col patch of this geometry does not fit the LDS tile kernel");
    return DDCN_OK;
}

size_t gweight16_workspace_bytes(const ddcn_geom &g)
{
    const Gw16Plan pl = gw16_plan(g);
    return pl.b_bytes + pl.a_bytes + pl.p_bytes;
}

template <typename T>
static int gweight_16(const void *x, const void *offset, const void *mask, const void *gout, float *gw, void *ws,
                      const ddcn_geom &g, hipStream_t stream)
{
    if (int rc = gweight16_supported(g)) return rc;
    const Gw16Plan pl = gw16_plan(g);
    const int Cog = g.Co / g.groups, Cg = g.C / g.groups, Kg = Cg * 9, P = g.Ho * g.Wo;
    bf16_t *colp = (bf16_t *)ws;
    bf16_t *ap = (bf16_t *)((char *)ws + pl.b_bytes);
    float *gwp = (float *)((char *)ws + pl.b_bytes + pl.a_bytes);

    // B: columns as pixel octets
    Bf16Args a = {};
    a.x = (const bf16_t *)x; a.offset = (const bf16_t *)offset; a.mask = (const bf16_t *)mask;
    a.col = colp;
    a.N = g.N; a.C = g.C; a.H = g.H; a.W = g.W; a.Co = g.Co; a.Ho = g.Ho; a.Wo = g.Wo; a.P = P;
    a.sh = g.sh; a.sw = g.sw; a.ph = g.ph; a.pw = g.pw; a.dh = g.dh; a.dw = g.dw;
    a.groups = g.groups; a.dg = g.dg; a.cpg = g.C / g.dg;
    a.KO = g.C * 9 / 8;
    a.Kg = Kg;
    a.pixoct = 1;
    if (!plan_bf16_tile(a, g)) return fail(DDCN_ERR_UNSUPPORTED, "16-bit G3: no LDS tile plan");
    {
        const dim3 grid((unsigned)(a.tiles * (g.C / a.CG)), 1, (unsigned)g.N);
        if (mask) run_bf16_tile<T, true>(a, grid, stream);
        else run_bf16_tile<T, false>(a, grid, stream);
        if (int rc = check_launch("im2col_16_pixoct")) return rc;
    }
    // A: grad_out in the same pixel order, blocked; the partial result starts at zero
    const int KT = (int)(pl.kpad / 32);
    const long atotal = (long)g.N * g.Co * pl.kpad;
    hipLaunchKernelGGL(gout_pixblock_kernel, dim3((unsigned)((atotal / 8 + 255) / 256)), dim3(256), 0, stream,
                       (const bf16_t *)gout, (u32x4 *)ap, atotal / 8, g.Co, Cog, KT, g.Ho, g.Wo);
    const long wtotal = (long)g.Co * Kg;
    hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, stream, gwp, wtotal);
    if (int rc = check_launch("gout_pixblock")) return rc;

    Bf16Args b = {};
    b.weight = ap; b.col = colp; b.outf = gwp; b.ldc = Kg; b.out_atomic = 1;
    b.N = g.N; b.groups = g.groups;
    b.P = Kg;                      // columns of the product: the rows of one group's columns
    b.Cog = Cog; b.Co = g.Co;      // rows: the group's output channels
    b.Kg = (int)pl.kpix;           // contraction: the pixels of one image (whole octets; the last 32-k tile may be partial)
    b.sA_img = (long)g.Co * pl.kpad; b.sA_grp = (long)Cog * pl.kpad;
    b.ldb = g.C * 9;
    b.sB_img = (pl.kpix / 8) * (long)b.ldb; b.sB_grp = Kg;
    // K splits: enough workgroups for the chip
    const bool sq = Cog >= 192;    // 256 x 256 tiles
    const long tiles = sq ? (long)((Cog + 255) / 256) * ((Kg + 255) / 256) : (long)((Cog + 127) / 128) * ((Kg + 127) / 128);
    long splits = std::max<long>(1, (sq ? 256 : 512) / std::max<long>(1, tiles * g.N * g.groups));
    splits = std::min<long>(splits, std::max(1, KT / 16));
    b.kt_per = (int)((KT + splits - 1) / splits);
    const unsigned nz = (unsigned)((KT + b.kt_per - 1) / b.kt_per);
    if (sq) {
        const dim3 g8((unsigned)tiles, (unsigned)(g.N * g.groups), nz);
        if (int rc = launch_gemm16_sq<T>(g8, stream, b, "gweight_16")) return rc;
    } else {
        const dim3 gg((unsigned)tiles, (unsigned)(g.N * g.groups), nz);
        hipLaunchKernelGGL((gemm16_dma_kernel<T, 4, 2>), gg, dim3(GT), 4 * 16384, stream, b);
    }
    if (int rc = check_launch("gemm_gweight_16")) return rc;
    hipLaunchKernelGGL(gweight_finish_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, stream, gwp, gw, wtotal, Cg);
    return check_launch("gweight_finish");
}

int launch_gweight_bf16(const void *x, const void *offset, const void *mask, const void *gout, float *gw, void *ws,
                        const ddcn_geom &g, hipStream_t stream)
{
    return gweight_16<BF16>(x, offset, mask, gout, gw, ws, g, stream);
}

int launch_gweight_f16(const void *x, const void *offset, const void *mask, const void *gout, float *gw, void *ws,
                       const ddcn_geom &g, hipStream_t stream)
{
    return gweight_16<F16>(x, offset, mask, gout, gw, ws, g, stream);
}

}  // namespace ddcn

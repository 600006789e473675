// Tiled backward scatter for 3x3 kernels: fused col2im + col2im_coord, everything staged in LDS (gfx950).
//
// Arithmetic: detseg/ops/dcn/src/deform_conv_cuda_kernel.cu K:278-334 / K:634-692 (col2im) and K:372-435 /
// K:694-766 (col2im_coord), helpers K:116-187.  Formula-by-formula citations are in ddcn_gather.hip.
//
// workgroup = (TH x TW tile of output pixels, group of CG input channels, image n); thread = output pixel.
//   registers : per tap the patch cell, the fractional parts and the mask of the pixel's sample, and four
//               running sums S_i = sum_c gcol[c,k,p] * x[c, corner_i].  grad_offset and grad_mask are linear
//               in the S_i (K:163-184, K:752), so they are formed once at the end, not per channel.
//   LDS       : xs[cell][4]  (float)  input patch of the tile for 4 channels, zero outside the image;
//               gs[cell][4]  (double) grad_input footprint for the same 4 channels.
//               ds_add_f64 because ds_add_f32 is ~20x slower on gfx950 (tools/ubench_lds_atomic.hip).
//   per chunk : 36 column loads (scalar row offsets) + the next patch in flight, then per tap
//               4 ds_read_b128 + 16 FMA + 16 ds_add_f64, no branch per corner: corners outside the image are
//               patch cells that hold x = 0 and whose gradient is simply never flushed (== the reference's
//               guards K:97-108, K:322-325).  Then the touched cells are flushed with one global atomic each.
//   escapees  : samples whose corners leave the patch are handled after the channel loop, from global memory.
#include <stdlib.h>

#include <algorithm>

#include "ddcn_common.h"

namespace ddcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int ST = 256;
#ifndef DDCN_S9_WPS
#define DDCN_S9_WPS 3
#endif
constexpr int SMAXCELLS = 8;
constexpr int SMAXCELLS_WIDE = 14;       // wide-offset mode (2-channel chunks, exact fixed point): up to 57 x 57 cells
constexpr unsigned SOOB = 0x80000000u;
constexpr int S9_LDS_MAX = 80 * 1024;   // planner's cap on the dynamic LDS of one workgroup (two per CU)

struct Tile9Args {
    const float *gcol, *x, *offset, *mask;
    float *gx, *goff, *gmask;
    ddcn_geom g;
    int P, cpg;
    long ld;
    int TH, TW, tiles_w, tiles, groups;
    int CG, M;
    int RH, RW, RS;
    int esc_cells;       // adaptive kernel: cost of one escaped sample in patch cells
    int Mw_max;          // adaptive kernel: largest margin the wide (2-channel) footprint can take; 0 = never wide
    // SKB mode (few output channels, no column gradients in memory): gcol[c,k,p] = sum_o sk_w[c][k][o] * sk_g[n][o][p]
    const float *sk_w;   // weights repacked [C][9][SKB] (zero beyond sk_co)
    const float *sk_g;   // gradient w.r.t. the convolution output [N][sk_co][P]
    int sk_co;
};

constexpr int SKB_MAX = 12;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t s9_rsrc(const void *base, long bytes)
{
    const unsigned n = bytes <= 0 ? 0u : (bytes > 0x7fffffffL ? 0x7fffffffu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, n, 0x00020000);
}

// FX (the default for 2- and 4-channel chunks): the footprint is kept as 32-bit fixed point, two channels per 64-bit LDS
// word, and accumulated with ds_add_u64 (19 cycles per wave instruction for two channels against 2 x 19 for the exact
// 64-bit fixed point XQ and 2 x 23.5 for ds_add_f64): half the LDS atomics, which are what bounds this kernel.
//   scale     per chunk the workgroup takes the largest |column value x mask| it is about to add, gmax < 2^e.
//   headroom  a field must hold every sum a cell can collect.  The workgroup counts, ONCE, how many of its samples have
//             their (hl, wl) corner in each patch cell (one ds_add_u32 per sample); a cell is touched by the samples
//             based at itself and at its left / upper / upper-left neighbour, and each adds at most gmax (a corner
//             weight is <= 1), so with cmax = the largest such 4-cell count, cmax < 2^hb, contributions scaled by
//             2^(sh - e), sh = 30 - hb, keep every field below cmax * (2^sh + 1/2) < 2^31: no overflow for ANY offset
//             field (all 2304 samples of a tile on one cell: sh = 18), and 2^-24 ... 2^-26 of the chunk's largest value
//             per add for ordinary ones (cmax 16 ... 64) -- the resolution of an fp32 atomic add.
//   packing   adding (hi << 32) + lo as ONE signed 64-bit number is exact for both fields (the borrow of a negative low
//             part is undone when the word is decoded as lo = int32(word), hi = (word - lo) >> 32).
// Integer sums do not depend on the order of the adds: the LDS part of the scatter is deterministic.
// A non-finite column value cannot be represented: the chunk then writes NaN to every cell it could have touched.
// SKB > 0: the column gradients are never in memory.  For a deformable convolution with few (<= SKB) output channels
// -- the DDCN layer's 9-channel mask branch -- gcol = W^T * grad_out is a 9-term dot product per element: each lane keeps
// its pixel's SKB output gradients in registers, the chunk's weights ([4 channels][9 taps][SKB], repacked by the host
// side) are staged in LDS next to the patch and read as broadcast ds_read_b128, and a tap's four column values are
// formed right before they are scattered (the reference: a 9-row GEMM into an 86.7 MB columns tensor, H:305-308).
// XQ: the footprint as EXACT 64-bit fixed point, one channel per word, accumulated with ds_add_u64 (19 cycles per wave
// instruction against 23-24 for ds_add_f64).  Scale per chunk and workgroup 2^(50 - e) with gmax < 2^e as for FX: a
// contribution is below 2^50, 2304 of them below 2^62.  A float times a power of two is exact in double, and
// fma(v, scale, 1.5 * 2^52) leaves round(v * scale) in the low mantissa bits (|v * scale| < 2^51), so the only rounding is
// to 2^-50 of the chunk's largest value -- 2^-26 of an fp32 ulp of it.  Integer sums do not depend on the order of the
// adds: the LDS part of the scatter is deterministic in this mode.
// Patch geometry of one workgroup: margin and rows x columns of the LDS footprint.  The plain kernels take the planner's
// (the same for every workgroup); the adaptive kernel picks one per workgroup from its own offsets.
struct PatchGeom { int M, RH, RW, RS; };

template <bool MOD, int NCELL, int SC, bool FX, int SKB = 0, bool XQ = false>
__device__ __forceinline__ void tile9_body(const Tile9Args &a, const PatchGeom pg)
{
    static_assert(!XQ || (!FX && SKB == 0), "exact fixed point: its own mode");
    constexpr bool MAXQ = FX || XQ;        // modes that need the chunk's largest |value|
    static_assert(!FX || SC == 4 || SC == 2, "fixed-point footprint packs channel pairs");
    static_assert(SKB == 0 || (SC == 4 && !FX && !MOD && SKB % 4 == 0), "direct mode: plain op, 4-channel chunks");
    typedef float xvec __attribute__((ext_vector_type(SC)));
    typedef unsigned long long u64;
    constexpr int K2 = 9;
    constexpr int GB = FX ? 4 : 8;                                           // footprint bytes per channel and cell
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ unsigned s_max[2];                                            // FX: bit pattern of gmax, per chunk parity
    __shared__ unsigned s_cmax;                                              // FX: largest 4-cell sample count of the patch
    double *gs = reinterpret_cast<double *>(smem_raw);                       // [SC][RS]: a wave's 64 cells -> 128 banks
    u64 *gq = reinterpret_cast<u64 *>(smem_raw);                             // FX: [SC/2][RS], channels (2j, 2j+1)
    xvec *xs = reinterpret_cast<xvec *>(smem_raw + (size_t)pg.RS * SC * GB);   // [RS][SC]
    float *wl = reinterpret_cast<float *>(smem_raw + (size_t)pg.RS * SC * (GB + 4));   // SKB: [SC][9][SKB] weights of the chunk
    const ddcn_geom &g = a.g;
    const int t = threadIdx.x;

    const int nwg = a.tiles * a.groups;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int grp = logical / a.tiles, tile = logical - grp * a.tiles;

    const int tyi = tile / a.tiles_w, txi = tile - tyi * a.tiles_w;
    const int ty = t / a.TW, tx = t - ty * a.TW;
    const int ho = tyi * a.TH + ty, wo = txi * a.TW + tx;
    const bool live = ty < a.TH && ho < g.Ho && wo < g.Wo;
    const int rh0 = tyi * a.TH * g.sh - g.ph - pg.M;
    const int rw0 = txi * a.TW * g.sw - g.pw - pg.M;
    const int n = blockIdx.z;
    const int cg0 = grp * a.CG;
    const int dgi = cg0 / a.cpg;
    const int HW = g.H * g.W;
    const int p = live ? ho * g.Wo + wo : 0;

    const __amdgpu_buffer_rsrc_t rx = s9_rsrc(a.x + (long)n * g.C * HW, (long)g.C * HW * 4);
    const __amdgpu_buffer_rsrc_t rgc = s9_rsrc(a.gcol + (long)n * g.C * K2 * a.ld, SKB ? 0 : (long)g.C * K2 * a.ld * 4);
    float *gxn = a.gx + (long)n * g.C * HW;

    // patch cell i = t + 256*j -> byte offset inside one channel plane (out of image: rejected by the descriptor)
    unsigned soff[NCELL];
#pragma unroll
    for (int j = 0; j < NCELL; ++j) {
        const int i = t + ST * j;
        const int rr = i / pg.RW, rc = i - rr * pg.RW;
        const int y = rh0 + rr, xx = rw0 + rc;
        soff[j] = (i < pg.RS && y >= 0 && y < g.H && xx >= 0 && xx < g.W) ? (unsigned)((y * g.W + xx) * 4) : SOOB;
    }

    // per-tap sample state and running sums
    int cell[K2];      // >= 0: patch cell of corner (hl, wl); -1: escapee; -2: sample outside (-1,H)x(-1,W)
    float flh[K2], flw[K2], fm[K2];
    float S1[K2], S2[K2], S3[K2], S4[K2];
    int ghw[K2];
    {
        const float *off = a.offset + ((long)n * g.dg + dgi) * 2 * K2 * a.P + p;
        const float *msk = MOD ? a.mask + ((long)n * g.dg + dgi) * K2 * a.P + p : nullptr;
        const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
        // all 27 values in flight before the first use (p is 0 for a dead lane: the loads are unconditional; a load
        // under `if (live)` becomes a branch and a wait per tap)
        float roh[K2], row_[K2], rm[K2];
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            roh[k] = off[(long)(2 * k) * a.P];
            row_[k] = off[(long)(2 * k + 1) * a.P];
            rm[k] = MOD ? msk[(long)k * a.P] : 1.f;
        }
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            const int i = k / 3, j = k - 3 * i;
            float h = -2.f, w = -2.f, m = 1.f;
            if (live) {
                h = (float)(h_in + i * g.dh) + roh[k];
                w = (float)(w_in + j * g.dw) + row_[k];
                if (MOD) m = rm[k];
            }
            const float fhl = floorf(h), fwl = floorf(w);
            flh[k] = h - fhl; flw[k] = w - fwl; fm[k] = m;
            const int hl = (int)fhl, wl = (int)fwl;
            ghw[k] = (hl << 16) | (wl & 0xffff);
            int c = -2;
            if (live && h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W) {
                const int rr = hl - rh0, rc = wl - rw0;
                c = (rr >= 0 && rr + 1 < pg.RH && rc >= 0 && rc + 1 < pg.RW) ? rr * pg.RW + rc : -1;
            }
            cell[k] = c;
            S1[k] = S2[k] = S3[k] = S4[k] = 0.f;
        }
    }

    float stg[NCELL][SC];
    auto stage_load = [&](int c0) {
#pragma unroll
        for (int j = 0; j < NCELL; ++j)
#pragma unroll
            for (int cc = 0; cc < SC; ++cc)
                if (NCELL <= 8 || ST * j < pg.RS)          // wide instantiations: skip the slots this patch does not use
                    stg[j][cc] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(rx, soff[j], (unsigned)((c0 + cc) * HW * 4), 0));
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int j = 0; j < NCELL; ++j) {
            const int i = t + ST * j;
            if (i < pg.RS) {
                xvec v;
#pragma unroll
                for (int cc = 0; cc < SC; ++cc) v[cc] = stg[j][cc];
                xs[i] = v;
            }
        }
    };

    constexpr int WCH = SKB ? SC * K2 * SKB : 1;              // weights per chunk
    constexpr int WPT = SKB ? (WCH + ST - 1) / ST : 1;        // ... per thread
    float wreg[WPT];
    float gp[SKB ? SKB : 1];
    auto wstage_load = [&](int c0) {
        if (SKB) {
#pragma unroll
            for (int r = 0; r < WPT; ++r) {
                const int i = t + ST * r;
                wreg[r] = i < WCH ? a.sk_w[(long)c0 * K2 * SKB + i] : 0.f;
            }
        }
    };
    auto wstage_store = [&]() {
        if (SKB) {
#pragma unroll
            for (int r = 0; r < WPT; ++r)
                if (t + ST * r < WCH) wl[t + ST * r] = wreg[r];
        }
    };
    if (SKB) {
#pragma unroll
        for (int o = 0; o < (SKB ? SKB : 1); ++o)
            gp[o] = (live && o < a.sk_co) ? a.sk_g[((long)n * a.sk_co + o) * a.P + p] : 0.f;
    }
    // column value (channel cc of the chunk, tap k) in direct mode
    auto sk_col = [&](int k, int cc) -> float {
        float v = 0.f;
        if (SKB) {
            const f32x4 *wv = reinterpret_cast<const f32x4 *>(wl + (cc * K2 + k) * SKB);
#pragma unroll
            for (int q4 = 0; q4 < SKB / 4; ++q4) {
                const f32x4 w4 = wv[q4];
                v += w4.x * gp[4 * q4] + w4.y * gp[4 * q4 + 1] + w4.z * gp[4 * q4 + 2] + w4.w * gp[4 * q4 + 3];
            }
        }
        return v;
    };

    int fx_sh = 19;                 // FX: contributions are scaled to below 2^fx_sh
    if (FX) {
        // headroom from the workgroup's own samples (see the FX note above); the counters borrow the footprint's memory
        unsigned *cnt32 = reinterpret_cast<unsigned *>(smem_raw);
        for (int i = t; i < pg.RS; i += ST) cnt32[i] = 0u;
        if (t == 0) s_cmax = 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K2; ++k)
            if (cell[k] >= 0) atomicAdd(&cnt32[cell[k]], 1u);
        __syncthreads();
        unsigned cm = 0u;
#pragma unroll
        for (int j = 0; j < NCELL; ++j) {
            const int i = t + ST * j;
            if (i < pg.RS) {
                const int rr = i / pg.RW, rc = i - rr * pg.RW;
                unsigned c4 = cnt32[i];
                if (rc > 0) c4 += cnt32[i - 1];
                if (rr > 0) c4 += cnt32[i - pg.RW];
                if (rr > 0 && rc > 0) c4 += cnt32[i - pg.RW - 1];
                cm = max(cm, c4);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cm = max(cm, (unsigned)__shfl_xor((int)cm, o));
        if ((t & 63) == 0 && cm) atomicMax(&s_cmax, cm);
        __syncthreads();
        fx_sh = 30 - (32 - __clz((int)s_cmax));          // cmax < 2^hb, hb = 32 - clz(cmax); cmax <= 2304 -> fx_sh >= 18
        __syncthreads();                                  // the counters are dead: the footprint takes their place
        for (int i = t; i < pg.RS * SC / 2; i += ST) gq[i] = 0ull;
        if (t < 2) s_max[t] = 0u;
    } else if (XQ) {
        for (int i = t; i < pg.RS * SC; i += ST) gq[i] = 0ull;
        if (t < 2) s_max[t] = 0u;
    } else {
        for (int i = t; i < pg.RS * SC; i += ST) gs[i] = 0.0;
    }
    stage_load(cg0);
    wstage_load(cg0);
    stage_store();
    wstage_store();

    const unsigned pvo = (live && !SKB) ? (unsigned)(p * 4) : SOOB;   // direct mode: every column load is rejected
    const int nchunks = a.CG / SC;
    // column values of the current chunk; each tap reloads its registers for the next chunk as soon as it is done
    float gcv[K2][SC];
#pragma unroll
    for (int k = 0; k < K2; ++k)
#pragma unroll
        for (int cc = 0; cc < SC; ++cc)
            gcv[k][cc] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rgc, pvo, (unsigned)(((cg0 + cc) * K2 + k) * a.ld * 4), 0));
    // FX: this thread's largest |value it will add| of the chunk held in gcv -> the workgroup's maximum
    auto publish_max = [&](int parity) {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < K2; ++k)
            if (cell[k] >= 0) {
#pragma unroll
                for (int cc = 0; cc < SC; ++cc) {
                    const float v = fabsf(MOD ? gcv[k][cc] * fm[k] : gcv[k][cc]);
                    m = (v > m || v != v) ? v : m;                       // NaN wins: it has to reach the flush
                }
            }
        // non-negative floats (and +inf, NaN) order like their bit patterns
        if (m != 0.f) atomicMax(&s_max[parity], __float_as_uint(m) & 0x7fffffffu);
    };
    if (MAXQ) {
        __syncthreads();            // s_max is zero before anyone publishes
        publish_max(0);
    }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = cg0 + ch * SC;
        const bool more = ch + 1 < nchunks;
        const unsigned pvn = more ? pvo : SOOB;
        int fx_e = 0;               // FX: contributions are scaled by 2^(fx_sh - fx_e)
        bool fx_bad = false;
        if (MAXQ) {
            const float gmax = __uint_as_float(s_max[ch & 1]);
            fx_bad = !(gmax <= 3.0e38f);                                  // inf or NaN in the chunk
            (void)frexpf(fx_bad ? 1.f : gmax, &fx_e);                     // gmax < 2^fx_e (gmax == 0: nothing is added)
        }
        const double xq_scale = XQ ? __builtin_ldexp(1.0, 50 - fx_e) : 0.0;
        // A. the next patch goes in flight
        if (more) { stage_load(c0 + SC); wstage_load(c0 + SC); }
        __builtin_amdgcn_sched_barrier(0);

        // B. taps: gather x, accumulate the S sums, scatter into the LDS footprint
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            if (cell[k] >= 0) {
                int ci = cell[k];
                float lh = flh[k], lw = flw[k];
                // keep the per-tap weights and addresses out of the loop-invariant hoist (they would cost 60 VGPRs)
                asm volatile("" : "+v"(ci), "+v"(lh), "+v"(lw));
                const xvec a1 = xs[ci], a2 = xs[ci + 1], a3 = xs[ci + pg.RW], a4 = xs[ci + pg.RW + 1];
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;   // == K:133-140 weights
                if (FX) {
                    float tgs[SC];
#pragma unroll
                    for (int cc = 0; cc < SC; ++cc) {
                        const float gc = gcv[k][cc];
                        S1[k] += gc * a1[cc]; S2[k] += gc * a2[cc]; S3[k] += gc * a3[cc]; S4[k] += gc * a4[cc];
                        tgs[cc] = ldexpf(MOD ? gc * fm[k] : gc, fx_sh - fx_e);
                    }
                    if (!fx_bad) {
                        u64 *q1 = gq + ci, *q3 = gq + ci + pg.RW;
#pragma unroll
                        for (int pr = 0; pr < SC / 2; ++pr) {
                            const float lo = tgs[2 * pr], hi = tgs[2 * pr + 1];
                            auto pack = [](float wl_, float wh_) -> u64 {
                                const long long l = (long long)__float2int_rn(wl_), h = (long long)__float2int_rn(wh_);
                                return (u64)((h << 32) + l);
                            };
                            atomicAdd(q1 + pr * pg.RS, pack(w1 * lo, w1 * hi));
                            atomicAdd(q1 + pr * pg.RS + 1, pack(w2 * lo, w2 * hi));
                            atomicAdd(q3 + pr * pg.RS, pack(w3 * lo, w3 * hi));
                            atomicAdd(q3 + pr * pg.RS + 1, pack(w4 * lo, w4 * hi));
                        }
                    }
                } else if (XQ) {
                    u64 *q1 = gq + ci, *q3 = gq + ci + pg.RW;
                    auto fixed = [&](float v) -> u64 {
                        const double r = __builtin_fma((double)v, xq_scale, 6755399441055744.0);   // + 1.5 * 2^52
                        return __builtin_bit_cast(u64, r) - 0x4338000000000000ull;
                    };
#pragma unroll
                    for (int cc = 0; cc < SC; ++cc) {
                        const float gc = gcv[k][cc];
                        S1[k] += gc * a1[cc]; S2[k] += gc * a2[cc]; S3[k] += gc * a3[cc]; S4[k] += gc * a4[cc];
                        const float tg = MOD ? gc * fm[k] : gc;
                        if (!fx_bad) {
                            atomicAdd(q1 + cc * pg.RS, fixed(w1 * tg));
                            atomicAdd(q1 + cc * pg.RS + 1, fixed(w2 * tg));
                            atomicAdd(q3 + cc * pg.RS, fixed(w3 * tg));
                            atomicAdd(q3 + cc * pg.RS + 1, fixed(w4 * tg));
                        }
                    }
                } else {
                    double *c1 = gs + ci, *c3 = gs + ci + pg.RW;
#pragma unroll
                    for (int cc = 0; cc < SC; ++cc) {
                        const float gc = SKB ? sk_col(k, cc) : gcv[k][cc];
                        S1[k] += gc * a1[cc]; S2[k] += gc * a2[cc]; S3[k] += gc * a3[cc]; S4[k] += gc * a4[cc];
                        const float tg = MOD ? gc * fm[k] : gc;
                        atomicAdd(c1 + cc * pg.RS, (double)(w1 * tg));
                        atomicAdd(c1 + cc * pg.RS + 1, (double)(w2 * tg));
                        atomicAdd(c3 + cc * pg.RS, (double)(w3 * tg));
                        atomicAdd(c3 + cc * pg.RS + 1, (double)(w4 * tg));
                    }
                }
            }
            if (!SKB) {
#pragma unroll
                for (int cc = 0; cc < SC; ++cc)
                    gcv[k][cc] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(rgc, pvn, (unsigned)(((c0 + SC + cc) * K2 + k) * a.ld * 4), 0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MAXQ && more) publish_max((ch + 1) & 1);   // gcv now holds the next chunk
        __syncthreads();

        // C. flush the footprint (one global atomic per touched in-image cell), re-zero, install the next patch
        if (FX) {
            if (t == 0) s_max[ch & 1] = 0u;          // everyone has read it; it is published again two chunks on
            const float inv = ldexpf(1.f, fx_e - fx_sh);
#pragma unroll
            for (int j = 0; j < NCELL; ++j) {
                const int i = t + ST * j;
                if (i < pg.RS) {
                    u64 wq[SC / 2 ? SC / 2 : 1];
                    bool any = fx_bad;
#pragma unroll
                    for (int pr = 0; pr < SC / 2; ++pr) { wq[pr] = gq[i + pr * pg.RS]; any |= wq[pr] != 0ull; }
                    if (any) {
                        float *dst = gxn + (long)c0 * HW + (soff[j] >> 2);
#pragma unroll
                        for (int pr = 0; pr < SC / 2; ++pr) {
                            gq[i + pr * pg.RS] = 0ull;
                            const int lo = (int)(unsigned)wq[pr];
                            const int hi = (int)(((long long)wq[pr] - (long long)lo) >> 32);
                            if (soff[j] != SOOB) {
                                if (fx_bad) {
                                    atomicAdd(dst + (2 * pr) * HW, __uint_as_float(0x7fc00000u));
                                    atomicAdd(dst + (2 * pr + 1) * HW, __uint_as_float(0x7fc00000u));
                                } else {
                                    if (lo != 0) atomicAdd(dst + (2 * pr) * HW, (float)lo * inv);
                                    if (hi != 0) atomicAdd(dst + (2 * pr + 1) * HW, (float)hi * inv);
                                }
                            }
                        }
                    }
                }
            }
        } else if (XQ) {
            if (t == 0) s_max[ch & 1] = 0u;
            const double inv = __builtin_ldexp(1.0, fx_e - 50);
#pragma unroll
            for (int j = 0; j < NCELL; ++j) {
                const int i = t + ST * j;
                if (i < pg.RS) {
                    u64 wq[SC];
                    bool any = fx_bad;
#pragma unroll
                    for (int cc = 0; cc < SC; ++cc) { wq[cc] = gq[i + cc * pg.RS]; any |= wq[cc] != 0ull; }
                    if (any) {
                        float *dst = gxn + (long)c0 * HW + (soff[j] >> 2);
#pragma unroll
                        for (int cc = 0; cc < SC; ++cc) {
                            gq[i + cc * pg.RS] = 0ull;
                            if (soff[j] != SOOB) {
                                if (fx_bad) atomicAdd(dst + cc * HW, __uint_as_float(0x7fc00000u));
                                else if (wq[cc] != 0ull) atomicAdd(dst + cc * HW, (float)((double)(long long)wq[cc] * inv));
                            }
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NCELL; ++j) {
                const int i = t + ST * j;
                if (i < pg.RS) {
                    double v[SC];
                    bool any = false;
#pragma unroll
                    for (int cc = 0; cc < SC; ++cc) { v[cc] = gs[i + cc * pg.RS]; any |= v[cc] != 0.0; }
                    if (any) {
                        float *dst = gxn + (long)c0 * HW + (soff[j] >> 2);
#pragma unroll
                        for (int cc = 0; cc < SC; ++cc) {
                            gs[i + cc * pg.RS] = 0.0;
                            if (soff[j] != SOOB && v[cc] != 0.0) atomicAdd(dst + cc * HW, (float)v[cc]);
                        }
                    }
                }
            }
        }
        if (more) { stage_store(); wstage_store(); }
        __syncthreads();
    }

    if (!live) return;

    // escapees: straight to global memory, whole channel group, same sums
#pragma unroll
    for (int k = 0; k < K2; ++k) {
        if (cell[k] == -1) {
            const int hl = ghw[k] >> 16, wl = (int)(short)(ghw[k] & 0xffff);
            const bool ok1 = hl >= 0 && wl >= 0, ok2 = hl >= 0 && wl + 1 <= g.W - 1;
            const bool ok3 = hl + 1 <= g.H - 1 && wl >= 0, ok4 = hl + 1 <= g.H - 1 && wl + 1 <= g.W - 1;
            const int gi = hl * g.W + wl;
            const float lh = flh[k], lw = flw[k];
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
            // Corner offsets inside a channel plane, 0 for a corner outside the map: every load below is
            // unconditional and four channels' loads are issued before the first use (a load under a condition
            // becomes a branch with its own wait: five memory latencies in a row per channel).
            const int o1 = ok1 ? gi : 0, o2 = ok2 ? gi + 1 : 0, o3 = ok3 ? gi + g.W : 0, o4 = ok4 ? gi + g.W + 1 : 0;
            const float *im = a.x + ((long)n * g.C + cg0) * HW;
            float *gim = gxn + (long)cg0 * HW;
            const float *gcp = a.gcol + (((long)n * g.C + cg0) * K2 + k) * a.ld + p;
            for (int c0 = 0; c0 < a.CG; c0 += 4) {
                float gc[4], r1[4], r2[4], r3[4], r4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = min(c0 + u, a.CG - 1);
                    if (SKB) {
                        const float *wr = a.sk_w + ((long)(cg0 + c) * K2 + k) * SKB;
                        float v = 0.f;
#pragma unroll
                        for (int o = 0; o < (SKB ? SKB : 1); ++o) v += wr[o] * gp[o];
                        gc[u] = v;
                    } else {
                        gc[u] = gcp[(long)c * K2 * a.ld];
                    }
                    r1[u] = im[(long)c * HW + o1]; r2[u] = im[(long)c * HW + o2];
                    r3[u] = im[(long)c * HW + o3]; r4[u] = im[(long)c * HW + o4];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c0 + u < a.CG) {
                        const float v1 = ok1 ? r1[u] : 0.f, v2 = ok2 ? r2[u] : 0.f;
                        const float v3 = ok3 ? r3[u] : 0.f, v4 = ok4 ? r4[u] : 0.f;
                        S1[k] += gc[u] * v1; S2[k] += gc[u] * v2; S3[k] += gc[u] * v3; S4[k] += gc[u] * v4;
                        const float tg = MOD ? gc[u] * fm[k] : gc[u];
                        float *gq = gim + (long)(c0 + u) * HW;
                        if (ok1) atomicAdd(gq + o1, w1 * tg);
                        if (ok2) atomicAdd(gq + o2, w2 * tg);
                        if (ok3) atomicAdd(gq + o3, w3 * tg);
                        if (ok4) atomicAdd(gq + o4, w4 * tg);
                    }
                }
            }
        }
    }

    // grad_offset / grad_mask from the sums (K:163-184 coordinate weights, K:752 mask gradient)
    float *goffb = a.goff + ((long)n * g.dg + dgi) * 2 * K2 * a.P + p;
    float *gmaskb = MOD ? a.gmask + ((long)n * g.dg + dgi) * K2 * a.P + p : nullptr;
#pragma unroll
    for (int k = 0; k < K2; ++k) {
        if (cell[k] != -2) {
            const float lh = flh[k], lw = flw[k];
            const float hh = 1.f - lh, hw = 1.f - lw;
            float gh = (-1.f * hw * S1[k]) + (-1.f * lw * S2[k]) + hw * S3[k] + lw * S4[k];
            float gw = (-1.f * hh * S1[k]) + hh * S2[k] + (-1.f * lh * S3[k]) + lh * S4[k];
            if (MOD) { gh *= fm[k]; gw *= fm[k]; }
            atomicAdd(goffb + (long)(2 * k) * a.P, gh);
            atomicAdd(goffb + (long)(2 * k + 1) * a.P, gw);
            if (MOD) atomicAdd(gmaskb + (long)k * a.P, hh * hw * S1[k] + hh * lw * S2[k] + lh * hw * S3[k] + lh * lw * S4[k]);
        }
    }
}

template <bool MOD, int NCELL, int SC, bool FX, int SKB = 0, bool XQ = false>
__global__ __launch_bounds__(ST, (SC == 4 || NCELL > 8) ? 2 : DDCN_S9_WPS) void col2im_coord_tile9_kernel(Tile9Args a)
{
    tile9_body<MOD, NCELL, SC, FX, SKB, XQ>(a, PatchGeom{a.M, a.RH, a.RW, a.RS});
}

// Adaptive footprint: every workgroup looks at its own 256 x 9 sample positions first and takes
//   - the planner's patch (margin a.M, 4-channel chunks) when next to none of them leave it, else
//   - a wider patch with 2-channel chunks in the same LDS allocation, margin a.M+2 ... a.Mw_max, the one that minimises
//     (cells of the patch) + ESC_CELLS x (samples still outside): a sample outside the footprint goes through the escapee
//     pass, which costs about as much as ESC_CELLS more cells of patch traffic per channel (profiles/r02c_*).
// Offsets with sigma = 8 px (the reference's literal Kaiming offset init gives 6) leave the margin-6 patch for most samples
// and took the scatter from 0.34 to 1.96 ms; with per-workgroup margins up to 16 it is 0.69 ms.
constexpr int ESC_CELLS = 120;
constexpr int NCELL_WIDE = 14;
constexpr int ADAPT_STEPS = 8;         // candidate margins a.M + 2 * j

// SCN / SCW: channels per chunk of the planner's and of the wide footprint: 4 / 2 in an 80 KiB allocation (two workgroups
// per CU), 2 / 1 in 40 KiB and 168 VGPRs (three per CU: the variant that runs beside a GEMM in the two-stream backward).
// FXA: packed 32-bit fixed-point footprints (two channels per LDS atomic) wherever a body has channel pairs; else (and for the
// 1-channel wide body) the exact 64-bit fixed point.
template <bool MOD, int NCELL, int SCN, int SCW, bool FXA>
__global__ __launch_bounds__(ST, SCN == 4 ? 2 : DDCN_S9_WPS) void col2im_coord_tile9_adaptive_kernel(Tile9Args a)
{
    __shared__ int s_esc[ADAPT_STEPS];
    const ddcn_geom &g = a.g;
    const int t = threadIdx.x;
    if (t < ADAPT_STEPS) s_esc[t] = 0;
    // same workgroup -> (tile, channel group) mapping as tile9_body
    const int nwg = a.tiles * a.groups;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int grp = logical / a.tiles, tile = logical - grp * a.tiles;
    const int tyi = tile / a.tiles_w, txi = tile - tyi * a.tiles_w;
    const int ty = t / a.TW, tx = t - ty * a.TW;
    const int ho = tyi * a.TH + ty, wo = txi * a.TW + tx;
    const bool live = ty < a.TH && ho < g.Ho && wo < g.Wo;
    const int n = blockIdx.z;
    const int dgi = grp * a.CG / a.cpg;
    const int p = live ? ho * g.Wo + wo : 0;
    // margin-0 patch of the tile: rows rh0 .. rh0 + RH0 - 1
    const int rh0 = tyi * a.TH * g.sh - g.ph, rw0 = txi * a.TW * g.sw - g.pw;
    const int RH0 = a.RH - 2 * a.M, RW0 = a.RW - 2 * a.M;
    int cnt[ADAPT_STEPS];
#pragma unroll
    for (int j = 0; j < ADAPT_STEPS; ++j) cnt[j] = 0;
    {
        const float *off = a.offset + ((long)n * g.dg + dgi) * 18 * a.P + p;
        const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
        float roh[9], row_[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { roh[k] = off[(long)(2 * k) * a.P]; row_[k] = off[(long)(2 * k + 1) * a.P]; }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int i = k / 3, j = k - 3 * i;
            const float h = (float)(h_in + i * g.dh) + roh[k], w = (float)(w_in + j * g.dw) + row_[k];
            if (live && h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W) {
                const int rr = (int)floorf(h) - rh0, rc = (int)floorf(w) - rw0;
                // the sample's corner cells are inside a margin-M patch iff M >= need
                const int need = max(max(-rr, rr + 2 - RH0), max(-rc, rc + 2 - RW0));
#pragma unroll
                for (int s = 0; s < ADAPT_STEPS; ++s) cnt[s] += need > a.M + 2 * s;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < ADAPT_STEPS; ++s)
        if (cnt[s]) atomicAdd(&s_esc[s], cnt[s]);
    __syncthreads();
    // cost of the planner's patch against every wide candidate (uniform over the workgroup)
    int best = 0, best_cost = a.RS + a.esc_cells * s_esc[0];
    if (best_cost > a.RS) {
#pragma unroll
        for (int s = 1; s < ADAPT_STEPS; ++s) {
            const int M = a.M + 2 * s;
            const int cost = (RH0 + 2 * M) * (RW0 + 2 * M) + a.esc_cells * s_esc[s] + a.RS / 8;   // + the 2-channel chunks' overhead
            if (M <= a.Mw_max && cost < best_cost) { best_cost = cost; best = s; }
        }
    }
    best = __builtin_amdgcn_readfirstlane(best);      // the same in every lane: make the branch a scalar one
    __syncthreads();
    constexpr bool FXW = FXA && SCW >= 2;
    if (best == 0) {
        tile9_body<MOD, NCELL, SCN, FXA, 0, !FXA>(a, PatchGeom{a.M, a.RH, a.RW, a.RS});
    } else {
        const int M = a.M + 2 * best, RH = RH0 + 2 * M, RW = RW0 + 2 * M;
        tile9_body<MOD, NCELL_WIDE, SCW, FXW, 0, !FXW>(a, PatchGeom{M, RH, RW, RH * RW});
    }
}

// weight[Co][C][9] -> wp[C][9][SKB_MAX] (zero beyond Co): the 9 x SKB block of a channel is contiguous for the LDS staging
__global__ __launch_bounds__(256) void repack_skinny_w_kernel(const float *w, float *wp, int Co, int C)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)C * 9 * SKB_MAX) return;
    const int o = (int)(i % SKB_MAX);
    const long ck = i / SKB_MAX;
    wp[i] = o < Co ? w[(long)o * C * 9 + ck] : 0.f;
}

size_t skinny_scatter_ws_bytes(const ddcn_geom &g)
{
    return round_up((size_t)g.C * 9 * SKB_MAX * sizeof(float), 256);
}

// Direct backward scatter for few output channels: grad_input / grad_offset straight from grad_out (no column
// gradients in memory).  `ws` receives the repacked weights (C * 9 * 12 floats).  *handled = false: not covered.
int launch_skinny_scatter(const float *x, const float *offset, const float *weight, const float *gout, float *gx,
                          float *goff, void *ws, size_t ws_bytes, const ddcn_geom &g, hipStream_t stream, bool *handled)
{
    *handled = false;
    static const int env_on = getenv("DDCN_SKINNY_SCATTER") ? atoi(getenv("DDCN_SKINNY_SCATTER")) : 1;
    const size_t need = (size_t)g.C * 9 * SKB_MAX * sizeof(float);
    if (!env_on || g.kh != 3 || g.kw != 3 || g.Co > SKB_MAX || g.groups != 1 || g.dg != 1 || ws_bytes < need) return DDCN_OK;
    // plan first (nothing is launched when the tile kernel does not cover the geometry)
    float *wp = (float *)ws;
    const long total = (long)g.C * 9 * SKB_MAX;
    hipLaunchKernelGGL(repack_skinny_w_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, weight, wp, g.Co, g.C);
    if (int rc = check_launch("repack_skinny_w")) return rc;
    return launch_col2im_coord_tile9(nullptr, x, offset, nullptr, gx, goff, nullptr, g, stream, handled, 4, wp, gout);
}

int launch_col2im_coord_tile9(const float *gcol, const float *x, const float *offset, const float *mask, float *gx,
                              float *goff, float *gmask, const ddcn_geom &g, hipStream_t stream, bool *handled,
                              int chunk_channels, const float *sk_w, const float *sk_g)
{
    *handled = false;
    const bool direct = sk_w != nullptr;
    if (direct && (mask || g.Co > SKB_MAX || g.groups != 1 || g.dg != 1 || !sk_g)) return DDCN_OK;
    if (direct) chunk_channels = 4;
    const Dims d = dims_of(g);
    if (g.kh != 3 || g.kw != 3) return DDCN_OK;
    static const int env_sc = getenv("DDCN_SCATTER_SC") ? atoi(getenv("DDCN_SCATTER_SC")) : 4;
    // experiment: DDCN_SCATTER_WIDE=<margin> -> 2-channel chunks, one 80 KiB footprint, margin up to 16
    const char *env_wide = getenv("DDCN_SCATTER_WIDE");
    const int wide_m = (env_wide && !direct && chunk_channels == 0) ? atoi(env_wide) : 0;
    const int SC = wide_m > 0 ? 2 : (chunk_channels > 0 ? chunk_channels : env_sc);
    if (SC != 2 && SC != 4) return DDCN_OK;
    if (g.H >= 32766 || g.W >= 32766 || d.cpg % SC) return DDCN_OK;
    if ((long)g.C * d.K2 * d.ld * 4 >= (1L << 31) || (long)g.C * g.H * g.W * 4 >= (1L << 31)) return DDCN_OK;
    static const int env_on = getenv("DDCN_SCATTER9") ? atoi(getenv("DDCN_SCATTER9")) : 1;
    if (!env_on) return DDCN_OK;
    Tile9Args a{};
    int TW;
    if (g.Wo <= 32) {
        TW = g.Wo;
    } else {
        int best = 32, waste = (g.Wo + 31) / 32 * 32 - g.Wo;
        for (int w = 31; w >= 16; --w) {
            const int ws = (g.Wo + w - 1) / w * w - g.Wo;
            if (ws < waste) { waste = ws; best = w; }
        }
        TW = best;
    }
    int TH = ST / TW;
    if (TH > g.Ho) TH = g.Ho;
    a.TW = TW; a.TH = TH;
    a.tiles_w = (g.Wo + TW - 1) / TW;
    a.tiles = a.tiles_w * ((g.Ho + TH - 1) / TH);
    static const int env_m = getenv("DDCN_TILE_MARGIN") ? atoi(getenv("DDCN_TILE_MARGIN")) : 6;
    static const int env_cg = getenv("DDCN_SCATTER_CG") ? atoi(getenv("DDCN_SCATTER_CG")) : 0;
    bool ok = false;
    const int max_cells = (wide_m > 0 ? SMAXCELLS_WIDE : SMAXCELLS) * ST;
    const size_t lds_cap = (SC == 4 || wide_m > 0) ? S9_LDS_MAX : S9_LDS_MAX / 2;
    for (int M = wide_m > 0 ? wide_m : env_m; M >= 0 && !ok; M -= 2) {
        const int RH = (TH - 1) * g.sh + 2 * g.dh + 2 + 2 * M;
        const int RW = (TW - 1) * g.sw + 2 * g.dw + 2 + 2 * M;
        const size_t wbytes = direct ? (size_t)4 * 9 * SKB_MAX * sizeof(float) : 0;
        if (RH * RW <= max_cells && (size_t)RH * RW * 12 * SC + wbytes <= lds_cap) {
            a.M = M; a.RH = RH; a.RW = RW; a.RS = RH * RW;
            ok = true;
        }
    }
    if (!ok) return DDCN_OK;
    int CG = d.cpg;
    while (CG % (2 * SC) == 0 && (long)a.tiles * g.N * (g.C / CG) < 1024) CG /= 2;
    if (env_cg > 0 && d.cpg % env_cg == 0 && env_cg % SC == 0) CG = env_cg;
    a.CG = CG;
    a.groups = g.C / CG;
    *handled = true;
    a.gcol = gcol; a.x = x; a.offset = offset; a.mask = mask; a.gx = gx; a.goff = goff; a.gmask = gmask;
    a.g = g; a.P = d.P; a.cpg = d.cpg; a.ld = d.ld;
    a.sk_w = sk_w; a.sk_g = sk_g; a.sk_co = g.Co;
    const long nwg = (long)a.tiles * a.groups;
    if (nwg > 0x7fffffffL || g.N > 65535) return fail(DDCN_ERR_UNSUPPORTED, "col2im_coord_tile9: grid too large");
    dim3 grid((unsigned)nwg, 1, g.N);
    // Footprint format (read per call so that tests can switch it): packed 32-bit fixed point with per-workgroup headroom
    // (FX, the default: half the LDS atomics), DDCN_SCATTER_FX=0 -> exact 64-bit fixed point (XQ), and with DDCN_SCATTER_XQ=0
    // on top doubles.  The direct (few-output-channel) mode keeps doubles.
    const char *env_fx = getenv("DDCN_SCATTER_FX");
    const bool fx = !(env_fx && atoi(env_fx) == 0) && !direct && wide_m == 0;
    const char *env_xq = getenv("DDCN_SCATTER_XQ");
    const bool xq = !fx && !direct && (wide_m > 0 || !(env_xq && atoi(env_xq) == 0));
    const size_t smem = (size_t)a.RS * (fx ? 8 : 12) * SC + (direct ? (size_t)4 * 9 * SKB_MAX * sizeof(float) : 0);
    const int ncell = (a.RS + ST - 1) / ST;
    bool ok_lds = true;
    // each instantiation's dynamic-LDS limit is raised once per device to the planner's cap (80 KiB), not per call
#define DDCN_S9_LAUNCH2(MOD, NC, SCV, FXV)                                                                       \
    do {                                                                                                         \
        static std::atomic<uint64_t> raised{0};                                                                  \
        ok_lds = raise_lds_limit((const void *)&col2im_coord_tile9_kernel<MOD, NC, SCV, FXV>, S9_LDS_MAX, raised); \
        if (ok_lds)                                                                                              \
            hipLaunchKernelGGL((col2im_coord_tile9_kernel<MOD, NC, SCV, FXV>), grid, dim3(ST), smem, stream, a); \
    } while (0)
#define DDCN_S9_LAUNCH_XQ2(MOD, NC, SCV)                                                                         \
    do {                                                                                                         \
        static std::atomic<uint64_t> raised{0};                                                                  \
        ok_lds = raise_lds_limit((const void *)&col2im_coord_tile9_kernel<MOD, NC, SCV, false, 0, true>, S9_LDS_MAX, raised); \
        if (ok_lds)                                                                                              \
            hipLaunchKernelGGL((col2im_coord_tile9_kernel<MOD, NC, SCV, false, 0, true>), grid, dim3(ST), smem, stream, a); \
    } while (0)
#define DDCN_S9_LAUNCH_XQ(MOD, NC)                                                                               \
    do {                                                                                                         \
        if (SC == 4) DDCN_S9_LAUNCH_XQ2(MOD, NC, 4);                                                             \
        else DDCN_S9_LAUNCH_XQ2(MOD, NC, 2);                                                                     \
    } while (0)
#define DDCN_S9_LAUNCH(NC)                                                                                       \
    do {                                                                                                         \
        if (xq && mask) DDCN_S9_LAUNCH_XQ(true, NC);                                                             \
        else if (xq) DDCN_S9_LAUNCH_XQ(false, NC);                                                               \
        else if (mask && fx && SC == 4) DDCN_S9_LAUNCH2(true, NC, 4, true);                                      \
        else if (fx && SC == 4) DDCN_S9_LAUNCH2(false, NC, 4, true);                                             \
        else if (mask && fx) DDCN_S9_LAUNCH2(true, NC, 2, true);                                                 \
        else if (fx) DDCN_S9_LAUNCH2(false, NC, 2, true);                                                        \
        else if (mask && SC == 4) DDCN_S9_LAUNCH2(true, NC, 4, false);                                           \
        else if (mask) DDCN_S9_LAUNCH2(true, NC, 2, false);                                                      \
        else if (SC == 4) DDCN_S9_LAUNCH2(false, NC, 4, false);                                                  \
        else DDCN_S9_LAUNCH2(false, NC, 2, false);                                                               \
    } while (0)
#define DDCN_S9_LAUNCH_DIRECT(NC)                                                                                \
    do {                                                                                                         \
        static std::atomic<uint64_t> raised{0};                                                                  \
        ok_lds = raise_lds_limit((const void *)&col2im_coord_tile9_kernel<false, NC, 4, false, SKB_MAX>, S9_LDS_MAX, raised); \
        if (ok_lds)                                                                                              \
            hipLaunchKernelGGL((col2im_coord_tile9_kernel<false, NC, 4, false, SKB_MAX>), grid, dim3(ST), smem, stream, a); \
    } while (0)
    // per-workgroup adaptive footprint: the default for the plain call (4-channel chunks, exact fixed point)
    static const int env_ad = getenv("DDCN_SCATTER_ADAPTIVE") ? atoi(getenv("DDCN_SCATTER_ADAPTIVE")) : 1;
    int Mw_max = 0;
    size_t smem_ad = smem;
    const size_t ad_cap = SC == 4 ? S9_LDS_MAX : S9_LDS_MAX / 2;
    // bytes per cell of the wide body's footprint + patch: SC/2 channels, packed when they still come in pairs
    const size_t wide_cell = (fx && SC == 4) ? 8 * 2 : (size_t)12 * (SC / 2);
    if (env_ad && (xq || fx) && wide_m == 0 && ncell <= 8) {
        const int RH0 = a.RH - 2 * a.M, RW0 = a.RW - 2 * a.M;
        for (int M = a.M + 2; M <= a.M + 2 * (ADAPT_STEPS - 1); M += 2) {
            const long cells = (long)(RH0 + 2 * M) * (RW0 + 2 * M);
            if (cells > NCELL_WIDE * ST || cells * (long)wide_cell > (long)ad_cap) break;
            Mw_max = M;
            smem_ad = std::max(smem, (size_t)cells * wide_cell);
        }
    }
    a.Mw_max = Mw_max;
    static const int env_esc = getenv("DDCN_SCATTER_ESC_CELLS") ? atoi(getenv("DDCN_SCATTER_ESC_CELLS")) : ESC_CELLS;
    a.esc_cells = env_esc;
#define DDCN_S9_LAUNCH_AD(MOD, NC, SCN, SCW, FXV)                                                                \
    do {                                                                                                         \
        static std::atomic<uint64_t> raised{0};                                                                  \
        ok_lds = raise_lds_limit((const void *)&col2im_coord_tile9_adaptive_kernel<MOD, NC, SCN, SCW, FXV>, S9_LDS_MAX, raised); \
        if (ok_lds)                                                                                              \
            hipLaunchKernelGGL((col2im_coord_tile9_adaptive_kernel<MOD, NC, SCN, SCW, FXV>), grid, dim3(ST), smem_ad, stream, a); \
    } while (0)
#define DDCN_S9_LAUNCH_ADF(MOD, NC, SCN, SCW)                                                                    \
    do {                                                                                                         \
        if (fx) DDCN_S9_LAUNCH_AD(MOD, NC, SCN, SCW, true);                                                      \
        else DDCN_S9_LAUNCH_AD(MOD, NC, SCN, SCW, false);                                                        \
    } while (0)
#define DDCN_S9_LAUNCH_ADN(NC)                                                                                   \
    do {                                                                                                         \
        if (mask && SC == 4) DDCN_S9_LAUNCH_ADF(true, NC, 4, 2);                                                 \
        else if (SC == 4) DDCN_S9_LAUNCH_ADF(false, NC, 4, 2);                                                   \
        else if (mask) DDCN_S9_LAUNCH_ADF(true, NC, 2, 1);                                                       \
        else DDCN_S9_LAUNCH_ADF(false, NC, 2, 1);                                                                \
    } while (0)
    if (Mw_max > 0) {
        if (ncell <= 2) DDCN_S9_LAUNCH_ADN(2);
        else if (ncell <= 4) DDCN_S9_LAUNCH_ADN(4);
        else if (ncell <= 6) DDCN_S9_LAUNCH_ADN(6);
        else DDCN_S9_LAUNCH_ADN(8);
    } else if (direct) {
        if (ncell <= 2) DDCN_S9_LAUNCH_DIRECT(2);
        else if (ncell <= 4) DDCN_S9_LAUNCH_DIRECT(4);
        else if (ncell <= 6) DDCN_S9_LAUNCH_DIRECT(6);
        else DDCN_S9_LAUNCH_DIRECT(8);
    } else if (ncell <= 2) DDCN_S9_LAUNCH(2);
    else if (ncell <= 4) DDCN_S9_LAUNCH(4);
    else if (ncell <= 6) DDCN_S9_LAUNCH(6);
    else if (ncell <= 8) DDCN_S9_LAUNCH(8);
    else { if (mask) DDCN_S9_LAUNCH_XQ2(true, 14, 2); else DDCN_S9_LAUNCH_XQ2(false, 14, 2); }
#undef DDCN_S9_LAUNCH_AD
#undef DDCN_S9_LAUNCH_ADF
#undef DDCN_S9_LAUNCH_ADN
#undef DDCN_S9_LAUNCH_DIRECT
#undef DDCN_S9_LAUNCH2
#undef DDCN_S9_LAUNCH
    if (!ok_lds) return fail(DDCN_ERR_LAUNCH, "col2im_coord_tile9: cannot raise the dynamic LDS limit to %d bytes", S9_LDS_MAX);
    return check_launch("col2im_coord_tile9");
}

}  // namespace ddcn

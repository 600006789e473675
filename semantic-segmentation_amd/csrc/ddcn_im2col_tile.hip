// LDS-staged deformable im2col for 3x3 kernels (gfx950, wave64).
//
// Same arithmetic as detseg/ops/dcn/src/deform_conv_cuda_kernel.cu K:189-242 (deformable_im2col) and
// K:569-632 (modulated), bilinear helper K:83-114; see ddcn_gather.hip for the formula citations.
//
// Work decomposition: workgroup = (TH x TW tile of output pixels, group of CG input channels, image n),
// 256 threads, thread = output pixel.
//   * The nine sample positions of the pixel (patch cell, fractional parts, mask) are computed ONCE and
//     kept in registers for the whole channel group (the reference recomputes them for every channel).
//   * The input patch the tile can reach -- tile extent * stride + kernel extent * dilation + a margin M
//     for the learned offsets -- is staged through LDS four channels at a time, channel-innermost
//     ([cell][4]), double buffered: while chunk c is gathered, the loads of chunk c+1 are in flight
//     (register staged, one barrier per chunk).  Cells outside the image are loaded through an
//     out-of-range buffer offset and read 0 == the reference's zero value for out-of-image corners.
//   * Per (tap, 4 channels) the gather is 4 ds_read_b128 (the four bilinear corners, all four channels
//     each), 16 FMAs and 4 buffer stores whose row offset is a scalar register: ~7 instructions per column
//     element, so the kernel is bound by the column writes, not by instruction issue.
//   * Samples whose corners leave the staged patch ("escapees": learned offsets are unbounded) gather
//     from global memory with the reference's own guards.
#include <stdlib.h>

#include <type_traits>

#include "ddcn_common.h"

namespace ddcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int IT = 256;
#ifndef DDCN_IM2COL_STORE_AUX
// Cache-policy bits of the column stores (bit 1 = non-temporal / streaming).  Measured at config 1 (profiles/r02b_*):
// with nt the kernel alone drops from 0.120 to 0.086 ms (56 % of HBM; sc0 / sc1 change nothing), but the columns then
// miss the Infinity Cache when G1 reads them back and G1 goes from 0.677 to 0.752 ms: the step is 0.025 ms SLOWER.  The
// default stays write-allocate; the same holds for the column gradients G2 writes (DDCN_GEMM_NT).
#define DDCN_IM2COL_STORE_AUX 0
#endif
constexpr int CC = 4;          // channels per LDS chunk (one ds_read_b128 per corner)
constexpr int MAXCELLS = 12;    // staging registers: patch cells per thread (patch <= 8 * 256 cells), template NCELL
constexpr unsigned OOBV = 0x80000000u;

struct Im2colTileArgs {
    const float *x, *offset, *mask;
    float *col;
    ddcn_geom g;
    int P, cpg;
    long ld;
    int TH, TW, tiles_w, tiles, groups;   // groups = C / CG
    int CG, M;
    int RH, RW, RS;
    int dbuf;       // 1: two LDS patch buffers (one barrier per chunk), 0: one buffer (two barriers, half the LDS)
    int strip;      // 1: tile = 256 consecutive pixels (full-width patch rows), 0: TH x TW tile
    // direct mode (SKW > 0): no columns; partial outputs part[grp][n][Co][P] = sum over the group's channels of w * sample
    const float *sk_w;   // weight [Co][C][3][3]
    float *sk_part;
    int sk_co;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void *base, long bytes)
{
    const unsigned n = bytes <= 0 ? 0u : (bytes > 0x7fffffffL ? 0x7fffffffu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, n, 0x00020000);
}

// SKW = 0: im2col (writes the columns).  SKW = 12: direct convolution for <= 12 output channels (the 9-channel mask
// branch of the DDCN layer, detseg/ops/dcn/deform_conv.py:438-446, 458): the sampled values never leave the
// registers, each is multiplied into the output channels with weights staged in LDS next to the patch
// ([channel][tap][12], read as broadcast ds_read_b128), and the workgroup writes one partial sum per output channel.
constexpr int SKMAX = 12;

template <bool MOD, int NCELL, int SKW>
__global__ __launch_bounds__(IT, 2) void im2col_tile_kernel(Im2colTileArgs a)
{
    constexpr int K2 = 9;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 buffers of [RS][CC]
    const ddcn_geom &g = a.g;
    const int t = threadIdx.x;

    // XCD-aware order: each XCD gets a contiguous run of (group, tile) pairs with the tile fastest, so the
    // partial cache lines that neighbouring tiles write into one column row meet in one L2
    const int nwg = a.tiles * a.groups;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int grp = logical / a.tiles, tile = logical - grp * a.tiles;

    int ho, wo, rh0, rw0;   // this thread's pixel; image row / column of patch cell (0, 0)
    bool live;
    if (a.strip) {
        // 256 consecutive pixels: every column-row store of a wave is one contiguous 256-B run (measured
        // 4.1 TB/s for this store shape against 2.2 TB/s for 25-pixel tile rows, tools/ubench_store_pattern.hip)
        const int pix = tile * IT + t;
        live = pix < a.P;
        ho = pix / g.Wo; wo = pix - ho * g.Wo;
        rh0 = (tile * IT / g.Wo) * g.sh - g.ph - a.M;
        rw0 = -1;                                      // patch spans columns -1 .. W (all a valid sample can touch)
    } else {
        const int tyi = tile / a.tiles_w, txi = tile - tyi * a.tiles_w;
        const int ty = t / a.TW, tx = t - ty * a.TW;
        ho = tyi * a.TH + ty; wo = txi * a.TW + tx;
        live = ty < a.TH && ho < g.Ho && wo < g.Wo;
        rh0 = tyi * a.TH * g.sh - g.ph - a.M;
        rw0 = txi * a.TW * g.sw - g.pw - a.M;
    }
    const int n = blockIdx.z;
    const int cg0 = grp * a.CG;
    const int dgi = cg0 / a.cpg;
    const int HW = g.H * g.W;
    const int p = live ? ho * g.Wo + wo : 0;

    const __amdgpu_buffer_rsrc_t rx = mk_rsrc(a.x + (long)n * g.C * HW, (long)g.C * HW * 4);
    const __amdgpu_buffer_rsrc_t rcol = mk_rsrc(a.col + (long)n * g.C * K2 * a.ld, (long)g.C * K2 * a.ld * 4);

    // ---- per-tap sample state (registers, whole channel group)
    // the offsets / masks go in flight first; the patch offsets below are computed while they travel
    float roh[K2], row_[K2], rm[K2];
    {
        const float *off = a.offset + ((long)n * g.dg + dgi) * 2 * K2 * a.P + p;
        const float *msk = MOD ? a.mask + ((long)n * g.dg + dgi) * K2 * a.P + p : nullptr;
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            roh[k] = live ? off[(long)(2 * k) * a.P] : 0.f;
            row_[k] = live ? off[(long)(2 * k + 1) * a.P] : 0.f;
            rm[k] = (MOD && live) ? msk[(long)k * a.P] : 1.f;
        }
    }
    // ---- per-thread staging offsets: cell i = t + 256*j of the patch -> byte offset inside one channel plane
    unsigned soff[NCELL];
#pragma unroll
    for (int j = 0; j < NCELL; ++j) {
        const int i = t + IT * j;
        const int rr = i / a.RW, rc = i - rr * a.RW;
        const int y = rh0 + rr, xx = rw0 + rc;
        soff[j] = (i < a.RS && y >= 0 && y < g.H && xx >= 0 && xx < g.W) ? (unsigned)((y * g.W + xx) * 4) : OOBV;
    }

    int cell[K2];        // patch cell of the (hl, wl) corner; -1: escapee (gather from global); -2: invalid sample
    float bw1[K2], bw2[K2], bw3[K2], bw4[K2], fm[K2];   // bilinear weights (K:93-110) and mask
    int ghw[K2];         // (hl << 16) | (wl & 0xffff) for the escapee path
    {
        const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            const int i = k / 3, j = k - 3 * i;
            float h = 0.f, w = 0.f;
            const float m = rm[k];
            if (live) {
                h = (float)(h_in + i * g.dh) + roh[k];
                w = (float)(w_in + j * g.dw) + row_[k];
            }
            const float fhl = floorf(h), fwl = floorf(w);
            const float lh = h - fhl, lw = w - fwl;
            const float hh = 1.f - lh, hw = 1.f - lw;
            bw1[k] = hh * hw; bw2[k] = hh * lw; bw3[k] = lh * hw; bw4[k] = lh * lw;
            fm[k] = m;
            const int hl = (int)fhl, wl = (int)fwl;
            ghw[k] = (hl << 16) | (wl & 0xffff);
            int c = -2;
            if (live && h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W) {
                const int rr = hl - rh0, rc = wl - rw0;
                c = (rr >= 0 && rr + 1 < a.RH && rc >= 0 && rc + 1 < a.RW) ? rr * a.RW + rc : -1;
            }
            cell[k] = c;
        }
    }

    f32x4 *buf0 = reinterpret_cast<f32x4 *>(smem);
    f32x4 *buf1 = buf0 + a.RS;
    float stg[NCELL][CC];
    // direct mode: weights of the chunk, wl[(cc * 9 + k) * 12 + co], two per thread (432 floats)
    float *wl = smem + (size_t)(a.dbuf ? 2 : 1) * a.RS * CC;
    float wreg[2] = {0.f, 0.f};
    float sacc[SKW ? SKW : 1];
#pragma unroll
    for (int i = 0; i < (SKW ? SKW : 1); ++i) sacc[i] = 0.f;
    auto wstage_load = [&](int c0) {
        if (SKW) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = t + IT * r;
                const int co = i % SKW, ck = i / SKW;          // ck = cc * 9 + k
                wreg[r] = (i < CC * K2 * SKW && co < a.sk_co) ? a.sk_w[((long)co * g.C + c0 + ck / K2) * K2 + ck % K2] : 0.f;
            }
        }
    };
    auto wstage_store = [&]() {
        if (SKW) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (t + IT * r < CC * K2 * SKW) wl[t + IT * r] = wreg[r];
        }
    };

    auto stage_load = [&](int c0) {
#pragma unroll
        for (int j = 0; j < NCELL; ++j) {
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
                stg[j][cc] = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(rx, soff[j], (unsigned)((c0 + cc) * HW * 4), 0));
        }
    };
    auto stage_store = [&](f32x4 *buf) {
#pragma unroll
        for (int j = 0; j < NCELL; ++j) {
            const int i = t + IT * j;
            if (i < a.RS) buf[i] = f32x4{stg[j][0], stg[j][1], stg[j][2], stg[j][3]};
        }
    };

    stage_load(cg0);
    wstage_load(cg0);
    stage_store(buf0);
    wstage_store();
    __syncthreads();

    const unsigned pvo = (unsigned)(p * 4);
    const int nchunks = a.CG / CC;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = cg0 + ch * CC;
        const f32x4 *cur = (a.dbuf && (ch & 1)) ? buf1 : buf0;
        f32x4 *nxt = (a.dbuf && !(ch & 1)) ? buf1 : buf0;
        if (ch + 1 < nchunks) { stage_load(c0 + CC); wstage_load(c0 + CC); }
        __builtin_amdgcn_sched_barrier(0);
        if (live) {
#pragma unroll
            for (int k = 0; k < K2; ++k) {
                const float w1 = bw1[k], w2 = bw2[k], w3 = bw3[k], w4 = bw4[k];
                const unsigned rowk = (unsigned)((c0 * K2 + k) * a.ld * 4);   // byte offset of column row (c0, k)
                const unsigned rstep = (unsigned)(K2 * a.ld * 4);
                if (cell[k] != -1) {
                    // in-patch (or invalid: weights irrelevant, result forced to 0 below); escapees: after the loop
                    const int ci = cell[k] < 0 ? 0 : cell[k];
                    const f32x4 a1 = cur[ci], a2 = cur[ci + 1], a3 = cur[ci + a.RW], a4 = cur[ci + a.RW + 1];
#pragma unroll
                    for (int cc = 0; cc < CC; ++cc) {
                        float val = (w1 * a1[cc] + w2 * a2[cc] + w3 * a3[cc] + w4 * a4[cc]);
                        if (MOD) val = val * fm[k];
                        val = cell[k] < 0 ? 0.f : val;
                        if (SKW) {
                            const f32x4 *wv = reinterpret_cast<const f32x4 *>(wl + (cc * K2 + k) * SKW);
#pragma unroll
                            for (int q = 0; q < SKW / 4; ++q) {
                                const f32x4 w4v = wv[q];
                                sacc[4 * q + 0] += w4v.x * val; sacc[4 * q + 1] += w4v.y * val;
                                sacc[4 * q + 2] += w4v.z * val; sacc[4 * q + 3] += w4v.w * val;
                            }
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rcol, pvo,
                                                                  rowk + cc * rstep, DDCN_IM2COL_STORE_AUX);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!a.dbuf) __syncthreads();   // every wave is done reading the single buffer
        if (ch + 1 < nchunks) { stage_store(nxt); wstage_store(); }
        __syncthreads();
    }
    // ---- escapees: samples whose corners leave the staged patch, gathered from global memory once per
    //      workgroup for the whole channel group (rare: |offset| beyond the margin)
    if (live) {
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            if (cell[k] == -1) {
                const int hl = ghw[k] >> 16, wl = (int)(short)(ghw[k] & 0xffff);
                const bool ok1 = hl >= 0 && wl >= 0, ok2 = hl >= 0 && wl + 1 <= g.W - 1;
                const bool ok3 = hl + 1 <= g.H - 1 && wl >= 0, ok4 = hl + 1 <= g.H - 1 && wl + 1 <= g.W - 1;
                // corner offsets inside a channel plane, 0 for a corner outside the map: the loads are unconditional
                // and four channels' loads are issued before the first use (a load under a condition becomes a
                // branch with its own wait)
                const int gi = hl * g.W + wl;
                const int o1 = ok1 ? gi : 0, o2 = ok2 ? gi + 1 : 0, o3 = ok3 ? gi + g.W : 0, o4 = ok4 ? gi + g.W + 1 : 0;
                const float *im = a.x + ((long)n * g.C + cg0) * HW;
                const float w1 = bw1[k], w2 = bw2[k], w3 = bw3[k], w4 = bw4[k];
                for (int c0 = 0; c0 < a.CG; c0 += 4) {
                    float r1[4], r2[4], r3[4], r4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float *pl = im + (long)min(c0 + u, a.CG - 1) * HW;
                        r1[u] = pl[o1]; r2[u] = pl[o2]; r3[u] = pl[o3]; r4[u] = pl[o4];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + u;
                        if (c < a.CG) {
                            const float v1 = ok1 ? r1[u] : 0.f, v2 = ok2 ? r2[u] : 0.f;
                            const float v3 = ok3 ? r3[u] : 0.f, v4 = ok4 ? r4[u] : 0.f;
                            float val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                            if (MOD) val = val * fm[k];
                            if (SKW) {
#pragma unroll
                                for (int co = 0; co < SKW; ++co)
                                    if (co < a.sk_co) sacc[co] += a.sk_w[((long)co * g.C + cg0 + c) * K2 + k] * val;
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rcol, pvo,
                                                                      (unsigned)(((cg0 + c) * K2 + k) * a.ld * 4), DDCN_IM2COL_STORE_AUX);
                            }
                        }
                    }
                }
            }
        }
    }
    if (SKW && live) {
        float *dst = a.sk_part + (((long)grp * g.N + n) * a.sk_co) * a.P + p;
#pragma unroll
        for (int co = 0; co < SKW; ++co)
            if (co < a.sk_co) dst[(long)co * a.P] = sacc[co];
    }
}

// direct mode, second step: out[n][co][p] = act(bias[co] + sum over channel groups of part[grp][n][co][p])
__global__ __launch_bounds__(256) void skinny_finish_kernel(const float *part, const float *bias, float *out, long per_grp,
                                                            int groups, int co_n, int P, int sigmoid)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_grp) return;
    float v = bias ? bias[(i / P) % co_n] : 0.f;
    for (int gq = 0; gq < groups; ++gq) v += part[(long)gq * per_grp + i];
    out[i] = sigmoid ? 1.f / (1.f + __expf(-v)) : v;
}

// Plans the tiling (fills the geometry part of a); false when this geometry must use the generic kernel.
static bool plan_tile(Im2colTileArgs &a, const ddcn_geom &g, long min_wgs)
{
    const Dims d = dims_of(g);
    if (g.kh != 3 || g.kw != 3) return false;
    if (g.H >= 32766 || g.W >= 32766 || d.cpg % CC) return false;
    if ((long)g.C * d.K2 * d.ld * 4 >= (1L << 31) || (long)g.C * g.H * g.W * 4 >= (1L << 31)) return false;
    // tile width: least padding waste among 16..32 (or the whole row when narrower)
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
    int TH = IT / TW;
    if (TH > g.Ho) TH = g.Ho;
    a.TW = TW; a.TH = TH;
    a.tiles_w = (g.Wo + TW - 1) / TW;
    a.tiles = a.tiles_w * ((g.Ho + TH - 1) / TH);
    static const int env_m = getenv("DDCN_TILE_MARGIN") ? atoi(getenv("DDCN_TILE_MARGIN")) : 6;   // tuning knobs
    static const int env_cg = getenv("DDCN_IM2COL_CG") ? atoi(getenv("DDCN_IM2COL_CG")) : 0;
    static const int env_strip = getenv("DDCN_IM2COL_STRIP") ? atoi(getenv("DDCN_IM2COL_STRIP")) : 1;
    bool ok = false;
    // strip mode (contiguous column stores) when the full-width patch fits the staging budget
    for (int M = env_m; env_strip && M >= 2 && !ok; M -= 2) {
        const int rows_span = (IT + g.Wo - 2) / g.Wo + 1;
        const int RH = (rows_span - 1) * g.sh + (g.kh - 1) * g.dh + 2 + 2 * M;
        const int RW = g.W + 2;
        if (RH * RW <= MAXCELLS * IT) {
            a.strip = 1; a.M = M; a.RH = RH; a.RW = RW; a.RS = RH * RW;
            a.TW = g.Wo; a.TH = 1; a.tiles_w = 1; a.tiles = (d.P + IT - 1) / IT;
            ok = true;
        }
    }
    for (int M = env_m; M >= 0 && !ok; M -= 2) {
        const int RH = (TH - 1) * g.sh + (g.kh - 1) * g.dh + 2 + 2 * M;
        const int RW = (TW - 1) * g.sw + (g.kw - 1) * g.dw + 2 + 2 * M;
        if (RH * RW <= MAXCELLS * IT) {
            a.M = M; a.RH = RH; a.RW = RW; a.RS = RH * RW;
            ok = true;
        }
    }
    if (!ok) return false;
    // channel group per workgroup: amortises the sample set-up, but keep enough workgroups in the launch
    int CG = d.cpg;
    while (CG % (2 * CC) == 0 && (long)a.tiles * g.N * (g.C / CG) < min_wgs) CG /= 2;
    if (env_cg > 0 && d.cpg % env_cg == 0 && env_cg % CC == 0) CG = env_cg;
    a.CG = CG;
    a.groups = g.C / CG;
    a.g = g; a.P = d.P; a.cpg = d.cpg; a.ld = d.ld;
    static const int env_dbuf = getenv("DDCN_IM2COL_DBUF") ? atoi(getenv("DDCN_IM2COL_DBUF")) : 0;
    a.dbuf = env_dbuf;
    return true;
}

template <int SKW>
static int run_tile(const Im2colTileArgs &a, bool mod, hipStream_t stream, const char *what)
{
    const long nwg = (long)a.tiles * a.groups;
    if (nwg > 0x7fffffffL || a.g.N > 65535) return fail(DDCN_ERR_UNSUPPORTED, "%s: grid too large", what);
    dim3 grid((unsigned)nwg, 1, a.g.N);
    const size_t smem = (size_t)(a.dbuf ? 2 : 1) * a.RS * CC * sizeof(float) + (SKW ? (size_t)CC * 9 * SKW * sizeof(float) : 0);
    const int ncell = (a.RS + IT - 1) / IT;
#define DDCN_IM2COL_LAUNCH(NC)                                                                              \
    do {                                                                                                    \
        if (mod)                                                                                            \
            hipLaunchKernelGGL((im2col_tile_kernel<true, NC, SKW>), grid, dim3(IT), smem, stream, a);       \
        else                                                                                                \
            hipLaunchKernelGGL((im2col_tile_kernel<false, NC, SKW>), grid, dim3(IT), smem, stream, a);      \
    } while (0)
    // > 64 KiB of dynamic LDS needs the opt-in: every instantiation is raised once per device to the largest patch the
    // planner admits (MAXCELLS * IT cells, double buffered, + the direct mode's weights), not per call
    constexpr int LDS_MAX = 2 * MAXCELLS * IT * CC * (int)sizeof(float) + CC * 9 * SKMAX * (int)sizeof(float);
    if (smem > (size_t)LDS_MAX) return fail(DDCN_ERR_UNSUPPORTED, "%s: patch of %zu bytes exceeds the LDS plan", what, smem);
    {
        static std::atomic<uint64_t> raised{0};
        const void *fns[] = {(const void *)&im2col_tile_kernel<true, 2, SKW>, (const void *)&im2col_tile_kernel<false, 2, SKW>,
                             (const void *)&im2col_tile_kernel<true, 4, SKW>, (const void *)&im2col_tile_kernel<false, 4, SKW>,
                             (const void *)&im2col_tile_kernel<true, 6, SKW>, (const void *)&im2col_tile_kernel<false, 6, SKW>,
                             (const void *)&im2col_tile_kernel<true, 8, SKW>, (const void *)&im2col_tile_kernel<false, 8, SKW>,
                             (const void *)&im2col_tile_kernel<true, 10, SKW>, (const void *)&im2col_tile_kernel<false, 10, SKW>,
                             (const void *)&im2col_tile_kernel<true, 12, SKW>, (const void *)&im2col_tile_kernel<false, 12, SKW>};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return fail(DDCN_ERR_NO_DEVICE, "%s: hipGetDevice failed", what);
        const uint64_t bit = dev < 64 ? (1ull << dev) : 0;
        if (!bit || !(raised.load(std::memory_order_acquire) & bit)) {
            for (const void *fn : fns) {
                std::atomic<uint64_t> scratch_flag{0};
                if (!raise_lds_limit(fn, LDS_MAX, scratch_flag))
                    return fail(DDCN_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit to %d bytes", what, LDS_MAX);
            }
            raised.fetch_or(bit, std::memory_order_release);
        }
    }
    if (ncell <= 2) DDCN_IM2COL_LAUNCH(2);
    else if (ncell <= 4) DDCN_IM2COL_LAUNCH(4);
    else if (ncell <= 6) DDCN_IM2COL_LAUNCH(6);
    else if (ncell <= 8) DDCN_IM2COL_LAUNCH(8);
    else if (ncell <= 10) DDCN_IM2COL_LAUNCH(10);
    else DDCN_IM2COL_LAUNCH(12);
#undef DDCN_IM2COL_LAUNCH
    return check_launch(what);
}

// *handled = false when this geometry must use the generic kernel.
int launch_im2col_tiled(const float *x, const float *offset, const float *mask, float *col, const ddcn_geom &g,
                        hipStream_t stream, bool *handled)
{
    *handled = false;
    Im2colTileArgs a{};
    if (!plan_tile(a, g, 1536)) return DDCN_OK;
    *handled = true;
    a.x = x; a.offset = offset; a.mask = mask; a.col = col;
    return run_tile<0>(a, mask != nullptr, stream, "im2col_tile");
}

// Direct deformable convolution for <= 12 output channels (no columns, no GEMM): out = act(bias + W * sample(x)).
// workspace: groups * N * Co * P floats of partial sums (groups = channel groups chosen by the planner).
// *handled = false: geometry not covered (the caller then composes im2col + GEMM).
int launch_skinny_forward(const float *x, const float *offset, const float *weight, const float *bias, float *out,
                          int sigmoid, void *ws, size_t ws_bytes, const ddcn_geom &g, hipStream_t stream, bool *handled)
{
    *handled = false;
    if (g.groups != 1 || g.dg != 1 || g.Co > SKMAX) return DDCN_OK;
    Im2colTileArgs a{};
    if (!plan_tile(a, g, 1024)) return DDCN_OK;
    const long per_grp = (long)g.N * g.Co * a.P;
    if ((size_t)a.groups * per_grp * sizeof(float) > ws_bytes) return DDCN_OK;
    *handled = true;
    a.x = x; a.offset = offset; a.mask = nullptr; a.col = nullptr;
    a.sk_w = weight; a.sk_part = (float *)ws; a.sk_co = g.Co;
    const int rc = run_tile<SKMAX>(a, false, stream, "skinny_forward");
    if (rc != DDCN_OK) return rc;
    const long blocks = (per_grp + 255) / 256;
    if (blocks > 0x7fffffffL) return fail(DDCN_ERR_UNSUPPORTED, "skinny_forward: grid too large");
    hipLaunchKernelGGL(skinny_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float *)ws, bias, out,
                       per_grp, a.groups, g.Co, a.P, sigmoid);
    return check_launch("skinny_finish");
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of a deformable convolution with FEW output channels, without columns (the backward twin of the
// direct forward above; the DDCN layer's mask branch, detseg/ops/dcn/deform_conv.py:438-446):
//     gw[o][c][k] += sum_{n,p} gpre[n][o][p] * sample(x[n][c], p, k)          (G3, deform_conv_cuda.cpp:341-346)
// where sample() is the bilinear value the forward's im2col would have written (K:189-242).  The reference re-runs
// im2col into its columns tensor (H:325-327: 86.7 MB per image for this branch) and multiplies by a 9-row matrix.
// Here a workgroup owns ONE input channel and a slice of the pixel strips of one image: per strip of 256 consecutive
// pixels it stages the channel's patch rows in LDS, every lane samples its pixel's nine taps and multiplies them into
// Co x 9 register accumulators; after the last strip the accumulators are summed over the workgroup (wave butterflies,
// then LDS) and added to gw with one atomic per (o, k).  Nothing but x, offset and gpre is read.
struct SkinnyGwArgs {
    const float *x, *gpre;
    const float *tab;    // sample table [N][9][3][P]: (hl << 16 | wl & 0xffff) bits, lh, lw; hl = -32768: sample invalid
    float *gw;
    ddcn_geom g;
    int P, co;
    int slices, tiles, tiles_per_slice;
    int M, RH, RW, RS;
};

// Sample table: the per-(pixel, tap) part of the bilinear gather -- integer corner and fractional parts, K:226-233 --
// computed once and streamed by every channel's workgroup below (the coordinate arithmetic is ~100 instructions per
// sample; recomputed per channel it was 10x the useful work of the weight-gradient kernel).
constexpr int TAB_INVALID = -32768;
__global__ __launch_bounds__(256) void sample_table_kernel(const float *offset, float *tab, ddcn_geom g, int P)
{
    constexpr int K2 = 9;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int k = blockIdx.y, n = blockIdx.z;
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    const int i = k / 3, j = k - 3 * i;
    const float *off = offset + ((long)n * 2 * K2 + 2 * k) * P + p;
    const float h = (float)(ho * g.sh - g.ph + i * g.dh) + off[0];
    const float w = (float)(wo * g.sw - g.pw + j * g.dw) + off[P];
    const float fhl = floorf(h), fwl = floorf(w);
    const bool valid = h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W;
    const int hl = valid ? (int)fhl : TAB_INVALID, wl = valid ? (int)fwl : 0;
    float *dst = tab + ((long)n * K2 + k) * 3 * P + p;
    dst[0] = __int_as_float((hl << 16) | (wl & 0xffff));
    dst[P] = h - fhl;
    dst[2 * (long)P] = w - fwl;
}

// Sum of v over the 64 lanes of a wave, valid in lane 63: four DPP steps reduce each row of 16 lanes, two row
// broadcasts carry the row sums up (the sequence rocPRIM uses on GFX9-class targets).
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    auto dpp = [](float x, auto ctrl, auto row_mask) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                     decltype(row_mask)::value, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});   // row_shr:1
    v += dpp(v, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});   // row_shr:2
    v += dpp(v, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});   // row_shr:4
    v += dpp(v, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});   // row_shr:8 -> lane 15 of each row
    v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1, 3
    v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 into rows 2, 3
    return v;
}

template <int COMAX, int NCELL>
__global__ __launch_bounds__(IT, 2) void skinny_gweight_kernel(SkinnyGwArgs a)
{
    constexpr int K2 = 9;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // xs[RS] patch of one channel, then red[4][COMAX * 9]
    const ddcn_geom &g = a.g;
    const int t = threadIdx.x;
    const int c = blockIdx.x, slice = blockIdx.y, n = blockIdx.z;
    const int HW = g.H * g.W;
    const __amdgpu_buffer_rsrc_t rx = mk_rsrc(a.x + ((long)n * g.C + c) * HW, (long)HW * 4);
    const float *xc = a.x + ((long)n * g.C + c) * HW;
    const float *tab = a.tab + (long)n * K2 * 3 * a.P;
    const float *gp_base = a.gpre + (long)n * a.co * a.P;
    float *xs = smem;

    float acc[COMAX][K2];
#pragma unroll
    for (int o = 0; o < COMAX; ++o)
#pragma unroll
        for (int k = 0; k < K2; ++k) acc[o][k] = 0.f;

    // patch cell i = t + 256 j -> (row, image column) inside the patch; the patch's first image row moves with the strip
    int prow[NCELL], pcol[NCELL];
#pragma unroll
    for (int j = 0; j < NCELL; ++j) {
        const int i = t + IT * j;
        prow[j] = i / a.RW;
        pcol[j] = i - prow[j] * a.RW - 1;          // patch columns are image columns -1 .. W
        if (i >= a.RS || pcol[j] < 0 || pcol[j] >= g.W) prow[j] = -(1 << 20);   // never inside the image
    }
    float stg[NCELL], tl[K2], th[K2], tw[K2], gp[COMAX];
    // everything strip `tile` needs from global memory goes in flight at once (out-of-image cells read 0 through an
    // out-of-range buffer offset == the reference's zero for out-of-image corners, K:97-108)
    auto fetch = [&](int tile) {
        const int rh0 = (tile * IT / g.Wo) * g.sh - g.ph - a.M;
        const int pix = tile * IT + t;
        const int p = pix < a.P ? pix : 0;
#pragma unroll
        for (int j = 0; j < NCELL; ++j) {
            const int y = rh0 + prow[j];
            const unsigned o = (y >= 0 && y < g.H) ? (unsigned)((y * g.W + pcol[j]) * 4) : OOBV;
            stg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, o, 0, 0));
        }
        const float *tp = tab + p;
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            tl[k] = tp[(long)(3 * k) * a.P];
            th[k] = tp[(long)(3 * k + 1) * a.P];
            tw[k] = tp[(long)(3 * k + 2) * a.P];
        }
#pragma unroll
        for (int o = 0; o < COMAX; ++o) gp[o] = (o < a.co) ? gp_base[(long)o * a.P + p] : 0.f;
    };

    const int tile_beg = slice * a.tiles_per_slice;
    const int tile_end = min(a.tiles, tile_beg + a.tiles_per_slice);
    if (tile_beg < tile_end) fetch(tile_beg);
    for (int tile = tile_beg; tile < tile_end; ++tile) {
        const bool live = tile * IT + t < a.P;
        const int rh0 = (tile * IT / g.Wo) * g.sh - g.ph - a.M;
#pragma unroll
        for (int j = 0; j < NCELL; ++j)
            if (t + IT * j < a.RS) xs[t + IT * j] = stg[j];
        // this strip's samples out of the prefetch registers, then the next strip's loads
        int chw[K2];
        float clh[K2], clw[K2], cgp[COMAX];
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            chw[k] = live ? __float_as_int(tl[k]) : (int)((unsigned)TAB_INVALID << 16);
            clh[k] = th[k]; clw[k] = tw[k];
        }
#pragma unroll
        for (int o = 0; o < COMAX; ++o) cgp[o] = gp[o];
        __syncthreads();
        if (tile + 1 < tile_end) fetch(tile + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            const int hl = chw[k] >> 16, wl = (int)(short)(chw[k] & 0xffff);
            float val = 0.f;
            if (hl != TAB_INVALID) {
                const float lh = clh[k], lw = clw[k];
                const float hh = 1.f - lh, hw = 1.f - lw;
                const int rr = hl - rh0, rc = wl + 1;
                float v1, v2, v3, v4;
                if (rr >= 0 && rr + 1 < a.RH) {          // wl in [-1, W-1] always holds for a valid sample
                    const int ci = rr * a.RW + rc;
                    v1 = xs[ci]; v2 = xs[ci + 1]; v3 = xs[ci + a.RW]; v4 = xs[ci + a.RW + 1];
                } else {                                  // the sample left the staged rows: the reference's own guards
                    const bool ok1 = hl >= 0 && wl >= 0, ok2 = hl >= 0 && wl + 1 <= g.W - 1;
                    const bool ok3 = hl + 1 <= g.H - 1 && wl >= 0, ok4 = hl + 1 <= g.H - 1 && wl + 1 <= g.W - 1;
                    const int gi = hl * g.W + wl;
                    const float r1 = xc[ok1 ? gi : 0], r2 = xc[ok2 ? gi + 1 : 0];
                    const float r3 = xc[ok3 ? gi + g.W : 0], r4 = xc[ok4 ? gi + g.W + 1 : 0];
                    v1 = ok1 ? r1 : 0.f; v2 = ok2 ? r2 : 0.f; v3 = ok3 ? r3 : 0.f; v4 = ok4 ? r4 : 0.f;
                }
                val = (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
            }
#pragma unroll
            for (int o = 0; o < COMAX; ++o) acc[o][k] += cgp[o] * val;
        }
        __syncthreads();   // every wave is done with the patch before the next strip overwrites it
    }

    // sum over the workgroup: DPP row reductions inside each wave (no LDS traffic), then the four waves through LDS
    float *red = smem;     // the patch is dead (barrier above)
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int o = 0; o < COMAX; ++o)
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            const float v = wave_sum_to_lane63(acc[o][k]);
            if (lane == 63) red[wave * (COMAX * K2) + o * K2 + k] = v;
        }
    __syncthreads();
    if (t < COMAX * K2) {
        const int o = t / K2, k = t - o * K2;
        if (o < a.co) {
            const float v = red[t] + red[COMAX * K2 + t] + red[2 * COMAX * K2 + t] + red[3 * COMAX * K2 + t];
            atomicAdd(a.gw + ((long)o * g.C + c) * K2 + k, v);
        }
    }
}

// `ws` (>= 108 * N * P bytes, any contents) receives the sample table.
int launch_skinny_gweight(const float *x, const float *offset, const float *gpre, float *gw, void *ws, size_t ws_bytes,
                          const ddcn_geom &g, hipStream_t stream, bool *handled)
{
    *handled = false;
    const Dims d = dims_of(g);
    if (g.kh != 3 || g.kw != 3 || g.groups != 1 || g.dg != 1 || g.Co > SKMAX) return DDCN_OK;
    if ((long)g.C * g.H * g.W >= (1L << 31) || g.N > 65535 || g.C > 0x7fffffff) return DDCN_OK;
    if (g.H >= 32766 || g.W >= 32766 || (size_t)g.N * 9 * 3 * d.P * sizeof(float) > ws_bytes) return DDCN_OK;
    static const int env_on = getenv("DDCN_SKINNY_GW") ? atoi(getenv("DDCN_SKINNY_GW")) : 1;
    if (!env_on) return DDCN_OK;
    SkinnyGwArgs a{};
    static const int env_m = getenv("DDCN_TILE_MARGIN") ? atoi(getenv("DDCN_TILE_MARGIN")) : 6;
    a.M = env_m < 0 ? 0 : env_m;
    const int rows_span = (IT + g.Wo - 2) / g.Wo + 1;
    a.RH = (rows_span - 1) * g.sh + 2 * g.dh + 2 + 2 * a.M;
    a.RW = g.W + 2;
    a.RS = a.RH * a.RW;
    const size_t smem = sizeof(float) * (size_t)max(a.RS, 4 * SKMAX * 9);   // patch, later the wave sums
    const int ncell = (a.RS + IT - 1) / IT;
    if (ncell > 16 || (long)g.H * g.W * 4 >= (1L << 31)) return DDCN_OK;   // very wide maps: columns path
    a.x = x; a.tab = (const float *)ws; a.gpre = gpre; a.gw = gw;
    a.g = g; a.P = d.P; a.co = g.Co;
    a.tiles = (d.P + IT - 1) / IT;
    // enough workgroups to fill the chip (two per CU) a few times over, but no more slices than strips
    int slices = (int)((2048 + (long)g.C * g.N - 1) / ((long)g.C * g.N));
    if (slices < 1) slices = 1;
    if (slices > a.tiles) slices = a.tiles;
    a.tiles_per_slice = (a.tiles + slices - 1) / slices;
    a.slices = (a.tiles + a.tiles_per_slice - 1) / a.tiles_per_slice;
    if (a.slices > 65535) return DDCN_OK;
    *handled = true;
    hipLaunchKernelGGL(sample_table_kernel, dim3((unsigned)((d.P + 255) / 256), 9, (unsigned)g.N), dim3(256), 0, stream,
                       offset, (float *)ws, g, d.P);
    const dim3 grid((unsigned)g.C, (unsigned)a.slices, (unsigned)g.N);
#define DDCN_SKGW_LAUNCH(CO)                                                                                  \
    do {                                                                                                      \
        if (ncell <= 4) hipLaunchKernelGGL((skinny_gweight_kernel<CO, 4>), grid, dim3(IT), smem, stream, a);  \
        else if (ncell <= 8) hipLaunchKernelGGL((skinny_gweight_kernel<CO, 8>), grid, dim3(IT), smem, stream, a);   \
        else if (ncell <= 12) hipLaunchKernelGGL((skinny_gweight_kernel<CO, 12>), grid, dim3(IT), smem, stream, a); \
        else hipLaunchKernelGGL((skinny_gweight_kernel<CO, 16>), grid, dim3(IT), smem, stream, a);            \
    } while (0)
    if (g.Co <= 9) DDCN_SKGW_LAUNCH(9);      // the DDCN layer's mask branch: 9 output channels
    else DDCN_SKGW_LAUNCH(SKMAX);
#undef DDCN_SKGW_LAUNCH
    return check_launch("skinny_gweight");
}

// out[i] = sigmoid(out[i]) (fallback of the direct entry point when the geometry is not covered)
int launch_sigmoid_inplace(float *out, long count, hipStream_t stream)
{
    const long blocks = (count + 255) / 256;
    if (blocks <= 0) return DDCN_OK;
    if (blocks > 0x7fffffffL) return fail(DDCN_ERR_UNSUPPORTED, "sigmoid: grid too large");
    hipLaunchKernelGGL(skinny_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float *)out,
                       (const float *)nullptr, out, count, 1, 1, 1, 1);
    return check_launch("sigmoid");
}

}  // namespace ddcn

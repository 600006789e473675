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

#include "ddcn_common.h"

namespace ddcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int IT = 256;
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
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void *base, long bytes)
{
    const unsigned n = bytes <= 0 ? 0u : (bytes > 0x7fffffffL ? 0x7fffffffu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, n, 0x00020000);
}

template <bool MOD, int NCELL>
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
    stage_store(buf0);
    __syncthreads();

    const unsigned pvo = (unsigned)(p * 4);
    const int nchunks = a.CG / CC;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = cg0 + ch * CC;
        const f32x4 *cur = (a.dbuf && (ch & 1)) ? buf1 : buf0;
        f32x4 *nxt = (a.dbuf && !(ch & 1)) ? buf1 : buf0;
        if (ch + 1 < nchunks) stage_load(c0 + CC);
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
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rcol, pvo,
                                                              rowk + cc * rstep, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!a.dbuf) __syncthreads();   // every wave is done reading the single buffer
        if (ch + 1 < nchunks) stage_store(nxt);
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
                const float *im = a.x + ((long)n * g.C + cg0) * HW + hl * g.W + wl;
                const float w1 = bw1[k], w2 = bw2[k], w3 = bw3[k], w4 = bw4[k];
                for (int c = 0; c < a.CG; ++c) {
                    const float v1 = ok1 ? im[0] : 0.f;
                    const float v2 = ok2 ? im[1] : 0.f;
                    const float v3 = ok3 ? im[g.W] : 0.f;
                    const float v4 = ok4 ? im[g.W + 1] : 0.f;
                    float val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                    if (MOD) val = val * fm[k];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rcol, pvo,
                                                          (unsigned)(((cg0 + c) * K2 + k) * a.ld * 4), 0);
                    im += HW;
                }
            }
        }
    }
}

// Plans the tiling; *handled = false when this geometry must use the generic kernel.
int launch_im2col_tiled(const float *x, const float *offset, const float *mask, float *col, const ddcn_geom &g,
                        hipStream_t stream, bool *handled)
{
    *handled = false;
    const Dims d = dims_of(g);
    if (g.kh != 3 || g.kw != 3) return DDCN_OK;
    if (g.H >= 32766 || g.W >= 32766 || d.cpg % CC) return DDCN_OK;
    if ((long)g.C * d.K2 * d.ld * 4 >= (1L << 31) || (long)g.C * g.H * g.W * 4 >= (1L << 31)) return DDCN_OK;
    Im2colTileArgs a{};
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
    if (!ok) return DDCN_OK;
    // channel group per workgroup: amortises the sample set-up, but keep >= ~1500 workgroups in the launch
    int CG = d.cpg;
    while (CG % (2 * CC) == 0 && (long)a.tiles * g.N * (g.C / CG) < 1536) CG /= 2;
    if (env_cg > 0 && d.cpg % env_cg == 0 && env_cg % CC == 0) CG = env_cg;
    a.CG = CG;
    a.groups = g.C / CG;
    *handled = true;
    a.x = x; a.offset = offset; a.mask = mask; a.col = col;
    a.g = g; a.P = d.P; a.cpg = d.cpg; a.ld = d.ld;
    const long nwg = (long)a.tiles * a.groups;
    if (nwg > 0x7fffffffL || g.N > 65535) return fail(DDCN_ERR_UNSUPPORTED, "im2col_tile: grid too large");
    dim3 grid((unsigned)nwg, 1, g.N);
    static const int env_dbuf = getenv("DDCN_IM2COL_DBUF") ? atoi(getenv("DDCN_IM2COL_DBUF")) : 0;
    a.dbuf = env_dbuf;
    const size_t smem = (size_t)(a.dbuf ? 2 : 1) * a.RS * CC * sizeof(float);
    const int ncell = (a.RS + IT - 1) / IT;
#define DDCN_IM2COL_LAUNCH(NC)                                                                         \
    do {                                                                                               \
        if (mask)                                                                                      \
            hipLaunchKernelGGL((im2col_tile_kernel<true, NC>), grid, dim3(IT), smem, stream, a);       \
        else                                                                                           \
            hipLaunchKernelGGL((im2col_tile_kernel<false, NC>), grid, dim3(IT), smem, stream, a);      \
    } while (0)
    hipError_t e = hipSuccess;
    if (smem > 64 * 1024) {
        // > 64 KiB of dynamic LDS needs the opt-in (set for every instantiation that can be launched below)
        const void *fns[] = {(const void *)&im2col_tile_kernel<true, 10>, (const void *)&im2col_tile_kernel<false, 10>,
                             (const void *)&im2col_tile_kernel<true, 12>, (const void *)&im2col_tile_kernel<false, 12>,
                             (const void *)&im2col_tile_kernel<true, 8>, (const void *)&im2col_tile_kernel<false, 8>,
                             (const void *)&im2col_tile_kernel<true, 6>, (const void *)&im2col_tile_kernel<false, 6>};
        for (const void *fn : fns)
            if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return fail(DDCN_ERR_LAUNCH, "im2col_tile: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    if (ncell <= 2) DDCN_IM2COL_LAUNCH(2);
    else if (ncell <= 4) DDCN_IM2COL_LAUNCH(4);
    else if (ncell <= 6) DDCN_IM2COL_LAUNCH(6);
    else if (ncell <= 8) DDCN_IM2COL_LAUNCH(8);
    else if (ncell <= 10) DDCN_IM2COL_LAUNCH(10);
    else DDCN_IM2COL_LAUNCH(12);
#undef DDCN_IM2COL_LAUNCH
    return check_launch("im2col_tile");
}

}  // namespace ddcn

// Tiled backward scatter of the DDCN operator: fused col2im + col2im_coord with the input-gradient
// accumulation privatised in LDS (gfx950, wave64).
//
// Same arithmetic as detseg/ops/dcn/src/deform_conv_cuda_kernel.cu K:278-334 / K:634-692 (col2im) and
// K:372-435 / K:694-766 (col2im_coord); see ddcn_gather.hip for the formula-by-formula citations.
//
// Work decomposition (MI355X-first, nothing like the reference's one-thread-per-column-element):
//   workgroup = (TH x TW tile of output pixels, group of CG input channels, image n), 256 threads;
//   thread    = one output pixel; it keeps the K2 sample positions of its pixel (h, w, mask) and the
//               3*K2 accumulators of d/d(offset_h), d/d(offset_w), d/d(mask) in registers for the whole
//               channel group -> grad_offset / grad_mask need no reduction inside the workgroup;
//   LDS       = gs[CC][RH x RW]: the grad_input footprint of the tile (tile extent * stride + kernel
//               extent * dilation + a margin M for the learned offsets) for CC channels at a time.
//               Bilinear scatter goes to LDS with ds_add_f32; samples whose corners leave the footprint
//               ("escapees") fall back to global atomics.  After each chunk the footprint is flushed with
//               one global atomic per touched cell -- neighbouring tiles overlap by the halo, so the flush
//               still has to be atomic, but it issues ~7x fewer global atomics than a direct scatter.
#include <stdlib.h>

#include "ddcn_common.h"

namespace ddcn {

constexpr int TT = 256;

struct TileArgs {
    const float *gcol, *x, *offset, *mask;
    float *gx, *goff, *gmask;
    ddcn_geom g;
    int P, cpg;
    long ld;
    int TH, TW, tiles_w;
    int CG, CC, M;
    int RH, RW, RS;
    int stepR, stepC;  // TT / RW, TT % RW
};

template <bool MOD>
__global__ __launch_bounds__(TT) void col2im_coord_tile_kernel(TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ddcn_geom &g = a.g;
    const int K2 = g.kh * g.kw;
    float *acc = smem;                       // [3*K2][TT]: d/d(off_h), d/d(off_w), d/d(mask) per tap, thread-private columns
    // [CC][RS] grad_input footprint of the tile, in DOUBLE: measured on gfx950 (tools/ubench_lds_atomic.hip),
    // ds_add_f32 retires ~1 lane per 3 clocks (195 cycles per wave-instruction) while ds_add_f64 runs at
    // ~10 cycles per wave-instruction -- the f64 atomic is 8-20x faster and also sums more accurately
    double *gs = reinterpret_cast<double *>(smem + 3 * K2 * TT);
    const int t = threadIdx.x;
    const int tyi = blockIdx.x / a.tiles_w, txi = blockIdx.x - tyi * a.tiles_w;
    const int ty = t / a.TW, tx = t - ty * a.TW;
    const int ho = tyi * a.TH + ty, wo = txi * a.TW + tx;
    const bool live = ty < a.TH && ho < g.Ho && wo < g.Wo;
    const int n = blockIdx.z;
    const int cg0 = blockIdx.y * a.CG;
    const int dgi = cg0 / a.cpg;
    const int rh0 = tyi * a.TH * g.sh - g.ph - a.M;   // image row of footprint row 0
    const int rw0 = txi * a.TW * g.sw - g.pw - a.M;
    const int HW = g.H * g.W;
    const int p = live ? ho * g.Wo + wo : 0;
    const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
    const float *off = a.offset + ((long)n * g.dg + dgi) * 2 * K2 * a.P + p;
    const float *msk = MOD ? a.mask + ((long)n * g.dg + dgi) * K2 * a.P + p : nullptr;

    for (int i = t; i < 3 * K2 * TT + 2 * a.CC * a.RS; i += TT) smem[i] = 0.f;
    const int fr0 = t / a.RW, fc0 = t - fr0 * a.RW;   // this thread's first footprint cell in the flush sweep
    __syncthreads();

    for (int c0 = cg0; c0 < cg0 + a.CG; c0 += a.CC) {
        const float *xb = a.x + ((long)n * g.C + c0) * HW;
        float *gxb = a.gx + ((long)n * g.C + c0) * HW;
        const float *gcb = a.gcol + ((long)n * g.C + c0) * K2 * a.ld + p;
        if (live) {
            int ki = 0, kj = 0;
            for (int k = 0; k < K2; ++k) {
                // sample position of (pixel, tap): recomputed per chunk (3 cached loads + ~30 VALU against
                // CC * ~35 instructions of channel work) instead of holding 9 taps of state in registers
                const float h = (float)(h_in + ki * g.dh) + off[(long)(2 * k) * a.P];
                const float w = (float)(w_in + kj * g.dw) + off[(long)(2 * k + 1) * a.P];
                const float m = MOD ? msk[(long)k * a.P] : 1.f;
                if (++kj == g.kw) { kj = 0; ++ki; }
                if (!(h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W)) continue;
                const int hl = (int)floorf(h), wl = (int)floorf(w);
                const bool ok1 = hl >= 0 && wl >= 0, ok2 = hl >= 0 && wl + 1 <= g.W - 1;
                const bool ok3 = hl + 1 <= g.H - 1 && wl >= 0, ok4 = hl + 1 <= g.H - 1 && wl + 1 <= g.W - 1;
                const int gi = hl * g.W + wl;                       // plane index of the (hl, wl) corner
                const int rr = hl - rh0, rc = wl - rw0;
                const bool inreg = rr >= 0 && rr + 1 < a.RH && rc >= 0 && rc + 1 < a.RW;
                const int ri = rr * a.RW + rc;
                const float fhl = (float)hl, fwl = (float)wl;
                const float gh_lo = fhl + 1.f - h, gh_hi = h + 1.f - (fhl + 1.f);
                const float gw_lo = fwl + 1.f - w, gw_hi = w + 1.f - (fwl + 1.f);
                const float s1 = gh_lo * gw_lo, s2 = gh_lo * gw_hi, s3 = gh_hi * gw_lo, s4 = gh_hi * gw_hi;
                const float lh = h - fhl, lw = w - fwl;
                const float hh = 1.f - lh, hw_ = 1.f - lw;
                const float b1 = hh * hw_, b2 = hh * lw, b3 = lh * hw_, b4 = lh * lw;
                const float cwl = fwl + 1.f - w, cwh = w - fwl;
                const float chl = fhl + 1.f - h, chh = h - fhl;
                float ah = 0.f, aw = 0.f, am = 0.f;
                const float *gck = gcb + (long)k * a.ld;
#pragma unroll 4
                for (int cc = 0; cc < a.CC; ++cc) {
                    // corner offsets are 0 for a corner outside the map: unconditional loads, one wait for the four
                    const float *im = xb + cc * HW;
                    const float gc = gck[(long)cc * K2 * a.ld];
                    const float r1 = im[ok1 ? gi : 0], r2 = im[ok2 ? gi + 1 : 0];
                    const float r3 = im[ok3 ? gi + g.W : 0], r4 = im[ok4 ? gi + g.W + 1 : 0];
                    const float v1 = ok1 ? r1 : 0.f;
                    const float v2 = ok2 ? r2 : 0.f;
                    const float v3 = ok3 ? r3 : 0.f;
                    const float v4 = ok4 ? r4 : 0.f;
                    float wh = 0.f, ww = 0.f;
                    wh += -1.f * cwl * v1; wh += -1.f * cwh * v2; wh += cwl * v3; wh += cwh * v4;
                    ww += -1.f * chl * v1; ww += chl * v2; ww += -1.f * chh * v3; ww += chh * v4;
                    float tg;
                    if (MOD) {
                        am += gc * (b1 * v1 + b2 * v2 + b3 * v3 + b4 * v4);
                        ah += wh * gc * m;
                        aw += ww * gc * m;
                        tg = gc * m;
                    } else {
                        ah += wh * gc;
                        aw += ww * gc;
                        tg = gc;
                    }
                    if (inreg) {
                        double *cell = gs + cc * a.RS + ri;
                        if (ok1) atomicAdd(cell, (double)(s1 * tg));
                        if (ok2) atomicAdd(cell + 1, (double)(s2 * tg));
                        if (ok3) atomicAdd(cell + a.RW, (double)(s3 * tg));
                        if (ok4) atomicAdd(cell + a.RW + 1, (double)(s4 * tg));
                    } else {
                        float *cell = gxb + cc * HW + gi;
                        if (ok1) atomicAdd(cell, s1 * tg);
                        if (ok2) atomicAdd(cell + 1, s2 * tg);
                        if (ok3) atomicAdd(cell + g.W, s3 * tg);
                        if (ok4) atomicAdd(cell + g.W + 1, s4 * tg);
                    }
                }
                float *ak = acc + (3 * k) * TT + t;   // thread-private: plain read-modify-write
                ak[0] += ah;
                ak[TT] += aw;
                if (MOD) ak[2 * TT] += am;
            }
        }
        __syncthreads();
        // flush the footprint: one global atomic per touched in-image cell, and re-zero for the next chunk
        for (int cc = 0; cc < a.CC; ++cc) {
            double *gsc = gs + cc * a.RS;
            float *gxc = gxb + cc * HW;
            int r = fr0, c = fc0;
            for (int i = t; i < a.RS; i += TT) {
                const double v = gsc[i];
                if (v != 0.0) {
                    gsc[i] = 0.0;
                    const int y = rh0 + r, xx = rw0 + c;
                    if (y >= 0 && y < g.H && xx >= 0 && xx < g.W) atomicAdd(gxc + y * g.W + xx, (float)v);
                }
                r += a.stepR; c += a.stepC;
                if (c >= a.RW) { c -= a.RW; r += 1; }
            }
        }
        __syncthreads();
    }

    if (live) {
        float *goffb = a.goff + ((long)n * g.dg + dgi) * 2 * K2 * a.P + p;
        float *gmaskb = MOD ? a.gmask + ((long)n * g.dg + dgi) * K2 * a.P + p : nullptr;
        for (int k = 0; k < K2; ++k) {
            // taps whose sample is invalid accumulated nothing: adding 0 keeps the cleared destination
            const float *ak = acc + (3 * k) * TT + t;
            const float vh = ak[0], vw = ak[TT];
            if (vh != 0.f) atomicAdd(goffb + (long)(2 * k) * a.P, vh);
            if (vw != 0.f) atomicAdd(goffb + (long)(2 * k + 1) * a.P, vw);
            if (MOD) {
                const float vm = ak[2 * TT];
                if (vm != 0.f) atomicAdd(gmaskb + (long)k * a.P, vm);
            }
        }
    }
}

// Picks tile / footprint / chunk sizes; returns false when this geometry should use the generic kernel.
static bool plan_tiles(const ddcn_geom &g, TileArgs &a)
{
    const Dims d = dims_of(g);
    if (d.K2 > 9) return false;   // accumulators: 3*K2 KiB of LDS
    if (g.H >= 32766 || g.W >= 32766) return false;
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
    int TH = TT / TW;
    if (TH > g.Ho) TH = g.Ho;
    a.TW = TW; a.TH = TH;
    a.tiles_w = (g.Wo + TW - 1) / TW;
    const int tiles = a.tiles_w * ((g.Ho + TH - 1) / TH);
    // channel group per workgroup: enough workgroups to fill 256 CUs a few times over
    int CG = d.cpg;
    while (CG % 2 == 0 && (long)tiles * g.N * (g.C / CG) < 1024 && CG > 8) CG /= 2;
    a.CG = CG;
    const size_t acc_bytes = (size_t)3 * d.K2 * TT * sizeof(float);
    static const int env_kb = getenv("DDCN_TILE_LDS_KB") ? atoi(getenv("DDCN_TILE_LDS_KB")) : 0;   // tuning knob
    static const int env_m = getenv("DDCN_TILE_MARGIN") ? atoi(getenv("DDCN_TILE_MARGIN")) : 6;
    const size_t lds_cap = (size_t)(env_kb > 0 ? env_kb : 76) * 1024;   // default: two workgroups per CU
    for (int M = env_m; M >= 0; M -= 2) {
        const int RH = (TH - 1) * g.sh + (g.kh - 1) * g.dh + 2 + 2 * M;
        const int RW = (TW - 1) * g.sw + (g.kw - 1) * g.dw + 2 + 2 * M;
        for (int CC = 8; CC >= 1; CC /= 2) {
            if (CG % CC) continue;
            if (acc_bytes + (size_t)CC * RH * RW * sizeof(double) <= lds_cap && (CC >= 4 || M == 0 || CG < 4)) {
                a.M = M; a.RH = RH; a.RW = RW; a.RS = RH * RW; a.CC = CC;
                a.stepR = TT / RW; a.stepC = TT % RW;
                return true;
            }
        }
    }
    return false;
}

int launch_col2im_coord_tiled(const float *gcol, const float *x, const float *offset, const float *mask, float *gx,
                              float *goff, float *gmask, const ddcn_geom &g, hipStream_t stream, bool *handled)
{
    TileArgs a{};
    *handled = false;
    if (!plan_tiles(g, a)) return DDCN_OK;
    *handled = true;
    const Dims d = dims_of(g);
    a.gcol = gcol; a.x = x; a.offset = offset; a.mask = mask; a.gx = gx; a.goff = goff; a.gmask = gmask;
    a.g = g; a.P = d.P; a.cpg = d.cpg; a.ld = d.ld;
    dim3 grid(a.tiles_w * ((g.Ho + a.TH - 1) / a.TH), g.C / a.CG, g.N);
    if (grid.y > 65535 || grid.z > 65535) return fail(DDCN_ERR_UNSUPPORTED, "col2im_coord_tile: grid too large");
    const size_t smem = (size_t)(3 * d.K2 * TT + 2 * a.CC * a.RS) * sizeof(float);
    const void *fn = mask ? reinterpret_cast<const void *>(&col2im_coord_tile_kernel<true>)
                          : reinterpret_cast<const void *>(&col2im_coord_tile_kernel<false>);
    // raised once per device to the planner's cap (plan_tiles never exceeds it), not per call to this call's size
    static const int env_kb = getenv("DDCN_TILE_LDS_KB") ? atoi(getenv("DDCN_TILE_LDS_KB")) : 0;
    const int lds_limit = (env_kb > 0 ? env_kb : 76) * 1024;
    static std::atomic<uint64_t> raised_mod{0}, raised_plain{0};
    if ((int)smem > lds_limit || !raise_lds_limit(fn, lds_limit, mask ? raised_mod : raised_plain))
        return fail(DDCN_ERR_LAUNCH, "col2im_coord_tile: cannot raise the dynamic LDS limit to %d bytes", lds_limit);
    if (mask)
        hipLaunchKernelGGL((col2im_coord_tile_kernel<true>), grid, dim3(TT), smem, stream, a);
    else
        hipLaunchKernelGGL((col2im_coord_tile_kernel<false>), grid, dim3(TT), smem, stream, a);
    return check_launch("col2im_coord_tile");
}

}  // namespace ddcn

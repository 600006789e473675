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

template <int K2, bool MOD>
__global__ __launch_bounds__(TT) void col2im_coord_tile_kernel(TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float gs[];
    const ddcn_geom &g = a.g;
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

    // ---- per-tap sample state, kept for the whole channel group
    int hw[K2];          // ((hl + 1) << 16) | (wl + 1), or -1 when the sample is outside (-1,H)x(-1,W)
    float sh_[K2], sw_[K2], sm_[K2];
    float acc_h[K2], acc_w[K2], acc_m[K2];
    {
        const float *off = a.offset + ((long)n * g.dg + dgi) * 2 * K2 * a.P + p;
        const float *msk = MOD ? a.mask + ((long)n * g.dg + dgi) * K2 * a.P + p : nullptr;
        const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            acc_h[k] = acc_w[k] = acc_m[k] = 0.f;
            hw[k] = -1; sh_[k] = sw_[k] = 0.f; sm_[k] = 1.f;
            if (live) {
                const int i = k / g.kw, j = k - i * g.kw;
                const float h = (float)(h_in + i * g.dh) + off[(long)(2 * k) * a.P];
                const float w = (float)(w_in + j * g.dw) + off[(long)(2 * k + 1) * a.P];
                if (MOD) sm_[k] = msk[(long)k * a.P];
                if (h > -1.f && w > -1.f && h < (float)g.H && w < (float)g.W) {
                    const int hl = (int)floorf(h), wl = (int)floorf(w);
                    hw[k] = ((hl + 1) << 16) | (wl + 1);
                    sh_[k] = h; sw_[k] = w;
                }
            }
        }
    }

    for (int i = t; i < a.CC * a.RS; i += TT) gs[i] = 0.f;
    const int fr0 = t / a.RW, fc0 = t - fr0 * a.RW;   // this thread's first footprint cell in the flush sweep
    __syncthreads();

    for (int c0 = cg0; c0 < cg0 + a.CG; c0 += a.CC) {
        const float *xb = a.x + ((long)n * g.C + c0) * HW;
        float *gxb = a.gx + ((long)n * g.C + c0) * HW;
        const float *gcb = a.gcol + ((long)n * g.C + c0) * K2 * a.ld + p;
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            if (hw[k] >= 0) {
                const int hl = (hw[k] >> 16) - 1, wl = (hw[k] & 0xffff) - 1;
                float h = sh_[k], w = sw_[k];
                const float m = sm_[k];
                int kk = k;
                // keep the per-tap weights from being hoisted out of the channel-group loop (9 taps x ~20
                // loop-invariant registers would spill): make their inputs opaque once per chunk
                asm volatile("" : "+v"(h), "+v"(w), "+s"(kk));
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
#pragma unroll 2
                for (int cc = 0; cc < a.CC; ++cc) {
                    const float *im = xb + cc * HW + gi;
                    const float gc = gcb[((long)cc * K2 + kk) * a.ld];
                    const float v1 = ok1 ? im[0] : 0.f;
                    const float v2 = ok2 ? im[1] : 0.f;
                    const float v3 = ok3 ? im[g.W] : 0.f;
                    const float v4 = ok4 ? im[g.W + 1] : 0.f;
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
                        float *cell = gs + cc * a.RS + ri;
                        if (ok1) atomicAdd(cell, s1 * tg);
                        if (ok2) atomicAdd(cell + 1, s2 * tg);
                        if (ok3) atomicAdd(cell + a.RW, s3 * tg);
                        if (ok4) atomicAdd(cell + a.RW + 1, s4 * tg);
                    } else {
                        float *cell = gxb + cc * HW + gi;
                        if (ok1) atomicAdd(cell, s1 * tg);
                        if (ok2) atomicAdd(cell + 1, s2 * tg);
                        if (ok3) atomicAdd(cell + g.W, s3 * tg);
                        if (ok4) atomicAdd(cell + g.W + 1, s4 * tg);
                    }
                }
                acc_h[k] += ah; acc_w[k] += aw; acc_m[k] += am;
            }
        }
        __syncthreads();
        // flush the footprint: one global atomic per touched in-image cell, and re-zero for the next chunk
        for (int cc = 0; cc < a.CC; ++cc) {
            float *gsc = gs + cc * a.RS;
            float *gxc = gxb + cc * HW;
            int r = fr0, c = fc0;
            for (int i = t; i < a.RS; i += TT) {
                const float v = gsc[i];
                if (v != 0.f) {
                    gsc[i] = 0.f;
                    const int y = rh0 + r, xx = rw0 + c;
                    if (y >= 0 && y < g.H && xx >= 0 && xx < g.W) atomicAdd(gxc + y * g.W + xx, v);
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
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            // invalid samples contribute exactly 0 (K:422-425, K:746-749); destination was cleared by the launcher
            if (hw[k] >= 0) {
                atomicAdd(goffb + (long)(2 * k) * a.P, acc_h[k]);
                atomicAdd(goffb + (long)(2 * k + 1) * a.P, acc_w[k]);
                if (MOD) atomicAdd(gmaskb + (long)k * a.P, acc_m[k]);
            }
        }
    }
}

// Picks tile / footprint / chunk sizes; returns false when this geometry should use the generic kernel.
static bool plan_tiles(const ddcn_geom &g, TileArgs &a)
{
    const Dims d = dims_of(g);
    if (d.K2 != 9 || g.kh != 3) return false;
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
    const size_t lds_cap = 48 * 1024;
    for (int M = 6; M >= 0; M -= 2) {
        const int RH = (TH - 1) * g.sh + (g.kh - 1) * g.dh + 2 + 2 * M;
        const int RW = (TW - 1) * g.sw + (g.kw - 1) * g.dw + 2 + 2 * M;
        for (int CC = 8; CC >= 1; CC /= 2) {
            if (CG % CC) continue;
            if ((size_t)CC * RH * RW * sizeof(float) <= lds_cap && (CC >= 4 || M == 0 || CG < 4)) {
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
    const size_t smem = (size_t)a.CC * a.RS * sizeof(float);
    if (mask)
        hipLaunchKernelGGL((col2im_coord_tile_kernel<9, true>), grid, dim3(TT), smem, stream, a);
    else
        hipLaunchKernelGGL((col2im_coord_tile_kernel<9, false>), grid, dim3(TT), smem, stream, a);
    return check_launch("col2im_coord_tile");
}

}  // namespace ddcn

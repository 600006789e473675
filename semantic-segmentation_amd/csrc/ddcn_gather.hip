// Bilinear gather / scatter kernels of the DDCN operator (gfx950, wave64).
//
// Arithmetic follows detseg/ops/dcn/src/deform_conv_cuda_kernel.cu (K below):
//   sample position  h = (ho*sh - ph + i*dh) + off_h   (int part first, then + offset, K:226-227)
//   sample valid iff h > -1 && w > -1 && h < H && w < W (K:228); corners guarded as K:97-108
//   scatter weights  K:116-142, coordinate weights K:144-187, mask gradient K:750-753.
// Unlike the reference (one thread per (c, pixel) recomputing every coordinate, one launch per
// image) each thread here owns a (pixel, tap) pair, derives the four corner indices/weights once
// and reuses them for a whole chunk of channels; all images are one launch.
#include "ddcn_common.h"

namespace ddcn {

constexpr int GT = 256;  // threads per workgroup

struct Sample {
    int i1, i2, i3, i4;      // corner offsets inside one H*W plane (clamped to 0 when the corner is unused)
    bool ok1, ok2, ok3, ok4; // corner contributes
    bool valid;              // sample inside (-1, H) x (-1, W)
    float h, w;
    int hl, wl;
};

__device__ __forceinline__ Sample make_sample(float h, float w, int H, int W)
{
    Sample s;
    s.h = h; s.w = w;
    s.valid = (h > -1.f && w > -1.f && h < (float)H && w < (float)W);
    const int hl = (int)floorf(h), wl = (int)floorf(w);
    const int hh = hl + 1, wh = wl + 1;
    s.hl = hl; s.wl = wl;
    s.ok1 = s.valid && hl >= 0 && wl >= 0;
    s.ok2 = s.valid && hl >= 0 && wh <= W - 1;
    s.ok3 = s.valid && hh <= H - 1 && wl >= 0;
    s.ok4 = s.valid && hh <= H - 1 && wh <= W - 1;
    s.i1 = s.ok1 ? hl * W + wl : 0;
    s.i2 = s.ok2 ? hl * W + wh : 0;
    s.i3 = s.ok3 ? hh * W + wl : 0;
    s.i4 = s.ok4 ? hh * W + wh : 0;
    return s;
}

struct GatherArgs {
    const float *x, *offset, *mask, *gcol, *depth;
    float *col, *gx, *goff, *gmask;
    ddcn_geom g;
    int K2, P, cpg, CH;  // CH: channels per workgroup chunk (divides cpg)
    long ld;
};

// ------------------------------------------------------------------------------------------ im2col
// grid: (ceil(P/256), C/CH, N).  thread = pixel; loops taps then the CH channels of the chunk.
__global__ __launch_bounds__(GT) void im2col_kernel(GatherArgs a)
{
    const ddcn_geom &g = a.g;
    const int p = blockIdx.x * GT + threadIdx.x;
    if (p >= a.P) return;
    const int c0 = blockIdx.y * a.CH;
    const int n = blockIdx.z;
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    const int dgi = c0 / a.cpg;
    const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
    const long HW = (long)g.H * g.W;
    const float *off = a.offset + ((long)n * g.dg + dgi) * 2 * a.K2 * a.P + p;
    const float *msk = a.mask ? a.mask + ((long)n * g.dg + dgi) * a.K2 * a.P + p : nullptr;
    const float *xb = a.x + ((long)n * g.C + c0) * HW;
    float *colb = a.col + ((long)n * g.C * a.K2 + (long)c0 * a.K2) * a.ld + p;

    for (int k = 0; k < a.K2; ++k) {
        const int i = k / g.kw, j = k - i * g.kw;
        const float oh = off[(long)(2 * k) * a.P];
        const float ow = off[(long)(2 * k + 1) * a.P];
        const float m = msk ? msk[(long)k * a.P] : 1.f;
        const float h = (float)(h_in + i * g.dh) + oh;
        const float w = (float)(w_in + j * g.dw) + ow;
        const Sample s = make_sample(h, w, g.H, g.W);
        const float lh = h - (float)s.hl, lw = w - (float)s.wl;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
        for (int c = 0; c < a.CH; ++c) {
            const float *im = xb + c * HW;
            // i1..i4 are 0 for a corner outside the map: unconditional loads, one wait for the four (a load under a
            // condition becomes a branch with a wait of its own)
            const float r1 = im[s.i1], r2 = im[s.i2], r3 = im[s.i3], r4 = im[s.i4];
            const float v1 = s.ok1 ? r1 : 0.f;
            const float v2 = s.ok2 ? r2 : 0.f;
            const float v3 = s.ok3 ? r3 : 0.f;
            const float v4 = s.ok4 ? r4 : 0.f;
            float val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
            if (msk) val = val * m;
            colb[((long)c * a.K2 + k) * a.ld] = val;
        }
    }
}

// ------------------------------------------------------------------------- fused col2im + col2im_coord
// grid: (ceil(P/64), C/CH, N).  256 threads = 64 pixels x 4 channel slices of the chunk (one wave each).
// Per (pixel, tap): corner data once, then for every channel of the slice: scatter into gx (atomics) and
// accumulate d/d(offset), d/d(mask) in registers; slices are combined in LDS, chunks with atomics.
__global__ __launch_bounds__(GT) void col2im_coord_kernel(GatherArgs a)
{
    const ddcn_geom &g = a.g;
    __shared__ float red[3][4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    const bool live = p < a.P;
    const int c0 = blockIdx.y * a.CH;
    const int n = blockIdx.z;
    const int pp = live ? p : 0;
    const int ho = pp / g.Wo, wo = pp - ho * g.Wo;
    const int dgi = c0 / a.cpg;
    const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
    const long HW = (long)g.H * g.W;
    const float *off = a.offset + ((long)n * g.dg + dgi) * 2 * a.K2 * a.P + pp;
    const float *msk = a.mask ? a.mask + ((long)n * g.dg + dgi) * a.K2 * a.P + pp : nullptr;
    // channels of the chunk are dealt round-robin to the 4 slices (works for any chunk size)
    const float *xb = a.x + ((long)n * g.C + c0) * HW;
    float *gxb = a.gx + ((long)n * g.C + c0) * HW;
    const float *gcb = a.gcol + ((long)n * g.C * a.K2 + (long)c0 * a.K2) * a.ld + pp;
    float *goffb = a.goff + ((long)n * g.dg + dgi) * 2 * a.K2 * a.P + pp;
    float *gmaskb = a.gmask ? a.gmask + ((long)n * g.dg + dgi) * a.K2 * a.P + pp : nullptr;

    for (int k = 0; k < a.K2; ++k) {
        const int i = k / g.kw, j = k - i * g.kw;
        float acc_h = 0.f, acc_w = 0.f, acc_m = 0.f;
        if (live) {
            const float oh = off[(long)(2 * k) * a.P];
            const float ow = off[(long)(2 * k + 1) * a.P];
            const float m = msk ? msk[(long)k * a.P] : 1.f;
            const float h = (float)(h_in + i * g.dh) + oh;
            const float w = (float)(w_in + j * g.dw) + ow;
            const Sample s = make_sample(h, w, g.H, g.W);
            if (s.valid) {
                const float fhl = (float)s.hl, fwl = (float)s.wl;
                // scatter weights, K:133-140 with (h, w) the corner and argmax the sample
                const float gh_lo = fhl + 1.f - h, gh_hi = h + 1.f - (fhl + 1.f);
                const float gw_lo = fwl + 1.f - w, gw_hi = w + 1.f - (fwl + 1.f);
                const float s1 = gh_lo * gw_lo, s2 = gh_lo * gw_hi, s3 = gh_hi * gw_lo, s4 = gh_hi * gw_hi;
                // bilinear weights for the mask gradient, K:93-110
                const float lh = h - fhl, lw = w - fwl;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const float b1 = hh * hw, b2 = hh * lw, b3 = lh * hw, b4 = lh * lw;
                // coordinate weights, K:163-184
                const float cwl = fwl + 1.f - w, cwh = w - fwl;   // dir 0 (d/dh)
                const float chl = fhl + 1.f - h, chh = h - fhl;   // dir 1 (d/dw)
                for (int c = slice; c < a.CH; c += 4) {
                    const float *im = xb + c * HW;
                    const float gc = gcb[((long)c * a.K2 + k) * a.ld];
                    const float r1 = im[s.i1], r2 = im[s.i2], r3 = im[s.i3], r4 = im[s.i4];   // i* = 0 when outside
                    const float v1 = s.ok1 ? r1 : 0.f;
                    const float v2 = s.ok2 ? r2 : 0.f;
                    const float v3 = s.ok3 ? r3 : 0.f;
                    const float v4 = s.ok4 ? r4 : 0.f;
                    float wh_ = 0.f, ww_ = 0.f;
                    wh_ += -1.f * cwl * v1; wh_ += -1.f * cwh * v2; wh_ += cwl * v3; wh_ += cwh * v4;
                    ww_ += -1.f * chl * v1; ww_ += chl * v2; ww_ += -1.f * chh * v3; ww_ += chh * v4;
                    if (msk) {
                        acc_m += gc * (b1 * v1 + b2 * v2 + b3 * v3 + b4 * v4);
                        acc_h += wh_ * gc * m;
                        acc_w += ww_ * gc * m;
                    } else {
                        acc_h += wh_ * gc;
                        acc_w += ww_ * gc;
                    }
                    const float tg = msk ? gc * m : gc;
                    float *gim = gxb + c * HW;
                    if (s.ok1) atomicAdd(gim + s.i1, s1 * tg);
                    if (s.ok2) atomicAdd(gim + s.i2, s2 * tg);
                    if (s.ok3) atomicAdd(gim + s.i3, s3 * tg);
                    if (s.ok4) atomicAdd(gim + s.i4, s4 * tg);
                }
            }
        }
        red[0][slice][lane] = acc_h;
        red[1][slice][lane] = acc_w;
        red[2][slice][lane] = acc_m;
        __syncthreads();
        if (slice < 3 && live) {
            const float v = red[slice][0][lane] + red[slice][1][lane] + red[slice][2][lane] + red[slice][3][lane];
            if (slice == 0) atomicAdd(goffb + (long)(2 * k) * a.P, v);
            if (slice == 1) atomicAdd(goffb + (long)(2 * k + 1) * a.P, v);
            if (slice == 2 && gmaskb) atomicAdd(gmaskb + (long)k * a.P, v);
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------- depth-aware gather
// K:868-918.  grid: (ceil(P/256), C/CH, N); thread = pixel.
__global__ __launch_bounds__(GT) void depthaware_im2col_kernel(GatherArgs a)
{
    const ddcn_geom &g = a.g;
    const int p = blockIdx.x * GT + threadIdx.x;
    if (p >= a.P) return;
    const int c0 = blockIdx.y * a.CH;
    const int n = blockIdx.z;
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
    const long HW = (long)g.H * g.W;
    const float *dp = a.depth + (long)n * HW;
    const float *xb = a.x + ((long)n * g.C + c0) * HW;
    float *colb = a.col + ((long)n * g.C * a.K2 + (long)c0 * a.K2) * a.ld + p;
    const int hc = h_in + g.dh * (g.kh - 1) / 2, wc = w_in + g.dw * (g.kw - 1) / 2;
    const bool valid = hc >= 0 && wc >= 0 && hc < g.H && wc < g.W;
    const float Di = valid ? dp[hc * g.W + wc] : 0.f;
    for (int k = 0; k < a.K2; ++k) {
        const int i = k / g.kw, j = k - i * g.kw;
        const int h_im = h_in + i * g.dh, w_im = w_in + j * g.dw;
        const bool inb = h_im >= 0 && w_im >= 0 && h_im < g.H && w_im < g.W;
        const int idx = inb ? h_im * g.W + w_im : 0;
        const float Dval = (inb && valid) ? dp[idx] : 0.f;
        const float sim = expf(-fabsf(Di - Dval));
        for (int c = 0; c < a.CH; ++c) {
            const float val = inb ? xb[c * HW + idx] : 0.f;
            colb[((long)c * a.K2 + k) * a.ld] = val * sim;
        }
    }
}

// K:952-996 as a GATHER.  The reference scatters col * exp(-|Di - Dval|) to gI[h_im, w_im] with one atomic per
// (channel, tap, pixel).  The taps sit on the integer grid, so every input cell (y, x) knows which output pixel reaches
// it through tap (i, j): ho = (y + ph - i*dh) / sh when that division is exact and in range (same for wo).  One lane owns
// one input cell: it resolves its <= K2 (pixel, similarity) pairs once, sums the column gradients of DA_CH channels in
// registers in ascending tap order and adds each sum to grad_input with a plain read-modify-write -- no atomics, no
// contention, deterministic.  grid: (ceil(H*W/256), C/DA_CH, N).
constexpr int DA_CH = 8;
__global__ __launch_bounds__(GT) void depthaware_col2im_kernel(GatherArgs a)
{
    const ddcn_geom &g = a.g;
    const int cell = blockIdx.x * GT + threadIdx.x;
    const int HWi = g.H * g.W;
    if (cell >= HWi) return;
    const int c0 = blockIdx.y * DA_CH;
    const int n = blockIdx.z;
    const int y = cell / g.W, x = cell - y * g.W;
    const float *dp = a.depth + (long)n * HWi;
    const float Dcell = dp[cell];
    const float *gcb = a.gcol + ((long)n * g.C + c0) * a.K2 * a.ld;
    const int nch = min(DA_CH, g.C - c0);
    float acc[DA_CH];
#pragma unroll
    for (int c = 0; c < DA_CH; ++c) acc[c] = 0.f;
    for (int k = 0; k < a.K2; ++k) {
        const int i = k / g.kw, j = k - i * g.kw;
        const int th = y + g.ph - i * g.dh, tw = x + g.pw - j * g.dw;
        if (th < 0 || tw < 0) continue;
        const int ho = th / g.sh, wo = tw / g.sw;
        if (ho * g.sh != th || wo * g.sw != tw || ho >= g.Ho || wo >= g.Wo) continue;
        // similarity of output pixel (ho, wo) and this tap, K:970-986: centre outside the map -> both depths read as 0
        const int hc = ho * g.sh - g.ph + g.dh * (g.kh - 1) / 2, wc = wo * g.sw - g.pw + g.dw * (g.kw - 1) / 2;
        const bool valid = hc >= 0 && wc >= 0 && hc < g.H && wc < g.W;
        const float Di = valid ? dp[hc * g.W + wc] : 0.f;
        const float Dval = valid ? Dcell : 0.f;
        const float sim = expf(-fabsf(Di - Dval));
        const float *src = gcb + (long)k * a.ld + (ho * g.Wo + wo);
#pragma unroll
        for (int c = 0; c < DA_CH; ++c)
            if (c < nch) acc[c] += src[(long)c * a.K2 * a.ld] * sim;
    }
    float *dst = a.gx + ((long)n * g.C + c0) * HWi + cell;
#pragma unroll
    for (int c = 0; c < DA_CH; ++c)
        if (c < nch) dst[(long)c * HWi] += acc[c];
}

// ---------------------------------------------------------------------------------------- bias grad
// gb[co] += sum_{n,p} gout[n][co][p]  (H:347-353: gout * ones).  grid: Co workgroups.
__global__ __launch_bounds__(GT) void bias_grad_kernel(const float *gout, float *gb, int N, int Co, int P)
{
    __shared__ float part[GT / 64];
    const int co = blockIdx.x;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) {
        const float *src = gout + ((long)n * Co + co) * P;
        for (int p = threadIdx.x; p < P; p += GT) acc += src[p];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = 0.f;
        for (int i = 0; i < GT / 64; ++i) v += part[i];
        gb[co] += v;
    }
}

// ------------------------------------------------------------------------------------------ launchers
static int pick_chunk(int cpg, int want, int multiple_of)
{
    // largest divisor of cpg that is <= want and a multiple of `multiple_of` (falls back to multiple_of==1)
    for (int ch = want; ch >= 1; --ch)
        if (cpg % ch == 0 && ch % multiple_of == 0) return ch;
    return 0;
}

static GatherArgs base_args(const ddcn_geom &g)
{
    GatherArgs a{};
    const Dims d = dims_of(g);
    a.g = g; a.K2 = d.K2; a.P = d.P; a.cpg = d.cpg; a.ld = d.ld;
    return a;
}

int launch_im2col(const float *x, const float *offset, const float *mask, float *col, const ddcn_geom &g,
                  hipStream_t stream)
{
    bool handled = false;
    const int rc = launch_im2col_tiled(x, offset, mask, col, g, stream, &handled);
    if (handled || rc != DDCN_OK) return rc;
    // generic path (input patch of one tile too large for LDS)
    GatherArgs a = base_args(g);
    a.x = x; a.offset = offset; a.mask = mask; a.col = col;
    a.CH = pick_chunk(a.cpg, 16, 1);
    dim3 grid((a.P + GT - 1) / GT, g.C / a.CH, g.N);
    if (grid.y > 65535 || grid.z > 65535) return fail(DDCN_ERR_UNSUPPORTED, "im2col: grid too large");
    hipLaunchKernelGGL(im2col_kernel, grid, dim3(GT), 0, stream, a);
    return check_launch("im2col");
}

int launch_col2im_coord(const float *gcol, const float *x, const float *offset, const float *mask, float *gx,
                        float *goff, float *gmask, const ddcn_geom &g, hipStream_t stream, bool clear_coord,
                        int chunk_channels)
{
    GatherArgs a = base_args(g);
    a.gcol = gcol; a.x = x; a.offset = offset; a.mask = mask; a.gx = gx; a.goff = goff; a.gmask = gmask;
    // goff / gmask are sums over channel chunks: clear, then accumulate (plain-store semantics of K:433, K:761-764)
    // (the whole-op backward entry points leave the clearing to the caller, who zero-fills every gradient anyway)
    hipError_t e = hipSuccess;
    if (clear_coord) e = hipMemsetAsync(goff, 0, sizeof(float) * (size_t)g.N * g.dg * 2 * a.K2 * a.P, stream);
    if (clear_coord && e == hipSuccess && gmask)
        e = hipMemsetAsync(gmask, 0, sizeof(float) * (size_t)g.N * g.dg * a.K2 * a.P, stream);
    if (e != hipSuccess) return fail(DDCN_ERR_LAUNCH, "col2im_coord: memset: %s", hipGetErrorString(e));
    bool handled = false;
    int rc = launch_col2im_coord_tile9(gcol, x, offset, mask, gx, goff, gmask, g, stream, &handled, chunk_channels);
    if (handled || rc != DDCN_OK) return rc;
    rc = launch_col2im_coord_tiled(gcol, x, offset, mask, gx, goff, gmask, g, stream, &handled);
    if (handled || rc != DDCN_OK) return rc;
    // generic path (any kernel size / footprint too large for LDS)
    a.CH = pick_chunk(a.cpg, 32, 1);
    dim3 grid((a.P + 63) / 64, g.C / a.CH, g.N);
    if (grid.y > 65535 || grid.z > 65535) return fail(DDCN_ERR_UNSUPPORTED, "col2im_coord: grid too large");
    hipLaunchKernelGGL(col2im_coord_kernel, grid, dim3(GT), 0, stream, a);
    return check_launch("col2im_coord");
}

int launch_depthaware_im2col(const float *x, const float *depth, float *col, const ddcn_geom &g,
                             hipStream_t stream)
{
    GatherArgs a = base_args(g);
    a.x = x; a.depth = depth; a.col = col;
    a.CH = pick_chunk(g.C, 16, 1);
    dim3 grid((a.P + GT - 1) / GT, g.C / a.CH, g.N);
    if (grid.y > 65535 || grid.z > 65535) return fail(DDCN_ERR_UNSUPPORTED, "depthaware_im2col: grid too large");
    hipLaunchKernelGGL(depthaware_im2col_kernel, grid, dim3(GT), 0, stream, a);
    return check_launch("depthaware_im2col");
}

int launch_depthaware_col2im(const float *gcol, const float *depth, float *gx, const ddcn_geom &g,
                             hipStream_t stream)
{
    GatherArgs a = base_args(g);
    a.gcol = gcol; a.depth = depth; a.gx = gx;
    const long cells = (long)g.H * g.W;
    dim3 grid((unsigned)((cells + GT - 1) / GT), (g.C + DA_CH - 1) / DA_CH, g.N);
    if (grid.y > 65535 || grid.z > 65535) return fail(DDCN_ERR_UNSUPPORTED, "depthaware_col2im: grid too large");
    hipLaunchKernelGGL(depthaware_col2im_kernel, grid, dim3(GT), 0, stream, a);
    return check_launch("depthaware_col2im");
}

int launch_bias_grad(const float *gout, float *gb, const ddcn_geom &g, hipStream_t stream)
{
    hipLaunchKernelGGL(bias_grad_kernel, dim3(g.Co), dim3(GT), 0, stream, gout, gb, g.N, g.Co, g.Ho * g.Wo);
    return check_launch("bias_grad");
}

}  // namespace ddcn

// fp64 instantiation of the DDCN operator for gfx950.
//
// The reference dispatches its kernels on double as well (AT_DISPATCH_FLOATING_TYPES_AND_HALF, K:258) and its own
// GPU self-test is an fp64 gradcheck (detseg/ops/dcn/test.py:208-267).  This file serves that use: validation in
// double precision.  It is written for clarity, not tuned like the fp32 / bf16 paths -- one lane per output element
// where the reference has one, a plain LDS-tiled FMA GEMM -- but it keeps the library's rules: all images in one
// launch per stage, no allocation, no synchronisation, caller-owned workspace, and it restates the same arithmetic:
//   im2col        K:189-242 / K:569-632 (bilinear helper K:83-114)
//   col2im        K:278-334 / K:634-692: each column element scatters to its (up to) four corners -- the same
//                 cells and weights the reference finds by scanning a 5x5 window with get_gradient_weight K:116-142
//   col2im_coord  K:372-435 / K:694-766: one lane per (image, deformable group, tap, pixel), a loop over the
//                 group's channels, no atomics; coordinate weights K:144-187, mask gradient K:750-753
//   depth-aware   K:868-918 / K:952-996
//   GEMMs         H:232-237, H:305-308, H:341-346 (+ bias rows H:347-353)
#include "ddcn_common.h"

namespace ddcn {
namespace {

constexpr int FT = 256;

struct F64Args {
    const double *x, *offset, *mask, *depth, *gcol;
    double *col, *gx, *goff, *gmask;
    ddcn_geom g;
    int K2, P, cpg;
    long ld;
};

struct Corner {
    bool valid, ok1, ok2, ok3, ok4;
    int hl, wl;
    double lh, lw;
};

__device__ __forceinline__ Corner corners(double h, double w, int H, int W)
{
    Corner c;
    c.valid = h > -1.0 && w > -1.0 && h < (double)H && w < (double)W;
    c.hl = (int)floor(h);
    c.wl = (int)floor(w);
    c.lh = h - c.hl;
    c.lw = w - c.wl;
    c.ok1 = c.valid && c.hl >= 0 && c.wl >= 0;
    c.ok2 = c.valid && c.hl >= 0 && c.wl + 1 <= W - 1;
    c.ok3 = c.valid && c.hl + 1 <= H - 1 && c.wl >= 0;
    c.ok4 = c.valid && c.hl + 1 <= H - 1 && c.wl + 1 <= W - 1;
    return c;
}

// sample position of (pixel p, tap k) of image n, deformable group dgi
__device__ __forceinline__ void position(const F64Args &a, int n, int dgi, int k, int p, double &h, double &w)
{
    const ddcn_geom &g = a.g;
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    const int i = k / g.kw, j = k - i * g.kw;
    const double *off = a.offset + ((long)n * g.dg + dgi) * 2 * a.K2 * a.P + p;
    h = (double)(ho * g.sh - g.ph + i * g.dh) + off[(long)(2 * k) * a.P];
    w = (double)(wo * g.sw - g.pw + j * g.dw) + off[(long)(2 * k + 1) * a.P];
}

__device__ __forceinline__ double mask_of(const F64Args &a, int n, int dgi, int k, int p)
{
    return a.mask ? a.mask[(((long)n * a.g.dg + dgi) * a.K2 + k) * a.P + p] : 1.0;
}

// one lane per column element (c, k, p); grid = (ceil(C*K2*P / FT), 1, N)
__global__ __launch_bounds__(FT) void im2col_f64_kernel(const F64Args a)
{
    const ddcn_geom &g = a.g;
    const long e = (long)blockIdx.x * FT + threadIdx.x;
    if (e >= (long)g.C * a.K2 * a.P) return;
    const int n = blockIdx.z;
    const int p = (int)(e % a.P), k = (int)(e / a.P % a.K2), c = (int)(e / ((long)a.P * a.K2));
    const int dgi = c / a.cpg;
    double h, w;
    position(a, n, dgi, k, p, h, w);
    const Corner s = corners(h, w, g.H, g.W);
    const double *im = a.x + ((long)n * g.C + c) * g.H * g.W + (long)s.hl * g.W + s.wl;
    const double v1 = s.ok1 ? im[0] : 0.0, v2 = s.ok2 ? im[1] : 0.0;
    const double v3 = s.ok3 ? im[g.W] : 0.0, v4 = s.ok4 ? im[g.W + 1] : 0.0;
    const double hh = 1.0 - s.lh, hw = 1.0 - s.lw;
    double val = hh * hw * v1 + hh * s.lw * v2 + s.lh * hw * v3 + s.lh * s.lw * v4;
    if (a.mask) val *= mask_of(a, n, dgi, k, p);
    a.col[(((long)n * g.C + c) * a.K2 + k) * a.ld + p] = val;
}

// one lane per column-gradient element (c, k, p): scatter to the four corners
__global__ __launch_bounds__(FT) void col2im_f64_kernel(const F64Args a)
{
    const ddcn_geom &g = a.g;
    const long e = (long)blockIdx.x * FT + threadIdx.x;
    if (e >= (long)g.C * a.K2 * a.P) return;
    const int n = blockIdx.z;
    const int p = (int)(e % a.P), k = (int)(e / a.P % a.K2), c = (int)(e / ((long)a.P * a.K2));
    const int dgi = c / a.cpg;
    double h, w;
    position(a, n, dgi, k, p, h, w);
    const Corner s = corners(h, w, g.H, g.W);
    if (!s.valid) return;
    const double top = a.gcol[(((long)n * g.C + c) * a.K2 + k) * a.ld + p] * mask_of(a, n, dgi, k, p);
    double *gim = a.gx + ((long)n * g.C + c) * g.H * g.W + (long)s.hl * g.W + s.wl;
    const double hh = 1.0 - s.lh, hw = 1.0 - s.lw;
    if (s.ok1) atomicAdd(gim, hh * hw * top);
    if (s.ok2) atomicAdd(gim + 1, hh * s.lw * top);
    if (s.ok3) atomicAdd(gim + g.W, s.lh * hw * top);
    if (s.ok4) atomicAdd(gim + g.W + 1, s.lh * s.lw * top);
}

// one lane per (deformable group, tap, pixel) of image n: goff (2 values) and gmask, summed over the group's channels
__global__ __launch_bounds__(FT) void coord_f64_kernel(const F64Args a)
{
    const ddcn_geom &g = a.g;
    const long e = (long)blockIdx.x * FT + threadIdx.x;
    if (e >= (long)g.dg * a.K2 * a.P) return;
    const int n = blockIdx.z;
    const int p = (int)(e % a.P), k = (int)(e / a.P % a.K2), dgi = (int)(e / ((long)a.P * a.K2));
    double h, w;
    position(a, n, dgi, k, p, h, w);
    const Corner s = corners(h, w, g.H, g.W);
    const double m = mask_of(a, n, dgi, k, p);
    const double hh = 1.0 - s.lh, hw = 1.0 - s.lw;
    double gh = 0.0, gw = 0.0, gm = 0.0;
    if (s.valid) {
        const long HW = (long)g.H * g.W;
        for (int cc = 0; cc < a.cpg; ++cc) {
            const int c = dgi * a.cpg + cc;
            const double *im = a.x + ((long)n * g.C + c) * HW + (long)s.hl * g.W + s.wl;
            const double v1 = s.ok1 ? im[0] : 0.0, v2 = s.ok2 ? im[1] : 0.0;
            const double v3 = s.ok3 ? im[g.W] : 0.0, v4 = s.ok4 ? im[g.W + 1] : 0.0;
            const double top = a.gcol[(((long)n * g.C + c) * a.K2 + k) * a.ld + p];
            // K:144-187: d val / d h and d val / d w of the bilinear form
            gh += top * m * (hw * (v3 - v1) + s.lw * (v4 - v2));
            gw += top * m * (hh * (v2 - v1) + s.lh * (v4 - v3));
            gm += top * (hh * hw * v1 + hh * s.lw * v2 + s.lh * hw * v3 + s.lh * s.lw * v4);
        }
    }
    double *goff = a.goff + ((long)n * g.dg + dgi) * 2 * a.K2 * a.P + p;
    goff[(long)(2 * k) * a.P] += gh;
    goff[(long)(2 * k + 1) * a.P] += gw;
    if (a.gmask) a.gmask[(((long)n * g.dg + dgi) * a.K2 + k) * a.P + p] += gm;
}

// depth-aware taps (integer positions), K:868-918 and K:952-996; one lane per (c, k, p)
template <bool SCATTER>
__global__ __launch_bounds__(FT) void depthaware_f64_kernel(const F64Args a)
{
    const ddcn_geom &g = a.g;
    const long e = (long)blockIdx.x * FT + threadIdx.x;
    if (e >= (long)g.C * a.K2 * a.P) return;
    const int n = blockIdx.z;
    const int p = (int)(e % a.P), k = (int)(e / a.P % a.K2), c = (int)(e / ((long)a.P * a.K2));
    const int ho = p / g.Wo, wo = p - ho * g.Wo;
    const int h_in = ho * g.sh - g.ph, w_in = wo * g.sw - g.pw;
    const int i = k / g.kw, j = k - i * g.kw;
    const long HW = (long)g.H * g.W;
    const double *dp = a.depth + (long)n * HW;
    const int hc = h_in + g.dh * (g.kh - 1) / 2, wc = w_in + g.dw * (g.kw - 1) / 2;
    const bool valid = hc >= 0 && wc >= 0 && hc < g.H && wc < g.W;
    const double Di = valid ? dp[hc * g.W + wc] : 0.0;
    const int h_im = h_in + i * g.dh, w_im = w_in + j * g.dw;
    const bool inb = h_im >= 0 && w_im >= 0 && h_im < g.H && w_im < g.W;
    const long idx = inb ? (long)h_im * g.W + w_im : 0;
    const double Dval = (inb && valid) ? dp[idx] : 0.0;
    const double sim = exp(-fabs(Di - Dval));
    const long at = (((long)n * g.C + c) * a.K2 + k) * a.ld + p;
    if (SCATTER) {
        if (inb) atomicAdd(a.gx + ((long)n * g.C + c) * HW + idx, a.gcol[at] * sim);
    } else {
        a.col[at] = (inb ? a.x[((long)n * g.C + c) * HW + idx] : 0.0) * sim;
    }
}

// C(m, n) (+)= sum_k A(m, k) * B(k, n) (+ bias[m]) with arbitrary element strides; 64x64x16 tiles, 4x4 per lane.
struct Gemm64 {
    const double *A, *B, *bias;
    double *C;
    int M, N, K;
    long sAm, sAk, sBk, sBn, ldc;
    long zA1, zB0, zB1, zC0, zC1;   // batch strides: z = z0 * nz1 + z1 (image, weight group)
    int nz1, accumulate;
};

__global__ __launch_bounds__(FT) void gemm_f64_kernel(const Gemm64 a)
{
    __shared__ double As[16][64 + 1], Bs[16][64 + 1];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int z0 = blockIdx.z / a.nz1, z1 = blockIdx.z % a.nz1;
    const double *A = a.A + z1 * a.zA1;
    const double *B = a.B + z0 * a.zB0 + z1 * a.zB1;
    double *C = a.C + z0 * a.zC0 + z1 * a.zC1;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < a.K; k0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = t + FT * r;              // 1024 elements of each 16x64 tile
            const int kk = i >> 6, mm = i & 63;
            As[kk][mm] = (m0 + mm < a.M && k0 + kk < a.K) ? A[(long)(m0 + mm) * a.sAm + (long)(k0 + kk) * a.sAk] : 0.0;
            Bs[kk][mm] = (n0 + mm < a.N && k0 + kk < a.K) ? B[(long)(k0 + kk) * a.sBk + (long)(n0 + mm) * a.sBn] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = As[kk][ty * 4 + i];
                bv[i] = Bs[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= a.M) continue;
        const double bv = a.bias ? a.bias[z1 * a.M + m] : 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= a.N) continue;
            double *dst = C + (long)m * a.ldc + n;
            *dst = (a.accumulate ? *dst : 0.0) + acc[i][j] + bv;
        }
    }
}

// gb[co] += sum_{n,p} gout  (H:347-353)
__global__ __launch_bounds__(FT) void bias_grad_f64_kernel(const double *gout, double *gb, int N, int Co, int P)
{
    __shared__ double part[FT];
    const int co = blockIdx.x;
    double acc = 0.0;
    for (int n = 0; n < N; ++n)
        for (int p = threadIdx.x; p < P; p += FT) acc += gout[((long)n * Co + co) * P + p];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = FT / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) gb[co] += part[0];
}

__global__ __launch_bounds__(FT) void sigmoid_f64_kernel(double *v, long count)
{
    const long i = (long)blockIdx.x * FT + threadIdx.x;
    if (i < count) v[i] = 1.0 / (1.0 + exp(-v[i]));
}

F64Args base(const ddcn_geom &g)
{
    F64Args a = {};
    const Dims d = dims_of(g);
    a.g = g; a.K2 = d.K2; a.P = d.P; a.cpg = d.cpg; a.ld = d.ld;
    return a;
}

dim3 element_grid(long per_image, int N) { return dim3((unsigned)((per_image + FT - 1) / FT), 1, (unsigned)N); }

int check_size(const ddcn_geom &g)
{
    const Dims d = dims_of(g);
    if ((long)g.C * d.K2 * d.P >= (1L << 31) * FT || g.N > 65535 || (long)g.N * g.groups > 65535)
        return fail(DDCN_ERR_UNSUPPORTED, "fp64 path: tensor too large");
    return DDCN_OK;
}

enum GemmKind { G_OUT, G_GCOL, G_GW };

// the three contractions; gw accumulates over the images one launch after the other (deterministic, no atomics)
int gemm(GemmKind kind, const double *weight, const double *colp, const double *gout, const double *bias, double *dst,
         const ddcn_geom &g, hipStream_t s)
{
    const Dims d = dims_of(g);
    Gemm64 a = {};
    a.nz1 = g.groups;
    if (kind == G_OUT) {            // out[n][grp] = W[grp] * col[n][grp] + bias
        a.A = weight; a.B = colp; a.C = dst; a.bias = bias;
        a.M = d.Cog; a.N = d.P; a.K = d.Kg;
        a.sAm = d.Kg; a.sAk = 1; a.sBk = d.ld; a.sBn = 1; a.ldc = d.P;
        a.zA1 = (long)d.Cog * d.Kg; a.zB0 = (long)d.Ktot * d.ld; a.zB1 = (long)d.Kg * d.ld;
        a.zC0 = (long)g.Co * d.P; a.zC1 = (long)d.Cog * d.P;
    } else if (kind == G_GCOL) {    // gcol[n][grp] = W[grp]^T * gout[n][grp]
        a.A = weight; a.B = gout; a.C = dst;
        a.M = d.Kg; a.N = d.P; a.K = d.Cog;
        a.sAm = 1; a.sAk = d.Kg; a.sBk = d.P; a.sBn = 1; a.ldc = d.ld;
        a.zA1 = (long)d.Cog * d.Kg; a.zB0 = (long)g.Co * d.P; a.zB1 = (long)d.Cog * d.P;
        a.zC0 = (long)d.Ktot * d.ld; a.zC1 = (long)d.Kg * d.ld;
    } else {                        // gw[grp] += gout[n][grp] * col[n][grp]^T, image by image
        a.M = d.Cog; a.N = d.Kg; a.K = d.P;
        a.sAm = d.P; a.sAk = 1; a.sBk = 1; a.sBn = d.ld; a.ldc = d.Kg;
        a.zA1 = (long)d.Cog * d.P; a.zB1 = (long)d.Kg * d.ld; a.zC1 = (long)d.Cog * d.Kg;
        a.accumulate = 1;
        for (int n = 0; n < g.N; ++n) {
            a.A = gout + (long)n * g.Co * d.P;
            a.B = colp + (long)n * d.Ktot * d.ld;
            a.C = dst;
            hipLaunchKernelGGL(gemm_f64_kernel, dim3((a.N + 63) / 64, (a.M + 63) / 64, g.groups), dim3(FT), 0, s, a);
        }
        return check_launch("gemm_f64 (gw)");
    }
    hipLaunchKernelGGL(gemm_f64_kernel, dim3((a.N + 63) / 64, (a.M + 63) / 64, g.N * g.groups), dim3(FT), 0, s, a);
    return check_launch("gemm_f64");
}

}  // namespace

size_t f64_workspace_bytes(const ddcn_geom &g)
{
    const Dims d = dims_of(g);
    return (size_t)g.N * d.Ktot * d.ld * sizeof(double);
}

int f64_forward(const double *x, const double *weight, const double *bias, const double *offset, const double *mask,
                const double *depth, double *out, int sigmoid, double *col, const ddcn_geom &g, hipStream_t s)
{
    if (int rc = check_size(g)) return rc;
    F64Args a = base(g);
    a.x = x; a.offset = offset; a.mask = mask; a.depth = depth; a.col = col;
    const dim3 grid = element_grid((long)g.C * a.K2 * a.P, g.N);
    if (depth) hipLaunchKernelGGL(depthaware_f64_kernel<false>, grid, dim3(FT), 0, s, a);
    else hipLaunchKernelGGL(im2col_f64_kernel, grid, dim3(FT), 0, s, a);
    if (int rc = check_launch("im2col_f64")) return rc;
    if (int rc = gemm(G_OUT, weight, col, nullptr, bias, out, g, s)) return rc;
    if (sigmoid) {
        const long count = (long)g.N * g.Co * a.P;
        hipLaunchKernelGGL(sigmoid_f64_kernel, dim3((unsigned)((count + FT - 1) / FT)), dim3(FT), 0, s, out, count);
        return check_launch("sigmoid_f64");
    }
    return DDCN_OK;
}

// Accumulates into gx, goff, gmask, gw, gb (the caller zero-fills, as for the fp32 entry points).
int f64_backward(const double *x, const double *weight, const double *offset, const double *mask, const double *depth,
                 const double *gout, const double *saved_cols, double *gx, double *goff, double *gmask, double *gw,
                 double *gb, double *col, const ddcn_geom &g, hipStream_t s)
{
    if (int rc = check_size(g)) return rc;
    F64Args a = base(g);
    a.x = x; a.offset = offset; a.mask = mask; a.depth = depth;
    a.gx = gx; a.goff = goff; a.gmask = gmask;
    const dim3 grid = element_grid((long)g.C * a.K2 * a.P, g.N);
    if (int rc = gemm(G_GCOL, weight, nullptr, gout, nullptr, col, g, s)) return rc;
    a.gcol = col;
    if (depth) {
        hipLaunchKernelGGL(depthaware_f64_kernel<true>, grid, dim3(FT), 0, s, a);
    } else {
        hipLaunchKernelGGL(col2im_f64_kernel, grid, dim3(FT), 0, s, a);
        hipLaunchKernelGGL(coord_f64_kernel, element_grid((long)g.dg * a.K2 * a.P, g.N), dim3(FT), 0, s, a);
    }
    if (int rc = check_launch("col2im_f64")) return rc;
    const double *cols = saved_cols;
    if (!cols) {
        a.col = col;
        if (depth) hipLaunchKernelGGL(depthaware_f64_kernel<false>, grid, dim3(FT), 0, s, a);
        else hipLaunchKernelGGL(im2col_f64_kernel, grid, dim3(FT), 0, s, a);
        if (int rc = check_launch("im2col_f64")) return rc;
        cols = col;
    }
    if (int rc = gemm(G_GW, nullptr, cols, gout, nullptr, gw, g, s)) return rc;
    if (gb) {
        hipLaunchKernelGGL(bias_grad_f64_kernel, dim3((unsigned)g.Co), dim3(FT), 0, s, gout, gb, g.N, g.Co, a.P);
        return check_launch("bias_grad_f64");
    }
    return DDCN_OK;
}

}  // namespace ddcn

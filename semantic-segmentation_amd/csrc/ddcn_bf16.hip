// Native bf16 forward of the (modulated) deformable convolution for gfx950 -- BASELINE config 5 (inference:
// bf16 activations / weights / columns, fp32 coordinate math, fp32 MFMA accumulation).  The reference has no
// bf16 path (deform_conv_cuda_kernel.cu:258 dispatches FLOATING_TYPES_AND_HALF); the arithmetic restated is the
// same K:569-632 (im2col) and H:430-435 (GEMM) as the fp32 path, with operands rounded to bf16.
//
// Two kernels:
//   im2col_bf16   one lane per (pixel, 8 channels): the 9 taps' corner indices and bilinear weights are computed
//                 once per tap in fp32 and reused for the 8 channels; the 72 samples leave as nine 16-byte stores.
//                 Columns layout: col[n][K/8][P][8] -- "k-octets": the 8 consecutive k of one pixel are contiguous,
//                 consecutive pixels follow.  That is exactly one B fragment of v_mfma_f32_32x32x16_bf16 per 16
//                 bytes, so the GEMM needs no transpose anywhere, and both this kernel's stores and the GEMM's
//                 loads are fully coalesced along the pixel axis.
//   gemm_bf16     out[Co][P] = W[Co][K] * col: 128x128x64 tiles, 4 waves of 64x64 (2x2 MFMA 32x32x16), operands
//                 staged global -> registers -> LDS (W rows XOR-swizzled by 16-byte chunk, columns linear),
//                 next tile's global loads in flight during the MFMAs, bias add and bf16 rounding in the epilogue.
//                 Workgroups that share a columns tile are consecutive on one XCD (its L2 serves the re-reads).
#include "ddcn_common.h"

namespace ddcn {
namespace {

typedef unsigned short bf16_t;  // storage type at the boundary
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
// round to nearest even; NaN stays NaN (quiet bit forced)
__device__ __forceinline__ unsigned f2bf(float f)
{
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return f2bf(lo) | (f2bf(hi) << 16); }

struct Bf16Args {
    const bf16_t *x, *offset, *mask, *weight, *bias;
    bf16_t *col, *out;
    int N, C, H, W, Co, Ho, Wo, P;
    int sh, sw, ph, pw, dh, dw, groups, dg, cpg;
    int KO;          // k-octets per image: C*9/8
    int Kg, Cog;     // per weight group: K extent, output channels
    int tiles_m, tiles_n;
};

constexpr int IT = 256;  // im2col threads

// K:569-632 with bf16 tensors.  grid = (ceil(P/IT), C/8, N)
template <bool MOD>
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

    // sample (c, k) is element c*9+k of the thread's 72: rounded and packed two per register as it is produced
    unsigned pk[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) pk[q] = 0u;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int i = k / 3, j = k % 3;
        const float oh = bf2f(offb[(long)(2 * k) * a.P]), ow = bf2f(offb[(long)(2 * k + 1) * a.P]);
        const float m = MOD ? bf2f(maskb[(long)k * a.P]) : 1.f;
        const float h = (float)(ho * a.sh - a.ph + i * a.dh) + oh;
        const float w = (float)(wo * a.sw - a.pw + j * a.dw) + ow;
        // K:602-610: zero unless -1 < h < H and -1 < w < W; corners outside the map contribute zero (K:466-496)
        const bool inside = h > -1.f && w > -1.f && h < (float)a.H && w < (float)a.W;
        const float hf = floorf(h), wf = floorf(w);
        const int hl = (int)hf, wl = (int)wf;
        const float lh = h - hf, lw = w - wf;
        const bool t = inside && hl >= 0, b = inside && hl + 1 <= a.H - 1;
        const bool l = wl >= 0, r = wl + 1 <= a.W - 1;
        // fold the mask into the four weights; a corner outside gets weight 0 and a clamped (valid) address
        const float w1 = (t && l) ? (1.f - lh) * (1.f - lw) * m : 0.f;
        const float w2 = (t && r) ? (1.f - lh) * lw * m : 0.f;
        const float w3 = (b && l) ? lh * (1.f - lw) * m : 0.f;
        const float w4 = (b && r) ? lh * lw * m : 0.f;
        const int hc0 = min(max(hl, 0), a.H - 1), hc1 = min(max(hl + 1, 0), a.H - 1);
        const int wc0 = min(max(wl, 0), a.W - 1), wc1 = min(max(wl + 1, 0), a.W - 1);
        const int i1 = hc0 * a.W + wc0, i2 = hc0 * a.W + wc1, i3 = hc1 * a.W + wc0, i4 = hc1 * a.W + wc1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bf16_t *pl = xb + c * HW;
            // a corner outside the map must also silence an Inf/NaN at its clamped address: select, then multiply
            const float x1 = (t && l) ? bf2f(pl[i1]) : 0.f, x2 = (t && r) ? bf2f(pl[i2]) : 0.f;
            const float x3 = (b && l) ? bf2f(pl[i3]) : 0.f, x4 = (b && r) ? bf2f(pl[i4]) : 0.f;
            const unsigned v = f2bf(w1 * x1 + w2 * x2 + w3 * x3 + w4 * x4);
            const int e = c * 9 + k;
            pk[e >> 1] |= (e & 1) ? v << 16 : v;   // pk starts at zero, each half is written once
        }
    }
    u32x4 *dst = (u32x4 *)(a.col + (((long)n * a.KO + (long)co * 9) * a.P + p) * 8);
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        u32x4 v;
        v.x = pk[4 * j + 0];
        v.y = pk[4 * j + 1];
        v.z = pk[4 * j + 2];
        v.w = pk[4 * j + 3];
        dst[(long)j * a.P] = v;
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
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wm * 64 + i * 32 + r32;
                const u32x4 v = As[row * 8 + ((s * 2 + hi) ^ (row & 7))];
                fa[i] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 v = Bs[(s * 2 + hi) * BN + wn * 64 + j * 32 + r32];
                fb[j] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
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
            const float bv = bias ? bf2f(bias[row]) : 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + r32;
                const float v = acc[i][j][e] + bv;
                if (col < a.P) C[(long)row * a.P + col] = (bf16_t)f2bf(v);
            }
        }
    }
}

}  // namespace

size_t bf16_workspace_bytes(const ddcn_geom &g)
{
    return (size_t)g.N * g.C * g.kh * g.kw * g.Ho * g.Wo * 2;
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
        g.N > 65535)
        return fail(DDCN_ERR_UNSUPPORTED, "bf16 path: tensor too large");
    return DDCN_OK;
}

int launch_forward_bf16(const void *x, const void *weight, const void *bias, const void *offset, const void *mask,
                        void *out, void *ws, const ddcn_geom &g, hipStream_t stream)
{
    if (int rc = bf16_supported(g)) return rc;
    Bf16Args a = {};
    a.x = (const bf16_t *)x; a.offset = (const bf16_t *)offset; a.mask = (const bf16_t *)mask;
    a.weight = (const bf16_t *)weight; a.bias = (const bf16_t *)bias;
    a.col = (bf16_t *)ws; a.out = (bf16_t *)out;
    a.N = g.N; a.C = g.C; a.H = g.H; a.W = g.W; a.Co = g.Co; a.Ho = g.Ho; a.Wo = g.Wo; a.P = g.Ho * g.Wo;
    a.sh = g.sh; a.sw = g.sw; a.ph = g.ph; a.pw = g.pw; a.dh = g.dh; a.dw = g.dw;
    a.groups = g.groups; a.dg = g.dg; a.cpg = g.C / g.dg;
    a.KO = g.C * 9 / 8;
    a.Kg = g.C / g.groups * 9;
    a.Cog = g.Co / g.groups;
    a.tiles_m = (a.Cog + BM - 1) / BM;
    a.tiles_n = (a.P + BN - 1) / BN;

    const dim3 ig((unsigned)((a.P + IT - 1) / IT), (unsigned)(g.C / 8), (unsigned)g.N);
    if (mask) hipLaunchKernelGGL(im2col_bf16_kernel<true>, ig, dim3(IT), 0, stream, a);
    else hipLaunchKernelGGL(im2col_bf16_kernel<false>, ig, dim3(IT), 0, stream, a);
    if (int rc = check_launch("im2col_bf16")) return rc;
    const long tiles = (long)a.tiles_m * a.tiles_n;
    if (tiles >= (1L << 31)) return fail(DDCN_ERR_UNSUPPORTED, "bf16 path: too many GEMM tiles");
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3((unsigned)tiles, (unsigned)(g.N * g.groups)), dim3(GT), 0, stream, a);
    return check_launch("gemm_bf16");
}

}  // namespace ddcn

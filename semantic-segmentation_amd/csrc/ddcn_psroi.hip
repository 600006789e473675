// Deformable position-sensitive RoI pooling for gfx950 (forward and backward).
//
// Replaces the two CUDA kernels of detseg/ops/dcn/src/deform_pool_cuda_kernel.cu (PK below):
//   forward   PK:53-139    one lane per output bin, lanes run along (pw, ph, ctop) so stores coalesce and the
//                          four corner reads of neighbouring lanes fall into the same lines of one score map;
//   backward  PK:141-263   two kernels, neither adds to global memory per sample as the CUDA kernel does:
//     offset gradient      one workgroup per (RoI, offset class, slice of the class's channels).  The bin
//                          geometry is computed once per workgroup into LDS instead of once per channel; the
//                          gradient -- two global atomics per SAMPLE in the CUDA kernel, all channels of a
//                          class onto the same 2*part^2 cells -- is summed per bin in registers, per
//                          workgroup in LDS, and leaves with one global atomic per cell;
//     feature gradient     "plane owner": one workgroup per (score map, RoI slice) keeps the whole H*W gradient
//                          plane in LDS as doubles (ds_add_f64 runs ~9x faster than ds_add_f32 on this chip),
//                          walks the RoIs of its image, and writes the plane out once.  Consecutive samples
//                          that fall into the same pixel cell are merged before they touch LDS.  A map too
//                          large for LDS falls back to global atomics inside the first kernel.
//
// Coordinate arithmetic is compiled with contraction off: whether a sample is dropped, and which cell it falls
// into, are discrete decisions, and they must not depend on where the compiler forms an FMA.
#include "ddcn_common.h"

#include <stdlib.h>

namespace ddcn {
namespace {

constexpr int RT = 256;          // threads per workgroup, both kernels
constexpr int MAX_BINS = 1024;   // pooled^2 and part^2 limits of the backward kernel (LDS tables)

// T = float or double: the reference instantiates both kernels for every floating type (PK:293, PK:342)
template <typename T>
struct RoiArgs {
    const T *data, *rois, *trans, *cnt, *gout;
    T *out, *out_cnt, *gdata, *gtrans;
    int B, C, H, W, R, OC, PS, GS, PART, SPP, NCLS, CEC, no_trans;
    T scale, tstd;
    int chunk;  // backward: channels per workgroup
};

template <typename T>
struct RoiBox {
    T x0, y0, w, h;  // roi_start_w, roi_start_h, roi_width, roi_height
    int batch;
};

// PK:79-89
template <typename T>
__device__ __forceinline__ RoiBox<T> roi_box(const T *roi, T scale)
{
#pragma clang fp contract(off)
    RoiBox<T> b;
    b.batch = (int)roi[0];
    b.x0 = round(roi[1]) * scale - T(0.5);
    b.y0 = round(roi[2]) * scale - T(0.5);
    const T x1 = (round(roi[3]) + T(1)) * scale - T(0.5);
    const T y1 = (round(roi[4]) + T(1)) * scale - T(0.5);
    // the reference compares against the DOUBLE literal 0.1
    b.w = (T)fmax((double)(x1 - b.x0), 0.1);
    b.h = (T)fmax((double)(y1 - b.y0), 0.1);
    return b;
}

template <typename T>
struct BinGeom {
    T wstart, hstart;
    int gofs;   // gh * GS + gw
    int pidx;   // part_h * PART + part_w
};

// PK:92-114 for one bin of one (RoI, class)
template <typename T>
__device__ __forceinline__ BinGeom<T> bin_geom(const RoiArgs<T> &a, const RoiBox<T> &b, int n, int cls, int ph, int pw)
{
#pragma clang fp contract(off)
    BinGeom<T> g;
    const T bin_h = b.h / (T)a.PS, bin_w = b.w / (T)a.PS;
    const int part_h = (int)floor((T)ph / (T)a.PS * (T)a.PART);
    const int part_w = (int)floor((T)pw / (T)a.PS * (T)a.PART);
    g.pidx = part_h * a.PART + part_w;
    T tx = T(0), ty = T(0);
    if (!a.no_trans) {
        const T *t = a.trans + ((long)n * a.NCLS + cls) * 2 * a.PART * a.PART;
        tx = t[g.pidx] * a.tstd;
        ty = t[a.PART * a.PART + g.pidx] * a.tstd;
    }
    T wstart = (T)pw * bin_w + b.x0;
    wstart += tx * b.w;
    T hstart = (T)ph * bin_h + b.y0;
    hstart += ty * b.h;
    g.wstart = wstart;
    g.hstart = hstart;
    int gw = (int)floor((T)pw * (T)a.GS / (T)a.PS);
    int gh = (int)floor((T)ph * (T)a.GS / (T)a.PS);
    gw = min(max(gw, 0), a.GS - 1);
    gh = min(max(gh, 0), a.GS - 1);
    g.gofs = gh * a.GS + gw;
    return g;
}

// PK:121-128: false when the sample is dropped; otherwise w, h clamped into the map
template <typename T>
__device__ __forceinline__ bool keep_sample(T &w, T &h, int W, int H)
{
    if (w < -T(0.5) || w > (T)W - T(0.5) || h < -T(0.5) || h > (T)H - T(0.5)) return false;
    w = fmin(fmax(w, T(0)), (T)W - T(1));
    h = fmin(fmax(h, T(0)), (T)H - T(1));
    return true;
}

template <typename T>
__global__ __launch_bounds__(RT) void psroi_forward_kernel(const RoiArgs<T> a)
{
    const long total = (long)a.R * a.OC * a.PS * a.PS;
    for (long index = (long)blockIdx.x * RT + threadIdx.x; index < total; index += (long)gridDim.x * RT) {
        const int pw = (int)(index % a.PS);
        const int ph = (int)(index / a.PS % a.PS);
        const int ctop = (int)(index / ((long)a.PS * a.PS) % a.OC);
        const int n = (int)(index / ((long)a.PS * a.PS * a.OC));
        const RoiBox<T> b = roi_box(a.rois + (long)n * 5, a.scale);
        T sum = T(0);
        int count = 0;
        if (b.batch >= 0 && b.batch < a.B) {  // a RoI naming an image the batch lacks pools nothing
            const BinGeom<T> g = bin_geom(a, b, n, ctop / a.CEC, ph, pw);
            T sub_w, sub_h;
            {
#pragma clang fp contract(off)
                sub_h = b.h / (T)a.PS / (T)a.SPP;
                sub_w = b.w / (T)a.PS / (T)a.SPP;
            }
            const T *plane = a.data + ((long)b.batch * a.C + (long)ctop * a.GS * a.GS + g.gofs) * a.H * a.W;
            for (int ih = 0; ih < a.SPP; ++ih) {
                for (int iw = 0; iw < a.SPP; ++iw) {
                    T w, h;
                    {
#pragma clang fp contract(off)
                        w = g.wstart + (T)iw * sub_w;
                        h = g.hstart + (T)ih * sub_h;
                    }
                    if (!keep_sample(w, h, a.W, a.H)) continue;
                    // PK:33-50
                    const int x1 = (int)floor(w), x2 = (int)ceil(w);
                    const int y1 = (int)floor(h), y2 = (int)ceil(h);
                    const T dx = w - (T)x1, dy = h - (T)y1;
                    const T v11 = plane[y1 * a.W + x1], v12 = plane[y2 * a.W + x1];
                    const T v21 = plane[y1 * a.W + x2], v22 = plane[y2 * a.W + x2];
                    sum += (T(1) - dx) * (T(1) - dy) * v11 + (T(1) - dx) * dy * v12 + dx * (T(1) - dy) * v21 +
                           dx * dy * v22;
                    ++count;
                }
            }
        }
        a.out[index] = count == 0 ? T(0) : sum / (T)count;
        a.out_cnt[index] = (T)count;
    }
}

// Adds the pending cell (base pixel p, its right/lower neighbours at +sx / +sy) to the feature gradient.
template <typename T>
__device__ __forceinline__ void flush_cell(T *gplane, int p, int sx, int sy, T a00, T a01, T a10,
                                           T a11)
{
    if (a00 != T(0)) atomicAdd(gplane + p, a00);
    if (a01 != T(0)) atomicAdd(gplane + p + sy, a01);
    if (a10 != T(0)) atomicAdd(gplane + p + sx, a10);
    if (a11 != T(0)) atomicAdd(gplane + p + sx + sy, a11);
}

// DATA: also scatter the feature gradient (global atomics); otherwise only the offset gradient.
template <typename T, bool DATA>
__global__ __launch_bounds__(RT) void psroi_backward_kernel(const RoiArgs<T> a)
{
    __shared__ T s_wstart[MAX_BINS], s_hstart[MAX_BINS];
    __shared__ int s_gofs[MAX_BINS], s_pidx[MAX_BINS];
    __shared__ T s_gt[2 * MAX_BINS];

    const int n = blockIdx.x / a.NCLS, cls = blockIdx.x % a.NCLS;
    const int c_begin = blockIdx.y * a.chunk, c_end = min(c_begin + a.chunk, a.CEC);
    const int bins = a.PS * a.PS, parts = a.PART * a.PART;
    const RoiBox<T> b = roi_box(a.rois + (long)n * 5, a.scale);
    if (b.batch < 0 || b.batch >= a.B) return;  // block-uniform: same rule as the forward kernel

    for (int i = threadIdx.x; i < bins; i += RT) {
        const BinGeom<T> g = bin_geom(a, b, n, cls, i / a.PS, i % a.PS);
        s_wstart[i] = g.wstart;
        s_hstart[i] = g.hstart;
        s_gofs[i] = g.gofs;
        s_pidx[i] = g.pidx;
    }
    for (int i = threadIdx.x; i < 2 * parts; i += RT) s_gt[i] = T(0);
    __syncthreads();

    T sub_w, sub_h;
    {
#pragma clang fp contract(off)
        sub_h = b.h / (T)a.PS / (T)a.SPP;
        sub_w = b.w / (T)a.PS / (T)a.SPP;
    }
    const long HW = (long)a.H * a.W;
    const T *img = a.data + (long)b.batch * a.C * HW;
    T *gimg = a.gdata + (long)b.batch * a.C * HW;
    const int items = (c_end - c_begin) * bins;
    for (int item = threadIdx.x; item < items; item += RT) {
        const int bin = item % bins;
        const int ctop = cls * a.CEC + c_begin + item / bins;
        const long oidx = ((long)n * a.OC + ctop) * bins + bin;
        const T cnt = a.cnt[oidx];
        if (cnt <= T(0)) continue;  // PK:203-206
        const T diff_val = a.gout[oidx] / cnt;
        const long pofs = ((long)ctop * a.GS * a.GS + s_gofs[bin]) * HW;
        const T *plane = img + pofs;
        T *gplane = gimg + pofs;
        const T wstart = s_wstart[bin], hstart = s_hstart[bin];
        T tdx = T(0), tdy = T(0);
        // pending cell of the feature gradient
        int pend = -1, pend_sx = 0, pend_sy = 0;
        T a00 = T(0), a01 = T(0), a10 = T(0), a11 = T(0);
        for (int ih = 0; ih < a.SPP; ++ih) {
            for (int iw = 0; iw < a.SPP; ++iw) {
                T w, h;
                {
#pragma clang fp contract(off)
                    w = wstart + (T)iw * sub_w;
                    h = hstart + (T)ih * sub_h;
                }
                if (!keep_sample(w, h, a.W, a.H)) continue;
                // PK:230-243
                const int x0 = (int)floor(w), x1 = (int)ceil(w);
                const int y0 = (int)floor(h), y1 = (int)ceil(h);
                const T dx = w - (T)x0, dy = h - (T)y0;
                const int p = y0 * a.W + x0, sx = x1 - x0, sy = (y1 - y0) * a.W;
                if (DATA) {
                    if (p != pend || sx != pend_sx || sy != pend_sy) {
                        if (pend >= 0) flush_cell(gplane, pend, pend_sx, pend_sy, a00, a01, a10, a11);
                        pend = p;
                        pend_sx = sx;
                        pend_sy = sy;
                        a00 = a01 = a10 = a11 = T(0);
                    }
                    a00 += (T(1) - dx) * (T(1) - dy) * diff_val;
                    a01 += (T(1) - dx) * dy * diff_val;
                    a10 += dx * (T(1) - dy) * diff_val;
                    a11 += dx * dy * diff_val;
                }
                if (a.no_trans) continue;
                // PK:249-256
                const T U00 = plane[p], U01 = plane[p + sy], U10 = plane[p + sx], U11 = plane[p + sx + sy];
                tdx += (U11 * dy + U10 * (T(1) - dy) - U01 * dy - U00 * (T(1) - dy)) * a.tstd * diff_val * b.w;
                tdy += (U11 * dx + U01 * (T(1) - dx) - U10 * dx - U00 * (T(1) - dx)) * a.tstd * diff_val * b.h;
            }
        }
        if (DATA && pend >= 0) flush_cell(gplane, pend, pend_sx, pend_sy, a00, a01, a10, a11);
        if (!a.no_trans) {
            if (tdx != T(0)) atomicAdd(&s_gt[s_pidx[bin]], tdx);
            if (tdy != T(0)) atomicAdd(&s_gt[parts + s_pidx[bin]], tdy);
        }
    }
    if (a.no_trans) return;
    __syncthreads();
    // PK:258-259, once per cell and workgroup
    T *gt = a.gtrans + ((long)n * a.NCLS + cls) * 2 * parts;
    for (int i = threadIdx.x; i < 2 * parts; i += RT) {
        const T v = s_gt[i];
        if (v != T(0)) atomicAdd(gt + i, v);
    }
}


// ---- feature gradient, plane-owner formulation ----
constexpr int ROI_CHUNK = 1024;           // RoI candidates compacted per pass
constexpr int PLANE_LDS_MAX = 144 * 1024; // bytes of the double plane a workgroup may hold

template <typename T>
__device__ __forceinline__ void lds_add(double *cell, T v)
{
    if (v != T(0)) __hip_atomic_fetch_add(cell, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// grid.x = B*C score maps, grid.y = RoI slices (RoI n belongs to slice n % gridDim.y).
// TRANS: the workgroup also keeps its score map in LDS and adds this channel's share of the offset gradient, two global
// atomics per (RoI, bin) instead of a second kernel that gathers every sample from global memory again; the item
// order is rotated by the channel so that the channels of a class do not hit one offset cell at the same moment.
template <typename T, bool TRANS>
__global__ __launch_bounds__(RT) void psroi_backward_plane_kernel(const RoiArgs<T> a)
{
    extern __shared__ double s_plane[];               // [H*W] gradient plane, then (TRANS) [H*W] score map as T
    __shared__ int s_bins[MAX_BINS], s_rois[ROI_CHUNK];
    __shared__ int s_nbins, s_nrois;

    const int b = blockIdx.x / a.C, c = blockIdx.x % a.C;
    const int gs2 = a.GS * a.GS;
    const int ctop = c / gs2, gcell = c % gs2, cls = ctop / a.CEC;
    const int bins = a.PS * a.PS, HW = a.H * a.W;
    const int slices = gridDim.y, slice = blockIdx.y;

    T *s_map = reinterpret_cast<T *>(s_plane + HW);
    for (int i = threadIdx.x; i < HW; i += RT) s_plane[i] = 0.0;
    if (TRANS) {
        const T *src = a.data + ((long)b * a.C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += RT) s_map[i] = src[i];
    }
    const int parts = a.PART * a.PART;
    if (threadIdx.x == 0) s_nbins = 0;
    __syncthreads();
    // bins of a RoI that read this score map (PK:111-114, 130): the same for every RoI
    for (int i = threadIdx.x; i < bins; i += RT) {
        int gw = (int)floor((T)(i % a.PS) * (T)a.GS / (T)a.PS);
        int gh = (int)floor((T)(i / a.PS) * (T)a.GS / (T)a.PS);
        gw = min(max(gw, 0), a.GS - 1);
        gh = min(max(gh, 0), a.GS - 1);
        if (gh * a.GS + gw == gcell) s_bins[atomicAdd(&s_nbins, 1)] = i;
    }
    __syncthreads();
    const int nbins = s_nbins;
    if (nbins == 0) return;  // block-uniform; the plane stays zero and gdata is left as it is

    const int per_pass = ROI_CHUNK * slices;
    for (int base = 0; base < a.R; base += per_pass) {
        if (threadIdx.x == 0) s_nrois = 0;
        __syncthreads();
        for (int k = threadIdx.x; k < ROI_CHUNK; k += RT) {
            const int n = base + k * slices + slice;
            if (n < a.R && (int)a.rois[(long)n * 5] == b) s_rois[atomicAdd(&s_nrois, 1)] = n;
        }
        __syncthreads();
        const int items = s_nrois * nbins;
        const int rot = items > 0 ? (int)(((long)c * 61 * RT) % items) : 0;
        for (int it = threadIdx.x; it < items; it += RT) {
            int item = it + rot;
            if (item >= items) item -= items;
            const int n = s_rois[item / nbins], bin = s_bins[item % nbins];
            const long oidx = ((long)n * a.OC + ctop) * bins + bin;
            const T cnt = a.cnt[oidx];
            if (cnt <= T(0)) continue;  // PK:203-206
            const T diff_val = a.gout[oidx] / cnt;
            const RoiBox<T> box = roi_box(a.rois + (long)n * 5, a.scale);
            const BinGeom<T> g = bin_geom(a, box, n, cls, bin / a.PS, bin % a.PS);
            T sub_w, sub_h;
            {
#pragma clang fp contract(off)
                sub_h = box.h / (T)a.PS / (T)a.SPP;
                sub_w = box.w / (T)a.PS / (T)a.SPP;
            }
            int pend = -1, pend_sx = 0, pend_sy = 0;
            T a00 = T(0), a01 = T(0), a10 = T(0), a11 = T(0);
            T tdx = T(0), tdy = T(0);
            for (int ih = 0; ih < a.SPP; ++ih) {
                for (int iw = 0; iw < a.SPP; ++iw) {
                    T w, h;
                    {
#pragma clang fp contract(off)
                        w = g.wstart + (T)iw * sub_w;
                        h = g.hstart + (T)ih * sub_h;
                    }
                    if (!keep_sample(w, h, a.W, a.H)) continue;
                    const int x0 = (int)floor(w), x1 = (int)ceil(w);
                    const int y0 = (int)floor(h), y1 = (int)ceil(h);
                    const T dx = w - (T)x0, dy = h - (T)y0;
                    const int p = y0 * a.W + x0, sx = x1 - x0, sy = (y1 - y0) * a.W;
                    if (p != pend || sx != pend_sx || sy != pend_sy) {
                        if (pend >= 0) {
                            lds_add(s_plane + pend, a00);
                            lds_add(s_plane + pend + pend_sy, a01);
                            lds_add(s_plane + pend + pend_sx, a10);
                            lds_add(s_plane + pend + pend_sx + pend_sy, a11);
                        }
                        pend = p;
                        pend_sx = sx;
                        pend_sy = sy;
                        a00 = a01 = a10 = a11 = T(0);
                    }
                    a00 += (T(1) - dx) * (T(1) - dy) * diff_val;
                    a01 += (T(1) - dx) * dy * diff_val;
                    a10 += dx * (T(1) - dy) * diff_val;
                    a11 += dx * dy * diff_val;
                    if (TRANS) {   // PK:249-256
                        const T U00 = s_map[p], U01 = s_map[p + sy], U10 = s_map[p + sx], U11 = s_map[p + sx + sy];
                        tdx += (U11 * dy + U10 * (T(1) - dy) - U01 * dy - U00 * (T(1) - dy)) * a.tstd * diff_val * box.w;
                        tdy += (U11 * dx + U01 * (T(1) - dx) - U10 * dx - U00 * (T(1) - dx)) * a.tstd * diff_val * box.h;
                    }
                }
            }
            if (pend >= 0) {
                lds_add(s_plane + pend, a00);
                lds_add(s_plane + pend + pend_sy, a01);
                lds_add(s_plane + pend + pend_sx, a10);
                lds_add(s_plane + pend + pend_sx + pend_sy, a11);
            }
            if (TRANS) {       // PK:258-259, once per (RoI, bin, channel)
                T *gt = a.gtrans + ((long)n * a.NCLS + cls) * 2 * parts + g.pidx;
                if (tdx != T(0)) atomicAdd(gt, tdx);
                if (tdy != T(0)) atomicAdd(gt + parts, tdy);
            }
        }
        __syncthreads();
    }
    // one pass over the plane: the only writer of these cells when there is one slice
    T *gplane = a.gdata + ((long)b * a.C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += RT) {
        const T v = (T)s_plane[i];
        if (v == T(0)) continue;
        if (slices == 1) gplane[i] += v;
        else atomicAdd(gplane + i, v);
    }
}

// ---- forward, plane-owner formulation: the score map is read once into LDS and every sample is taken from there
// (four ds_read_b32 instead of four scattered global loads).  Same arithmetic and summation order as
// psroi_forward_kernel, so both give the same bits.  grid.x = B*C score maps, grid.y = RoI slices.
template <typename T>
__global__ __launch_bounds__(RT) void psroi_forward_plane_kernel(const RoiArgs<T> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_fraw[];
    T *s_map = reinterpret_cast<T *>(s_fraw);          // [H*W]
    __shared__ int s_bins[MAX_BINS], s_rois[ROI_CHUNK];
    __shared__ int s_nbins, s_nrois;

    const int b = blockIdx.x / a.C, c = blockIdx.x % a.C;
    const int gs2 = a.GS * a.GS;
    const int ctop = c / gs2, gcell = c % gs2, cls = ctop / a.CEC;
    const int bins = a.PS * a.PS, HW = a.H * a.W;
    const int slices = gridDim.y, slice = blockIdx.y;

    const T *src = a.data + ((long)b * a.C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += RT) s_map[i] = src[i];
    if (threadIdx.x == 0) s_nbins = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += RT) {     // bins that read this score map (PK:111-114, 130)
        int gw = (int)floor((T)(i % a.PS) * (T)a.GS / (T)a.PS);
        int gh = (int)floor((T)(i / a.PS) * (T)a.GS / (T)a.PS);
        gw = min(max(gw, 0), a.GS - 1);
        gh = min(max(gh, 0), a.GS - 1);
        if (gh * a.GS + gw == gcell) s_bins[atomicAdd(&s_nbins, 1)] = i;
    }
    __syncthreads();
    const int nbins = s_nbins;
    if (nbins == 0) return;   // block-uniform: no bin of any RoI reads this map

    const int per_pass = ROI_CHUNK * slices;
    for (int base = 0; base < a.R; base += per_pass) {
        if (threadIdx.x == 0) s_nrois = 0;
        __syncthreads();
        for (int k = threadIdx.x; k < ROI_CHUNK; k += RT) {
            const int n = base + k * slices + slice;
            if (n < a.R && (int)a.rois[(long)n * 5] == b) s_rois[atomicAdd(&s_nrois, 1)] = n;
        }
        __syncthreads();
        const int items = s_nrois * nbins;
        for (int item = threadIdx.x; item < items; item += RT) {
            const int n = s_rois[item / nbins], bin = s_bins[item % nbins];
            const RoiBox<T> box = roi_box(a.rois + (long)n * 5, a.scale);
            const BinGeom<T> g = bin_geom(a, box, n, cls, bin / a.PS, bin % a.PS);
            T sub_w, sub_h;
            {
#pragma clang fp contract(off)
                sub_h = box.h / (T)a.PS / (T)a.SPP;
                sub_w = box.w / (T)a.PS / (T)a.SPP;
            }
            T sum = T(0);
            int count = 0;
            for (int ih = 0; ih < a.SPP; ++ih) {
                for (int iw = 0; iw < a.SPP; ++iw) {
                    T w, h;
                    {
#pragma clang fp contract(off)
                        w = g.wstart + (T)iw * sub_w;
                        h = g.hstart + (T)ih * sub_h;
                    }
                    if (!keep_sample(w, h, a.W, a.H)) continue;
                    const int x1 = (int)floor(w), x2 = (int)ceil(w);
                    const int y1 = (int)floor(h), y2 = (int)ceil(h);
                    const T dx = w - (T)x1, dy = h - (T)y1;
                    const T v11 = s_map[y1 * a.W + x1], v12 = s_map[y2 * a.W + x1];
                    const T v21 = s_map[y1 * a.W + x2], v22 = s_map[y2 * a.W + x2];
                    sum += (T(1) - dx) * (T(1) - dy) * v11 + (T(1) - dx) * dy * v12 + dx * (T(1) - dy) * v21 +
                           dx * dy * v22;
                    ++count;
                }
            }
            const long oidx = ((long)n * a.OC + ctop) * bins + bin;
            a.out[oidx] = count == 0 ? T(0) : sum / (T)count;
            a.out_cnt[oidx] = (T)count;
        }
        __syncthreads();
    }
}

// RoIs whose batch index names no image of the batch are visited by no plane owner: their outputs are zero.
template <typename T>
__global__ __launch_bounds__(RT) void psroi_forward_orphans_kernel(const RoiArgs<T> a)
{
    const int n = blockIdx.x;
    const int batch = (int)a.rois[(long)n * 5];
    if (batch >= 0 && batch < a.B) return;
    const long per = (long)a.OC * a.PS * a.PS;
    for (long i = threadIdx.x; i < per; i += RT) {
        a.out[n * per + i] = T(0);
        a.out_cnt[n * per + i] = T(0);
    }
}

template <typename T>
int check_roi_geom(const ddcn_roi_geom *g, RoiArgs<T> &a, bool backward)
{
    if (!g) return fail(DDCN_ERR_BAD_ARG, "roi geom is NULL");
    if (g->B <= 0 || g->C <= 0 || g->H <= 0 || g->W <= 0)
        return fail(DDCN_ERR_SHAPE, "non-positive data size B=%d C=%d H=%d W=%d", g->B, g->C, g->H, g->W);
    if (g->R < 0) return fail(DDCN_ERR_SHAPE, "negative RoI count %d", g->R);
    if (g->out_channels <= 0 || g->pooled <= 0 || g->group_size <= 0 || g->part_size <= 0 || g->sample_per_part <= 0)
        return fail(DDCN_ERR_SHAPE, "non-positive out_channels=%d out_size=%d group_size=%d part_size=%d "
                    "sample_per_part=%d", g->out_channels, g->pooled, g->group_size, g->part_size, g->sample_per_part);
    // the kernels read score map (ctop*gs + gh)*gs + gw (PK:130): it has to exist
    if ((long)g->out_channels * g->group_size * g->group_size != g->C)
        return fail(DDCN_ERR_SHAPE, "data has %d channels, out_channels*group_size^2 = %ld", g->C,
                    (long)g->out_channels * g->group_size * g->group_size);
    int ncls = 1;
    if (!g->no_trans) {
        if (g->trans_channels < 2 || g->trans_channels % 2)
            return fail(DDCN_ERR_SHAPE, "offset has %d channels, need a positive even count", g->trans_channels);
        ncls = g->trans_channels / 2;  // PK:289
        if (g->out_channels % ncls)
            return fail(DDCN_ERR_SHAPE, "out_channels %d not divisible by the %d offset classes", g->out_channels, ncls);
    }
    if (!(g->trans_std >= 0.f && g->trans_std <= 1.f))  // deform_pool.py:40
        return fail(DDCN_ERR_BAD_ARG, "trans_std %g outside [0, 1]", (double)g->trans_std);
    if ((long)g->H * g->W >= (1L << 31) || (long)g->R * g->out_channels * g->pooled * g->pooled >= (1L << 40))
        return fail(DDCN_ERR_UNSUPPORTED, "tensor too large");
    if (backward && (g->pooled * g->pooled > MAX_BINS || g->part_size * g->part_size > MAX_BINS))
        return fail(DDCN_ERR_UNSUPPORTED, "backward supports out_size, part_size <= 32 (got %d, %d)", g->pooled,
                    g->part_size);
    a.B = g->B; a.C = g->C; a.H = g->H; a.W = g->W; a.R = g->R;
    a.OC = g->out_channels; a.PS = g->pooled; a.GS = g->group_size; a.PART = g->part_size;
    a.SPP = g->sample_per_part; a.NCLS = ncls; a.CEC = g->out_channels / ncls; a.no_trans = g->no_trans ? 1 : 0;
    a.scale = (T)g->spatial_scale; a.tstd = (T)g->trans_std;
    a.chunk = a.CEC;
    return DDCN_OK;
}


template <typename T>
static int psroi_forward(const T *data, const T *rois, const T *trans, T *out, T *out_count, const ddcn_roi_geom *g,
                         hipStream_t stream)
{
    RoiArgs<T> a = {};
    if (int rc = check_roi_geom(g, a, false)) return rc;
    if (a.R == 0) return DDCN_OK;
    if (!data || !rois || !out || !out_count || (!a.no_trans && !trans))
        return fail(DDCN_ERR_BAD_ARG, "psroi forward: NULL tensor pointer");
    a.data = data; a.rois = rois; a.trans = trans; a.out = out; a.out_cnt = out_count;
    // plane owners when a score map fits in LDS and the launch gets enough of them (DDCN_PSROI_PLANE_FWD=0: never)
    const size_t map_bytes = (size_t)a.H * a.W * sizeof(T);
    const long planes = (long)a.B * a.C;
    static const int env_plane = getenv("DDCN_PSROI_PLANE_FWD") ? atoi(getenv("DDCN_PSROI_PLANE_FWD")) : 1;
    if (env_plane && map_bytes <= (size_t)PLANE_LDS_MAX && a.PS * a.PS <= MAX_BINS && planes < (1L << 31) &&
        planes * ((a.R + 31) / 32) >= 256 && a.R <= 65535 * 32) {
        static std::atomic<uint64_t> raised{0};   // one per instantiation (T)
        if (!raise_lds_limit((const void *)psroi_forward_plane_kernel<T>, PLANE_LDS_MAX, raised))
            return fail(DDCN_ERR_LAUNCH, "psroi forward: cannot raise the LDS limit to %d bytes", PLANE_LDS_MAX);
        long rs = (2048 + planes - 1) / planes;
        const long rs_max = (a.R + 31) / 32;
        if (rs > rs_max) rs = rs_max;
        if (rs < 1) rs = 1;
        if (rs > 64) rs = 64;
        hipLaunchKernelGGL(psroi_forward_orphans_kernel<T>, dim3((unsigned)a.R), dim3(RT), 0, stream, a);
        hipLaunchKernelGGL(psroi_forward_plane_kernel<T>, dim3((unsigned)planes, (unsigned)rs), dim3(RT), map_bytes,
                           stream, a);
        return check_launch("psroi_forward_plane");
    }
    const long total = (long)a.R * a.OC * a.PS * a.PS;
    const long blocks = (total + RT - 1) / RT;
    hipLaunchKernelGGL(psroi_forward_kernel<T>, dim3((unsigned)(blocks < (1L << 20) ? blocks : (1L << 20))), dim3(RT), 0,
                       stream, a);
    return check_launch("psroi_forward");
}

template <typename T>
static int psroi_backward(const T *grad_out, const T *data, const T *rois, const T *trans, const T *out_count,
                          T *grad_data, T *grad_trans, const ddcn_roi_geom *g, hipStream_t stream)
{
    RoiArgs<T> a = {};
    if (int rc = check_roi_geom(g, a, true)) return rc;
    if (a.R == 0) return DDCN_OK;
    if (!grad_out || !data || !rois || !out_count || !grad_data || (!a.no_trans && (!trans || !grad_trans)))
        return fail(DDCN_ERR_BAD_ARG, "psroi backward: NULL tensor pointer");
    if ((long)a.R * a.NCLS >= (1L << 31)) return fail(DDCN_ERR_UNSUPPORTED, "too many RoIs");
    a.gout = grad_out; a.data = data; a.rois = rois; a.trans = trans; a.cnt = out_count;
    a.gdata = grad_data; a.gtrans = grad_trans;
    // Channels of a class are sliced over blockIdx.y until the launch has a few workgroups per CU; each slice still
    // spans at least one full pass of the workgroup over the bins.
    const long base_wgs = (long)a.R * a.NCLS;
    const int bins = a.PS * a.PS;
    const int min_chunk = (RT + bins - 1) / bins;
    int slices = (int)((1024 + base_wgs - 1) / base_wgs);
    int chunk = (a.CEC + slices - 1) / slices;
    if (chunk < min_chunk) chunk = min_chunk;
    if (chunk > a.CEC) chunk = a.CEC;
    a.chunk = chunk;
    slices = (a.CEC + chunk - 1) / chunk;
    if (slices > 65535) return fail(DDCN_ERR_UNSUPPORTED, "too many channel slices");
    // feature gradient: plane-owner kernel when a double plane fits in LDS, else atomics inside the offset kernel
    const size_t plane_bytes = (size_t)a.H * a.W * sizeof(double);
    const long planes = (long)a.B * a.C;
    const bool owner = plane_bytes <= (size_t)PLANE_LDS_MAX && planes < (1L << 31) && !getenv("DDCN_PSROI_ATOMIC");
    if (owner) {
        static std::atomic<uint64_t> raised0{0}, raised1{0};
        if (!raise_lds_limit((const void *)psroi_backward_plane_kernel<T, false>, PLANE_LDS_MAX, raised0) ||
            !raise_lds_limit((const void *)psroi_backward_plane_kernel<T, true>, PLANE_LDS_MAX, raised1))
            return fail(DDCN_ERR_LAUNCH, "psroi backward: cannot raise the LDS limit to %d bytes", PLANE_LDS_MAX);
        // RoI slices: enough workgroups to fill the chip, but each slice keeps >= 32 RoIs to amortise its plane
        long rs = (2048 + planes - 1) / planes;
        const long rs_max = (a.R + 31) / 32;
        if (rs > rs_max) rs = rs_max;
        if (rs < 1) rs = 1;
        if (rs > 64) rs = 64;
        // offset gradient inside the plane owners when the score map fits next to the gradient plane
        // (DDCN_PSROI_FUSED_TRANS=0: separate kernel)
        static const int env_fused = getenv("DDCN_PSROI_FUSED_TRANS") ? atoi(getenv("DDCN_PSROI_FUSED_TRANS")) : 1;
        const size_t both = plane_bytes + (size_t)a.H * a.W * sizeof(T);
        if (!a.no_trans && env_fused && both <= (size_t)PLANE_LDS_MAX) {
            hipLaunchKernelGGL((psroi_backward_plane_kernel<T, true>), dim3((unsigned)planes, (unsigned)rs), dim3(RT),
                               both, stream, a);
            return check_launch("psroi_backward_plane");
        }
        hipLaunchKernelGGL((psroi_backward_plane_kernel<T, false>), dim3((unsigned)planes, (unsigned)rs), dim3(RT),
                           plane_bytes, stream, a);
        if (int rc = check_launch("psroi_backward_plane")) return rc;
        if (a.no_trans) return DDCN_OK;
        hipLaunchKernelGGL((psroi_backward_kernel<T, false>), dim3((unsigned)base_wgs, (unsigned)slices), dim3(RT), 0,
                           stream, a);
    } else {
        hipLaunchKernelGGL((psroi_backward_kernel<T, true>), dim3((unsigned)base_wgs, (unsigned)slices), dim3(RT), 0,
                           stream, a);
    }
    return check_launch("psroi_backward");
}

}  // namespace

int launch_psroi_forward(const float *data, const float *rois, const float *trans, float *out, float *out_count,
                         const ddcn_roi_geom *g, hipStream_t stream)
{
    return psroi_forward<float>(data, rois, trans, out, out_count, g, stream);
}

int launch_psroi_backward(const float *grad_out, const float *data, const float *rois, const float *trans,
                          const float *out_count, float *grad_data, float *grad_trans, const ddcn_roi_geom *g,
                          hipStream_t stream)
{
    return psroi_backward<float>(grad_out, data, rois, trans, out_count, grad_data, grad_trans, g, stream);
}

int launch_psroi_forward_f64(const double *data, const double *rois, const double *trans, double *out,
                             double *out_count, const ddcn_roi_geom *g, hipStream_t stream)
{
    return psroi_forward<double>(data, rois, trans, out, out_count, g, stream);
}

int launch_psroi_backward_f64(const double *grad_out, const double *data, const double *rois, const double *trans,
                              const double *out_count, double *grad_data, double *grad_trans, const ddcn_roi_geom *g,
                              hipStream_t stream)
{
    return psroi_backward<double>(grad_out, data, rois, trans, out_count, grad_data, grad_trans, g, stream);
}

}  // namespace ddcn

// ddcn_modulated_deform_conv_forward_static_f32: the modulated / plain DDCN forward for callers that cannot allocate
// (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same kernels as
// ddcn_modulated_deform_conv_forward_f32 (im2col -> G1); the columns scratch that entry point takes as a caller
// workspace lives here in a file-scope device array of DDCN_STATIC_SCRATCH_BYTES per GPU.
//
// `scratch` below is a HOST-side stand-in for the device array: its device address is looked up once per device with
// hipGetSymbolAddress (each GPU holds its own copy at its own address) and only that address is ever handed to a
// kernel.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race.
// The columns are fully overwritten by im2col before G1 reads them, so nothing has to be cleared between calls.
#include <mutex>

#include "ddcn_common.h"

namespace ddcn {

static __device__ __align__(256) unsigned char scratch[DDCN_STATIC_SCRATCH_BYTES];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static void *g_scratch_ptr[kMaxDevices];   // device address of `scratch` on device i, nullptr until looked up

// Device address of the current device's copy of `scratch` (looked up on the first call per device).
static int scratch_of_current_device(void **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(DDCN_ERR_NO_DEVICE, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return fail(DDCN_ERR_UNSUPPORTED, "device id %d beyond %d", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(DDCN_ERR_LAUNCH, "hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return DDCN_OK;
}

}  // namespace ddcn

using namespace ddcn;

extern "C" {

size_t ddcn_static_scratch_bytes(void) { return (size_t)DDCN_STATIC_SCRATCH_BYTES; }

int ddcn_modulated_deform_conv_forward_static_f32(const float *x, const float *weight, const float *bias,
                                                  const float *offset, const float *mask, float *out,
                                                  const ddcn_geom *g, void *stream)
{
    if (!x || !weight || !offset || !out) return fail(DDCN_ERR_BAD_ARG, "%s: x, weight, offset and out must not be NULL", __func__);
    int rc = ddcn_check_geom(g);
    if (rc != DDCN_OK) return rc;
    const size_t need = ddcn_workspace_bytes(g);
    if (need > (size_t)DDCN_STATIC_SCRATCH_BYTES)
        return fail(DDCN_ERR_UNSUPPORTED,
                    "%s: the columns of this call need %zu bytes, the static scratch holds %zu (N*C*kh*kw*ld*4 must fit; "
                    "use ddcn_modulated_deform_conv_forward_f32 with a caller workspace)",
                    __func__, need, (size_t)DDCN_STATIC_SCRATCH_BYTES);
    void *ptr = nullptr;
    rc = scratch_of_current_device(&ptr);
    if (rc != DDCN_OK) return rc;
    float *col = static_cast<float *>(ptr);
    hipStream_t s = (hipStream_t)stream;
    rc = launch_im2col(x, offset, mask, col, *g, s);
    if (rc != DDCN_OK) return rc;
    return launch_gemm_out(weight, col, bias, out, *g, s);
}

}  // extern "C"

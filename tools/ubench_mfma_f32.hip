// Practical ceiling of v_mfma_f32_32x32x2_f32 on the whole chip: register-only MFMA loops, no memory traffic.
// usage: ubench_mfma_f32 [waves_per_simd] [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float seed)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a0 = seed * (1.f + 0.37f * threadIdx.x), a1 = seed * 2.13f - threadIdx.x * 0.011f, b0 = seed * 3.7f + threadIdx.x * 0.003f, b1 = -seed * 5.1f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    const float seed = argc > 3 ? (float)atof(argv[3]) : 0.f;
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float *out;
    hipMalloc(&out, 4);
    const int blocks = cus * wps;   // 256 threads = 4 waves = 1 wave per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, seed);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
        const double cyc = (double)iters * 64 * 64 * wps;   // MFMA cycles per SIMD if back to back
        printf("cus %d waves/simd %d: %.3f ms  %.1f TFLOP/s  implied clock %.2f GHz\n", cus, wps, ms, flops / ms / 1e9,
               cyc / ms / 1e6);
    }
    return 0;
}

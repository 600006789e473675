// Microbenchmark: LDS float-atomic throughput on gfx950 (diagnostic tool, not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, int region, int spread)
{
    extern __shared__ float s[];
    for (int i = threadIdx.x; i < region; i += 256) s[i] = 0.f;
    __syncthreads();
    unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.f;
    const int base = (threadIdx.x * 7) % (region - spread - 64);
    for (int it = 0; it < iters; ++it) {
        st = st * 1664525u + 1013904223u;
        const int idx = base + (int)((st >> 8) % (unsigned)spread);
        const float v = (float)(st & 255) * 0.001f;
        if (MODE == 0) atomicAdd(&s[idx], v);                                   // ds_add_f32
        if (MODE == 1) { float o = s[idx]; s[idx] = o + v; }                    // racy read-modify-write
        if (MODE == 2) atomicAdd((unsigned *)&s[idx], (unsigned)(st & 255));    // ds_add_u32
        if (MODE == 3) acc += s[idx];                                           // ds_read only
        if (MODE == 4) acc += atomicAdd(&s[idx], v);                            // ds_add_rtn_f32
        if (MODE == 5) {                                                        // CAS loop on the bit pattern
            unsigned *a = (unsigned *)&s[idx];
            unsigned old = *a, assumed;
            do {
                assumed = old;
                old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) + v));
            } while (old != assumed);
        }
        if (MODE == 6) atomicAdd((double *)&s[idx & ~1], (double)v);            // ds_add_f64
        if (MODE == 8) atomicAdd((double *)&s[((threadIdx.x + it * 3) * 2) % (region - 2)], (double)v);   // ds_add_f64, lanes on consecutive doubles
        if (MODE == 9) atomicAdd((unsigned *)&s[(threadIdx.x + it * 3) % region], (unsigned)(st & 255));      // ds_add_u32, consecutive
        if (MODE == 10) atomicAdd((unsigned long long *)&s[idx & ~1], (unsigned long long)(st & 255) << 20);   // ds_add_u64
        if (MODE == 11) atomicAdd((unsigned long long *)&s[((threadIdx.x + it * 3) * 2) % (region - 2)], (unsigned long long)(st & 255));  // ds_add_u64, consecutive
        if (MODE == 7) {                                                        // 4 CAS loops interleaved (4 corners)
            unsigned *a0 = (unsigned *)&s[idx], *a1 = a0 + 1, *a2 = a0 + 46, *a3 = a0 + 47;
            unsigned o0 = *a0, o1 = *a1, o2 = *a2, o3 = *a3;
            bool d0 = false, d1 = false, d2 = false, d3 = false;
            while (!(d0 && d1 && d2 && d3)) {
                unsigned r0 = o0, r1 = o1, r2 = o2, r3 = o3;
                if (!d0) r0 = atomicCAS(a0, o0, __float_as_uint(__uint_as_float(o0) + v));
                if (!d1) r1 = atomicCAS(a1, o1, __float_as_uint(__uint_as_float(o1) + v));
                if (!d2) r2 = atomicCAS(a2, o2, __float_as_uint(__uint_as_float(o2) + v));
                if (!d3) r3 = atomicCAS(a3, o3, __float_as_uint(__uint_as_float(o3) + v));
                d0 = d0 || r0 == o0; d1 = d1 || r1 == o1; d2 = d2 || r2 == o2; d3 = d3 || r3 == o3;
                o0 = r0; o1 = r1; o2 = r2; o3 = r3;
            }
        }
    }
    __syncthreads();
    float r = acc;
    for (int i = threadIdx.x; i < region; i += 256) r += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
static void run(const char *name, int iters, int region, int spread)
{
    float *out;
    hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 2048;
    k<MODE><<<grid, 256, region * 4>>>(out, iters, region, spread);
    hipEventRecord(a);
    k<MODE><<<grid, 256, region * 4>>>(out, iters, region, spread);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)grid * 4 * iters;
    // cycles per wave-instruction per CU at 2.4 GHz, 256 CUs
    printf("%-16s region=%5d spread=%5d  %.3f ms  %.1f cycles/wave-instr/CU\n", name, region, spread, ms,
           ms * 1e-3 * 2.4e9 / (wave_instr / 256));
    hipFree(out);
}

int main()
{
    for (int spread : {64, 1024}) {
        run<0>("ds_add_f32", 2000, 11408, spread);
        run<1>("read+write", 2000, 11408, spread);
        run<8>("ds_add_f64 linear", 2000, 11408, spread);
        run<9>("ds_add_u32 linear", 2000, 11408, spread);
        run<2>("ds_add_u32", 2000, 11408, spread);
        run<3>("ds_read", 2000, 11408, spread);
        run<4>("ds_add_rtn_f32", 2000, 11408, spread);
        run<5>("cas_loop", 2000, 11408, spread);
        run<6>("ds_add_f64", 2000, 11408, spread);
        run<10>("ds_add_u64", 2000, 11408, spread);
        run<11>("ds_add_u64 linear", 2000, 11408, spread);
        run<7>("cas_loop_x4 (4 adds)", 2000, 11408, spread);
    }
    return 0;
}

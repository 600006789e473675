// Microbenchmark: HBM write throughput of the im2col column-store pattern vs a linear fill (diagnostic tool).
#include <hip/hip_runtime.h>
#include <stdio.h>

// mode 0: linear fill, one dword per lane, grid-stride.  mode 1: 2-D tile pattern: WG = (tile of TH x TW pixels,
// 4 channels) writes 36 rows; lanes = tile pixels.  mode 2: as 1 but 256 consecutive pixels per WG (1-D strip).
// mode 3: as 2 with float4 stores (thread = 4 consecutive pixels, 64 threads per row-store).
__global__ __launch_bounds__(256) void k(float *col, long ld, int P, int Wo, int TW, int TH, int tiles_w, int tiles,
                                         int rows_total, int mode, long total)
{
    const int t = threadIdx.x;
    if (mode == 0) {
        for (long i = (long)blockIdx.x * 256 + t; i < total; i += (long)gridDim.x * 256) col[i] = 1.f;
        return;
    }
    const int tile = blockIdx.x % tiles, grp = blockIdx.x / tiles;
    const long row0 = (long)grp * 36;
    if (mode == 1) {
        const int tyi = tile / tiles_w, txi = tile % tiles_w;
        const int ty = t / TW, tx = t % TW;
        const int ho = tyi * TH + ty, wo = txi * TW + tx;
        if (ty >= TH || ho >= P / Wo || wo >= Wo) return;
        const int p = ho * Wo + wo;
        for (int r = 0; r < 36; ++r) col[(row0 + r) * ld + p] = (float)r;
    } else if (mode == 2) {
        const int p = tile * 256 + t;
        if (p >= P) return;
        for (int r = 0; r < 36; ++r) col[(row0 + r) * ld + p] = (float)r;
    } else {
        const int p = tile * 1024 + t * 4;
        if (p + 3 >= P) return;
        for (int r = 0; r < 36; ++r) *(float4 *)&col[(row0 + r) * ld + p] = make_float4(r, r, r, r);
    }
}

int main()
{
    const int Wo = 97, Ho = 97, P = Wo * Ho;
    const long ld = 9440, rows = 4608L * 2;
    const long total = rows * ld;
    float *col;
    hipMalloc(&col, total * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) {
        int TW = 25, TH = 10, tiles_w = 4, tiles = 40;
        int grid;
        if (mode == 0) grid = 256 * 16;
        else if (mode == 1) grid = tiles * (int)(rows / 36);
        else if (mode == 2) { tiles = (P + 255) / 256; grid = tiles * (int)(rows / 36); }
        else { tiles = (P + 1023) / 1024; grid = tiles * (int)(rows / 36); }
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            k<<<grid, 256>>>(col, ld, P, Wo, TW, TH, tiles_w, tiles, (int)rows, mode, total);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = mode == 0 ? total * 4.0 : (double)rows * P * 4.0;
        printf("mode %d: %.3f ms  %.0f GB/s\n", mode, ms, bytes / ms / 1e6);
    }
    return 0;
}

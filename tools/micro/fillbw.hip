// HBM write bandwidth of MI355X for the float64 K store (800 MB): plain / nontemporal 16-B stores,
// and the Gram epilogue's pattern (8 B per lane, 32 lanes = 256 contiguous bytes, two rows per wave store).
//   hipcc --offload-arch=gfx950 -O3 -o fillbw fillbw.hip && ./fillbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void fill16(double2* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) p[i] = make_double2(1.0, 2.0);
}
__global__ void fill16nt(double2* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) { __builtin_nontemporal_store(1.0, &p[i].x); __builtin_nontemporal_store(2.0, &p[i].y); }
}
// one workgroup (256 threads) per 128x128 tile of an N x N matrix, written like the MFMA epilogue
__global__ void filltile(double* K, int N, int tiles_n) {
    const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 2; ++nt) {
        const size_t col = (size_t)bn * 128 + (wn * 2 + nt) * 32 + (lane & 31);
        for (int q = 0; q < 4; ++q) for (int j = 0; j < 4; ++j) {
            const size_t row = (size_t)bm * 128 + (wm * 2 + mt) * 32 + 8 * q + 4 * (lane >> 5) + j;
            if (row < (size_t)N && col < (size_t)N) K[row * N + col] = (double)j;
        }
    }
}
int main() {
    const int N = 10000;
    const size_t bytes = (size_t)N * N * 8;
    double* p; hipMalloc(&p, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    for (int grid : {1024, 4096, 16384}) {
        for (int v = 0; v < 2; ++v) {
            float best = 1e9;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(a);
                if (v == 0) fill16<<<grid, 256>>>((double2*)p, bytes / 16); else fill16nt<<<grid, 256>>>((double2*)p, bytes / 16);
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("fill16%s grid %5d: %.3f ms  %.2f TB/s\n", v ? "nt" : "  ", grid, best, bytes / best / 1e9);
        }
    }
    const int tiles = (N + 127) / 128;
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(a); filltile<<<tiles * tiles, 256>>>(p, N, tiles);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("epilogue-pattern tile store: %.3f ms  %.2f TB/s\n", best, bytes / best / 1e9);
    return 0;
}

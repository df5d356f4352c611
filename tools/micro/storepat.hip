// Store-pattern micro-benchmark for the float64 Gram output: how fast can 256 persistent workgroups
// write an N x N float64 matrix when every workgroup writes TILES (rows of TS x 8 bytes, row stride 8 N bytes) instead
// of one linear stream?  The Gram kernel's store waves write exactly this pattern (tile rows + mirrored tile rows).
//   hipcc --offload-arch=gfx950 -O3 -o storepat storepat.hip && ./storepat [N]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void fill_linear(v2d* p, size_t n, int nt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    const v2d v = {1.0, 2.0};
    for (; i < n; i += s) { if (nt) __builtin_nontemporal_store(v, &p[i]); else p[i] = v; }
}

// upper-triangle tile t (row-major over bm <= bn) -> (bm, bn)
__device__ __forceinline__ void tri_tile(int t, int T, int& bm, int& bn) {
    bm = 0;
    while (t >= T - bm) { t -= T - bm; ++bm; }
    bn = bm + t;
}

// order 0: linear over the upper triangle; 1: 8x8 patches dealt to the XCDs (workgroup b runs on XCD b % 8) as the Gram kernel
// TSR x TSC tile: rows of TSC doubles.  WAVES store waves per workgroup; a wave writes whole rows (TSC * 8 / 1024 instr.)
template <int TSR, int TSC>
__global__ __launch_bounds__(512) void store_tiles(double* __restrict__ K, int N, int T_r, int T_c, int nt, int mirror, int order,
                                                  int waves) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w >= waves) return;
    const int n_tiles = mirror ? T_r * (T_r + 1) / 2 : T_r * T_c;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        int bm, bn;
        int tt = t;
        if (order == 1) {       // permute: consecutive ids of one XCD walk an 8x8 neighbourhood (approximation: block-cyclic)
            const int xcd = t & 7, i = t >> 3;
            tt = (i / 64) * 512 + xcd * 64 + (i % 64);
            if (tt >= n_tiles) tt = t;
        }
        if (mirror) tri_tile(tt, T_r, bm, bn); else { bm = tt / T_c; bn = tt % T_c; }
        const v2d v = {(double)bm, (double)bn};
        for (int pass = 0; pass < (mirror && bm != bn ? 2 : 1); ++pass) {
            const size_t r0 = (size_t)(pass ? bn : bm) * (pass ? TSC : TSR), c0 = (size_t)(pass ? bm : bn) * (pass ? TSR : TSC);
            const int rows = pass ? TSC : TSR, cols = pass ? TSR : TSC;
            for (int r = w; r < rows; r += waves) {
                if (r0 + r >= (size_t)N) break;
                double* dst = K + (r0 + r) * (size_t)N + c0;
                for (int c = 2 * lane; c < cols; c += 128)
                    if (c0 + c + 1 < (size_t)N) { if (nt) __builtin_nontemporal_store(v, (v2d*)(dst + c)); else *(v2d*)(dst + c) = v; }
            }
        }
    }
}

template <int TSR, int TSC>
static void run(const char* name, double* K, int N, int nt, int mirror, int order, int waves, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int T_r = (N + TSR - 1) / TSR, T_c = (N + TSC - 1) / TSC;
    float best = 1e9, ms;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(a);
        store_tiles<TSR, TSC><<<grid, 512>>>(K, N, T_r, T_c, nt, mirror, order, waves);
        hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-34s nt %d mirror %d order %d waves %d grid %4d: %.3f ms  %.2f TB/s\n", name, nt, mirror, order, waves, grid, best,
           (double)N * N * 8 / best / 1e9);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 10000;
    const size_t bytes = (size_t)N * N * 8;
    double* K; hipMalloc(&K, bytes + 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float ms;
    for (int nt = 0; nt < 2; ++nt) {
        float best = 1e9;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(a); fill_linear<<<4096, 256>>>((v2d*)K, bytes / 16, nt);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("linear fill nt %d: %.3f ms  %.2f TB/s\n", nt, best, bytes / best / 1e9);
    }
    for (int nt = 0; nt < 2; ++nt)
        for (int order = 0; order < 2; ++order) {
            run<128, 128>("tiles 128x128 (1 KiB rows)", K, N, nt, 1, order, 4, 256);
            run<128, 128>("tiles 128x128 (1 KiB rows)", K, N, nt, 1, order, 8, 256);
        }
    run<128, 128>("tiles 128x128 (1 KiB rows)", K, N, 1, 0, 0, 4, 256);
    run<128, 128>("tiles 128x128, 2 WG/CU", K, N, 1, 1, 0, 4, 512);
    run<256, 256>("tiles 256x256 (2 KiB rows)", K, N, 1, 1, 0, 4, 256);
    run<256, 256>("tiles 256x256 (2 KiB rows)", K, N, 1, 1, 0, 8, 256);
    run<128, 512>("tiles 128x512 (4 KiB rows)", K, N, 1, 0, 0, 4, 256);
    run<64, 1024>("tiles 64x1024 (8 KiB rows)", K, N, 1, 0, 0, 4, 256);
    run<128, 128>("tiles 128x128 full (no mirror)", K, N, 1, 0, 0, 8, 256);
    return 0;
}

// Micro-benchmark behind the Gram kernel's operand choice:
//   (1) exactness of small integer counts carried as MX fp4 (e2m1) operands with unit scales,
//   (2) issue rate of v_mfma_i32_32x32x32_i8, v_mfma_i32_16x16x64_i8 and
//       v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4) at 1 and 2 waves per SIMD on random data.
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// count 0..4 -> e2m1 code (0, 1.0, 2.0, 3.0, 4.0)
__host__ __device__ inline unsigned fp4_code(unsigned c) { return c == 0 ? 0u : (c == 1 ? 2u : (c == 2 ? 4u : (c == 3 ? 5u : 6u))); }

// one wave: D[32x32] = A[32x64] . B[32x64]^T with fp4 operands; A, B given as nibble images
// [32 rows][32 bytes] (two columns per byte, low nibble first)
__global__ void fp4_check_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, float* __restrict__ D) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const v4i a = *(const v4i*)(A + r * 32 + h * 16);
    const v4i b = *(const v4i*)(B + r * 32 + h * 16);
    v8i xa = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
    v8i xb = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, xb, acc, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * h, col = r;
        D[row * 32 + col] = acc[q];
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(const v4i* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    v4i a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[(lane + 64 * i) & 1023];
    for (int i = 0; i < 2; ++i) b[i] = src[(lane + 64 * (4 + i)) & 1023];
    float s = 0;
    if (MODE == 0) {            // i8 32x32x32
        v16i acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += (float)acc[i][j][q];
    } else if (MODE == 1) {     // i8 16x16x64
        v4i acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) acc[i][j][q] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) s += (float)acc[i][j][q];
    } else {                    // fp4 32x32x64 (scaled, unit scales)
        v16f acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0;
        v8i xa[4], xb[2];
        for (int i = 0; i < 4; ++i) { v4i t = a[i] & 0x66666666; xa[i] = (v8i){t[0], t[1], t[2], t[3], 0, 0, 0, 0}; }
        for (int i = 0; i < 2; ++i) { v4i t = b[i] & 0x66666666; xb[i] = (v8i){t[0], t[1], t[2], t[3], 0, 0, 0, 0}; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[i], xb[j], acc[i][j], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    }
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run_rate(const char* name, double ops_per_mfma, const v4i* src, float* out, int waves_per_simd) {
    const int iters = 4000, blocks = 256 * waves_per_simd;      // 256-thread blocks: 4 waves = one per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rate_kernel<MODE><<<blocks, 256>>>(src, out, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    rate_kernel<MODE><<<blocks, 256>>>(src, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double n_mfma = (double)blocks * 4 * iters * 8;
    const double tops = n_mfma * ops_per_mfma / (ms * 1e-3) / 1e12;
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 8.0 * waves_per_simd);
    printf("%-28s waves/SIMD %d: %8.1f TOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, waves_per_simd, tops, cyc);
}


template <int NACC>
__global__ __launch_bounds__(256) void dep_kernel(const v4i* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    v4i a = src[lane & 1023] & 0x66666666, b = src[(lane + 64) & 1023] & 0x66666666;
    v8i xa = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, xb = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    v16f acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, xb, acc[i], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    if (s == 12345.678f) out[0] = s;
}
template <int NACC>
static void run_dep(const v4i* src, float* out) {
    const int iters = 8000 / NACC;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dep_kernel<NACC><<<256, 256>>>(src, out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    dep_kernel<NACC><<<256, 256>>>(src, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("fp4 32x32x64, %d independent accumulators, 1 wave/SIMD: %.1f ns per MFMA (%.1f cycles at 2.4 GHz)\n", NACC,
           ms * 1e6 / (iters * NACC), ms * 1e-3 * 2.4e9 / (iters * NACC));
}

int main() {
    // ---- (1) exactness
    std::vector<unsigned char> A(32 * 32), B(32 * 32);
    std::vector<int> ca(32 * 64), cb(32 * 64);
    srand(7);
    for (int i = 0; i < 32 * 64; ++i) ca[i] = rand() % 5, cb[i] = rand() % 5;
    for (int r = 0; r < 32; ++r)
        for (int q = 0; q < 32; ++q) {
            A[r * 32 + q] = (unsigned char)(fp4_code(ca[r * 64 + 2 * q]) | (fp4_code(ca[r * 64 + 2 * q + 1]) << 4));
            B[r * 32 + q] = (unsigned char)(fp4_code(cb[r * 64 + 2 * q]) | (fp4_code(cb[r * 64 + 2 * q + 1]) << 4));
        }
    unsigned char *dA, *dB; float* dD;
    CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dD, 32 * 32 * 4));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    fp4_check_kernel<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(32 * 32);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int ref = 0;
            for (int k = 0; k < 64; ++k) ref += ca[i * 64 + k] * cb[j * 64 + k];
            if ((float)ref != D[i * 32 + j]) { if (bad < 5) printf("mismatch (%d,%d): %d vs %g\n", i, j, ref, D[i * 32 + j]); ++bad; }
        }
    printf("fp4 (e2m1, unit scale) integer counts 0..4, 32x32x64: %s (%d mismatches)\n", bad ? "WRONG" : "exact", bad);
    // ---- (2) rates
    std::vector<int> h(1024 * 4);
    for (auto& x : h) x = rand() * 65537 + rand();
    v4i* src; float* out;
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int w = 1; w <= 2; ++w) {
        run_rate<0>("v_mfma_i32_32x32x32_i8", 2.0 * 32 * 32 * 32, src, out, w);
        run_rate<1>("v_mfma_i32_16x16x64_i8", 2.0 * 16 * 16 * 64, src, out, w);
        run_rate<2>("v_mfma_scale_f32_32x32x64 fp4", 2.0 * 32 * 32 * 64, src, out, w);
    }
    run_dep<1>(src, out); run_dep<2>(src, out); run_dep<4>(src, out); run_dep<8>(src, out);
    return bad ? 1 : 0;
}

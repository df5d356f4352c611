// How fast can the CUs pull L2/MALL-resident operand panels into LDS (or VGPRs)?  The Gram kernel's
// K loop is bounded by exactly this path.  Every workgroup streams `rows` rows x 64 B per
// K-step out of a panel buffer, K-step after K-step, as the Gram kernel does (16-byte chunks, 4 lanes
// per row); workgroups of one XCD share panels.
//   mode 0: global_load_lds_dwordx4 into a 4-stage LDS ring, counted vmcnt + s_barrier per K-step
//   mode 1: global_load_dwordx4 into VGPRs (no LDS), results xor-reduced
//   mode 2: as 0 without the per-step barrier (vmcnt only)
// build: hipcc --offload-arch=gfx950 -O3 l2lds.hip -o l2lds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef int v4i __attribute__((ext_vector_type(4)));

// panel buffer: n_rows x ld bytes.  Workgroup b reads rows [r0(b), r0(b) + ROWS) (wrapping), all k_steps.
template <int WAVES, int PPW, int MODE, int CH>
__global__ __launch_bounds__(64 * WAVES) void stream_kernel(const int8_t* __restrict__ P, int64_t ld, int n_rows,
                                                           int k_steps, int reps, int row_stride, int* __restrict__ sink) {
    constexpr int LPR = CH / 16, RPI = 64 / LPR;   // lanes per row, rows per wave instruction
    constexpr int ROWS = WAVES * PPW * RPI, STAGE = ROWS * CH, NS = 4;
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int srow = lane / LPR, schunk = lane % LPR;
    // workgroups of one XCD (b % 8) walk neighbouring panels so that they share L2 lines
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int r0 = ((xcd * 37 + idx / row_stride) * ROWS) % (n_rows - ROWS);   // row_stride = workgroups sharing a panel
    const int8_t* g[PPW];
    int dst[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int r = (wave * PPW + q) * RPI;
        g[q] = P + (int64_t)(r0 + r + srow) * ld + schunk * 16;
        dst[q] = r * CH;
    }
    v4i acc = {0, 0, 0, 0};
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 1) {
            for (int kt = 0; kt < k_steps; ++kt) {
#pragma unroll
                for (int q = 0; q < PPW; ++q) acc ^= *(const v4i*)(g[q] + (int64_t)kt * CH);
            }
        } else {
            for (int p = 0; p < NS - 1 && p < k_steps; ++p) {
#pragma unroll
                for (int q = 0; q < PPW; ++q)
                    __builtin_amdgcn_global_load_lds((glb_void_t*)(g[q] + (int64_t)p * CH), (lds_void_t*)(smem + (p % NS) * STAGE + dst[q]), 16, 0, 0);
            }
            for (int kt = 0; kt < k_steps; ++kt) {
                if (kt + NS - 1 < k_steps) {
#pragma unroll
                    for (int q = 0; q < PPW; ++q)
                        __builtin_amdgcn_global_load_lds((glb_void_t*)(g[q] + (int64_t)(kt + NS - 1) * CH),
                                                         (lds_void_t*)(smem + ((kt + NS - 1) % NS) * STAGE + dst[q]), 16, 0, 0);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PPW) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (MODE == 0) __builtin_amdgcn_s_barrier();
                acc[0] ^= *(const int*)(smem + (kt % NS) * STAGE + lane * 4);
            }
        }
    }
    if (acc[0] == 0x12345 && acc[1] == 7) sink[0] = acc[2] + acc[3];
}

template <int WAVES, int PPW, int MODE, int CH>
static void run(const char* name, const int8_t* P, int64_t ld, int n_rows, int k_steps, int wg_per_cu, int row_stride, int* sink) {
    constexpr int ROWS = WAVES * PPW * (1024 / CH), LDS = 4 * ROWS * CH;
    auto kern = stream_kernel<WAVES, PPW, MODE, CH>;
    if (LDS > 160 * 1024) return;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int blocks = 256 * wg_per_cu, reps = 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<<<blocks, 64 * WAVES, LDS>>>(P, ld, n_rows, k_steps, 1, row_stride, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    kern<<<blocks, 64 * WAVES, LDS>>>(P, ld, n_rows, k_steps, reps, row_stride, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)blocks * reps * k_steps * ROWS * CH;
    printf("%-34s CH %3d rows/step %4d waves %2d wg/CU %d share %2d: %7.2f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", name, CH, ROWS, WAVES,
           wg_per_cu, row_stride, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    const int n_rows = 10368;
    const int64_t ld = 2432;            // config 3: 38 K-steps of 64 B
    const int k_steps = 38;
    int8_t* P; int* sink;
    CK(hipMalloc(&P, (size_t)n_rows * ld)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(P, 1, (size_t)n_rows * ld));
    // round 6: does the L2 -> LDS rate of a CU depend on HOW MANY waves issue the pieces?  (the Gram kernel's load role is four
    // waves x 8 pieces per 128-byte K-step of a 128 + 128-row stage; "share 8" is its panel sharing)
    for (int mode : {0, 2}) {
        printf("-- 256 rows x 128 B per K-step (32 KiB stage), %s\n", mode == 0 ? "barrier per K-step" : "no barrier");
        if (mode == 0) {
            run<2, 16, 0, 128>("glds  2 waves x 16 pieces", P, ld, n_rows, 19, 1, 8, sink);
            run<4, 8, 0, 128>("glds  4 waves x 8 pieces", P, ld, n_rows, 19, 1, 8, sink);
            run<8, 4, 0, 128>("glds  8 waves x 4 pieces", P, ld, n_rows, 19, 1, 8, sink);
            run<16, 2, 0, 128>("glds 16 waves x 2 pieces", P, ld, n_rows, 19, 1, 8, sink);
        } else {
            run<2, 16, 2, 128>("glds  2 waves x 16 pieces", P, ld, n_rows, 19, 1, 8, sink);
            run<4, 8, 2, 128>("glds  4 waves x 8 pieces", P, ld, n_rows, 19, 1, 8, sink);
            run<8, 4, 2, 128>("glds  8 waves x 4 pieces", P, ld, n_rows, 19, 1, 8, sink);
            run<16, 2, 2, 128>("glds 16 waves x 2 pieces", P, ld, n_rows, 19, 1, 8, sink);
        }
    }
    for (int share : {1, 2, 4, 8, 32}) {
        run<4, 4, 0, 64>("glds 256 rows x 64 B", P, ld, n_rows, 38, 2, share, sink);
        run<8, 4, 0, 64>("glds 512 rows x 64 B", P, ld, n_rows, 38, 1, share, sink);
        run<4, 8, 0, 128>("glds 256 rows x 128 B", P, ld, n_rows, 19, 1, share, sink);
        run<8, 4, 0, 128>("glds 256 rows x 128 B (8 waves)", P, ld, n_rows, 19, 1, share, sink);
        run<8, 4, 1, 64>("global_load 512 rows x 64 B", P, ld, n_rows, 38, 2, share, sink);
    }
    return 0;
}

// Micro-benchmark: cost of a cooperative launch and of a global (all-workgroup) barrier on
// MI355X, to decide whether multi-pass kernels (radix passes, scans) should be fused.
//   hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar.hip && ./gridbar
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32;

__device__ __forceinline__ bool grid_barrier(u32* bar, u32 nblocks) {
    bool ok = true;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const u32 gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u32 arrived = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (arrived == nblocks) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            u32 spins = 0;
            while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { ok = false; break; }
            }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void bar_kernel(u32* bar, u32* data, int nbar, u32* err) {
    u32 acc = 0;
    for (int i = 0; i < nbar; ++i) {
        data[(size_t)blockIdx.x * 256 + threadIdx.x] = i + blockIdx.x;      // something to publish
        if (!grid_barrier(bar, gridDim.x)) { if (threadIdx.x == 0) atomicAdd(err, 1u); return; }
        acc += data[(size_t)((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];   // read the neighbour's value
    }
    // neighbour wrote i + its block id at every step
    u32 want = 0;
    for (int i = 0; i < nbar; ++i) want += i + (blockIdx.x + 1) % gridDim.x;
    if (acc != want) atomicAdd(err + 1, 1u);
}

__global__ void empty_kernel(u32* p) { if (p == nullptr) p[0] = 1; }

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bar_kernel, 256, 0);
    printf("CUs %d, cooperativeLaunch %d, resident blocks/CU %d\n", prop.multiProcessorCount, prop.cooperativeLaunch, per_cu);
    u32 *bar, *data, *err;
    hipMalloc(&bar, 8); hipMemset(bar, 0, 8);
    hipMalloc(&data, 2048 * 256 * 4); hipMalloc(&err, 8); hipMemset(err, 0, 8);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grids[] = {25, 64, 256, 489, 512, 1024};
    for (int gi = 0; gi < 6; ++gi) {
        int grid = grids[gi];
        if (grid > per_cu * prop.multiProcessorCount) continue;
        for (int coop = 0; coop < 2; ++coop)
            for (int nbar : {0, 10, 40}) {
                float best = 1e9;
                for (int rep = 0; rep < 6; ++rep) {
                    void* args[] = {&bar, &data, &nbar, &err};
                    hipEventRecord(e0, s);
                    for (int k = 0; k < 10; ++k) {
                        if (coop) {
                            if (hipLaunchCooperativeKernel((void*)bar_kernel, dim3(grid), dim3(256), args, 0, s) != hipSuccess) { printf("coop launch failed\n"); return 1; }
                        } else hipLaunchKernelGGL(bar_kernel, dim3(grid), dim3(256), 0, s, bar, data, nbar, err);
                    }
                    hipEventRecord(e1, s); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("grid %4d coop %d barriers %2d: %.2f us per launch\n", grid, coop, nbar, best * 100.0f);
            }
    }
    u32 h[2]; hipMemcpy(h, err, 8, hipMemcpyDeviceToHost);
    printf("barrier timeouts %u, wrong sums %u\n", h[0], h[1]);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, s);
        for (int k = 0; k < 100; ++k) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, bar);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("empty kernel back-to-back: %.2f us per launch\n", best * 10.0f);
    return 0;
}

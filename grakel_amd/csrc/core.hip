// k-core decomposition of every graph of a batch (CoreFramework, SURVEY.md 8f-2;
// reference: core_number(), grakel/kernels/core_framework.py:376-416 -- the Batagelj-Zaversnik
// bin walk, sequential per graph).  Here one workgroup per graph peels in parallel: with the
// degrees in LDS, every vertex whose remaining degree is <= k leaves with core number k and
// lowers its neighbours' degrees; when a sweep removes nothing, k grows.  Same core numbers,
// O(n + m) LDS work per sweep.  Graphs are taken as undirected (symmetric CSR), like the
// reference's neighbour lists.
#include "common.h"

#define CORE_THREADS 256
#define CORE_MAX_N 16384          // 2 x int32 per vertex in LDS (128 KiB)

__global__ __launch_bounds__(CORE_THREADS) void core_number_kernel(
    const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx,
    i32* __restrict__ core_out, u32* __restrict__ too_big) {
    extern __shared__ i32 core_lds[];
    __shared__ int changed, remaining;
    const int tid = threadIdx.x;
    const i32 v0 = graph_ptr[blockIdx.x];
    const int n = graph_ptr[blockIdx.x + 1] - v0;
    if (n > CORE_MAX_N) {
        if (tid == 0) atomicMax(too_big, (u32)n);
        return;
    }
    i32* deg = core_lds;
    i32* core = core_lds + n;
    for (int v = tid; v < n; v += CORE_THREADS) {
        deg[v] = row_ptr[v0 + v + 1] - row_ptr[v0 + v];
        core[v] = -1;
    }
    if (tid == 0) remaining = n;
    __syncthreads();
    for (int k = 0; remaining > 0; ++k) {
        for (;;) {
            __syncthreads();
            if (tid == 0) changed = 0;
            __syncthreads();
            int gone = 0;
            for (int v = tid; v < n; v += CORE_THREADS) {
                if (core[v] < 0 && deg[v] <= k) {
                    core[v] = k;
                    ++gone;
                    for (i32 e = row_ptr[v0 + v]; e < row_ptr[v0 + v + 1]; ++e) {
                        const int u = col_idx[e] - v0;
                        if (u != v) atomicSub(&deg[u], 1);      // leaving vertices may be hit too: harmless
                    }
                }
            }
            if (gone) { atomicSub(&remaining, gone); changed = 1; }
            __syncthreads();
            if (!changed) break;
        }
    }
    for (int v = tid; v < n; v += CORE_THREADS) core_out[v0 + v] = core[v];
}

extern "C" int gk_core_numbers(gk_ctx* ctx, gk_batch* b, int32_t* out_core) {
    GK_ARG(ctx && b && out_core, "gk_core_numbers: null argument");
    GK_ARG(!b->is_pair_batch, "gk_core_numbers: needs a graph batch");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ProfScope prof(ctx, "core");
    const i64 N = b->n_graphs, V = b->n_nodes;
    if (V == 0) return GK_OK;
    Tmp<i32> core(ctx);
    Tmp<u32> too_big(ctx);
    GK_TRY(core.alloc(V)); GK_TRY(too_big.alloc(1));
    GK_TRY(gk_zero_async(ctx, too_big.p, 4));
    i64 nmax = b->max_graph_nodes < CORE_MAX_N ? b->max_graph_nodes : CORE_MAX_N;
    const size_t lds = (size_t)(nmax > 0 ? nmax : 1) * 8;
    GK_TRY(gk_func_lds(ctx, (const void*)core_number_kernel, (int)lds));
    core_number_kernel<<<dim3((unsigned)N), CORE_THREADS, lds, ctx->stream>>>(b->graph_ptr, b->row_ptr, b->col_idx, core.p, too_big.p);
    GK_HIP_CHECK(hipGetLastError());
    u32 h_big = 0;
    GK_HIP_CHECK(hipMemcpyAsync(&h_big, too_big.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipMemcpyAsync(out_core, core.p, (size_t)V * 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (h_big) {
        gk_set_error("gk_core_numbers: a graph has %u vertices, the LDS peeling kernel takes at most %d", h_big, CORE_MAX_N);
        return GK_ERR_ARG;
    }
    return GK_OK;
}

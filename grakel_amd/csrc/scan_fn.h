// Prefix sum over values produced on the fly by a functor, with the consumer fused in:
//     struct F { __device__ T value(i64 i) const;  __device__ void emit(i64 i, T v, T inclusive) const;
//                __device__ void finish(T total) const;        // one thread, after the last emit
//                __device__ i64 seg_first_tile(i64 tile) const; };  // first tile of the tile's segment
// Segments (e.g. one per WL level, each padded to whole tiles) restart the sum at zero; an
// unsegmented functor returns 0.
// Two kernels (tile sums, then apply-with-direct-offset as in scan_sort.hip); no flag / scan
// arrays ever touch HBM.  Used for run-head -> label id, triple emission and column ids.
#pragma once
#include "common.h"

#define SF_THREADS 256
#define SF_ITEMS 8
#define SF_TILE (SF_THREADS * SF_ITEMS)

template <typename T>
__device__ __forceinline__ T sf_wave_incl_scan(T x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    return x;
}

template <typename T, typename F>
__global__ __launch_bounds__(SF_THREADS) void scan_fn_sums_kernel(F f, T* __restrict__ partial, i64 n) {
    __shared__ T wsum[SF_THREADS / 64];
    const i64 base = (i64)blockIdx.x * SF_TILE;
    T s = 0;
#pragma unroll
    for (int i = 0; i < SF_ITEMS; ++i) {
        const i64 idx = base + (i64)i * SF_THREADS + threadIdx.x;   // striped: coalesced
        if (idx < n) s += f.value(idx);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = 0;
        for (int i = 0; i < SF_THREADS / 64; ++i) t += wsum[i];
        partial[blockIdx.x] = t;
    }
}

// Striped tile: in row i thread t owns element tile0 + i*256 + t, so value()/emit() of a wave
// touch 64 consecutive elements (coalesced gathers/scatters in the functors).  One block-wide
// scan per row; the carry links the rows.
// PREFIXED: partial[] already holds the EXCLUSIVE prefix of the tile sums (sf_scan_partials_kernel) -- the carry of a tile is
// one subtraction.  Otherwise (few tiles) every block adds up the sums of the tiles before it in its segment itself, which
// saves the middle kernel but is quadratic in the tile count: 643 M ShortestPath pair items (314 k tiles) spent 22 ms per scan
// there (round 5).
template <typename T, typename F, bool PREFIXED>
__global__ __launch_bounds__(SF_THREADS) void scan_fn_apply_kernel(F f, const T* __restrict__ partial, i64 n,
                                                                   T* __restrict__ total) {
    __shared__ T wsum[SF_THREADS / 64];
    __shared__ T bsum[SF_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T carry = 0;
    if (PREFIXED) {
        carry = partial[blockIdx.x] - partial[f.seg_first_tile(blockIdx.x)];
    } else {
        T s = 0;
        for (int i = (int)f.seg_first_tile(blockIdx.x) + threadIdx.x; i < (int)blockIdx.x; i += SF_THREADS) s += partial[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) bsum[w] = s;
        __syncthreads();
        for (int q = 0; q < SF_THREADS / 64; ++q) carry += bsum[q];
    }
    const i64 tile0 = (i64)blockIdx.x * SF_TILE;
#pragma unroll
    for (int i = 0; i < SF_ITEMS; ++i) {
        const i64 idx = tile0 + (i64)i * SF_THREADS + threadIdx.x;
        const T v = idx < n ? f.value(idx) : (T)0;
        const T inc = sf_wave_incl_scan(v);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        T woff = 0, row = 0;
#pragma unroll
        for (int q = 0; q < SF_THREADS / 64; ++q) {
            const T x = wsum[q];
            if (q < w) woff += x;
            row += x;
        }
        if (idx < n) f.emit(idx, v, carry + woff + inc);
        carry += row;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        if (total) *total = carry;
        f.finish(carry);
    }
}

// one block: partial[0 .. m) -> its exclusive prefix, in place
template <typename T>
__global__ __launch_bounds__(1024) void sf_scan_partials_kernel(T* __restrict__ partial, i64 m) {
    __shared__ T wsum[16];
    __shared__ T carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (i64 c0 = 0; c0 < m; c0 += 1024) {
        const i64 idx = c0 + threadIdx.x;
        const T x = idx < m ? partial[idx] : (T)0;
        const T inc = sf_wave_incl_scan(x);
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        T woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        const T carry = carry_s;
        if (idx < m) partial[idx] = carry + woff + inc - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
}
#define SF_DIRECT_MAX 4096          // tiles up to which every block sums its predecessors itself

// total (device, may be null) receives the grand total.  n == 0: total = 0.
template <typename T, typename F>
static int gk_scan_fn(gk_ctx* ctx, const F& f, i64 n, T* total) {
    if (n <= 0) {
        if (total) GK_TRY(gk_zero_async(ctx, total, sizeof(T)));
        return GK_OK;
    }
    const i64 nblk = cdiv(n, SF_TILE);
    Tmp<T> partial(ctx);
    GK_TRY(partial.alloc(nblk));
    if (nblk > 1)      // a single tile has nothing before it (the apply kernel reads partial[0 .. blockIdx))
        scan_fn_sums_kernel<T, F><<<dim3((unsigned)nblk), dim3(SF_THREADS), 0, ctx->stream>>>(f, partial.p, n);
    if (nblk > (ctx->opt.scan_direct_max > 0 ? (i64)ctx->opt.scan_direct_max : (i64)SF_DIRECT_MAX)) {
        sf_scan_partials_kernel<T><<<dim3(1), dim3(1024), 0, ctx->stream>>>(partial.p, nblk);
        scan_fn_apply_kernel<T, F, true><<<dim3((unsigned)nblk), dim3(SF_THREADS), 0, ctx->stream>>>(f, partial.p, n, total);
    } else
        scan_fn_apply_kernel<T, F, false><<<dim3((unsigned)nblk), dim3(SF_THREADS), 0, ctx->stream>>>(f, partial.p, n, total);
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

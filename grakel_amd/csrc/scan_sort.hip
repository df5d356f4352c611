// Device-wide prefix sums and a stable LSD radix sort for (u64 key, u32 value) pairs.
// Hand-written for gfx950: 64-lane wave scans via DPP shuffles, wave-ballot digit matching
// for the stable scatter.  Used by the WL dictionary (sort node signatures), the feature
// builder (head flags -> run ids) and the ShortestPath pair dictionary.
#include "common.h"

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

template <typename T>
__device__ __forceinline__ T wave_incl_scan(T x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    return x;
}

template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums_kernel(const T* __restrict__ in,
                                                                       T* __restrict__ partial,
                                                                       i64 n) {
    __shared__ T wsum[SCAN_THREADS / 64];
    const i64 base = (i64)blockIdx.x * SCAN_TILE;
    T s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        i64 idx = base + (i64)i * SCAN_THREADS + threadIdx.x;   // striped: coalesced
        if (idx < n) s += in[idx];
    }
    T w = wave_incl_scan(s);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        T t = 0;
        for (int i = 0; i < SCAN_THREADS / 64; ++i) t += wsum[i];
        partial[blockIdx.x] = t;
    }
}

// One block: partial[] -> exclusive prefix in place; grand total to *total (if non-null).
// (only used when there are more tiles than SCAN_DIRECT_MAX)
template <typename T>
__global__ __launch_bounds__(1024) void scan_partials_kernel(T* __restrict__ partial, i64 m,
                                                              T* __restrict__ total) {
    __shared__ T wsum[16];
    __shared__ T carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (i64 c0 = 0; c0 < m; c0 += 1024) {
        i64 idx = c0 + threadIdx.x;
        T x = idx < m ? partial[idx] : (T)0;
        T inc = wave_incl_scan(x);
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        T woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        T carry = carry_s;
        if (idx < m) partial[idx] = carry + woff + inc - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry_s;
}

// DIRECT: partial[] holds raw tile sums; every block first adds up the sums of the tiles
// before it (a few hundred values, L2 resident) -- saves the single-block middle kernel.
template <typename T, bool EXCL, bool DIRECT>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const T* in, T* out, i64 n,
                                                                   const T* __restrict__ partial,
                                                                   T* __restrict__ total) {
    // in and out may alias (in-place scans): no __restrict__ on them
    __shared__ T wsum[SCAN_THREADS / 64];
    __shared__ T bsum[SCAN_THREADS / 64];
    T block_off;
    if (DIRECT) {
        T s = 0;
        for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_THREADS) s += partial[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) bsum[threadIdx.x >> 6] = s;
        __syncthreads();
        block_off = 0;
        for (int w = 0; w < SCAN_THREADS / 64; ++w) block_off += bsum[w];
    } else {
        block_off = partial[blockIdx.x];
    }
    const i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_ITEMS;   // blocked
    T v[SCAN_ITEMS];
    T run = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        i64 idx = base + i;
        v[i] = idx < n ? in[idx] : (T)0;
        run += v[i];
    }
    T inc = wave_incl_scan(run);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    T woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    T acc = inc - run + woff + block_off;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        i64 idx = base + i;
        if (idx < n) {
            if (EXCL) out[idx] = acc;
            acc += v[i];
            if (!EXCL) out[idx] = acc;
        }
    }
    if (DIRECT && total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1)
        *total = acc;   // last thread of the last tile: grand total
}

#define SCAN_DIRECT_MAX 8192

template <typename T>
static int scan_impl(gk_ctx* ctx, const T* in, T* out, i64 n, bool exclusive, T* total) {
    if (n <= 0) {
        if (total) GK_TRY(gk_zero_async(ctx, total, sizeof(T)));
        return GK_OK;
    }
    i64 nblk = cdiv(n, SCAN_TILE);
    Tmp<T> partial(ctx);
    GK_TRY(partial.alloc(nblk));
    scan_tile_sums_kernel<T><<<dim3((unsigned)nblk), dim3(SCAN_THREADS), 0, ctx->stream>>>(in, partial.p, n);
    const dim3 g((unsigned)nblk), t(SCAN_THREADS);
    if (nblk <= SCAN_DIRECT_MAX) {
        if (exclusive) scan_apply_kernel<T, true, true><<<g, t, 0, ctx->stream>>>(in, out, n, partial.p, total);
        else scan_apply_kernel<T, false, true><<<g, t, 0, ctx->stream>>>(in, out, n, partial.p, total);
    } else {
        scan_partials_kernel<T><<<dim3(1), dim3(1024), 0, ctx->stream>>>(partial.p, nblk, total);
        if (exclusive) scan_apply_kernel<T, true, false><<<g, t, 0, ctx->stream>>>(in, out, n, partial.p, total);
        else scan_apply_kernel<T, false, false><<<g, t, 0, ctx->stream>>>(in, out, n, partial.p, total);
    }
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

int gk_scan_u32(gk_ctx* ctx, const u32* in, u32* out, i64 n, bool exclusive, u32* total) {
    return scan_impl<u32>(ctx, in, out, n, exclusive, total);
}
int gk_scan_u64(gk_ctx* ctx, const u64* in, u64* out, i64 n, bool exclusive, u64* total) {
    return scan_impl<u64>(ctx, in, out, n, exclusive, total);
}

// ---------------------------------------------------------------------------------------
// Radix sort: 8-bit digits, per pass  histogram -> exclusive scan -> stable scatter.
// A block owns a tile of RS_TILE consecutive keys and processes it in RS_ROUNDS rounds of
// 256 keys (one per thread, in index order) so that stability only needs (a) the rank of a
// key among equal digits inside its wave (ballot match) and (b) a per-digit running count
// across waves and rounds, which thread d keeps in a register for digit d.
// ---------------------------------------------------------------------------------------
#define RS_THREADS 256
#define RS_ROUNDS 8
#define RS_TILE (RS_THREADS * RS_ROUNDS)

__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const u64* __restrict__ kin, i64 n,
                                                                int shift, u32* __restrict__ hist,
                                                                int nblk) {
    __shared__ u32 h[256];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const i64 tile0 = (i64)blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        i64 idx = tile0 + r * RS_THREADS + tid;
        if (idx < n) {
            u32 d = (u32)(kin[idx] >> shift) & 255u;
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    hist[(i64)tid * nblk + blockIdx.x] = h[tid];
}

// Stable scatter of one 2048-key tile.  Keys are taken in index order: round r, thread t holds
// key tile0 + r*256 + t.  Phase 1: for every (round, wave) the lanes with equal digits find
// each other with 8 ballots; the lowest lane of a group records the group size in
// cnt[r][w][digit] and every lane keeps its rank inside the group.  Phase 2: thread d turns the
// 32 counts of digit d into exclusive offsets in (round, wave) order -- exactly the order the
// keys must keep.  Phase 3: position = global base[d] + offset[r][w][d] + rank.  Three barriers
// per tile instead of four per round.
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(
    const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout,
    u32* __restrict__ vout, i64 n, int shift, const u32* __restrict__ offs, int nblk) {
    constexpr int NWAVE = RS_THREADS / 64;
    __shared__ u32 cnt[RS_ROUNDS * NWAVE * 256];     // 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const i64 tile0 = (i64)blockIdx.x * RS_TILE;
    u64 key[RS_ROUNDS];
    u32 val[RS_ROUNDS], rank[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const i64 idx = tile0 + r * RS_THREADS + tid;
        const bool act = idx < n;
        key[r] = act ? kin[idx] : 0ull;
        val[r] = act ? (vin ? vin[idx] : (u32)idx) : 0u;
    }
#pragma unroll
    for (int q = 0; q < RS_ROUNDS * NWAVE; ++q) cnt[q * 256 + tid] = 0;
    __syncthreads();
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const bool act = tile0 + r * RS_THREADS + tid < n;
        const u32 d = (u32)(key[r] >> shift) & 255u;
        u64 m = __ballot(act);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const u64 bb = __ballot(act && bit);
            m &= bit ? bb : ~bb;
        }
        rank[r] = (u32)__popcll(m & lt);
        if (act && rank[r] == 0) cnt[(r * NWAVE + w) * 256 + d] = (u32)__popcll(m);
    }
    __syncthreads();
    {
        u32 run = offs[(i64)tid * nblk + blockIdx.x];    // global base of digit `tid` for this tile
#pragma unroll
        for (int q = 0; q < RS_ROUNDS * NWAVE; ++q) {
            const u32 c = cnt[q * 256 + tid];
            cnt[q * 256 + tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        if (tile0 + r * RS_THREADS + tid < n) {
            const u32 d = (u32)(key[r] >> shift) & 255u;
            const u32 pos = cnt[(r * NWAVE + w) * 256 + d] + rank[r];
            kout[pos] = key[r];
            vout[pos] = val[r];
        }
    }
}

int gk_radix_sort_pairs(gk_ctx* ctx, u64* keys_in, u32* vals_in, u64* keys_out, u32* vals_out,
                        i64 n, int key_bits, bool implicit_iota) {
    // Result always lands in keys_out/vals_out; keys_in/vals_in are scratch afterwards.
    if (n <= 0) return GK_OK;
    if (key_bits < 0) key_bits = 0;
    if (key_bits > 64) key_bits = 64;
    int passes = (key_bits + 7) / 8;
    if (passes == 0) passes = 1;   // nothing to distinguish: one (trivial) pass keeps the code paths uniform
    int nblk = (int)cdiv(n, RS_TILE);
    Tmp<u32> hist(ctx);
    GK_TRY(hist.alloc((size_t)256 * nblk));
    u64 *ks = keys_in, *kd = keys_out;
    u32 *vs = vals_in, *vd = vals_out;
    // vals_in may be null on entry: the values are then the indices 0..n-1 (implicit iota);
    // vals_scratch is the ping-pong partner of vals_out in that case
    if ((passes & 1) == 0) {   // even number of passes: start from the out buffers
        GK_HIP_CHECK(hipMemcpyAsync(keys_out, keys_in, n * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
        if (vals_in && !implicit_iota)
            GK_HIP_CHECK(hipMemcpyAsync(vals_out, vals_in, n * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
        ks = keys_out, kd = keys_in, vs = vals_out, vd = vals_in;
    }
    for (int p = 0; p < passes; ++p) {
        int shift = p * 8;
        radix_hist_kernel<<<dim3(nblk), dim3(RS_THREADS), 0, ctx->stream>>>(ks, n, shift, hist.p, nblk);
        GK_TRY(gk_scan_u32(ctx, hist.p, hist.p, (i64)256 * nblk, true, nullptr));
        radix_scatter_kernel<<<dim3(nblk), dim3(RS_THREADS), 0, ctx->stream>>>(
            ks, (implicit_iota && p == 0) ? nullptr : vs, kd, vd, n, shift, hist.p, nblk);
        u64* tk = ks; ks = kd; kd = tk;
        u32* tv = vs; vs = vd; vd = tv;
    }
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

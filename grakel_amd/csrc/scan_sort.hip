// Device-wide prefix sums and a stable radix sort for (u64 key, u32 value) pairs.
// Hand-written for gfx950: 64-lane wave scans via DPP shuffles, wave-ballot digit matching
// for the stable scatter, an in-LDS finish per top-digit bucket.  Used by the WL dictionary (sort node signatures), the feature
// builder (head flags -> run ids) and the ShortestPath pair dictionary.
#include "common.h"
#include <stdlib.h>

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

// Digit histogram without same-address pile-ups: the lanes of a wave that hold the same digit find
// each other with 8 ballots and their leader adds the group size once.  Equal keys (a WL class of
// thousands of nodes has ONE hash) would otherwise serialise thousands of LDS atomics on one counter.
__device__ __forceinline__ void wave_digit_add(u32* __restrict__ counters, bool act, u32 d) {
    u64 m = __ballot(act);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const u64 bb = __ballot(act && bit);
        m &= bit ? bb : ~bb;
    }
    const u64 lt = (1ull << (threadIdx.x & 63)) - 1ull;
    if (act && (m & lt) == 0ull) atomicAdd(&counters[d], (u32)__popcll(m));
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
// Radix sort: 8-bit digits, stable.  A kernel boundary (~2.6 us back to back on MI355X) is the
// cheapest device-wide synchronisation available -- measured: a global atomic barrier costs
// ~0.14 us PER WORKGROUP and a cooperative launch +18 us (tools/micro/gridbar.hip) -- so the
// passes are organised to need as few kernels as possible:
//   n <= RS_SMALL_TILES tiles : ONE kernel per pass.  Every block re-counts the digits of the
//       whole (L2-resident) key array -- split into "tiles before mine" and "the rest" -- instead
//       of reading a histogram somebody else produced.
//   larger n : three kernels per pass -- per-tile histogram, one block per digit scanning its
//       row of tile counts, scatter (which prefix-sums the 256 digit totals itself).
// A block owns a tile of RS_TILE consecutive keys and processes it in RS_ROUNDS rounds of
// 256 keys (one per thread, in index order) so that stability only needs (a) the rank of a
// key among equal digits inside its wave (ballot match) and (b) a per-digit running count
// across waves and rounds, which thread d keeps in a register for digit d.
// ---------------------------------------------------------------------------------------
#define RS_THREADS 256
#define RS_ROUNDS 8
#define RS_TILE (RS_THREADS * RS_ROUNDS)
#define RS_SMALL_TILES 32

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
        const i64 idx = tile0 + r * RS_THREADS + tid;
        if (idx < n) atomicAdd(&h[(u32)(kin[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(i64)tid * nblk + blockIdx.x] = h[tid];
}

// block d: exclusive prefix of digit d's tile counts (in place) and the digit total
__global__ __launch_bounds__(256) void radix_rowscan_kernel(u32* __restrict__ hist, int nblk,
                                                            u32* __restrict__ totals) {
    __shared__ u32 wsum[4];
    const int tid = threadIdx.x;
    u32* row = hist + (i64)blockIdx.x * nblk;
    u32 carry = 0;
    for (int c0 = 0; c0 < nblk; c0 += 256) {
        const int i = c0 + tid;
        const u32 v = i < nblk ? row[i] : 0u;
        const u32 inc = wave_incl_scan(v);
        if ((tid & 63) == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        u32 woff = 0;
        for (int w = 0; w < (tid >> 6); ++w) woff += wsum[w];
        const u32 all = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (i < nblk) row[i] = carry + woff + inc - v;
        carry += all;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry;
}

// Stable scatter of one 2048-key tile.  Keys are taken in index order: round r, thread t holds
// key tile0 + r*256 + t.  Phase 0: thread d obtains the global base of digit d for this tile
// (SMALL: by counting the whole array; else: row offset + prefix of the digit totals).
// Phase 1: for every (round, wave) the lanes with equal digits find each other with 8 ballots;
// the lowest lane of a group records the group size in cnt[r][w][digit] and every lane keeps its
// rank inside the group.  Phase 2: thread d turns the 32 counts of digit d into exclusive offsets
// in (round, wave) order -- exactly the order the keys must keep.  Phase 3: position = global
// base[d] + offset[r][w][d] + rank.
template <int THREADS, bool SMALL>
__global__ __launch_bounds__(THREADS) void radix_scatter_kernel(
    const u64* __restrict__ kin, const u32* __restrict__ vin, u64* __restrict__ kout,
    u32* __restrict__ vout, i64 n, int shift, const u32* __restrict__ offs,
    const u32* __restrict__ totals, int nblk, u32* __restrict__ totals_out, u32* __restrict__ max_out) {
    constexpr int NWAVE = THREADS / 64, ROUNDS = RS_TILE / THREADS, NQ = ROUNDS * NWAVE;   // NQ == 32
    __shared__ u32 cnt[NQ * 256];     // 32 KiB
    __shared__ u32 dsum[4];
    __shared__ u32 dmaxv[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const i64 tile0 = (i64)blockIdx.x * RS_TILE;
    u64 key[ROUNDS];
    u32 val[ROUNDS], rank[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const i64 idx = tile0 + r * THREADS + tid;
        const bool act = idx < n;
        key[r] = act ? kin[idx] : 0ull;
        val[r] = act ? (vin ? vin[idx] : (u32)idx) : 0u;
    }
    u32 before = 0, total = 0;
    if (SMALL) {
        // cnt[0..255]: digits of the keys in tiles before this one, cnt[256..511]: all the others
        if (tid < 512) cnt[tid] = 0;
        __syncthreads();
        for (i64 i0 = 0; i0 < n; i0 += 4 * THREADS) {
            u64 k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const i64 idx = i0 + u * THREADS + tid;
                k[u] = idx < n ? kin[idx] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const i64 idx = i0 + u * THREADS + tid;
                if (idx < n) atomicAdd(&cnt[(idx < tile0 ? 0u : 256u) + ((u32)(k[u] >> shift) & 255u)], 1u);
            }
        }
        __syncthreads();
        if (tid < 256) {
            before = cnt[tid];
            total = before + cnt[256 + tid];
        }
        __syncthreads();
    } else if (tid < 256) {
        before = offs[(i64)tid * nblk + blockIdx.x];
        total = totals[tid];
    }
    const u32 dincl = wave_incl_scan(total);
    if (lane == 63 && w < 4) dsum[w] = dincl;
    if (blockIdx.x == 0 && (totals_out || max_out)) {      // digit statistics of this pass (one block reports)
        if (totals_out && tid < 256) totals_out[tid] = total;
        if (max_out && w < 4) {     // plain store by the last of the four digit waves to arrive (no zeroing launch)
            u32 m = total;
            for (int off = 32; off > 0; off >>= 1) { const u32 o = __shfl_down(m, off, 64); m = o > m ? o : m; }
            if (lane == 0) dmaxv[w] = m;
        }
    }
#pragma unroll
    for (int q = 0; q < NQ * 256 / THREADS; ++q) cnt[q * THREADS + tid] = 0;
    __syncthreads();
    if (max_out && blockIdx.x == 0 && tid == 0) {
        u32 m = dmaxv[0];
        for (int q = 1; q < 4; ++q) m = dmaxv[q] > m ? dmaxv[q] : m;
        *max_out = m;
    }
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const bool act = tile0 + r * THREADS + tid < n;
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
    if (tid < 256) {
        u32 run = dincl - total + before;    // keys with smaller digits + equal digits in earlier tiles
        for (int q = 0; q < w; ++q) run += dsum[q];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const u32 c = cnt[q * 256 + tid];
            cnt[q * 256 + tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (tile0 + r * THREADS + tid < n) {
            const u32 d = (u32)(key[r] >> shift) & 255u;
            const u32 pos = cnt[(r * NWAVE + w) * 256 + d] + rank[r];
            kout[pos] = key[r];
            vout[pos] = val[r];
        }
    }
}

// ---------------------------------------------------------------------------------------
// Bucket finish: after ONE stable pass on the TOP digit the array is 256 contiguous buckets; a
// workgroup then owns a bucket and runs every remaining (low to high) digit pass on it by itself,
// ping-ponging between the two global buffers with only workgroup barriers in between.  Six
// passes over 1 M keys become 4 launches instead of 18.  Correct for any bucket size; fast when
// the buckets are balanced, which the caller judges from the class sizes of the previous level.
// ---------------------------------------------------------------------------------------
#define BK_THREADS 1024
#define BK_ROUNDS 12
#define BK_SUB 4                                // rounds ranked per sub-step (bounds the counter array)
#define BK_CAP (BK_THREADS * BK_ROUNDS)        // bucket size the in-LDS path takes (12288 keys)
#define BK_IDX_BITS 14
#define BK_SMALL 512                             // buckets up to this size are ranked by comparison
#define BK_LDS_BYTES (BK_CAP * 8 + BK_SUB * (BK_THREADS / 64) * 256 * 2 + 4 * 256 * 4)

// One workgroup per top-digit bucket of (ks, vs); the sorted bucket lands in (kd, vd).
//  * bucket <= 512 keys: every element is ranked against all others (no passes at all).
//  * bucket <= 12288 keys and <= 6 remaining digits: everything happens in LDS.  An element is
//    (remaining key bits << 14 | position in the bucket), up to 12 per thread in registers in index
//    order.  A pass (a) finds each element's rank inside its (round, wave) group with 8 ballots
//    and adds the group sizes to the digit totals, (b) prefix-sums the totals, (c) four rounds at
//    a time turns the group sizes (u16 counters) into offsets and scatters the elements into the
//    LDS array, (d) reads them back in index order.  Work is proportional to the rounds in use.
//    Keys and values are gathered from the (untouched) source by position at the very end.
//  * anything else: the passes ping-pong through the two global buffers (workgroup barriers
//    only); correct for any size, slow for a big bucket -- the caller avoids that case.
__global__ __launch_bounds__(BK_THREADS) void radix_bucket_kernel(
    u64* __restrict__ ks, u32* __restrict__ vs, u64* __restrict__ kd, u32* __restrict__ vd,
    const u32* __restrict__ totals, int n_passes) {
    constexpr int NWAVE = BK_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char bk_lds[];
    __shared__ u32 base[2][256];
    __shared__ u32 dsum[4];
    __shared__ u32 bstart;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // bucket range: exclusive prefix of the digit totals
    {
        const u32 t = tid < 256 ? totals[tid] : 0u;
        const u32 inc = wave_incl_scan(t);
        if (lane == 63 && w < 4) dsum[w] = inc;
        __syncthreads();
        if (tid == (int)blockIdx.x) {
            u32 off = inc - t;
            for (int q = 0; q < w; ++q) off += dsum[q];
            bstart = off;
        }
        __syncthreads();
    }
    const u32 size = totals[blockIdx.x];
    if (size == 0) return;
    const i64 start = bstart;
    const u64 lt = (1ull << lane) - 1ull;

    if (size <= BK_SMALL && n_passes <= 6) {
        // small bucket (the active levels of a WL job): rank every element against all others --
        // (key bits, position) is a strict total order, so the rank is the output position.
        // All 1024 threads take part: an element is served by 1024/S2 threads (S2 = 256 or 512
        // element slots), each counting the smaller elements in its share of the bucket with
        // broadcast 16-byte LDS reads; the partial ranks meet in an LDS counter.
        u64* elems = (u64*)bk_lds;                              // [BK_SMALL], padded with the largest value
        u32* rk = (u32*)(bk_lds + BK_SMALL * 8);                // [BK_SMALL]
        const u64 keymask = (1ull << (n_passes * 8)) - 1ull;
        if (tid < BK_SMALL) {
            elems[tid] = (u32)tid < size ? (((ks[start + tid] & keymask) << BK_IDX_BITS) | (u64)tid) : ~0ull;
            rk[tid] = 0;
        }
        __syncthreads();
        const u32 S2 = size <= 256 ? 256u : 512u;
        const u32 e = (u32)tid & (S2 - 1u), part = (u32)tid / S2, parts = BK_THREADS / S2;
        const u32 per = (((size + parts - 1u) / parts) + 1u) & ~1u;      // even: 16-byte reads
        if (e < size) {
            const u64 mine = elems[e];
            u32 j = part * per;
            const u32 j1 = j + per < (u32)BK_SMALL ? j + per : (u32)BK_SMALL;
            u32 r = 0;
#pragma unroll 8
            for (; j < j1; j += 2) {
                const ulonglong2 ab = *(const ulonglong2*)(elems + j);
                r += (ab.x < mine ? 1u : 0u) + (ab.y < mine ? 1u : 0u);
            }
            if (r) atomicAdd(&rk[e], r);
        }
        __syncthreads();
        if ((u32)tid < size) {
            const u32 rnk = rk[tid];
            kd[start + rnk] = ks[start + tid];
            vd[start + rnk] = vs[start + tid];
        }
        return;
    }
    if (size <= BK_CAP && n_passes <= 6) {
        u64* elems = (u64*)bk_lds;                                         // [BK_CAP]
        unsigned short* cnt16 = (unsigned short*)(bk_lds + BK_CAP * 8);    // [BK_SUB][NWAVE][256]
        u32* qsum = (u32*)(bk_lds + BK_CAP * 8 + BK_SUB * NWAVE * 256 * 2);   // [4][256]
        const int nr = (int)((size + BK_THREADS - 1) / BK_THREADS);        // rounds in use (block-uniform)
        const u64 keymask = (1ull << (n_passes * 8)) - 1ull;
        u64 e[BK_ROUNDS];
#pragma unroll
        for (int r = 0; r < BK_ROUNDS; ++r) {
            const u32 i = r * BK_THREADS + tid;
            e[r] = (r < nr && i < size) ? (((ks[start + i] & keymask) << BK_IDX_BITS) | (u64)i) : ~0ull;
        }
        int cur = 0;
        for (int p = 0; p < n_passes; ++p) {
            const int shift = BK_IDX_BITS + 8 * p;
            if (tid < 256) base[cur][tid] = 0;
            __syncthreads();
            // (a) rank inside the (round, wave) group; the group leader keeps the group size
            u32 rank[BK_ROUNDS], gsz[BK_ROUNDS];
#pragma unroll
            for (int r = 0; r < BK_ROUNDS; ++r) {
                rank[r] = 0, gsz[r] = 0;
                if (r < nr) {
                    const bool act = (u32)(r * BK_THREADS + tid) < size;
                    const u32 d = (u32)(e[r] >> shift) & 255u;
                    u64 m = __ballot(act);
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const bool bit = (d >> b) & 1u;
                        const u64 bb = __ballot(act && bit);
                        m &= bit ? bb : ~bb;
                    }
                    rank[r] = (u32)__popcll(m & lt);
                    if (act && rank[r] == 0) {
                        gsz[r] = (u32)__popcll(m);
                        atomicAdd(&base[cur][d], gsz[r]);
                    }
                }
            }
            __syncthreads();
            // (b) digit offsets
            {
                const u32 t = tid < 256 ? base[cur][tid] : 0u;
                const u32 inc = wave_incl_scan(t);
                if (lane == 63 && w < 4) dsum[w] = inc;
                __syncthreads();
                if (tid < 256) {
                    u32 off = inc - t;
                    for (int q = 0; q < w; ++q) off += dsum[q];
                    base[cur][tid] = off;
                }
            }
            // (c) BK_SUB rounds at a time: group sizes -> offsets -> scatter
            for (int r0 = 0; r0 < nr; r0 += BK_SUB) {
                const int rs = nr - r0 < BK_SUB ? nr - r0 : BK_SUB;          // rounds of this sub-step
                {
                    uint4* z = (uint4*)cnt16;
                    for (int q = tid; q < rs * NWAVE * 256 * 2 / 16; q += BK_THREADS) z[q] = make_uint4(0, 0, 0, 0);
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < BK_ROUNDS; ++r)
                    if (r >= r0 && r < r0 + rs && gsz[r])
                        cnt16[((r - r0) * NWAVE + w) * 256 + ((u32)(e[r] >> shift) & 255u)] = (unsigned short)gsz[r];
                __syncthreads();
                {   // four threads per digit, each a contiguous quarter of the rs*16 groups
                    const int d = tid & 255, qt = tid >> 8;
                    const int per = rs * NWAVE / 4, g0 = qt * per;
                    u32 ssum = 0;
                    for (int g = 0; g < per; ++g) ssum += cnt16[(g0 + g) * 256 + d];
                    qsum[qt * 256 + d] = ssum;
                    __syncthreads();
                    u32 run = base[cur][d];
                    for (int q = 0; q < qt; ++q) run += qsum[q * 256 + d];
                    for (int g = 0; g < per; ++g) {
                        const u32 c = cnt16[(g0 + g) * 256 + d];
                        cnt16[(g0 + g) * 256 + d] = (unsigned short)run;
                        run += c;
                    }
                    if (qt == 3) base[cur ^ 1][d] = run;       // digit offsets for the next sub-step
                }
                __syncthreads();
                cur ^= 1;
#pragma unroll
                for (int r = 0; r < BK_ROUNDS; ++r)
                    if (r >= r0 && r < r0 + rs && (u32)(r * BK_THREADS + tid) < size) {
                        const u32 d = (u32)(e[r] >> shift) & 255u;
                        elems[(u32)cnt16[((r - r0) * NWAVE + w) * 256 + d] + rank[r]] = e[r];
                    }
                __syncthreads();
            }
            // (d) back into registers in index order
#pragma unroll
            for (int r = 0; r < BK_ROUNDS; ++r) {
                const u32 i = r * BK_THREADS + tid;
                if (r < nr && i < size) e[r] = elems[i];
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < BK_ROUNDS; ++r) {
            const u32 i = r * BK_THREADS + tid;
            if (r < nr && i < size) {
                const u32 src = (u32)(e[r] & (u64)((1u << BK_IDX_BITS) - 1));
                kd[start + i] = ks[start + src];
                vd[start + i] = vs[start + src];
            }
        }
        return;
    }

    // ---- general path: ping-pong through global memory
    constexpr int ROUNDS = RS_TILE / BK_THREADS, NQ = ROUNDS * NWAVE;   // 32 groups per tile
    u32* cnt = (u32*)bk_lds;                                            // [NQ * 256]
    u64* ka = ks; u32* va = vs; u64* kb = kd; u32* vb = vd;
    for (int p = 0; p < n_passes; ++p) {
        const int shift = 8 * p;
        if (tid < 256) base[0][tid] = 0;
        __syncthreads();
        for (u32 i0 = 0; i0 < size; i0 += BK_THREADS) {
            const u32 i = i0 + tid;
            const bool act = i < size;
            wave_digit_add(base[0], act, act ? (u32)(ka[start + i] >> shift) & 255u : 0u);
        }
        __syncthreads();
        {
            const u32 t = tid < 256 ? base[0][tid] : 0u;
            const u32 inc = wave_incl_scan(t);
            if (lane == 63 && w < 4) dsum[w] = inc;
            __syncthreads();
            if (tid < 256) {
                u32 off = inc - t;
                for (int q = 0; q < w; ++q) off += dsum[q];
                base[0][tid] = off;
            }
        }
        for (u32 tile0 = 0; tile0 < size; tile0 += RS_TILE) {
            u64 key[ROUNDS];
            u32 val[ROUNDS], rank[ROUNDS];
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const u32 i = tile0 + r * BK_THREADS + tid;
                const bool act = i < size;
                key[r] = act ? ka[start + i] : 0ull;
                val[r] = act ? va[start + i] : 0u;
            }
#pragma unroll
            for (int q = 0; q < NQ * 256 / BK_THREADS; ++q) cnt[q * BK_THREADS + tid] = 0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                const bool act = tile0 + r * BK_THREADS + tid < size;
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
            if (tid < 256) {
                u32 run = base[0][tid];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const u32 c = cnt[q * 256 + tid];
                    cnt[q * 256 + tid] = run;
                    run += c;
                }
                base[0][tid] = run;      // carried into the bucket's next tile
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                if (tile0 + r * BK_THREADS + tid < size) {
                    const u32 d = (u32)(key[r] >> shift) & 255u;
                    const u32 pos = cnt[(r * NWAVE + w) * 256 + d] + rank[r];
                    kb[start + pos] = key[r];
                    vb[start + pos] = val[r];
                }
            }
            __syncthreads();
        }
        // the bucket's next pass reads what this workgroup just wrote (same CU, workgroup scope)
        __threadfence_block();
        __syncthreads();
        u64* tk = ka; ka = kb; kb = tk;
        u32* tv = va; va = vb; vb = tv;
    }
    if (ka != kd) {       // an even number of passes ended in the source buffer: move the bucket over
        for (u32 i = tid; i < size; i += BK_THREADS) {
            kd[start + i] = ka[start + i];
            vd[start + i] = va[start + i];
        }
    }
}

int gk_radix_sort_pairs(gk_ctx* ctx, const u64* keys_in, const u32* vals_in, u64* keys_out, u32* vals_out,
                        i64 n, int key_bits, int use_buckets, u32* top_digit_max) {
    // use_buckets: 0 pass by pass, 1 bucket finish, 2 probe -- histogram the top digit, read the
    // 256 bucket sizes back (one host sync) and take the bucket finish when the largest bucket is
    // at most `probe_limit` keys; only worth it on large arrays (the sync costs about one pass).
    // Stable sort by the low key_bits of the keys.  The inputs are only read (vals_in == nullptr:
    // the values are the indices 0..n-1); the passes ping-pong between the out buffers and a
    // temporary pair, ordered so that the last pass lands in keys_out / vals_out.
    // top_digit_max (device, may be null) receives the size of the largest top-digit bucket.
    if (n <= 0) {
        if (top_digit_max) GK_TRY(gk_zero_async(ctx, top_digit_max, 4));
        return GK_OK;
    }
    if (key_bits < 0) key_bits = 0;
    if (key_bits > 64) key_bits = 64;
    int passes = (key_bits + 7) / 8;
    if (passes == 0) passes = 1;   // nothing to distinguish: one (trivial) pass keeps the code paths uniform
    int nblk = (int)cdiv(n, RS_TILE);
    const bool small = nblk <= RS_SMALL_TILES;
    Tmp<u32> hist(ctx), vtmp(ctx);
    Tmp<u64> ktmp(ctx);
    GK_TRY(hist.alloc((size_t)256 * nblk + 512));
    if (passes > 1) { GK_TRY(ktmp.alloc(n)); GK_TRY(vtmp.alloc(n)); }
    u32* totals = hist.p + (size_t)256 * nblk;
    u32* bucket_totals = totals + 256;
    bool probed = false;
    if (use_buckets == 2) {
        use_buckets = 0;
        if (n <= BK_CAP) {
            use_buckets = 1;                   // no bucket can exceed what the LDS path holds
        } else if (passes >= 4) {
            const int shift = 8 * (passes - 1);
            radix_hist_kernel<<<dim3(nblk), dim3(RS_THREADS), 0, ctx->stream>>>(keys_in, n, shift, hist.p, nblk);
            radix_rowscan_kernel<<<dim3(256), dim3(256), 0, ctx->stream>>>(hist.p, nblk, totals);
            u32 h_tot[256];
            GK_TRY(gk_readback(ctx, totals, h_tot, 256));
            u32 mx = 0;
            for (int d = 0; d < 256; ++d) mx = h_tot[d] > mx ? h_tot[d] : mx;
            if (mx <= BK_CAP) use_buckets = 1, probed = !small;     // large arrays reuse the histogram below
        }
    }
    if (use_buckets && passes >= 3) {
        // top digit first (stable), then every bucket finishes on its own
        const int inner = passes - 1, shift = 8 * inner;
        u64* kx = ktmp.p;  u32* vx = vtmp.p;         // top-digit pass: in -> tmp; buckets: tmp -> out
        u64* ky = keys_out;  u32* vy = vals_out;
        if (small) {
            radix_scatter_kernel<1024, true><<<dim3(nblk), dim3(1024), 0, ctx->stream>>>(
                keys_in, vals_in, kx, vx, n, shift, nullptr, nullptr, nblk, bucket_totals, top_digit_max);
        } else {
            if (!probed) {
                radix_hist_kernel<<<dim3(nblk), dim3(RS_THREADS), 0, ctx->stream>>>(keys_in, n, shift, hist.p, nblk);
                radix_rowscan_kernel<<<dim3(256), dim3(256), 0, ctx->stream>>>(hist.p, nblk, totals);
            }
            radix_scatter_kernel<1024, false><<<dim3(nblk), dim3(1024), 0, ctx->stream>>>(
                keys_in, vals_in, kx, vx, n, shift, hist.p, totals, nblk, bucket_totals, top_digit_max);
        }
        GK_TRY(gk_func_lds(ctx, (const void*)radix_bucket_kernel, BK_LDS_BYTES));
        radix_bucket_kernel<<<dim3(256), dim3(BK_THREADS), BK_LDS_BYTES, ctx->stream>>>(kx, vx, ky, vy, bucket_totals, inner);
        GK_HIP_CHECK(hipGetLastError());
        return GK_OK;
    }
    const u64* ks = keys_in;
    const u32* vs = vals_in;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;
        u64* kd = to_out ? keys_out : ktmp.p;
        u32* vd = to_out ? vals_out : vtmp.p;
        int shift = p * 8;
        u32* mx = p == passes - 1 ? top_digit_max : nullptr;
        if (small) {
            radix_scatter_kernel<1024, true><<<dim3(nblk), dim3(1024), 0, ctx->stream>>>(
                ks, vs, kd, vd, n, shift, nullptr, nullptr, nblk, nullptr, mx);
        } else {
            radix_hist_kernel<<<dim3(nblk), dim3(RS_THREADS), 0, ctx->stream>>>(ks, n, shift, hist.p, nblk);
            radix_rowscan_kernel<<<dim3(256), dim3(256), 0, ctx->stream>>>(hist.p, nblk, totals);
            radix_scatter_kernel<1024, false><<<dim3(nblk), dim3(1024), 0, ctx->stream>>>(
                ks, vs, kd, vd, n, shift, hist.p, totals, nblk, nullptr, mx);
        }
        ks = kd, vs = vd;
    }
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

// ---------------------------------------------------------------------------------------
// Dictionary WITHOUT a sort (full WL levels of graph batches, wl.hip): equal keys only have to MEET, no
// consumer needs them in order once the label-count features are built graph-major (features_gm.hip).
// One stable pass on the top digit (as above) leaves 256 buckets; a workgroup then owns a bucket and
// runs an open-addressing table in LDS over the remaining key bits.  The table holds DISTINCT keys -- a slot is
// (key + 1 as a 64-bit word, 0 = empty | one member / several) -- and the bucket's items stream through it from HBM in
// 1024-item chunks (twice: insert, then look up), so a bucket may hold any number of items as long as its
// distinct keys fit: a class of 14 000 isolated vertices is one slot.  (The first version kept every ITEM's key in
// LDS and overflowed at 8 192 items per bucket: config 5, 50 000 graphs with 70 000 isolated vertices, redid every
// level on the sorting path -- profiles/r03a_config5_*.)  An item claims an empty slot with one 64-bit
// compare-and-swap that publishes its key at the same time, or finds its key and bumps the counter.  Slots in
// use, ranked by a workgroup prefix sum, are the bucket's classes; bucket offsets (a 256-entry prefix) make the
// ids dense.  Replaces the remaining 3-5 digit passes of the bucket finish AND the run-head scan.
//   per item (bucket_dict_kernel -> bucket_assign_kernel): bits 0-13 rank of its class in the bucket,
//   bit 30 the item claimed the class (its node becomes the representative), bit 31 singleton
// More distinct keys than max_distinct (or more than 64 chunks of items): the kernel raises BD_OVERFLOW in
// *overflow (and keeps the outputs memory-safe); the caller redoes the level with the sorting path.
// gk_bucket_dictionary_fits() is the caller's a-priori test (all-distinct worst case).
// ---------------------------------------------------------------------------------------
#define BD_SLOTS 12288          // 12 B per slot: 144 KiB of LDS
#define BD_MAX_DISTINCT 9216    // load factor 0.75
#define BD_MAX_CHUNKS 64        // claim flags of a thread: one bit per chunk

__global__ __launch_bounds__(1024) void bucket_dict_kernel(const u64* __restrict__ kx, const u32* __restrict__ totals,
                                                           int shift, u32* __restrict__ item_out, u32* __restrict__ nd,
                                                           u32* __restrict__ overflow, unsigned long long* __restrict__ ticket,
                                                           u32 max_distinct) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bd_lds[];
    unsigned long long* key_s = (unsigned long long*)bd_lds;                   // [BD_SLOTS] key + 1, 0 = empty
    u32* word_s = (u32*)(bd_lds + (size_t)BD_SLOTS * 8);                       // [BD_SLOTS] 1 / 2 = one / several members, later rank | singleton << 31
    __shared__ u32 dsum[4];
    __shared__ u32 wsum[16];
    __shared__ u32 bstart, n_claimed, ovf;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (blockIdx.x == 0 && tid == 0) *ticket = 0;           // bucket_assign_kernel's arrival counter
    if (tid == 0) n_claimed = 0, ovf = 0;
    {   // bucket range: exclusive prefix of the digit totals
        const u32 t = tid < 256 ? totals[tid] : 0u;
        const u32 inc = wave_incl_scan(t);
        if (lane == 63 && w < 4) dsum[w] = inc;
        __syncthreads();
        if (tid == (int)blockIdx.x) {
            u32 off = inc - t;
            for (int q = 0; q < w; ++q) off += dsum[q];
            bstart = off;
        }
        __syncthreads();
    }
    const u32 size = totals[blockIdx.x];
    const i64 start = bstart;
    if (size == 0) { if (tid == 0) nd[blockIdx.x] = 0; return; }
    const u64 kmask = shift >= 64 ? ~0ull : ((1ull << shift) - 1ull);       // shift <= 56: key + 1 never wraps to 0
    for (int t = tid; t < BD_SLOTS; t += 1024) key_s[t] = 0ull, word_s[t] = 0u;
    __syncthreads();
    const bool too_long = size > (u32)BD_MAX_CHUNKS * 1024u;
    // ---- insert: the thread keeps one "I claimed the slot" bit per chunk
    u64 claimed = 0;
    if (!too_long) {
        int c = 0;
        // the next chunk's key is in flight while this one probes the table: the largest bucket (a heavy class sends
        // 14 000 equal keys into ONE bucket: 3-4x the average size) is the kernel's critical path, one HBM round trip
        // per chunk otherwise
        u64 k_cur = tid < size ? kx[start + tid] : 0ull;
        for (u32 i = tid; i < size; i += 1024, ++c) {
            const u64 k_nxt = i + 1024 < size ? kx[start + i + 1024] : 0ull;
            const u64 k1 = (k_cur & kmask) + 1ull;
            k_cur = k_nxt;
            u32 h = (u32)((((k1 * 0x9E3779B97F4A7C15ull) >> 32) * (u64)BD_SLOTS) >> 32);
            for (;;) {
                unsigned long long v = key_s[h];
                if (v == 0ull) {
                    if (*(volatile u32*)&ovf) break;                                       // table declared full: stop claiming
                    v = atomicCAS(&key_s[h], 0ull, (unsigned long long)k1);
                    if (v == 0ull) {                                      // claimed: this item owns the class
                        claimed |= 1ull << c;
                        atomicMax(&word_s[h], 1u);
                        if (atomicAdd(&n_claimed, 1u) + 1u > max_distinct) ovf = 1u;
                        break;
                    }
                }
                // a second member makes the class shared; members are not counted: the thousands of items of a hot
                // class (isolated vertices, degree-one nodes) then only READ the slot (a broadcast) instead of
                // queueing on one LDS atomic
                if (v == k1) { if (*(volatile u32*)&word_s[h] != 2u) atomicMax(&word_s[h], 2u); break; }
                h = h + 1u == (u32)BD_SLOTS ? 0u : h + 1u;
            }
        }
    }
    __syncthreads();
    if (too_long || ovf) {              // not handled here: one class for the whole bucket (memory-safe), level redone by the caller
        for (u32 i = tid; i < size; i += 1024) item_out[start + i] = i == 0 ? (1u << 30) : 0u;
        if (tid == 0) { nd[blockIdx.x] = 1; atomicOr(overflow, 0x80000000u); }
        return;
    }
    // ---- rank the slots in use: thread t owns slots [12t, 12t + 12)
    constexpr int PER = BD_SLOTS / 1024;
    u32 mine = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) mine += word_s[PER * tid + q] ? 1u : 0u;
    const u32 inc = wave_incl_scan(mine);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    u32 before = inc - mine, all = 0;
    for (int q = 0; q < 16; ++q) {
        if (q < w) before += wsum[q];
        all += wsum[q];
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const u32 members = word_s[PER * tid + q];                 // 1: one member, 2: two or more
        if (members) {
            word_s[PER * tid + q] = before | (members == 1u ? 0x80000000u : 0u);
            ++before;
        }
    }
    if (tid == 0) nd[blockIdx.x] = all;
    __syncthreads();
    // ---- every item looks its class up again
    int c = 0;
    u64 k_cur = tid < size ? kx[start + tid] : 0ull;
    for (u32 i = tid; i < size; i += 1024, ++c) {
        const u64 k_nxt = i + 1024 < size ? kx[start + i + 1024] : 0ull;
        const u64 k1 = (k_cur & kmask) + 1ull;
        k_cur = k_nxt;
        u32 h = (u32)((((k1 * 0x9E3779B97F4A7C15ull) >> 32) * (u64)BD_SLOTS) >> 32);
        while (key_s[h] != k1) h = h + 1u == (u32)BD_SLOTS ? 0u : h + 1u;
        const u32 v = word_s[h];
        item_out[start + i] = (v & 0x3fffu) | (((claimed >> c) & 1ull) ? 1u << 30 : 0u) | (v & 0x80000000u);
    }
}

// all-distinct worst case of a bucket: mean + 6 standard deviations of the binomial split over 256 buckets
bool gk_bucket_dictionary_fits(gk_ctx* ctx, i64 n) {
    const double m = (double)n / 256.0;
    const double cap = ctx->opt.bd_slots > 0 ? 1e30 : (double)BD_MAX_DISTINCT;      // test hook: overflow is the point
    double sd = 1.0;
    while (sd * sd < m) sd += 1.0;
    return m + 6.0 * sd <= cap;
}

// second half: dense ids = bucket offset + rank; labels, representatives, singleton / shared flags, the level's
// class count and number of nodes in shared classes; the last workgroup to finish posts {listed, *extra} to the
// host mailbox (as HeadAssignSplit's finish hook does on the sorting path)
__global__ __launch_bounds__(1024) void bucket_assign_kernel(const u32* __restrict__ vx, const u32* __restrict__ item_in,
                                                             const u32* __restrict__ totals, const u32* __restrict__ nd,
                                                             i32* __restrict__ lab, i32* __restrict__ rep, u32* __restrict__ frozen,
                                                             unsigned char* __restrict__ shared_out, u32* __restrict__ listed_out,
                                                             u32* __restrict__ count_out, unsigned long long* __restrict__ ticket,
                                                             u32* __restrict__ mbox, u32 seq, const u32* __restrict__ extra, i64 n,
                                                             int flag_in_lab) {
    // one item per thread (the scattered stores need many workgroups in flight); the bucket of an item is found
    // in the two 256-entry prefixes every workgroup rebuilds in LDS
    __shared__ u32 starts[257], bases[257];
    __shared__ u32 dsum[4], esum[4];
    __shared__ u32 red[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    {
        const u32 t = tid < 256 ? totals[tid] : 0u, e = tid < 256 ? nd[tid] : 0u;
        const u32 inct = wave_incl_scan(t), ince = wave_incl_scan(e);
        if (lane == 63 && w < 4) dsum[w] = inct, esum[w] = ince;
        __syncthreads();
        if (tid < 256) {
            u32 off = inct, base = ince;
            for (int q = 0; q < w; ++q) off += dsum[q], base += esum[q];
            starts[tid + 1] = off, bases[tid + 1] = base;            // inclusive -> entry tid + 1
        }
        if (tid == 0) starts[0] = 0, bases[0] = 0;
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid == 0) *count_out = bases[256];
    const i64 i = (i64)blockIdx.x * 1024 + tid;
    u32 listed = 0;
    if (i < n) {
        int lo = 0, hi = 256;                                       // largest b with starts[b] <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if ((i64)starts[mid] <= i) lo = mid; else hi = mid;
        }
        const u32 node = vx[i], t = item_in[i];
        const u32 id = bases[lo] + (t & 0x3fffu);
        const u32 single = t >> 31;
        // flag_in_lab: the singleton flag rides in bit 31 of the label word until the caller's verification pass
        // (one thread per node, in node order) strips it and writes the flag byte coalesced; a singleton is its own
        // representative, so rep[] is only written for the shared classes -- at a deep level that is a tenth of the
        // classes, i.e. a million scattered 4-byte stores fewer
        lab[node] = (i32)(id | (flag_in_lab ? single << 31 : 0u));
        if (frozen) frozen[node] = single;
        if (shared_out) shared_out[node] = single ? 0 : 1;
        if (((t >> 30) & 1u) && !(flag_in_lab && single)) rep[id] = (i32)node;
        listed = single ? 0u : 1u;
    }
    for (int off = 32; off > 0; off >>= 1) listed += __shfl_down(listed, off, 64);
    if (lane == 0) red[w] = listed;
    __syncthreads();
    if (tid == 0) {
        u32 s = 0;
        for (int q = 0; q < 16; ++q) s += red[q];
        // count and arrival in ONE 64-bit atomic: no fence between two atomics (a device-scope fence writes the
        // XCD's dirty L2 lines back -- all the scattered stores above -- once per workgroup: measured 3x the kernel)
        const unsigned long long old = atomicAdd(ticket, ((unsigned long long)s << 32) | 1ull);
        if ((u32)old == gridDim.x - 1) {                    // last arrival: everybody's count is in
            const u32 total = (u32)(old >> 32) + s;
            *listed_out = total;
            if (!mbox) return;
            __hip_atomic_store(&mbox[1], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mbox[2], extra ? *extra : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mbox[3], bases[256], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // labels of the level (wl.hip: convergence)
            __threadfence_system();
            __hip_atomic_store(&mbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// keys[i] belongs to item i (0..n-1).  lab / frozen / shared_out are indexed by item, rep by class id.
// *overflow gets 0x80000000 or-ed in when a bucket did not fit.
int gk_bucket_dictionary(gk_ctx* ctx, const u64* keys, i64 n, int key_bits, i32* lab, i32* rep, u32* frozen,
                         unsigned char* shared_out, u32* count_dev, u32* listed_dev, u32* top_digit_max, u32* overflow,
                         u32* mbox, u32 seq, int flag_in_lab) {
    if (key_bits > 64) key_bits = 64;
    const int passes = (key_bits + 7) / 8;
    const int shift = 8 * (passes - 1);
    const int nblk = (int)cdiv(n, RS_TILE);
    Tmp<u32> hist(ctx), vx(ctx), item(ctx), nd(ctx);
    Tmp<u64> kx(ctx);
    GK_TRY(hist.alloc((size_t)256 * nblk + 768)); GK_TRY(vx.alloc(n)); GK_TRY(item.alloc(n)); GK_TRY(nd.alloc(256));
    GK_TRY(kx.alloc(n));
    u32* totals = hist.p + (size_t)256 * nblk;
    u32* bucket_totals = totals + 256;
    unsigned long long* ticket = (unsigned long long*)(totals + 512);      // 8-byte aligned: 256 * nblk + 512 words in
    if (nblk <= RS_SMALL_TILES) {
        radix_scatter_kernel<1024, true><<<dim3(nblk), dim3(1024), 0, ctx->stream>>>(
            keys, nullptr, kx.p, vx.p, n, shift, nullptr, nullptr, nblk, bucket_totals, top_digit_max);
    } else {
        radix_hist_kernel<<<dim3(nblk), dim3(RS_THREADS), 0, ctx->stream>>>(keys, n, shift, hist.p, nblk);
        radix_rowscan_kernel<<<dim3(256), dim3(256), 0, ctx->stream>>>(hist.p, nblk, totals);
        radix_scatter_kernel<1024, false><<<dim3(nblk), dim3(1024), 0, ctx->stream>>>(
            keys, nullptr, kx.p, vx.p, n, shift, hist.p, totals, nblk, bucket_totals, top_digit_max);
    }
    {
        const int lds = BD_SLOTS * 12;
        GK_TRY(gk_func_lds(ctx, (const void*)bucket_dict_kernel, lds));
        u32 max_distinct = BD_MAX_DISTINCT;
        if (ctx->opt.bd_slots > 0 && (u32)ctx->opt.bd_slots < max_distinct) max_distinct = (u32)ctx->opt.bd_slots;     // test hook
        bucket_dict_kernel<<<dim3(256), dim3(1024), lds, ctx->stream>>>(kx.p, bucket_totals, shift, item.p, nd.p, overflow, ticket,
                                                                      max_distinct);
    }
    bucket_assign_kernel<<<dim3((unsigned)cdiv(n, 1024)), dim3(1024), 0, ctx->stream>>>(
        vx.p, item.p, bucket_totals, nd.p, lab, rep, frozen, shared_out, listed_dev, count_dev, ticket, mbox, seq, top_digit_max, n,
        flag_in_lab);
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

// Signature pieces shared by the relabel routes (wl.hip: host-driven, wl_stream.hip: without host round trips):
// the 64-bit multiset hash of (own label, degree, sorted neighbour labels) and the in-register sorting network.
#pragma once
#include "common.h"

#define WL_DEG_SMALL 32       // nodes up to this degree: one thread sorts its list in LDS
#define SIG_THREADS 256
#define SIG_LDS_CAP 6144      // ints staged per 256-node chunk (24 KiB)
#define BIG_THREADS 1024     // (round 5: 256 -> 1024: a hub's bitonic network is ~80 barrier-separated stages whatever the width)
#define BIG_LDS_CAP 16384     // ints: one workgroup bitonic-sorts a big node's list in LDS

__device__ __forceinline__ u64 mix64(u64 z) {
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ u64 sig_elem(u32 lab, u64 seed) {
    return mix64(((u64)lab + 1ull) * 0x9E3779B97F4A7C15ULL + seed);
}
__device__ __forceinline__ u64 sig_head(u32 own, u32 deg, u64 seed) {
    return mix64(mix64((u64)own + 0x632BE59BD9B4E019ULL * (seed | 1ull)) ^
                 ((u64)deg * 0xD6E8FEB86659FD93ULL));
}

template <typename P>
__device__ __forceinline__ void insertion_sort(P x, int d) {
    for (int i = 1; i < d; ++i) {
        i32 key = x[i];
        int j = i - 1;
        while (j >= 0 && x[j] > key) {
            x[j + 1] = x[j];
            --j;
        }
        x[j + 1] = key;
    }
}

// bitonic compare-exchange network on the first N (power of two) registers of x: every index is a compile-time
// constant after unrolling, so the elements stay in VGPRs (80 comparators for N = 16, 24 for N = 8)
template <int N>
__device__ __forceinline__ void sort_regs(i32 (&x)[16]) {
#pragma unroll
    for (int k = 2; k <= N; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const i32 a = x[i], b = x[l];
                    const i32 lo = a < b ? a : b, hi = a < b ? b : a;
                    if ((i & k) == 0) x[i] = lo, x[l] = hi;
                    else x[i] = hi, x[l] = lo;
                }
            }
}

// Bitonic network over ONE WAVE, 64 R elements striped over the lanes (element i = 64 r + lane): partners less than 64
// apart by a lane shuffle, farther apart in the lane's own registers (static indices).  Pad with 0x7fffffff.
#define WAVE_DEG_MAX 1024
template <int R>
__device__ __forceinline__ void wave_bitonic_sort(i32 (&x)[R], int lane) {
#pragma unroll
    for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int q = r ^ (j >> 6);
                    if (q > r) {
                        const bool up = ((r * 64) & k) == 0;          // k >= 128 here: the bit only depends on r
                        const i32 a = x[r], b = x[q];
                        const i32 lo = a < b ? a : b, hi = a < b ? b : a;
                        x[r] = up ? lo : hi, x[q] = up ? hi : lo;
                    }
                }
            } else {
                const bool lower = (lane & j) == 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const i32 a = x[r];
                    const i32 b = __shfl_xor(a, j, 64);
                    const bool up = ((r * 64 + lane) & k) == 0;
                    x[r] = (lower == up) ? (a < b ? a : b) : (a < b ? b : a);
                }
            }
        }
    }
}


// One node's signature key with its neighbour labels gathered straight into registers (degree <= 16): the 16
// gathers are independent loads, the sort is the fixed network, the sorted list goes to nbr_sorted for the verifier.
// (An insertion sort in global memory pays two memory latencies per step.)
__device__ __forceinline__ u64 node_key_regs(const i32* __restrict__ col_idx, const i32* __restrict__ lab_prev,
                                             i32* __restrict__ x, i32 s, int d, u32 own, u64 seed) {
    i32 r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = k < d ? lab_prev[col_idx[s + k]] : 0x7fffffff;
    sort_regs<16>(r);
    u64 acc = sig_head(own, (u32)d, seed);
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (k < d) { x[k] = r[k]; acc += sig_elem((u32)r[k], seed); }
    return acc;
}


static inline int bits_for(u64 max_value) {
    int b = 0;
    while (b < 64 && (max_value >> b)) ++b;
    return b;
}

static inline u64 level_seed(int level, int round) {
    u64 z = 0x243F6A8885A308D3ULL + (u64)level * 0x9E3779B97F4A7C15ULL + (u64)round * 0xC2B2AE3D27D4EB4FULL;
    z ^= z >> 31; z *= 0xff51afd7ed558ccdULL; z ^= z >> 29;
    return z;
}

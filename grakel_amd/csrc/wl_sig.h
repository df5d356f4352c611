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

// lane i <- lane (i ^ J) of a wave.  __shfl_xor is a ds_bpermute: every call goes through the CU's LDS crossbar, and a
// wave-per-vertex signature makes 21 (sort) + 12 (64-bit reduction) of them per vertex -- round 6: the COLLAB-like batch spent
// 222 us per level there, bound by that one pipe (360 k vertices x 33 crossbar passes per level).  Distances 1, 2, 4, 8 are
// data-parallel primitives of the vector ALU (quad_perm, row_shl / row_shr under bank masks, row_ror), 16 is a ds_swizzle
// (no address, no bank traffic); 32 is v_permlane32_swap: no crossbar pass left.
template <int J>
__device__ __forceinline__ i32 lane_xor(i32 x) {
    if (J == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);            // quad_perm [1, 0, 3, 2]
    if (J == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);            // quad_perm [2, 3, 0, 1]
    if (J == 4) {                                                                      // banks 0, 2 <- lane + 4; banks 1, 3 <- lane - 4
        const i32 t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);      // row_shl:4
        return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);             // row_shr:4
    }
    if (J == 8) return __builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, true);            // row_ror:8
    if (J == 16) return __builtin_amdgcn_ds_swizzle(x, 0x401F);                       // bit mode: and 0x1f, xor 0x10
    // 32: v_permlane32_swap (gfx950) exchanges the upper half of one register with the lower half of another; on two copies of x
    // the first result holds x[0..31] in both halves, the second x[32..63] (tools/micro/lanexor.hip checks every distance)
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return (i32)((threadIdx.x & 32u) ? r[0] : r[1]);
}
__device__ __forceinline__ i32 lane_xor_by(i32 x, int j) {      // j a constant after unrolling
    switch (j) {
        case 1: return lane_xor<1>(x);
        case 2: return lane_xor<2>(x);
        case 4: return lane_xor<4>(x);
        case 8: return lane_xor<8>(x);
        case 16: return lane_xor<16>(x);
        default: return lane_xor<32>(x);
    }
}
// sum of a 64-bit value over the wave, in every lane
__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
    for (int j = 1; j < 64; j <<= 1) {
        const u32 lo = (u32)lane_xor_by((i32)(u32)v, j), hi = (u32)lane_xor_by((i32)(u32)(v >> 32), j);
        v += ((u64)hi << 32) | lo;
    }
    return v;
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
                    const i32 b = lane_xor_by(a, j);
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

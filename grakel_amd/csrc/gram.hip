// Gram matrix  K = Phi_rows . Phi_cols^T  on the MFMA units of gfx950.
//
// Label counts are small non-negative integers, so the product is EXACT integer arithmetic.
// A launch sequence (gk_gram_launch) is:
//   1. gram_ws_kernel      : dense columns as MFMA operands -- counts 0..4 as MX fp4 codes
//                            (v_mfma_scale_f32_32x32x64_f8f6f4, exact), counts 5..127 as int8
//                            (v_mfma_i32_32x32x32_i8); features.hip only allows fp4 when every entry stays
//                            below 2^24 (exact float32 accumulation) and int8 below 2^31: bit-exact versus
//                            the reference's float64 result.  Persistent, warp-specialised, three roles of
//                            four waves: multiply / fill the LDS-DMA operand ring / write the parked previous
//                            tile to K (float64) under the current tile's K loop.
//      gram_dd_kernel      : the same job without parked tile and store waves: the multiplying waves write the
//                            previous tile from a second accumulator set between their K-steps; picked for fp4
//                            jobs of <= 600 tiles (option gram.dd forces / forbids it).
//      gram_tile_kernel    : the plain one-tile-per-workgroup form (option gram.no_ws).
//   2. gram_f64_kernel     : dense columns holding a count > 127 (typical for ShortestPath
//                            histograms) form a narrow float64 side operand,
//                            v_mfma_f64_16x16x4_f64, accumulated onto K (exact while K < 2^53).
//   3. gram_low_kernel     : useful columns present in < 24 graphs (option feat.low_df) never enter a dense operand;
//                            their df*(df-1) pair products are added as float64 atomics.
//   4. gram_normalize_kernel, only when 2. or 3. ran and normalisation was requested.
// Histogram-intersection features (kind 1) arrive unary-expanded (features.hip), so step 1 computes
// sum_l min(c_il, c_jl) exactly; step 2 never applies and step 3 adds min(c_a, c_b) per pair.
// The store epilogue fuses what the reference does in three extra N^2 passes: the per-level sum
// (all levels are concatenated along K), the diagonal (graph-unique label columns are not in
// Phi_s; K_ii is written from the exact selfk vector instead) and -- when no extra term
// follows -- the normalisation K_ij / sqrt(K_ii K_jj) (weisfeiler_lehman.py:323-328,
// kernel.py:195-204).
#include "common.h"
#include "cpu_budget.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>
#include <immintrin.h>
#include <sched.h>
#include <stdio.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double finish_entry(double val, i64 grow, i64 gcol, bool symmetric,
                                               const u64* __restrict__ selfk, i64 n_fit, int normalize) {
    // rows index graphs [row_base, ...), cols index graphs [0, n_cols)
    if (symmetric && grow == gcol) val = (double)selfk[grow];
    if (normalize) {
        double dr = (double)selfk[symmetric ? grow : n_fit + grow];
        double dc = (double)selfk[gcol];
        val = val / sqrt(dr * dc);
        if (normalize == 2) {   // numpy.nan_to_num
            if (val != val) val = 0.0;
            else if (val > 1.7976931348623157e308) val = 1.7976931348623157e308;
        }
    }
    return val;
}

// Block -> output tile.  Workgroup b is observed to run on XCD b % 8 (speed assumption only),
// so consecutive ids of ONE XCD (b>>3) walk 8x8-tile patches: the ~128 tiles resident on an
// XCD then share 2x(8+8) operand panels instead of ~90, and its 4 MiB L2 serves the re-reads.
// Symmetric jobs only visit patches/tiles on or above the diagonal.
#define GI_PATCH 8
#define GI_STRIP 64          // patch >= GI_STRIP: the strip walk with strips of (patch - GI_STRIP) tile columns
__device__ __forceinline__ bool gram_map_tile(int b, int tiles_m, int tiles_n, int sym, int patch,
                                              int& bm, int& bn) {
    if (!patch) {
        bm = b / tiles_n, bn = b % tiles_n;
        return !(sym && bn < bm);
    }
    if (patch >= GI_STRIP) {
        // Strip walk (round 6).  The job's tiles in STRIP-MAJOR order -- strips of W tile columns, inside a strip row by
        // row (a symmetric job: only the tiles on / above the diagonal, and they are the only ones numbered: no holes) --
        // are cut into eight contiguous ranges, one per XCD; the 32 workgroups of an XCD take consecutive tiles of its range,
        // i.e. at any time 32 / W rows x W columns of one strip.  The strip's W column panels are re-read by every one of those
        // rounds and stay in the XCD's L2 for the whole walk down the strip; only the 32 / W row panels of a round are
        // new.  An 8 x 8 patch walked as two rounds of 4 x 8 fetched 12 panels per 32 tiles; this fetches 32 / W (+ W once per strip).
        const int W = patch - GI_STRIP;
        const int xcd = b & 7, i = b >> 3;
        const i64 total = sym ? (i64)tiles_n * (tiles_n + 1) / 2 : (i64)tiles_m * tiles_n;
        const i64 per = (total + 7) / 8;
        i64 g = (i64)xcd * per + i;
        if (i >= per || g >= total) return false;
        if (!sym) {
            const i64 full = (i64)tiles_m * W;
            const int s = (int)(g / full);
            g -= (i64)s * full;
            const int ws = tiles_n - s * W < W ? tiles_n - s * W : W;
            bm = (int)(g / ws), bn = s * W + (int)(g % ws);
            return true;
        }
        int s = 0, ws = W;
        for (;; ++s) {                   // strip s: rows 0 .. s W - 1 in full, then the triangle of its diagonal block
            ws = tiles_n - s * W < W ? tiles_n - s * W : W;
            const i64 n_s = (i64)s * W * ws + (i64)ws * (ws + 1) / 2;
            if (g < n_s) break;
            g -= n_s;
        }
        const i64 rect = (i64)s * W * ws;
        if (g < rect) {
            bm = (int)(g / ws), bn = s * W + (int)(g % ws);
        } else {
            int j = 0, r = (int)(g - rect);
            while (r >= ws - j) r -= ws - j, ++j;
            bm = s * W + j, bn = s * W + j + r;
        }
        return true;
    }
    const int P = patch;
    const int pm = (tiles_m + P - 1) / P, pn = (tiles_n + P - 1) / P;
    const int xcd = b & 7, i = b >> 3;
    const int pid = (i / (P * P)) * 8 + xcd, t = i % (P * P);
    int pr, pc;
    if (sym) {
        int rem = pid;
        pr = 0;
        while (pr < pm && rem >= pn - pr) { rem -= pn - pr; ++pr; }
        if (pr >= pm) return false;
        pc = pr + rem;
    } else {
        if (pid >= pm * pn) return false;
        pr = pid / pn, pc = pid % pn;
    }
    bm = pr * P + t / P, bn = pc * P + t % P;
    if (bm >= tiles_m || bn >= tiles_n) return false;
    return !(sym && bn < bm);
}

static inline i64 gram_grid_blocks(int tiles_m, int tiles_n, int sym, int patch) {
    if (!patch) return (i64)tiles_m * tiles_n;
    if (patch >= GI_STRIP) {
        const i64 total = sym ? (i64)tiles_n * (tiles_n + 1) / 2 : (i64)tiles_m * tiles_n;
        return ((total + 7) / 8) * 8;
    }
    const int P = patch;
    const i64 pm = (tiles_m + P - 1) / P, pn = (tiles_n + P - 1) / P;
    const i64 np = sym ? (pm * pn - pm * (pm - 1) / 2) : pm * pn;   // sym: pm == pn
    return ((np + 7) / 8) * 8 * P * P;
}

// ---------------------------------------------------------------------------------------
// Dense path: 128x128 output tile per 256-thread workgroup (2x2 waves, 64x64 per wave = 2x2 MFMA
// 32x32 tiles), two workgroups per CU so that one tile's float64 store epilogue overlaps the other's
// K loop.  What bounds it, in this order: the float64 store of K (HBM), the operand
// bytes pulled through L2 -> LDS, then the MFMA pipe.  Hence:
//   * operand K-steps are 128 B per row -- whole cache lines: tools/micro/l2lds.hip measures 26 TB/s
//     chip-wide for 128-B row pieces against 16-17 TB/s for 64-B pieces;
//   * counts 0..4 travel as MX fp4 (e2m1) codes with unit block scales, two columns per byte:
//     v_mfma_scale_f32_32x32x64_f8f6f4 multiplies them exactly (tools/micro/mfma_rate.hip) at twice
//     the int8 rate and half the bytes; float32 accumulation is exact because features.hip only
//     chooses fp4 when every Gram entry stays below 2^24.  The few columns with counts 5..127 form a
//     leading int8 region (v_mfma_i32_32x32x32_i8 on the same registers, converted once);
//   * tiles go L2 -> LDS directly (global_load_lds_dwordx4: no VGPR round trip) into a 2-stage ring,
//     one s_barrier per K-step.  The LDS image is linear per wave instruction (8 rows x 128 B =
//     1 KiB), so the bank-conflict-free layout comes from XOR-swizzling the 16-byte chunk index on
//     the SOURCE address and on the fragment read:  physical chunk = logical chunk ^ ((row >> 1) & 7),
//     which spreads every 16-lane ds_read_b128 group over all 16 slots of the 256-B bank row.
// ---------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define GT_BM 128
#define GT_BK 128                                   // operand bytes per row per K-step
#define GT_STAGE (2 * GT_BM * GT_BK)                // A rows then B rows: 32 KiB
#define GT_LDT (GT_BM + 4)                          // padded column of the transposed tile (epilogue)
#define GT_LDS_BYTES (GT_BM * GT_LDT * 4 > 2 * GT_STAGE ? GT_BM * GT_LDT * 4 : 2 * GT_STAGE)

// by value: __builtin_bit_cast applied directly to a vector ELEMENT expression reads element 0 (clang, ROCm 7.2)
__device__ __forceinline__ int gt_bits(float x) { return __builtin_bit_cast(int, x); }

template <bool FP4>
__global__ __launch_bounds__(256) void gram_tile_kernel(
    const int8_t* __restrict__ A, const int8_t* __restrict__ B, i64 ld, int k_steps, int k8_steps,
    const u64* __restrict__ selfk, double* __restrict__ K, i64 M, i64 N, i64 row_base,
    int symmetric, i64 n_fit, int normalize, int tiles_m, int tiles_n, int tri, int patch, i64 ldk, i64 col_base,
    int even) {
    // K points at the job's entry (0, 0); ldk = elements between two of its rows; (row_base, col_base) = the
    // job's origin in the whole matrix (diagonal / normalisation look-ups)
    constexpr int BM = GT_BM, BN = GT_BM, TM = 2, TN = 2, PPW = 8;
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bm, bn;
    if (!gram_map_tile(blockIdx.x, tiles_m, tiles_n, tri, patch, bm, bn)) return;

    // staging: wave w fills stage rows [64w, 64w + 64) (waves 0,1: the A rows, 2,3: the B rows), eight
    // 1-KiB pieces of 8 rows; lane -> (row lane>>3, physical chunk lane&7) fetches the logical chunk
    // physical ^ ((stage row >> 1) & 7)
    const int8_t* gsrc[PPW];
    {
        const int srow = lane >> 3, pch = lane & 7;
        const int8_t* base = wave < 2 ? A + ((i64)bm * BM + wave * 64) * ld : B + ((i64)bn * BN + (wave - 2) * 64) * ld;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int r = q * 8 + srow;                           // row inside the wave's 64 (64w is a multiple of 16)
            gsrc[q] = base + (i64)r * ld + ((pch ^ ((r >> 1) & 7)) << 4);
        }
    }
#define GT_ISSUE(KT)                                                                          \
    {                                                                                         \
        const i64 go = (i64)(KT) * GT_BK;                                                     \
        int8_t* st = smem + ((KT) & 1) * GT_STAGE + wave * (64 * GT_BK);                      \
        _Pragma("unroll") for (int q = 0; q < PPW; ++q)                                       \
            __builtin_amdgcn_global_load_lds((glb_void_t*)(gsrc[q] + go), (lds_void_t*)(st + q * 1024), 16, 0, 0); \
    }

    // fragment addresses: row rr of the stage, slice s (32 B = one MFMA K-slice), half fh = lane >> 5
    const int fr = lane & 31, fh = lane >> 5;
    int offa[TM][4], offb[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rr = wm * 64 + i * 32 + fr;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) offa[i][sl] = rr * GT_BK + (((2 * sl + fh) ^ ((rr >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int rr = BM + wn * 64 + j * 32 + fr;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) offb[j][sl] = rr * GT_BK + (((2 * sl + fh) ^ ((rr >> 1) & 7)) << 4);
    }

    // one register set for both accumulator types: int32 while the int8 region runs, float32 after
    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;       // bit pattern 0 == int32 0

#define GT_MFMA(FA, FB, AS_FP4)                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                            \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                      \
            if (AS_FP4) {                                                                     \
                const v8i xa = {FA[i][0], FA[i][1], FA[i][2], FA[i][3], 0, 0, 0, 0};          \
                const v8i xb = {FB[j][0], FB[j][1], FB[j][2], FB[j][3], 0, 0, 0, 0};          \
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, xb, acc[i][j], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
            } else {                                                                          \
                acc[i][j] = __builtin_bit_cast(v16f, __builtin_amdgcn_mfma_i32_32x32x32_i8(   \
                    FA[i], FB[j], __builtin_bit_cast(v16i, acc[i][j]), 0, 0, 0));             \
            }                                                                                 \
        }
    // K-step: wait for the stage, one barrier (everybody also finished reading the other buffer), queue the
    // next stage into that buffer, then four K-slices with the fragments of slice s+1 in flight while
    // the MFMAs of slice s run
#define GT_STEP(AS_FP4)                                                                       \
    {                                                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        __builtin_amdgcn_s_barrier();                                                         \
        if (kt + 1 < k_steps) GT_ISSUE(kt + 1);                                               \
        const int8_t* st = smem + (kt & 1) * GT_STAGE;                                        \
        v4i fa[2][TM], fb[2][TN];                                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[0][i] = *(const v4i*)(st + offa[i][0]); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[0][j] = *(const v4i*)(st + offb[j][0]); \
        _Pragma("unroll") for (int sl = 0; sl < 4; ++sl) {                                    \
            if (sl < 3) {                                                                     \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[(sl + 1) & 1][i] = *(const v4i*)(st + offa[i][(sl + 1) & 3]); \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[(sl + 1) & 1][j] = *(const v4i*)(st + offb[j][(sl + 1) & 3]); \
            }                                                                                 \
            GT_MFMA(fa[sl & 1], fb[sl & 1], AS_FP4)                                           \
        }                                                                                     \
    }
    GT_ISSUE(0);
    int kt = 0;
    if (FP4) {
        for (; kt < k8_steps; ++kt) GT_STEP(false)
        if (k8_steps > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = (float)gt_bits(acc[i][j][r]);
        }
        for (; kt < k_steps; ++kt) GT_STEP(true)
    } else {
        for (; kt < k_steps; ++kt) GT_STEP(false)
    }
#undef GT_STEP
#undef GT_ISSUE
#undef GT_MFMA
#define GT_VAL(X) (FP4 ? (double)(X) : (double)gt_bits(X))

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool mirror = tri && bm != bn;     // off-diagonal tile of a symmetric job: also write K^T
    // The transposed copy goes through LDS (the operand ring is free after the last K-step): the 32-bit
    // accumulators are written column-major with a 4-word pad per column (conflict-free 16-byte
    // writes), then every wave streams whole columns back -- 128 consecutive K^T entries, 1 KiB of
    // float64 per store instruction -- instead of 16-byte pieces scattered over 32 rows.  Plain counts
    // only (normalised epilogues keep the register path).
    constexpr int LDT = GT_LDT;
    const bool lds_mirror = mirror && normalize == 0 && even;
    float* tsm = (float*)smem;
    if (lds_mirror) __syncthreads();          // the last K-step's fragment reads are done (block-uniform)
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int ctile = (wn * TN + nt) * 32 + (lane & 31);
            const i64 col = (i64)bn * BN + ctile;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rtile = (wm * TM + mt) * 32 + 8 * q + 4 * (lane >> 5);
                const i64 row0 = (i64)bm * BM + rtile;
                double v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i64 row = row0 + j;
                    v[j] = 0.0;
                    if (row < M && col < N) {
                        v[j] = finish_entry(GT_VAL(acc[mt][nt][4 * q + j]), row_base + row, col_base + col,
                                            symmetric != 0, selfk, n_fit, normalize);
                        K[row * ldk + col] = v[j];         // 32 lanes -> 256 contiguous bytes
                    }
                }
                if (lds_mirror) {
                    float4 t;
                    t.x = acc[mt][nt][4 * q], t.y = acc[mt][nt][4 * q + 1];
                    t.z = acc[mt][nt][4 * q + 2], t.w = acc[mt][nt][4 * q + 3];
                    *(float4*)(tsm + ctile * LDT + rtile) = t;
                } else if (mirror && col < N) {             // K[col][row0..row0+3]: 32 B per lane
                    double* dst = K + col * ldk + row0;
                    if (even && row0 + 3 < M) {
                        *(double2*)(dst) = make_double2(v[0], v[1]);
                        *(double2*)(dst + 2) = make_double2(v[2], v[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (row0 + j < M) dst[j] = v[j];
                    }
                }
            }
        }
    if (lds_mirror) {          // block-uniform
        __syncthreads();
        const i64 r = (i64)bm * BM + 2 * lane;              // two consecutive entries of K^T's row per lane
        for (int c = wave; c < BN; c += 4) {
            const i64 krow = (i64)bn * BN + c;
            if (krow >= N) break;
            const float2 t = *(const float2*)(tsm + c * LDT + 2 * lane);
            double* dst = K + krow * ldk + r;
            if (r + 1 < M) *(double2*)dst = make_double2(GT_VAL(t.x), GT_VAL(t.y));
            else if (r < M) dst[0] = GT_VAL(t.x);
        }
    }
#undef GT_VAL
}

// ---------------------------------------------------------------------------------------
// Warp-specialised persistent form of the same tile product (the default): ONE 512-thread workgroup
// per CU walks a strided sequence of tiles.  Waves 0-3 only run K loops; waves 4-7 only write K.  A
// finished tile's 32-bit accumulators are parked in a 64-KiB LDS buffer (column-major, XOR-swizzled)
// and the store waves stream them out -- rows of the tile and, for an off-diagonal tile of a
// symmetric job, rows of its transpose, 1 KiB of float64 per store instruction -- a few rows per
// K-step of the NEXT tile, in lockstep with the compute waves' one barrier per K-step.  So the
// float64 store of K (the HBM roof of this kernel) runs under the K loops instead of after them,
// the operand ring can be three stages deep (two K-steps = 64 KiB in flight per CU; the L2 -> LDS
// path needs that much to stream), and the load pipeline does not drain between tiles.
//   LDS: ring 3 x 32 KiB | parked tile 64 KiB = 160 KiB (all of it).
// ---------------------------------------------------------------------------------------
#define WS_THREADS 768          // 12 waves: 4 multiply, 4 store, 4 load
#define WS_RING 3
#define WS_OUT_OFF (WS_RING * GT_STAGE)
#define WS_LDS_BYTES (WS_OUT_OFF + GT_BM * GT_BM * 4)

struct WsTile { int id, bm, bn, ok; };

__device__ __forceinline__ WsTile ws_next_tile(int id, int stride, int n_ids, int tiles_m, int tiles_n, int tri, int patch) {
    WsTile t;
    t.ok = 0, t.bm = 0, t.bn = 0;
    for (id += stride; id < n_ids; id += stride)
        if (gram_map_tile(id, tiles_m, tiles_n, tri, patch, t.bm, t.bn)) { t.ok = 1; break; }
    t.id = id;
    return t;
}

// numpy.nan_to_num of a normalised entry (mode 2), as finish_entry does
__device__ __forceinline__ double ws_fix(double v, int normalize) {
    if (normalize == 2) {          // wave-uniform; selects, no divergent branches in the store role
        const double m = __builtin_fmin(v, 1.7976931348623157e308);
        v = (v != v) ? 0.0 : m;
    }
    return v;
}
__device__ __forceinline__ double ws_readlane_f64(double v, int l) {      // l wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// parked tile: entry (row, col) of the 128x128 tile as a 32-bit word at
__device__ __forceinline__ int ws_out_addr(int row, int col) {
    return col * (GT_BM * 4) + ((((row >> 2) ^ (col & 31)) << 4) | ((row & 3) << 2));
}

#ifdef GK_ABLATION
// workgroup 0's cycle stamps (s_memtime), per role: [role][0] total, [1] waiting at barriers, [2] K-step body, [3] tile change
// (load role: waiting for its own pieces).  Reading the counter drains lgkmcnt, i.e. it serialises outstanding LDS reads
// into the "body" figure: compare totals across ablations, not bodies in isolation.
__device__ unsigned long long g_ws_dbg[3][4];
#define WS_DBG_DECL unsigned long long t_bar = 0, t_body = 0, t_walk = 0, t_x = 0; const unsigned long long t_start = __builtin_readcyclecounter();
#define WS_DBG_T0() t_x = __builtin_readcyclecounter();
#define WS_DBG_ADD(acc) { const unsigned long long t_y = __builtin_readcyclecounter(); acc += t_y - t_x; t_x = t_y; }
#define WS_DBG_OUT(role_)                                                                       \
    if (blockIdx.x == 0 && (tid & 255) == 0) {                                                 \
        g_ws_dbg[role_][0] = __builtin_readcyclecounter() - t_start, g_ws_dbg[role_][1] = t_bar; \
        g_ws_dbg[role_][2] = t_body, g_ws_dbg[role_][3] = t_walk;                               \
    }
#else
#define WS_DBG_DECL
#define WS_DBG_T0()
#define WS_DBG_ADD(acc)
#define WS_DBG_OUT(role_)
#endif
// ABL (tools' build only, timing ablations with WRONG results), a bit mask: 1 no operand loads, 2 the multiplying waves
// skip their K-steps, 4 MFMAs on fabricated fragments (no LDS reads), 8 no stores, 16 no per-K-step barrier, 32 the store
// waves skip their chunks altogether, 64 no parking of finished tiles, 128 K-step barrier only every second step (races: timing only)
// FULL: the variant that can also take the rare labels' pair updates into its tiles, normalise in the lean store path and
// keep diagonal tiles lean; the plain variant carries none of that code -- its store role is instruction-bound, and every
// wave-uniform test more costs the unnormalised default ~4 %
template <bool FP4, int ABL, bool FULL = false>
__global__ __launch_bounds__(WS_THREADS) void gram_ws_kernel(
    const int8_t* __restrict__ A, const int8_t* __restrict__ B, i64 ld, int k_steps, int k8_steps,
    const u64* __restrict__ selfk, double* __restrict__ K, i64 M, i64 N, i64 row_base,
    int symmetric, i64 n_fit, int normalize, int tiles_m, int tiles_n, int tri, int patch, int n_ids, i64 M_store, unsigned* __restrict__ xcc_ticket,
    i64 ldk, i64 col_base, int even_in, const u32* __restrict__ pair_cnt, const uint2* __restrict__ pairs, int pair_T, int pair_cap,
    const double* __restrict__ rs) {
    constexpr int BM = GT_BM, BN = GT_BM, TM = 2, TN = 2, PPW = 8;
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably uniform: scalar branches, scalar tile walk
    const int role = wave >> 2;                             // wave-uniform: 0 multiply, 1 store, 2 load
    const bool is_compute = role == 0, is_loader = role == 2;
    const int cw = wave & 3, wm = cw >> 1, wn = cw & 1;
    constexpr bool NO_LOAD = (ABL & 1) != 0, NO_COMPUTE = (ABL & 2) != 0, FAKE_READ = (ABL & 4) != 0, NO_STORE = (ABL & 8) != 0,
                   NO_BAR = (ABL & 16) != 0, NO_CHUNK = (ABL & 32) != 0, NO_PARK = (ABL & 64) != 0, HALF_BAR = (ABL & 128) != 0;

    // ---- compute role state -------------------------------------------------------------
    // staging (as gram_tile_kernel): compute wave w fills stage rows [64w, 64w + 64)
    i64 roff[PPW];
    {
        const int srow = lane >> 3, pch = lane & 7;
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int r = q * 8 + srow;
            roff[q] = (i64)r * ld + ((pch ^ ((r >> 1) & 7)) << 4);
        }
    }
    const int fr = lane & 31, fh = lane >> 5;
    int offa[TM][4], offb[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rr = wm * 64 + i * 32 + fr;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) offa[i][sl] = rr * GT_BK + (((2 * sl + fh) ^ ((rr >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int rr = BM + wn * 64 + j * 32 + fr;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) offb[j][sl] = rr * GT_BK + (((2 * sl + fh) ^ ((rr >> 1) & 7)) << 4);
    }
    v16f acc[TM][TN];
#define WS_ZERO_ACC()                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                         \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    WS_ZERO_ACC()

    // ---- the tile sequences: `cur` is being multiplied, `ldt` is being loaded (at most one tile ahead),
    // `prv` is being written out by the store waves
    const int stride = gridDim.x;
    int first_id = (int)blockIdx.x;
    if (xcc_ticket) {          // experiment: identity from the XCD this workgroup really runs on
        int& s_first = *(int*)(smem + WS_OUT_OFF);      // the parked-tile buffer is idle at start
        if (tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 7u;
            const unsigned slot = atomicAdd(&xcc_ticket[xcc], 1u);
            s_first = (int)((slot << 3) | xcc);
        }
        __syncthreads();
        first_id = __builtin_amdgcn_readfirstlane(s_first);
    }
    WsTile cur = ws_next_tile(first_id - stride, stride, n_ids, tiles_m, tiles_n, tri, patch);
    WsTile ldt = cur, prv;
    prv.ok = 0, prv.bm = prv.bn = 0, prv.id = 0;
    int kt_ld = 0, buf_ld = 0, buf_cp = 0, ahead = 0;       // ahead: stages issued and not yet consumed
    const int8_t* src_base = nullptr;
#define WS_SRC_BASE()                                                                          \
    src_base = cw < 2 ? A + ((i64)ldt.bm * BM + cw * 64) * ld : B + ((i64)ldt.bn * BN + (cw - 2) * 64) * ld;
#define WS_ISSUE_NEXT()                                                                        \
    if (ldt.ok) {                                                                              \
        const int8_t* gp = src_base + (i64)kt_ld * GT_BK;                                      \
        int8_t* st = smem + buf_ld * GT_STAGE + cw * (64 * GT_BK);                             \
        if (!NO_LOAD) {                                                                        \
            _Pragma("unroll") for (int q = 0; q < PPW; ++q)                                    \
                __builtin_amdgcn_global_load_lds((glb_void_t*)(gp + roff[q]), (lds_void_t*)(st + q * 1024), 16, 0, 0); \
        }                                                                                      \
        buf_ld = buf_ld == WS_RING - 1 ? 0 : buf_ld + 1;                                       \
        ++ahead;                                                                               \
        if (++kt_ld == k_steps) {                                                              \
            kt_ld = 0;                                                                         \
            ldt = ws_next_tile(ldt.id, stride, n_ids, tiles_m, tiles_n, tri, patch);           \
            WS_SRC_BASE()                                                                      \
        }                                                                                      \
    }
    if (is_loader) {
        WS_SRC_BASE()
        WS_ISSUE_NEXT()
        WS_ISSUE_NEXT()
    }

#define WS_MFMA(FA, FB, AS_FP4)                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
            if (AS_FP4) {                                                                      \
                const v8i xa = {FA[i][0], FA[i][1], FA[i][2], FA[i][3], 0, 0, 0, 0};           \
                const v8i xb = {FB[j][0], FB[j][1], FB[j][2], FB[j][3], 0, 0, 0, 0};           \
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, xb, acc[i][j], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
            } else {                                                                           \
                acc[i][j] = __builtin_bit_cast(v16f, __builtin_amdgcn_mfma_i32_32x32x32_i8(    \
                    FA[i], FB[j], __builtin_bit_cast(v16i, acc[i][j]), 0, 0, 0));              \
            }                                                                                  \
        }
// K-step = 4 K-slices.  One compute wave per SIMD has nobody to hide its LDS latency behind, so the
// fragment reads run two slices ahead of the MFMAs: R0 R1 | R2 M0 | R3 M1 | M2 | M3 (sched_barrier keeps
// the groups apart; the waits the compiler inserts are counted lgkmcnt, LDS returns in order).
#define WS_READ(SL, BUF)                                                                       \
    if (!FAKE_READ) {                                                                          \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[BUF][i] = *(const v4i*)(st + offa[i][SL]); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[BUF][j] = *(const v4i*)(st + offb[j][SL]); \
    } else {                                                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[BUF][i] = (v4i){offa[i][SL], buf_cp, kt_ld, SL};  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[BUF][j] = (v4i){offb[j][SL], buf_cp, kt_ld, SL};  \
    }
#define WS_COMPUTE(AS_FP4)                                                                     \
    {                                                                                          \
        const int8_t* st = smem + buf_cp * GT_STAGE;                                           \
        v4i fa[3][TM], fb[3][TN];                                                              \
        WS_READ(0, 0) WS_READ(1, 1)                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        WS_READ(2, 2) WS_MFMA(fa[0], fb[0], AS_FP4)                                            \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        WS_READ(3, 0) WS_MFMA(fa[1], fb[1], AS_FP4)                                            \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        WS_MFMA(fa[2], fb[2], AS_FP4)                                                          \
        WS_MFMA(fa[0], fb[0], AS_FP4)                                                          \
    }

    // ---- store role: the parked tile `prv` goes out in k_steps chunks, one per K-step of `cur`.
    // Store wave s owns tile rows [32s, 32s+32) (units 0..31) and, for a mirrored tile, the rows
    // [32s, 32s+32) of the transposed tile (units 32..63).  A unit is one 1-KiB row: lane l holds the
    // entries 2l, 2l+1.
    const int sw = wave - 4;
    const bool even = even_in != 0;
    int8_t* const outb = smem + WS_OUT_OFF;
#define WS_VAL(X) (FP4 ? (double)(X) : (double)gt_bits(X))
    auto store_units = [&](int u0, int u1) __attribute__((always_inline)) {
        const bool mirror = tri && prv.bm != prv.bn;
        const int n_units = mirror ? 64 : 32;
        // the plain path has no global load, so the store waves never wait on their own stores (loads and
        // stores share vmcnt and retire in order)
        const i64 d0 = row_base + (i64)prv.bm * BM - col_base - (i64)prv.bn * BN;      // matrix row - column at the tile origin
        const bool slow = normalize != 0 || (symmetric && d0 > -BM && d0 < BN);
        if (u1 > n_units) u1 = n_units;
        for (int u = u0; u < u1; ++u) {
            const bool tr = u >= 32;                       // a row of the transposed tile
            const int r = sw * 32 + (u & 31);              // row of the (transposed) tile
            // global position of the 128-entry row: K[grow][gcol0 + 0..127]
            const i64 grow = (tr ? (i64)prv.bn * BN : (i64)prv.bm * BM) + r;
            const i64 gcol0 = tr ? (i64)prv.bm * BM : (i64)prv.bn * BN;
            const i64 lim_r = tr ? N : M_store, lim_c = tr ? M_store : N;     // transposed rows only exist when M == N
            if (grow >= lim_r) continue;
            const int e = 2 * lane;
            float x0, x1;
            if (tr) {
                const float2 t = *(const float2*)(outb + ws_out_addr(e, r));      // tile entries (row e, col r), (e+1, r)
                x0 = t.x, x1 = t.y;
            } else {
                x0 = *(const float*)(outb + ws_out_addr(r, e));
                x1 = *(const float*)(outb + ws_out_addr(r, e + 1));
            }
            const i64 gc = gcol0 + e;
            double v0 = WS_VAL(x0), v1 = WS_VAL(x1);
            if (slow) {       // tiles that touch the diagonal, normalised jobs: selfk look-ups (global loads)
                // tile coordinates for finish_entry: rows index the job's rows, cols its columns
                const i64 jr0 = tr ? gc : grow, jc0 = tr ? grow : gc, jr1 = tr ? gc + 1 : grow, jc1 = tr ? grow : gc + 1;
                if (gc < lim_c) v0 = finish_entry(v0, row_base + jr0, col_base + jc0, symmetric != 0, selfk, n_fit, normalize);
                if (gc + 1 < lim_c) v1 = finish_entry(v1, row_base + jr1, col_base + jc1, symmetric != 0, selfk, n_fit, normalize);
            }
            double* dst = K + grow * ldk + gc;
            if (even && gc + 1 < lim_c) *(double2*)dst = make_double2(v0, v1);
            else {
                if (gc < lim_c) dst[0] = v0;
                if (gc + 1 < lim_c) dst[1] = v1;
            }
        }
    };
    // Interior tiles of plain jobs (no selfk look-up, even N, tile completely inside the matrix): LEAN batches of four
    // rows.  The store role is instruction-bound -- one wave per SIMD issues a vector instruction every 4-8 cycles and a
    // taken branch costs more -- and it is the role the other two wait for at the K-step barriers (ablation: with the
    // store waves idle the kernel takes 0.18 ms, with the round-2 store code but no store instructions 0.25).  So a
    // batch is straight-line code: the parked layout keeps four consecutive rows of a column in one 16-byte group, i.e.
    // a batch of tile rows is TWO ds_read_b128 (columns e, e + 1) and a batch of rows of the transposed tile FOUR
    // ds_read_b64; lane-constant address parts are computed once per kernel, the tile's destination rows once per tile.
    // The LDS reads are inline asm: in a kernel that also uses LDS-DMA the compiler puts s_waitcnt vmcnt(0) in front of
    // every LDS read, i.e. the store waves would wait for their own stores every row.
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    const unsigned out_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int8_t*)outb;
    const int e = 2 * lane;
    const unsigned lane_c0 = out_lds + (unsigned)e * (GT_BM * 4), lane_c1 = lane_c0 + GT_BM * 4;      // column bases of e, e + 1
    const unsigned lane_x0 = (unsigned)(e & 31), lane_x1 = (unsigned)((e + 1) & 31);
    const unsigned lane_t = out_lds + (unsigned)((e & 3) << 2), lane_g = (unsigned)(e >> 2);          // rows e, e + 1 of a column
    bool prv_plain = false;
    int prv_batches = 0;                 // 8 batches of tile rows, 8 more of transposed rows for a mirrored tile
    double* dst_rows = nullptr;          // K row of tile row 32 sw, column of tile column e
    double* dst_cols = nullptr;          // the same for the transposed tile
    // normalised jobs (rs = 1 / sqrt(self similarity) per graph, float64): entry * rs[row] * rs[col] in the lean path --
    // within 2 ulp of the reference's entry / sqrt(K_ii K_jj); tiles that touch the diagonal take the general path, whose
    // finish_entry is the reference's formula (the diagonal is exactly 1)
    const bool scaled = FULL && normalize != 0 && rs != nullptr;
    const i64 rs_row0 = (symmetric ? 0 : n_fit) + row_base;       // rs index of the job's row 0
    double cf0 = 1.0, cf1 = 1.0, tf0 = 1.0, tf1 = 1.0;   // column factors of the lane: tile columns e, e + 1 | transposed tile
    bool prv_diag = false;
    int fix_mode = 0;
    double dgv = 0.0;                    // diagonal tile: the value of entry (r, r) for tile row r = 32 sw + (lane & 31)
    double rfv = 1.0;                    // row factors, one per lane: lanes 0-31 rs of tile rows 32 sw + lane, lanes 32-63 of the
                                         // transposed tile's rows 32 sw + lane - 32 (read back with v_readlane: no memory
                                         // access, hence no wait for the wave's own stores, per row)
    auto tile_setup = [&]() __attribute__((always_inline)) {
        const bool mirror = tri && prv.bm != prv.bn;
        const i64 d0 = row_base + (i64)prv.bm * BM - col_base - (i64)prv.bn * BN;
        // a tile exactly on the diagonal of a symmetric job stays lean: entry (r, r) is the exact self similarity
        // (normalised: 1, or what 0/0 gives for a graph without features), one value per lane for the wave's 32 rows
        prv_diag = FULL && symmetric && d0 == 0;
        fix_mode = 0;
        if (scaled) {
            const double* cr = rs + col_base + (i64)prv.bn * BN;           // factors of the tile's columns
            const double* rr = rs + rs_row0 + (i64)prv.bm * BM;            // ... of its rows
            cf0 = cr[e], cf1 = cr[e + 1], tf0 = rr[e], tf1 = rr[e + 1];
            rfv = lane < 32 ? rr[sw * 32 + lane] : cr[sw * 32 + lane - 32];
            // numpy.nan_to_num only matters when a factor is infinite (a graph without features): wave-uniform, rare
            const bool odd = !(cf0 <= 1.0) || !(cf1 <= 1.0) || !(tf0 <= 1.0) || !(tf1 <= 1.0) || !(rfv <= 1.0);
            fix_mode = (normalize == 2 && __builtin_amdgcn_ballot_w64(odd) != 0ull) ? 2 : 0;
        }
        if (prv_diag) {
            const u64 sk = selfk[rs_row0 + (i64)prv.bm * BM + sw * 32 + (lane & 31)];
            dgv = normalize == 0 ? (double)sk : (sk != 0 ? 1.0 : (normalize == 2 ? 0.0 : __builtin_nan("")));
        }
        prv_plain = (normalize == 0 || scaled) && !(symmetric && d0 > -BM && d0 < BN && !prv_diag) && even &&
                    ((i64)prv.bm + 1) * BM <= M_store && ((i64)prv.bn + 1) * BN <= N &&
                    (!mirror || (((i64)prv.bn + 1) * BN <= M_store && ((i64)prv.bm + 1) * BM <= N));
        prv_batches = mirror ? 16 : 8;
        dst_rows = K + ((i64)prv.bm * BM + sw * 32) * ldk + (i64)prv.bn * BN + e;
        dst_cols = K + ((i64)prv.bn * BN + sw * 32) * ldk + (i64)prv.bm * BM + e;
    };
    auto store_batch = [&](int b) __attribute__((always_inline)) {          // b wave-uniform
        if (b < 8) {
            const unsigned g = (unsigned)(sw * 8 + b);
            v4f x0, x1;
            asm volatile("ds_read_b128 %0, %1" : "=v"(x0) : "v"(lane_c0 + ((g ^ lane_x0) << 4)));
            asm volatile("ds_read_b128 %0, %1" : "=v"(x1) : "v"(lane_c1 + ((g ^ lane_x1) << 4)));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1));
            double* const d = dst_rows + (i64)(4 * b) * ldk;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v2d v = {WS_VAL(x0[q]), WS_VAL(x1[q])};
                if (scaled) {
                    const double rf = ws_readlane_f64(rfv, 4 * b + q);
                    v.x = ws_fix(v.x * rf * cf0, fix_mode), v.y = ws_fix(v.y * rf * cf1, fix_mode);
                }
                if (prv_diag) {                                        // wave-uniform
                    const int rt = sw * 32 + 4 * b + q;
                    const double dv = ws_readlane_f64(dgv, 4 * b + q);
                    v.x = e == rt ? dv : v.x, v.y = e + 1 == rt ? dv : v.y;
                }
                if (!NO_STORE) __builtin_nontemporal_store(v, (v2d*)(d + (i64)q * ldk));
                else if (v.x == 1.2345e300) d[0] = v.y;
            }
        } else {
            const unsigned c0 = (unsigned)(sw * 32 + 4 * (b - 8));
            float2 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned c = c0 + (unsigned)q;
                asm volatile("ds_read_b64 %0, %1" : "=v"(x[q]) : "v"(lane_t + c * (GT_BM * 4) + ((lane_g ^ (c & 31u)) << 4)));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
            double* const d = dst_cols + (i64)(4 * (b - 8)) * ldk;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v2d v = {WS_VAL(x[q].x), WS_VAL(x[q].y)};
                if (scaled) {
                    const double rf = ws_readlane_f64(rfv, 32 + 4 * (b - 8) + q);
                    v.x = ws_fix(v.x * rf * tf0, fix_mode), v.y = ws_fix(v.y * rf * tf1, fix_mode);
                }
                if (!NO_STORE) __builtin_nontemporal_store(v, (v2d*)(d + (i64)q * ldk));
                else if (v.x == 1.2345e300) d[0] = v.y;
            }
        }
    };
    const int k_first = 0, k_out = k_steps;
    const int upc = k_out > 0 ? (64 + k_out - 1) / k_out : 64;                 // general path: units per K-step (a mirrored tile has 64 per wave)

    // ---- rare labels' pair updates, folded into the tile (pairs != nullptr): right after parking its quadrant of a tile a
    // multiplying wave reads the tile's pair list and adds the values that fall into ITS OWN quadrant to the parked entries
    // with LDS atomics -- LDS executes a wave's instructions in order, so no barrier is needed between its own park and its
    // own adds, and the adds are complete (lgkmcnt) before the wave reaches the next K-step barrier, after which the store
    // waves start on the tile.  The multiplying waves have no other vector-memory traffic and wait at that barrier anyway.
    // Integer values: exact in the float32 / int32 the tile is parked in.  A tile on the diagonal holds (r, c) and (c, r).
    const bool fold = FULL && pairs != nullptr;
    auto pairs_apply = [&](const WsTile& t) __attribute__((always_inline)) {
        const int idx = t.bm * pair_T + t.bn;
        const int n = (int)pair_cnt[idx];
        const int hi = n < pair_cap ? n : pair_cap;          // the rest sits in the overflow list
        const uint2* __restrict__ bucket = pairs + (i64)idx * pair_cap;
        const bool diag = t.bm == t.bn;
        for (int j0 = 0; j0 < hi; j0 += 256) {
            uint2 p[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = j0 + lane + 64 * i;
                p[i] = j < hi ? bucket[j] : make_uint2(0u, 0u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (j0 + lane + 64 * i >= hi) continue;
                const int r = (int)(p[i].x & 127u), c = (int)((p[i].x >> 7) & 127u);
                const bool own = (r >> 6) == wm && (c >> 6) == wn, own_t = diag && (c >> 6) == wm && (r >> 6) == wn;
                const unsigned a0 = out_lds + (unsigned)ws_out_addr(r, c), a1 = out_lds + (unsigned)ws_out_addr(c, r);
                if (FP4) {
                    const float v = (float)p[i].y;
                    if (own) asm volatile("ds_add_f32 %0, %1" ::"v"(a0), "v"(v) : "memory");
                    if (own_t) asm volatile("ds_add_f32 %0, %1" ::"v"(a1), "v"(v) : "memory");
                } else {
                    if (own) asm volatile("ds_add_u32 %0, %1" ::"v"(a0), "v"(p[i].y) : "memory");
                    if (own_t) asm volatile("ds_add_u32 %0, %1" ::"v"(a1), "v"(p[i].y) : "memory");
                }
            }
        }
    };

    WS_DBG_DECL
    // ---- main loops: one per role, with the same barrier sequence (k_steps + 1 per tile, one at the end).
    // Separate loop nests keep the accumulators of the compute role in one straight-line K loop.
    if (is_compute) {
#define WS_STEP(AS_FP4)                                                                        \
    {                                                                                          \
        WS_DBG_T0()                                                                            \
        if (!NO_BAR && !(HALF_BAR && (kt & 1))) __builtin_amdgcn_s_barrier();       /* the load waves saw this stage land */ \
        WS_DBG_ADD(t_bar)                                                                      \
        if (!NO_COMPUTE) WS_COMPUTE(AS_FP4)                                                    \
        buf_cp = buf_cp == WS_RING - 1 ? 0 : buf_cp + 1;                                       \
        WS_DBG_ADD(t_body)                                                                     \
    }
        while (cur.ok) {
            int kt = 0;
            if (FP4) {
                for (; kt < k8_steps; ++kt) WS_STEP(false)
                if (k8_steps > 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][j][r] = (float)gt_bits(acc[i][j][r]);
                }
                for (; kt < k_steps; ++kt) WS_STEP(true)
            } else {
                for (; kt < k_steps; ++kt) WS_STEP(false)
            }
            // hand-over: the store waves have read the parked tile completely, park this one
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WS_DBG_T0()
            __builtin_amdgcn_s_barrier();
            WS_DBG_ADD(t_bar)
            if (!NO_PARK)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    const int ctile = (wn * TN + nt) * 32 + (lane & 31);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rtile = (wm * TM + mt) * 32 + 8 * q + 4 * (lane >> 5);
                        typedef float v4f __attribute__((ext_vector_type(4)));
                        const v4f t = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
                        // asm: a plain LDS store would make the compiler drain the LDS-DMA prefetch (vmcnt(0)) first
                        asm volatile("ds_write_b128 %0, %1" ::"v"(out_lds + (unsigned)ws_out_addr(rtile, ctile)), "v"(t) : "memory");
                    }
                }
            if (fold) {
                pairs_apply(cur);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the adds are in before the next barrier
            }
            WS_ZERO_ACC()
            cur = ws_next_tile(cur.id, stride, n_ids, tiles_m, tiles_n, tri, patch);
            WS_DBG_ADD(t_walk)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        WS_DBG_OUT(0)
#undef WS_STEP
    } else if (is_loader) {
        // ---- load role: the operand ring.  Issuing a 1-KiB LDS-DMA piece costs the issuing wave ~100 cycles
        // (MI355X_MICROARCH.md "LDS-DMA piece"; measured here: 820 cycles per K-step for eight pieces), so the pieces
        // stay out of the multiplying waves' instruction streams: a wave per SIMD does nothing else.  Wait until the own
        // pieces of the stage have landed, meet the others at the K-step's barrier, queue the stage after next into the
        // buffer the multiplying waves have just left.
        while (cur.ok) {
            for (int kt = 0; kt < k_steps; ++kt) {
                WS_DBG_T0()
                if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WS_DBG_ADD(t_walk)
                if (!NO_BAR && !(HALF_BAR && (kt & 1))) __builtin_amdgcn_s_barrier();
                WS_DBG_ADD(t_bar)
                --ahead;
                WS_ISSUE_NEXT()
                WS_DBG_ADD(t_body)
            }
            WS_DBG_T0()
            __builtin_amdgcn_s_barrier();
            WS_DBG_ADD(t_bar)
            cur = ws_next_tile(cur.id, stride, n_ids, tiles_m, tiles_n, tri, patch);
        }
        __builtin_amdgcn_s_barrier();
        WS_DBG_OUT(2)
    } else {
        while (cur.ok) {
            int due = 0, next_b = 0;               // batches spread evenly over the K-steps: due += batches, one leaves per k_out
            for (int kt = 0; kt < k_steps; ++kt) {
                WS_DBG_T0()
                if (!NO_BAR && !(HALF_BAR && (kt & 1))) __builtin_amdgcn_s_barrier();
                WS_DBG_ADD(t_bar)
                if (prv.ok && !NO_CHUNK && kt >= k_first) {          // folding: the pair updates land during the first K-step
                    if (prv_plain) {
                        for (due += prv_batches; due >= k_out; due -= k_out) store_batch(next_b++);
                    } else store_units((kt - k_first) * upc, (kt - k_first + 1) * upc);
                }
                WS_DBG_ADD(t_body)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WS_DBG_T0()
            __builtin_amdgcn_s_barrier();
            WS_DBG_ADD(t_bar)
            prv = cur;
            tile_setup();
            cur = ws_next_tile(cur.id, stride, n_ids, tiles_m, tiles_n, tri, patch);
            WS_DBG_ADD(t_walk)
        }
        // drain: the last parked tile
        __builtin_amdgcn_s_barrier();
        if (prv.ok) {
            if (prv_plain) {
                for (int b = 0; b < prv_batches; ++b) store_batch(b);
            } else store_units(0, 64);
        }
        WS_DBG_OUT(1)
    }
#undef WS_VAL
#undef WS_COMPUTE
#undef WS_READ
#undef WS_MFMA
#undef WS_ISSUE_NEXT
#undef WS_SRC_BASE
#undef WS_ZERO_ACC
}

// ---------------------------------------------------------------------------------------
// Third form (option gram.dd): NO parked tile and no store waves.  Four multiplying waves keep TWO accumulator sets: the
// tile being multiplied and the finished previous tile, whose entries they convert and store a few registers at a time
// between the K-steps of the current one (one (row group, 32x32 block) per K-step: four 256-byte row stores + two
// 16-byte pieces of the mirrored tile per lane).  The 64 KiB of LDS the parked tile took go to the operand ring: five
// stages, four of them in flight (128 KiB instead of 64), filled by four load waves.  Eight waves per workgroup.
// What it trades: the stores leave from the multiplying waves' own instruction streams (an HBM back-pressure stall there
// stalls their MFMA issue too), and the mirrored half goes out as 64-byte runs instead of 1-KiB rows.
// ---------------------------------------------------------------------------------------
#define DD_THREADS 512
#define DD_RING 5
#define DD_LDS_BYTES (DD_RING * GT_STAGE)

template <bool FP4>
__global__ __launch_bounds__(DD_THREADS) void gram_dd_kernel(
    const int8_t* __restrict__ A, const int8_t* __restrict__ B, i64 ld, int k_steps, int k8_steps,
    const u64* __restrict__ selfk, double* __restrict__ K, i64 M, i64 N, i64 row_base,
    int symmetric, i64 n_fit, int normalize, int tiles_m, int tiles_n, int tri, int patch, int n_ids, i64 ldk, i64 col_base,
    int even_in) {
    constexpr int BM = GT_BM, BN = GT_BM, TM = 2, TN = 2, PPW = 8;
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_compute = wave < 4;
    const int cw = wave & 3, wm = cw >> 1, wn = cw & 1;
    const int stride = gridDim.x;
    WsTile cur = ws_next_tile((int)blockIdx.x - stride, stride, n_ids, tiles_m, tiles_n, tri, patch);
    if (!is_compute) {
        // ---- load role: stage g (global K-step counter) goes to ring slot g % 5; at most four stages in flight
        i64 roff[PPW];
        {
            const int srow = lane >> 3, pch = lane & 7;
#pragma unroll
            for (int q = 0; q < PPW; ++q) {
                const int r = q * 8 + srow;
                roff[q] = (i64)r * ld + ((pch ^ ((r >> 1) & 7)) << 4);
            }
        }
        WsTile ldt = cur;
        int kt_ld = 0, buf_ld = 0, ahead = 0;
        const int8_t* src_base = nullptr;
        auto set_base = [&]() __attribute__((always_inline)) {
            src_base = cw < 2 ? A + ((i64)ldt.bm * BM + cw * 64) * ld : B + ((i64)ldt.bn * BN + (cw - 2) * 64) * ld;
        };
        auto issue = [&]() __attribute__((always_inline)) {
            if (!ldt.ok) return;
            const int8_t* gp = src_base + (i64)kt_ld * GT_BK;
            int8_t* st = smem + buf_ld * GT_STAGE + cw * (64 * GT_BK);
#pragma unroll
            for (int q = 0; q < PPW; ++q)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(gp + roff[q]), (lds_void_t*)(st + q * 1024), 16, 0, 0);
            buf_ld = buf_ld == DD_RING - 1 ? 0 : buf_ld + 1;
            ++ahead;
            if (++kt_ld == k_steps) {
                kt_ld = 0;
                ldt = ws_next_tile(ldt.id, stride, n_ids, tiles_m, tiles_n, tri, patch);
                set_base();
            }
        };
        set_base();
        issue(); issue(); issue(); issue();
        while (cur.ok) {
            for (int kt = 0; kt < k_steps; ++kt) {
                // the oldest stage in flight has landed when at most (ahead - 1) stages' pieces are outstanding
                if (ahead >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
                else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
                else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                --ahead;
                issue();          // into the slot the multiplying waves left at the previous step
            }
            cur = ws_next_tile(cur.id, stride, n_ids, tiles_m, tiles_n, tri, patch);
        }
        return;
    }
    // ---- multiply role
    const int fr = lane & 31, fh = lane >> 5;
    int offa[TM][4], offb[TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rr = wm * 64 + i * 32 + fr;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) offa[i][sl] = rr * GT_BK + (((2 * sl + fh) ^ ((rr >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int rr = BM + wn * 64 + j * 32 + fr;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) offb[j][sl] = rr * GT_BK + (((2 * sl + fh) ^ ((rr >> 1) & 7)) << 4);
    }
    v16f acc[TM][TN], old[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f, old[i][j][r] = 0.0f;
    WsTile prv;
    prv.ok = 0, prv.bm = prv.bn = 0, prv.id = 0;
    const bool even = even_in != 0;
#define DD_VAL(X) (FP4 ? (double)(X) : (double)gt_bits(X))
    // rows 8 Q + 4 (lane >> 5) + 0..3 of block (MT, NT) of the PREVIOUS tile: four row stores (32 lanes = 256 contiguous
    // bytes each) and, for an off-diagonal tile of a symmetric job, the same four values as 32 contiguous bytes of the
    // mirrored tile's row
#define DD_GROUP(MT, NT, Q)                                                                       \
    {                                                                                             \
        const int ctile = (wn * TN + (NT)) * 32 + (lane & 31);                                    \
        const i64 col = (i64)prv.bn * BN + ctile;                                                 \
        const int rtile = (wm * TM + (MT)) * 32 + 8 * (Q) + 4 * (lane >> 5);                      \
        const i64 row0 = (i64)prv.bm * BM + rtile;                                                \
        double v[4];                                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
            const i64 row = row0 + j;                                                             \
            v[j] = DD_VAL(old[MT][NT][4 * (Q) + j]);                                              \
            if (row < M && col < N) {                                                             \
                if (slow) v[j] = finish_entry(v[j], row_base + row, col_base + col, symmetric != 0, selfk, n_fit, normalize); \
                __builtin_nontemporal_store(v[j], &K[row * ldk + col]);                           \
            }                                                                                     \
        }                                                                                         \
        if (mirror && col < N) {                                                                  \
            double* dst = K + col * ldk + row0;                                                   \
            if (even && row0 + 3 < M) {                                                           \
                typedef double v2d_ __attribute__((ext_vector_type(2)));                          \
                __builtin_nontemporal_store((v2d_){v[0], v[1]}, (v2d_*)dst);                      \
                __builtin_nontemporal_store((v2d_){v[2], v[3]}, (v2d_*)(dst + 2));                \
            } else {                                                                              \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                     \
                    if (row0 + j < M) dst[j] = v[j];                                              \
            }                                                                                     \
        }                                                                                         \
    }
    // group g = 0..15 of the previous tile: (MT, NT, Q) = (g >> 3, (g >> 2) & 1, g & 3); g is a compile-time constant at
    // every use (the K loop is unrolled sixteen-fold), so the accumulator registers are addressed statically.  Interior
    // tiles of plain jobs (no diagonal, no normalisation, no ragged edge: prv_plain) take the short form DD_FAST from
    // per-tile lane pointers; every other tile is written in one go by DD_STORE before its successor starts.
#define DD_STORE(G)                                                                               \
    {                                                                                             \
        const bool mirror = tri && prv.bm != prv.bn;                                              \
        const i64 d0 = row_base + (i64)prv.bm * BM - col_base - (i64)prv.bn * BN;                 \
        const bool slow = normalize != 0 || (symmetric && d0 > -BM && d0 < BN);                   \
        DD_GROUP(((G) >> 3), (((G) >> 2) & 1), ((G) & 3))                                         \
    }
    typedef double v2d_ __attribute__((ext_vector_type(2)));
    double* p_dir = nullptr;         // previous tile: &K[row of this lane in block (0,0), group 0][its column]
    double* p_mir = nullptr;         // ... and the mirrored position &K[its column][that row]
    bool prv_plain = false, prv_mirror = false;
#define DD_FAST(G)                                                                                \
    {                                                                                             \
        const int MT_ = (G) >> 3, NT_ = ((G) >> 2) & 1, Q_ = (G) & 3;                             \
        double* pd = p_dir + (i64)(MT_ * 32 + 8 * Q_) * ldk + NT_ * 32;                           \
        const double v0 = DD_VAL(old[MT_][NT_][4 * Q_]), v1 = DD_VAL(old[MT_][NT_][4 * Q_ + 1]);  \
        const double v2 = DD_VAL(old[MT_][NT_][4 * Q_ + 2]), v3 = DD_VAL(old[MT_][NT_][4 * Q_ + 3]); \
        __builtin_nontemporal_store(v0, pd);                                                      \
        __builtin_nontemporal_store(v1, pd + ldk);                                                \
        __builtin_nontemporal_store(v2, pd + 2 * ldk);                                            \
        __builtin_nontemporal_store(v3, pd + 3 * ldk);                                            \
        if (prv_mirror) {                                                                         \
            double* pm = p_mir + (i64)(NT_ * 32) * ldk + MT_ * 32 + 8 * Q_;                       \
            __builtin_nontemporal_store((v2d_){v0, v1}, (v2d_*)pm);                               \
            __builtin_nontemporal_store((v2d_){v2, v3}, (v2d_*)(pm + 2));                         \
        }                                                                                         \
    }
    int buf_cp = 0;
#define DD_MFMA(FA, FB, AS_FP4)                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
            if (AS_FP4) {                                                                      \
                const v8i xa = {FA[i][0], FA[i][1], FA[i][2], FA[i][3], 0, 0, 0, 0};           \
                const v8i xb = {FB[j][0], FB[j][1], FB[j][2], FB[j][3], 0, 0, 0, 0};           \
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, xb, acc[i][j], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
            } else {                                                                           \
                acc[i][j] = __builtin_bit_cast(v16f, __builtin_amdgcn_mfma_i32_32x32x32_i8(    \
                    FA[i], FB[j], __builtin_bit_cast(v16i, acc[i][j]), 0, 0, 0));              \
            }                                                                                  \
        }
#define DD_READ(SL, BUF)                                                                       \
    {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[BUF][i] = *(const v4i*)(st + offa[i][SL]); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[BUF][j] = *(const v4i*)(st + offb[j][SL]); \
    }
#define DD_STEP(AS_FP4)                                                                        \
    {                                                                                          \
        __builtin_amdgcn_s_barrier();                                                          \
        const int8_t* st = smem + buf_cp * GT_STAGE;                                           \
        v4i fa[3][TM], fb[3][TN];                                                              \
        DD_READ(0, 0) DD_READ(1, 1)                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        DD_READ(2, 2) DD_MFMA(fa[0], fb[0], AS_FP4)                                            \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        DD_READ(3, 0) DD_MFMA(fa[1], fb[1], AS_FP4)                                            \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        DD_MFMA(fa[2], fb[2], AS_FP4)                                                          \
        DD_MFMA(fa[0], fb[0], AS_FP4)                                                          \
        buf_cp = buf_cp == DD_RING - 1 ? 0 : buf_cp + 1;                                       \
    }
#define DD_CONVERT()                                                                           \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                             \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                         \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = (float)gt_bits(acc[i][j][r]);
    while (cur.ok) {
        int kt = 0;
        if (FP4) {                                                 // the (few) int8 K-steps first; the stores ride the others
            for (; kt < k8_steps; ++kt) DD_STEP(false)
            if (k8_steps > 0) { DD_CONVERT() }
        }
        const int k_rest = k_steps - kt;
        for (int kt0 = 0; kt0 < k_rest; kt0 += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (kt0 + u < k_rest) {                           // wave-uniform
                    DD_STEP(FP4)
                    if (prv_plain && kt0 == 0) DD_FAST(u)         // one group of the previous tile per K-step
                }
            }
        }
        if (prv_plain) {                                          // fewer than 16 such K-steps: the rest of the previous tile
#pragma unroll
            for (int g = 0; g < 16; ++g)
                if (g >= k_rest) DD_FAST(g)
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                old[i][j] = acc[i][j];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            }
        prv = cur;
        cur = ws_next_tile(cur.id, stride, n_ids, tiles_m, tiles_n, tri, patch);
        {
            const i64 d0 = row_base + (i64)prv.bm * BM - col_base - (i64)prv.bn * BN;
            prv_mirror = tri && prv.bm != prv.bn;
            prv_plain = normalize == 0 && !(symmetric && d0 > -BM && d0 < BN) && even && ((i64)prv.bm + 1) * BM <= M &&
                        ((i64)prv.bn + 1) * BN <= N && (!prv_mirror || (((i64)prv.bn + 1) * BN <= M && ((i64)prv.bm + 1) * BM <= N));
            const i64 r_l = (i64)prv.bm * BM + wm * 64 + 4 * (lane >> 5), c_l = (i64)prv.bn * BN + wn * 64 + (lane & 31);
            p_dir = K + r_l * ldk + c_l;
            p_mir = K + c_l * ldk + r_l;
        }
        if (!prv_plain) {                                         // diagonal / edge / normalised tile: all of it now
#pragma unroll
            for (int g = 0; g < 16; ++g) DD_STORE(g)
        }
    }
    if (prv.ok && prv_plain) {
#pragma unroll
        for (int g = 0; g < 16; ++g) DD_FAST(g)
    }
#undef DD_FAST
#undef DD_CONVERT
#undef DD_STORE
#undef DD_STEP
#undef DD_READ
#undef DD_MFMA
#undef DD_GROUP
#undef DD_VAL
}

// ---------------------------------------------------------------------------------------
// Rare labels' pair updates binned by output tile (for gram_ws_kernel's fold-in), in ONE pass: a thread per entry
// (label, graph a) walks the later entries (graph b) of its label's list and drops {row in tile | col in tile << 7, value}
// into the bucket of tile (min(a,b) / 128, max(a,b) / 128) -- a slot from the tile's counter.  Buckets have a fixed
// capacity (four times the mean load + 128); a pair that finds its bucket full goes to an overflow list that a small
// kernel applies as float64 atomics after the tile kernel (normalised there if the job is).  No host synchronisation.
// ---------------------------------------------------------------------------------------
__global__ void gram_bin_pairs_kernel(const i32* __restrict__ low_lab, const u32* __restrict__ roff, const u32* __restrict__ df,
                                      const i32* __restrict__ lgraph, const i32* __restrict__ lcnt, i64 n_entries, int minsum,
                                      int T, int cap, u32* __restrict__ tile_cnt, uint2* __restrict__ bucket,
                                      u32* __restrict__ ovf_n, uint4* __restrict__ ovf, u32 ovf_cap) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const i32 q = low_lab[e];
    const u32 t0 = roff[q];
    const int m = (int)df[q], ia = (int)(e - (i64)t0);
    const i32 ga = lgraph[e];
    const u32 ca = (u32)lcnt[e];
#pragma unroll 4
    for (int ib = ia + 1; ib < m; ++ib) {
        const i32 gb = lgraph[t0 + ib];
        const u32 cb = (u32)lcnt[t0 + ib];
        if (ga == gb) continue;
        const i32 r = ga < gb ? ga : gb, c = ga < gb ? gb : ga;
        const int tile = (r >> 7) * T + (c >> 7);
        const u32 value = minsum ? (ca < cb ? ca : cb) : ca * cb;
        const u32 pos = atomicAdd(&tile_cnt[tile], 1u);
        if (pos < (u32)cap) bucket[(i64)tile * cap + pos] = make_uint2((u32)(r & 127) | ((u32)(c & 127) << 7), value);
        else {
            const u32 o = atomicAdd(ovf_n, 1u);
            if (o < ovf_cap) ovf[o] = make_uint4((u32)r, (u32)c, value, 0u);
        }
    }
}

// the overflow list (usually empty): K[r][c] += v, K[c][r] += v; a normalised job gets the normalised contribution
__global__ void gram_pairs_overflow_kernel(const u32* __restrict__ ovf_n, const uint4* __restrict__ ovf, u32 ovf_cap,
                                           double* __restrict__ K, i64 ldk, const u64* __restrict__ selfk, int normalize) {
    const u32 n = *ovf_n < ovf_cap ? *ovf_n : ovf_cap;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint4 p = ovf[i];
        double v = (double)p.z;
        if (normalize) v /= sqrt((double)selfk[p.x] * (double)selfk[p.y]);
        atomicAdd(&K[(i64)p.x * ldk + p.y], v);
        atomicAdd(&K[(i64)p.y * ldk + p.x], v);
    }
}

// rs[g] = 1 / sqrt(self similarity of graph g): the lean store path of a normalised job multiplies by two of these
__global__ void gram_rs_kernel(const u64* __restrict__ selfk, i64 n, double* __restrict__ rs) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rs[i] = 1.0 / sqrt((double)selfk[i]);
}

// Does the fold-in pay?  Measured (config 3, 10 000 graphs): binning 48 us + 8 us inside the tile kernel against 46 us of
// float64 atomics -- a loss -- unless the job is normalised: the atomics route then needs a separate pass over the whole
// matrix (0.3 ms there).  The binning kernel is bound by same-address atomics on the tile counters (~0.1 us each in
// series: ShortestPath's 4 110 graphs with 2 000 pairs per tile took 232 us), the normalisation pass by 16 N^2 bytes.
static bool gram_fold_pays(const gk_feat* f, int normalize) {
    if (!normalize) return false;
    const double N = (double)f->n_graphs, T = ceil(N / GT_BM), tiles = T * (T + 1) / 2;
    const double bound = (double)f->rare_entries * (double)(f->low_df > 1 ? f->low_df - 1 : 1) / 2;
    return 20.0 + 0.1 * bound / tiles < 16.0 * N * N / 5.0e6;          // microseconds
}

static int gram_build_pairs(gk_ctx* ctx, gk_feat* f) {
    const i64 N = f->n_graphs;
    const int T = (int)cdiv(N, GT_BM);
    const i64 bound = std::max<i64>(1, f->rare_entries * (i64)(f->low_df > 1 ? f->low_df - 1 : 1) / 2);     // pairs at most
    const i64 tiles = (i64)T * (T + 1) / 2;
    i64 cap = 4 * cdiv(bound, tiles) + 128;
    if (cap > 4096) cap = 4096;
    if (ctx->opt.gram_pair_cap > 0) cap = ctx->opt.gram_pair_cap;
    void* q = nullptr;
    GK_TRY(gk_dev_alloc(ctx, &q, ((size_t)T * T + 1) * 4));
    f->pair_cnt = (u32*)q, f->arena.push_back(q);
    f->pair_ovf_n = f->pair_cnt + (size_t)T * T;
    GK_TRY(gk_dev_alloc(ctx, &q, (size_t)T * T * (size_t)cap * 8));
    f->pairs = (uint2*)q, f->arena.push_back(q);
    GK_TRY(gk_dev_alloc(ctx, &q, (size_t)bound * 16));
    f->pair_ovf = (uint4*)q, f->arena.push_back(q);
    f->pair_ovf_cap = bound, f->pair_cap = (int)cap;
    GK_TRY(gk_zero_async(ctx, f->pair_cnt, ((size_t)T * T + 1) * 4));
    gram_bin_pairs_kernel<<<dim3((unsigned)cdiv(f->rare_entries, 256)), dim3(256), 0, ctx->stream>>>(
        f->gm_low_lab, f->gm_roff, f->gm_df, f->gm_low_graph, f->gm_low_cnt, f->rare_entries, f->kind == GK_FEAT_MINSUM ? 1 : 0, T,
        (int)cap, f->pair_cnt, f->pairs, f->pair_ovf_n, f->pair_ovf, (u32)std::min<i64>(bound, 0xffffffffll));
    GK_HIP_CHECK(hipGetLastError());
    f->pair_T = T;
    return GK_OK;
}

// which kernel form launch_tiles takes for a job: 0 plain tile kernel, 1 gram_ws_kernel, 2 gram_dd_kernel
static int tiles_kernel_form(gk_ctx* ctx, gk_feat* f, i64 M, i64 n_cols, int tri) {
    if (ctx->opt.gram_no_ws) return 0;
    // the direct-store form wins for small jobs (round 3, fp4 operands: N = 2000 0.035 vs 0.050 ms,
    // N = 4000 0.070 vs 0.079, N = 6000 0.132 vs 0.110; int8-only operands lose at 561 tiles: 0.170 vs 0.140);
    // option gram.dd: 0 = this rule, 1 = always, 2 = never
    // Round 5 (tools/dev/dd_vs_ws.py): the rule was "up to 600 tiles"; with few K-steps per tile the direct-store form has no
    // K loop to hide its stores under and loses as soon as a CU gets a second tile -- 528 tiles of 8 K-steps 0.083 vs 0.072 ms,
    // the NCI1-like set (561 tiles, 8 K-steps, a ragged last tile row) 0.135 vs 0.065 ms; it still wins while every CU has at
    // most ONE tile (136 tiles: 0.038 vs 0.054, 36 tiles: 0.028 vs 0.048)
    const i64 tiles_m = cdiv(M, GT_BM), tiles_n = cdiv(n_cols, GT_BM);
    const i64 real_tiles = tri ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
    const i64 one_round = ctx->n_cu > 0 ? ctx->n_cu : 256;
    return (ctx->opt.gram_dd == 1 || (ctx->opt.gram_dd == 0 && f->phi_fp4 && real_tiles <= one_round)) ? 2 : 1;
}

static int launch_tiles(gk_ctx* ctx, gk_feat* f, const int8_t* a, const int8_t* b, i64 M, i64 n_cols,
                        i64 row_lo, int normalize, double* K, int tri, int patch, double* entries_done, i64 ldk, i64 col_lo,
                        bool want_fold, bool* folded) {
    // a / b: operand rows of the job's first row / first column; K: the job's entry (0, 0)
    const int even = ((uintptr_t)K % 16 == 0 && ldk % 2 == 0) ? 1 : 0;      // 16-byte stores of two float64
    const int tiles_m = (int)cdiv(M, GT_BM), tiles_n = (int)cdiv(n_cols, GT_BM);
    // Tile order (round 6): the strip walk (gram_map_tile) instead of the 8 x 8 patches of rounds 1-5.  The kernel is bound by
    // what crosses the fabric behind the L2s -- the float64 store of K AND the operand panels every XCD-round has to fetch anew
    // (PMC: 538 MB of fetches for a 25.5 MB operand at config 3, next to 800 MB of stores) -- and a walk down a strip fetches
    // 32 / W new panels per 32 tiles where a patch fetched 12.  Strip width per job, measured (profiles/r06_strip_sweep.txt;
    // tile kernel ms, patches -> best strip): symmetric 4 000 graphs 0.089 -> 0.069 (W = 4), config 3 0.258 -> 0.222 (8),
    // 20 000 graphs 1.07 -> 0.85 (16), 50 000 graphs 4.61 -> 4.18 (16; W = 4 / 8 lose there: 6.7 / 4.9); row blocks
    // 1 250 x 10 000 0.083 -> 0.063, 6 250 x 50 000 0.645 -> 0.541, 25 000 x 200 000 17.5 -> 12.2 (8; 16: 14.3).
    // Option gram.strip: 1 = the patches, 2 .. 32 = that width.
    int patch_sz = patch ? GI_PATCH : 0;
    if (patch && ctx->opt.gram_strip != 1) {
        int W = 8;
        if (tri) W = tiles_n >= 128 ? 16 : (tiles_n <= 40 ? 4 : 8);
        if (ctx->opt.gram_strip >= 2 && ctx->opt.gram_strip <= 32 && (ctx->opt.gram_strip & (ctx->opt.gram_strip - 1)) == 0) W = ctx->opt.gram_strip;
        patch_sz = GI_STRIP + W;
    }
    const i64 blocks = gram_grid_blocks(tiles_m, tiles_n, tri, patch_sz);
    int k_all = f->k1_steps + f->k8_steps, k8 = f->k8_steps;
    i64 M_store = M;
    int abl_bits = 0;
#ifdef GK_ABLATION
    // timing ablations (WRONG results by construction): only in the tools' build of the library
    // (make -C grakel_amd/csrc abl -> libgk_hip_abl.so, tools/gram_only.py --abl, tools/gram_ablate.sh); the shipped
    // library has none of this.  GK_GRAM_ABL = bit mask, see gram_ws_kernel
    if (const char* abl = getenv("GK_GRAM_ABL")) abl_bits = atoi(abl);
#endif
    const int form = tiles_kernel_form(ctx, f, M, n_cols, tri);
    const bool use_ws = form != 0, use_dd = form == 2;
    *folded = false;
    if (use_dd) {
        const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
        const i64 grid = blocks < n_cu ? blocks : n_cu;
        auto kern = f->phi_fp4 ? gram_dd_kernel<true> : gram_dd_kernel<false>;
        GK_TRY(gk_func_lds(ctx, (const void*)kern, DD_LDS_BYTES));
        kern<<<dim3((unsigned)grid), dim3(DD_THREADS), DD_LDS_BYTES, ctx->stream>>>(
            a, b, f->n_cols_pad, k_all, k8, f->selfk, K, M, n_cols, row_lo, f->symmetric ? 1 : 0, f->n_fit,
            normalize, tiles_m, tiles_n, tri, patch_sz, (int)blocks, ldk, col_lo, even);
    } else if (use_ws) {
        const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
        const i64 grid = blocks < n_cu ? blocks : n_cu;
        unsigned* ticket = nullptr;
        Tmp<unsigned> ticket_buf(ctx);
        if (ctx->opt.gram_xcc) {
            GK_TRY(ticket_buf.alloc(8));
            GK_TRY(gk_zero_async(ctx, ticket_buf.p, 32));
            ticket = ticket_buf.p;
        }
        void (*kern)(const int8_t*, const int8_t*, i64, int, int, const u64*, double*, i64, i64, i64, int, i64, int, int, int,
                     int, int, int, i64, unsigned*, i64, i64, int, const u32*, const uint2*, int, int, const double*) = f->phi_fp4 ? gram_ws_kernel<true, 0> : gram_ws_kernel<false, 0>;
        if (want_fold || normalize) kern = f->phi_fp4 ? gram_ws_kernel<true, 0, true> : gram_ws_kernel<false, 0, true>;
#ifdef GK_ABLATION
#define WS_ABL_CASE(X) if (abl_bits == X) kern = gram_ws_kernel<true, X>;
        WS_ABL_CASE(1) WS_ABL_CASE(2) WS_ABL_CASE(3) WS_ABL_CASE(8) WS_ABL_CASE(9) WS_ABL_CASE(10) WS_ABL_CASE(11) WS_ABL_CASE(13)
        WS_ABL_CASE(24) WS_ABL_CASE(25) WS_ABL_CASE(32) WS_ABL_CASE(43) WS_ABL_CASE(107) WS_ABL_CASE(128) WS_ABL_CASE(136)
        // round 4: the floor of the two memory streams alone -- 18 = no multiply, no K-step barrier: the operand DMA stream (L2 ->
        // LDS) and the float64 tile-pattern store run free inside a tile and meet once per tile; 82 = the same without parking;
        // 16 = everything but the K-step barrier (races: timing only); 26 / 17 = one of the two streams alone, free-running
        WS_ABL_CASE(16) WS_ABL_CASE(18) WS_ABL_CASE(82) WS_ABL_CASE(26) WS_ABL_CASE(19)
#undef WS_ABL_CASE
#endif
        (void)abl_bits;
        GK_TRY(gk_func_lds(ctx, (const void*)kern, WS_LDS_BYTES));
        // the rare labels' pair updates go into the tiles of a full symmetric job (binned on the job's first launch)
        const bool fold = want_fold;
        *folded = fold;
        kern<<<dim3((unsigned)grid), dim3(WS_THREADS), WS_LDS_BYTES, ctx->stream>>>(
            a, b, f->n_cols_pad, k_all, k8, f->selfk, K, M, n_cols, row_lo, f->symmetric ? 1 : 0, f->n_fit,
            normalize, tiles_m, tiles_n, tri, patch_sz, (int)blocks, M_store, ticket, ldk, col_lo, even,
            fold ? f->pair_cnt : nullptr, fold ? f->pairs : nullptr, fold ? f->pair_T : 0, fold ? f->pair_cap : 0,
            normalize ? f->rs : nullptr);
    } else if (f->phi_fp4) {
        auto kern = gram_tile_kernel<true>;
        GK_TRY(gk_func_lds(ctx, (const void*)kern, GT_LDS_BYTES));
        kern<<<dim3((unsigned)blocks), dim3(256), GT_LDS_BYTES, ctx->stream>>>(
            a, b, f->n_cols_pad, k_all, k8, f->selfk, K, M_store, n_cols, row_lo, f->symmetric ? 1 : 0, f->n_fit,
            normalize, tiles_m, tiles_n, tri, patch_sz, ldk, col_lo, even);
    } else {
        auto kern = gram_tile_kernel<false>;
        GK_TRY(gk_func_lds(ctx, (const void*)kern, GT_LDS_BYTES));
        kern<<<dim3((unsigned)blocks), dim3(256), GT_LDS_BYTES, ctx->stream>>>(
            a, b, f->n_cols_pad, k_all, k8, f->selfk, K, M_store, n_cols, row_lo, f->symmetric ? 1 : 0, f->n_fit,
            normalize, tiles_m, tiles_n, tri, patch_sz, ldk, col_lo, even);
    }
    // entries actually multiplied (real rows and columns; the zero padding of edge tiles is not work)
    *entries_done = tri ? (double)M * (M + 1) / 2 : (double)M * n_cols;
    return GK_OK;
}

// ---------------------------------------------------------------------------------------
// float64 path: 64x64 output tile per 256-thread workgroup (2x2 waves, 32x32 per wave as 2x2
// MFMA 16x16x4 tiles), K-step 16.  A[i][k]: lane l holds i = l&15, k = l>>4; D: col = l&15,
// row = (l>>4) + 4*reg.
// ---------------------------------------------------------------------------------------
#define GD_BM 64
#define GD_BN 64
#define GD_BK 16
#define GD_LD 17

// Round 5: a diagonal block of a symmetric job only multiplies the tiles on and above its diagonal and adds the result to
// both halves (tri), and the K loop is split over ksplit workgroups per tile whenever the tiles alone would leave CUs
// idle -- 19 x 19 tiles of the D&D-like ShortestPath job (1 178 graphs, 9.2 ms) were 1.4 workgroups per CU, the second
// round of them running with 60 % of the chip empty.  Split partial sums meet in K through float64 atomics (integers
// below 2^53: exact in any order).  The next operand tile is fetched into registers while the current one multiplies.
__global__ __launch_bounds__(256) void gram_f64_kernel(
    const double* __restrict__ A, const double* __restrict__ B, i64 ld, int k_tiles,
    const u64* __restrict__ selfk, double* __restrict__ K, i64 M, i64 N, i64 row_base,
    int symmetric, i64 n_fit, int normalize, int tiles_n, int accumulate, i64 ldk, i64 col_base, int tri, int ksplit) {
    __shared__ double sA[GD_BM * GD_LD];
    __shared__ double sB[GD_BN * GD_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int t = blockIdx.x / ksplit;
    const int ks = blockIdx.x - t * ksplit;
    int bm, bn;
    if (tri) {                                         // upper triangle, row by row
        bm = 0;
        for (int len = tiles_n; t >= len; --len) t -= len, ++bm;
        bn = bm + t;
    } else
        bm = t / tiles_n, bn = t % tiles_n;
    const int kper = (k_tiles + ksplit - 1) / ksplit;
    const int kt0 = ks * kper, kt1 = kt0 + kper < k_tiles ? kt0 + kper : k_tiles;
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;
    // 64 rows x 16 doubles per operand tile = 1024 elements, 4 per thread
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    const double* gA = A + ((i64)bm * GD_BM + lrow) * ld + lk;
    const double* gB = B + ((i64)bn * GD_BN + lrow) * ld + lk;
    double ra[4], rb[4];
    if (kt0 < kt1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) ra[q] = gA[(i64)kt0 * GD_BK + q], rb[q] = gB[(i64)kt0 * GD_BK + q];
    }
    for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sA[lrow * GD_LD + lk + q] = ra[q];
            sB[lrow * GD_LD + lk + q] = rb[q];
        }
        __syncthreads();
        if (kt + 1 < kt1) {
            const i64 go = (i64)(kt + 1) * GD_BK;
#pragma unroll
            for (int q = 0; q < 4; ++q) ra[q] = gA[go + q], rb[q] = gB[go + q];
        }
#pragma unroll
        for (int ks4 = 0; ks4 < 4; ++ks4) {
            const int kk = ks4 * 4 + (lane >> 4);
            double a0 = sA[(wm * 32 + (lane & 15)) * GD_LD + kk];
            double a1 = sA[(wm * 32 + 16 + (lane & 15)) * GD_LD + kk];
            double b0 = sB[(wn * 32 + (lane & 15)) * GD_LD + kk];
            double b1 = sB[(wn * 32 + 16 + (lane & 15)) * GD_LD + kk];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    const bool shared_cell = ksplit > 1;               // other workgroups add to the same entries
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const i64 col = (i64)bn * GD_BN + wn * 32 + nt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const i64 row = (i64)bm * GD_BM + wm * 32 + mt * 16 + (lane >> 4) + 4 * r;
                if (row < M && col < N) {
                    const double v = acc[mt][nt][r];
                    if (accumulate) {       // K already holds the int8 product (and selfk on the diagonal)
                        if (symmetric && row_base + row == col_base + col) continue;
                        if (shared_cell) unsafeAtomicAdd(&K[row * ldk + col], v);
                        else K[row * ldk + col] += v;
                        if (tri && bm != bn) {             // the mirrored tile (a diagonal tile holds both halves itself)
                            if (shared_cell) unsafeAtomicAdd(&K[col * ldk + row], v);
                            else K[col * ldk + row] += v;
                        }
                    } else {
                        K[row * ldk + col] = finish_entry(v, row_base + row, col_base + col, symmetric != 0, selfk, n_fit, normalize);
                    }
                }
            }
        }
}

// Rare columns (colid == -2): K[g_a][g_b] += c_a * c_b for every ordered pair of graphs that
// share the label.  One wave per rare label run (df < GK_LOW_DF triples): lanes walk the df*df pairs.
// Integer-valued float64 atomics: exact and order independent.
__global__ void gram_low_kernel(const LevelPack P, double* __restrict__ K, i64 ldk, i64 row_lo, i64 row_hi,
                                int symmetric, i64 n_fit, int minsum, i64 col_lo, i64 col_hi) {
    i64 w = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= P.first[P.n]) return;
    int l = 0;
    while (w >= P.first[l + 1]) ++l;          // wave-uniform: which level this rare run belongs to
    w -= P.first[l];
    const i32* __restrict__ tri_pos = P.tri_pos[l];
    const i32* __restrict__ tri_graph = P.tri_graph[l];
    const i32* __restrict__ tstart = P.tstart[l];
    const i32 r = P.low_runs[l][w];
    const i32 t0 = tstart[r];
    const int m = tstart[r + 1] - t0;
    for (int p = lane; p < m * m; p += 64) {
        const int ia = p / m, ib = p - ia * m;
        const i32 a = t0 + ia, b = t0 + ib;
        const i64 ga = tri_graph[a], gb = tri_graph[b];
        const i64 row = symmetric ? ga : ga - n_fit;       // rectangular job: rows are the target graphs
        if (row < row_lo || row >= row_hi || gb < col_lo || gb >= col_hi) continue;
        if (symmetric ? (ia == ib) : (gb >= n_fit)) continue;
        const i32 ca = tri_pos[a + 1] - tri_pos[a], cb = tri_pos[b + 1] - tri_pos[b];
        atomicAdd(&K[(row - row_lo) * ldk + (gb - col_lo)], minsum ? (double)(ca < cb ? ca : cb) : (double)ca * (double)cb);
    }
}

// the same for the graph-major feature builder: a rare label's entries are a (graph, count) list
__global__ void gram_low_gm_kernel(const i32* __restrict__ low_q, i64 n_low, const u32* __restrict__ roff,
                                   const u32* __restrict__ df, const i32* __restrict__ lgraph, const i32* __restrict__ lcnt,
                                   double* __restrict__ K, i64 ldk, i64 row_lo, i64 row_hi, int symmetric, i64 n_fit,
                                   int minsum, i64 col_lo, i64 col_hi) {
    const i64 w = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n_low) return;
    const i32 q = low_q[w];
    const u32 t0 = roff[q];
    const int m = (int)df[q];
    for (int p = lane; p < m * m; p += 64) {
        const int ia = p / m, ib = p - ia * m;
        const i64 ga = lgraph[t0 + ia], gb = lgraph[t0 + ib];
        const i64 row = symmetric ? ga : ga - n_fit;       // rectangular job: rows are the target graphs
        if (row < row_lo || row >= row_hi || gb < col_lo || gb >= col_hi) continue;
        if (symmetric ? (ia == ib) : (gb >= n_fit)) continue;
        const i32 ca = lcnt[t0 + ia], cb = lcnt[t0 + ib];
        atomicAdd(&K[(row - row_lo) * ldk + (gb - col_lo)], minsum ? (double)(ca < cb ? ca : cb) : (double)ca * (double)cb);
    }
}

__global__ void gram_normalize_kernel(double* __restrict__ K, const u64* __restrict__ selfk, i64 M, i64 n_cols,
                                      i64 row_lo, int symmetric, i64 n_fit, int normalize) {
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * n_cols) return;
    const i64 row = idx / n_cols, col = idx - row * n_cols;
    const double dr = (double)selfk[(symmetric ? 0 : n_fit) + row_lo + row];
    const double dc = (double)selfk[col];
    double val = K[idx] / sqrt(dr * dc);
    if (normalize == 2) {
        if (val != val) val = 0.0;
        else if (val > 1.7976931348623157e308) val = 1.7976931348623157e308;
    }
    K[idx] = val;
}

// Block [row_lo,row_hi) x [col_lo,col_hi) of the job's Gram matrix into K (the block's entry (0,0); ldk elements
// between rows).  Every term is included (dense MFMA product, float64 side operand, rare-column pair updates).
// A block on the diagonal of a symmetric job (same row and column range) only multiplies the tiles on/above
// its diagonal and stores both halves.
static int gram_block_launch(gk_ctx* ctx, gk_feat* f, i64 row_lo, i64 row_hi, i64 col_lo, i64 col_hi, int normalize,
                             double* K, i64 ldk, bool accumulate_stats) {
    const i64 M = row_hi - row_lo, NC = col_hi - col_lo;
    if (M <= 0 || NC <= 0) return GK_OK;
    const i64 first_row_graph = (f->symmetric ? 0 : f->n_fit) + row_lo;   // row of Phi
    // the MFMA kernel is bracketed by two events that are only READ in gk_gram_last_stats: the call
    // returns as soon as everything is queued, so the host can already queue the next job
    if (!f->ev0) {
        GK_HIP_CHECK(hipEventCreate(&f->ev0));
        GK_HIP_CHECK(hipEventCreate(&f->ev1));
    }
    hipEvent_t e0 = f->ev0, e1 = f->ev1;
    double entries_done = (double)M * NC;
    const int normalize_req = normalize;
    const bool has_low = f->n_low_cols > 0, has_wide = f->n_cols_wide > 0;
    // the rare labels' pair updates can go INTO the tiles (gram_ws_kernel) when this is the whole symmetric matrix of a
    // graph-major feature job; the epilogue then normalises as well, unless a float64 side operand follows
    const bool want_fold = has_low && f->gm && f->symmetric && row_lo == 0 && col_lo == 0 && M == f->n_graphs && NC == f->n_graphs &&
                           ctx->opt.gram_fold != 2 && !ctx->opt.gram_no_sym && f->rare_entries > 0 && f->k1_steps + f->k8_steps >= 2 &&
                           tiles_kernel_form(ctx, f, M, NC, 1) == 1 && (ctx->opt.gram_fold == 1 || gram_fold_pays(f, normalize_req));
    bool folded = false;
    if ((has_low && !want_fold) || has_wide) normalize = 0;   // normalise after the extra terms instead of in the epilogue
    if (want_fold && f->pair_T == 0) GK_TRY(gram_build_pairs(ctx, f));          // once per feature job
    if (normalize && !f->rs) {
        void* q = nullptr;
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)f->n_graphs * 8));
        f->rs = (double*)q, f->arena.push_back(q);
        gram_rs_kernel<<<dim3((unsigned)cdiv(f->n_graphs, 256)), dim3(256), 0, ctx->stream>>>(f->selfk, f->n_graphs, f->rs);
    }
    GK_HIP_CHECK(hipEventRecord(e0, ctx->stream));
    {
        const int8_t* phi = (const int8_t*)f->phi;
        const int8_t* pa = phi + first_row_graph * f->n_cols_pad;
        const int8_t* pb = (f->phi_r ? (const int8_t*)f->phi_r : phi) + col_lo * f->n_cols_pad;       // split columns: right operand
        const int tri = (f->symmetric && row_lo == col_lo && row_hi == col_hi && !ctx->opt.gram_no_sym) ? 1 : 0;
        const int patch = ctx->opt.gram_no_patch ? 0 : 1;
        GK_TRY(launch_tiles(ctx, f, pa, pb, M, NC, row_lo, normalize, K, tri, patch, &entries_done, ldk, col_lo, want_fold, &folded));
    }
    GK_HIP_CHECK(hipGetLastError());
    GK_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    if (has_wide) {   // float64 side operand: K += Phi_w . Phi_w^T (diagonal excluded: it is selfk)
        const double* pw = f->phi_w;
        const int tiles_m = (int)cdiv(M, GD_BM), tiles_n = (int)cdiv(NC, GD_BN);
        const int k_tiles = (int)(f->n_cols_wide_pad / GD_BK);
        const int tri64 = (f->symmetric && row_lo == col_lo && row_hi == col_hi && !ctx->opt.gram_no_sym) ? 1 : 0;
        const i64 tiles = tri64 ? (i64)tiles_n * (tiles_n + 1) / 2 : (i64)tiles_m * tiles_n;
        // four workgroups per CU hide one another's operand fetches; a slice of the K loop keeps at least 32 steps
        const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
        int ksplit = (int)std::min<i64>(cdiv(4 * (i64)n_cu, tiles), std::max(1, k_tiles / 32));
        if (ksplit < 1 || ctx->opt.gram_no_split64) ksplit = 1;
        gram_f64_kernel<<<dim3((unsigned)(tiles * ksplit)), dim3(256), 0, ctx->stream>>>(
            pw + first_row_graph * f->n_cols_wide_pad, pw + col_lo * f->n_cols_wide_pad, f->n_cols_wide_pad,
            k_tiles, f->selfk, K, M, NC, row_lo, f->symmetric ? 1 : 0, f->n_fit, 0, tiles_n, 1,
            ldk, col_lo, tri64, ksplit);
    }
    if (has_low && folded) {
        // applied inside the tile kernel; what did not fit its tile's bucket (usually nothing) follows here
        gram_pairs_overflow_kernel<<<dim3(64), dim3(256), 0, ctx->stream>>>(f->pair_ovf_n, f->pair_ovf, (u32)std::min<i64>(f->pair_ovf_cap, 0xffffffffll),
                                                                          K, ldk, f->selfk, has_wide ? 0 : normalize_req);
    } else if (has_low && f->gm) {
        gram_low_gm_kernel<<<dim3((unsigned)cdiv(f->n_low_cols * 64, 256)), dim3(256), 0, ctx->stream>>>(
            f->gm_low_q, f->n_low_cols, f->gm_roff, f->gm_df, f->gm_low_graph, f->gm_low_cnt, K, ldk, row_lo, row_hi,
            f->symmetric ? 1 : 0, f->n_fit, f->kind == GK_FEAT_MINSUM ? 1 : 0, col_lo, col_hi);
    } else if (has_low) {
        for (int l0 = 0; l0 < f->n_levels; l0 += GK_PACK_LEVELS) {     // one launch per 16 levels
            LevelPack P = {};
            P.n = 0, P.first[0] = 0;
            for (int l = l0; l < f->n_levels && l < l0 + GK_PACK_LEVELS; ++l) {
                LevelTriples& L = f->lev[l];
                if (!L.tri_pos || L.n_low == 0) continue;
                P.tri_pos[P.n] = L.tri_pos, P.tri_graph[P.n] = L.tri_graph, P.tstart[P.n] = L.tstart;
                P.low_runs[P.n] = L.low_runs;
                P.first[P.n + 1] = P.first[P.n] + L.n_low;
                ++P.n;
            }
            if (P.n == 0) continue;
            gram_low_kernel<<<dim3((unsigned)cdiv(P.first[P.n] * 64, 256)), dim3(256), 0, ctx->stream>>>(
                P, K, ldk, row_lo, row_hi, f->symmetric ? 1 : 0, f->n_fit, f->kind == GK_FEAT_MINSUM ? 1 : 0, col_lo, col_hi);
        }
    }
    if ((has_low && !folded) || has_wide) {
        if (normalize_req) {
            GK_ARG(col_lo == 0 && ldk == NC, "gram: normalisation of a column block is applied by gk_gram_normalize_rows");
            gram_normalize_kernel<<<dim3((unsigned)cdiv(M * NC, 256)), dim3(256), 0, ctx->stream>>>(
                K, f->selfk, M, NC, row_lo, f->symmetric ? 1 : 0, f->n_fit, normalize_req);
        }
        GK_HIP_CHECK(hipGetLastError());
    }
    f->last_ms = -1.0;      // not read yet
    // work actually executed: a diagonal block only runs the tiles on/above its diagonal
    const double fl = 2.0 * entries_done * (double)f->n_cols;     // columns actually holding a label (padding excluded)
    f->last_flops = accumulate_stats ? f->last_flops + fl : fl;
    return GK_OK;
}

// rows [row_lo,row_hi) of the job's Gram matrix into K ([row_hi-row_lo] x n_cols, row major)
int gk_gram_launch(gk_ctx* ctx, gk_feat* f, i64 row_lo, i64 row_hi, int normalize, double* K) {
    const i64 n_cols = f->symmetric ? f->n_graphs : f->n_fit;
    return gram_block_launch(ctx, f, row_lo, row_hi, 0, n_cols, normalize, K, n_cols, false);
}

// ---- block-wise entry points of the multi-GPU path (grakel_amd/dist.py): a rank computes some blocks of
// its row block, ships them to the ranks owning the mirrored blocks, and places what it receives transposed
extern "C" int gk_gram_block(gk_ctx* ctx, gk_feat* f, int64_t row_lo, int64_t row_hi, int64_t col_lo, int64_t col_hi,
                             double* out_dev, int64_t ld) {
    GK_ARG(ctx && f && out_dev, "gk_gram_block: null argument");
    const i64 n_rows = f->symmetric ? f->n_graphs : f->n_graphs - f->n_fit;
    const i64 n_cols = f->symmetric ? f->n_graphs : f->n_fit;
    GK_ARG(row_lo >= 0 && row_hi <= n_rows && row_lo <= row_hi && col_lo >= 0 && col_hi <= n_cols && col_lo <= col_hi,
           "gk_gram_block: bad block");
    GK_ARG(ld >= col_hi - col_lo, "gk_gram_block: leading dimension smaller than the block");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ProfScope prof(ctx, "gram");
    return gram_block_launch(ctx, f, row_lo, row_hi, col_lo, col_hi, 0, out_dev, ld, true);
}

extern "C" int gk_gram_reset_stats(gk_feat* f) {
    GK_ARG(f, "gk_gram_reset_stats: null");
    f->last_flops = 0.0;
    return GK_OK;
}

// dst[i][j] = src[i][j] (transpose == 0, rows x cols) or dst[j][i] = src[i][j] (transpose != 0): 32x32 tiles through LDS
__global__ __launch_bounds__(256) void block_copy_kernel(const double* __restrict__ src, i64 rows, i64 cols, i64 ld_src,
                                                         double* __restrict__ dst, i64 ld_dst, int transpose) {
    __shared__ double t[32][33];
    const i64 tiles_c = (cols + 31) / 32;
    const i64 ti = blockIdx.x / tiles_c, tj = blockIdx.x % tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const i64 i = ti * 32 + r, j = tj * 32 + tx;
        if (i < rows && j < cols) t[r][tx] = src[i * ld_src + j];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        if (!transpose) {
            const i64 i = ti * 32 + r, j = tj * 32 + tx;
            if (i < rows && j < cols) dst[i * ld_dst + j] = t[r][tx];
        } else {
            const i64 j = tj * 32 + r, i = ti * 32 + tx;       // dst row j, dst column i
            if (i < rows && j < cols) dst[j * ld_dst + i] = t[tx][r];
        }
    }
}

extern "C" int gk_block_copy(gk_ctx* ctx, const double* src_dev, int64_t rows, int64_t cols, int64_t ld_src,
                             double* dst_dev, int64_t ld_dst, int transpose) {
    GK_ARG(ctx && src_dev && dst_dev && rows >= 0 && cols >= 0, "gk_block_copy: bad argument");
    if (rows == 0 || cols == 0) return GK_OK;
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    const i64 blocks = ((rows + 31) / 32) * ((cols + 31) / 32);
    GK_ARG(blocks < (1ll << 31), "gk_block_copy: block too large");
    block_copy_kernel<<<dim3((unsigned)blocks), 256, 0, ctx->stream>>>(src_dev, rows, cols, ld_src, dst_dev, ld_dst, transpose);
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

// K[row_hi-row_lo x n_cols] (device, leading dimension n_cols) /= sqrt(selfk_row selfk_col); mode as gk_gram
extern "C" int gk_gram_normalize_rows(gk_ctx* ctx, gk_feat* f, int64_t row_lo, int64_t row_hi, double* K_dev, int mode) {
    GK_ARG(ctx && f && K_dev && (mode == 1 || mode == 2), "gk_gram_normalize_rows: bad argument");
    const i64 n_cols = f->symmetric ? f->n_graphs : f->n_fit;
    const i64 M = row_hi - row_lo;
    if (M <= 0) return GK_OK;
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    gram_normalize_kernel<<<dim3((unsigned)cdiv(M * n_cols, 256)), dim3(256), 0, ctx->stream>>>(
        K_dev, f->selfk, M, n_cols, row_lo, f->symmetric ? 1 : 0, f->n_fit, mode);
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

// ---------------------------------------------------------------------------------------
// Device -> host copy of an integer-valued Gram matrix in a narrow type.  The float64 matrix is 8 N^2 bytes and the
// copy runs at the PCIe rate (57 GB/s into pinned memory: 14 ms of the 15.7 ms a caller waits for 10 000 graphs).
// Every entry is a non-negative integer below the job's bound (features.hip: k_bound), so below 2^16 the matrix travels as
// uint16 (a quarter of the bytes), below 2^31 as int32 (half): one kernel narrows K into a device buffer, the buffer
// crosses in 8-MiB chunks through a ring of four pinned staging blocks, and host threads widen chunk c into the caller's
// float64 array (non-temporal stores) while chunk c + 1 is on the bus.  Same values, bit for bit: integers convert exactly.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void gram_narrow_kernel(const double* __restrict__ K, i64 n, T* __restrict__ out) {
    const i64 i = ((i64)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const double2 a = *(const double2*)(K + i), b = *(const double2*)(K + i + 2);
        out[i] = (T)a.x, out[i + 1] = (T)a.y, out[i + 2] = (T)b.x, out[i + 3] = (T)b.y;
    } else {
        for (i64 j = i; j < n; ++j) out[j] = (T)K[j];
    }
}

static int host_cpu_budget();
template <typename T>
static void widen_slice(const T* __restrict__ src, double* __restrict__ dst, size_t n) {
    size_t i = 0;
    for (; i < n && ((uintptr_t)(dst + i) & 15); ++i) dst[i] = (double)src[i];
    for (; i + 2 <= n; i += 2) _mm_stream_pd(dst + i, _mm_set_pd((double)src[i + 1], (double)src[i]));
    for (; i < n; ++i) dst[i] = (double)src[i];
}

#define GC_CHUNK ((size_t)8 << 20)      // bytes of narrow data per chunk
#define GC_SLOTS 4

template <typename T>
static int gram_copy_out_narrow(gk_ctx* ctx, const double* K_dev, i64 n_entries, double* out_host) {
    if (!ctx->stage_host) {
        void* h = nullptr;
        GK_HIP_CHECK(hipHostMalloc(&h, GC_CHUNK * GC_SLOTS, hipHostMallocDefault));
        ctx->stage_host = h;
        for (int i = 0; i < GC_SLOTS; ++i) GK_HIP_CHECK(hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming));
    }
    const auto t_narrow0 = std::chrono::steady_clock::now();
    Tmp<T> narrow(ctx);
    GK_TRY(narrow.alloc((size_t)n_entries));
    gram_narrow_kernel<T><<<dim3((unsigned)cdiv(cdiv(n_entries, 4), 256)), dim3(256), 0, ctx->stream>>>(K_dev, n_entries, narrow.p);
    GK_HIP_CHECK(hipGetLastError());
    const size_t per_chunk = GC_CHUNK / sizeof(T);
    const int n_chunks = (int)cdiv(n_entries, (i64)per_chunk);
    int n_thr = ctx->opt.gram_copy_threads > 0 ? ctx->opt.gram_copy_threads : host_cpu_budget();
    if (ctx->opt.gram_copy_threads <= 0 && n_thr > 16) n_thr = 16;      // measured: 8 threads keep up with the bus, 48 are slower
    if (n_thr < 1) n_thr = 1;
    std::atomic<int> ready(0), stop(0);
    std::vector<std::atomic<int>> done((size_t)n_chunks);
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    const T* stage = (const T*)ctx->stage_host;
    auto chunk_len = [&](int c) { return (size_t)std::min<i64>((i64)per_chunk, n_entries - (i64)c * (i64)per_chunk); };
    auto worker = [&](int w) {
        for (int c = 0; c < n_chunks; ++c) {
            for (unsigned spins = 0; ready.load(std::memory_order_acquire) <= c; ++spins) {
                if (stop.load(std::memory_order_relaxed)) return;
                if (spins < 4096) _mm_pause(); else std::this_thread::yield();
            }
            const size_t len = chunk_len(c);
            const size_t lo = len * (size_t)w / (size_t)n_thr, hi = len * (size_t)(w + 1) / (size_t)n_thr;
            widen_slice(stage + (size_t)(c % GC_SLOTS) * per_chunk + lo, out_host + (size_t)c * per_chunk + lo, hi - lo);
            _mm_sfence();
            done[(size_t)c].fetch_add(1, std::memory_order_release);
        }
    };
    // (the workers read n_thr only after `ready` has been raised: it may still shrink here when thread creation fails --
    // container pid / ulimit limits --, which must not leave the extern "C" entry point as an exception)
    std::vector<std::thread> pool;
    pool.reserve((size_t)n_thr);
    try {
        for (int w = 0; w < n_thr; ++w) pool.emplace_back(worker, w);
    } catch (...) {}
    const bool inline_widen = pool.empty();
    n_thr = inline_widen ? 1 : (int)pool.size();
    int rc = GK_OK;
    auto queue_chunk = [&](int c) -> bool {
        const int slot = c % GC_SLOTS;
        return hipMemcpyAsync((char*)ctx->stage_host + (size_t)slot * GC_CHUNK, narrow.p + (size_t)c * per_chunk,
                              chunk_len(c) * sizeof(T), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
               hipEventRecord(ctx->stage_ev[slot], ctx->stream) == hipSuccess;
    };
    for (int c = 0; c < n_chunks && c < GC_SLOTS; ++c)
        if (!queue_chunk(c)) { rc = GK_ERR_HIP; break; }
    for (int c = 0; c < n_chunks && rc == GK_OK; ++c) {
        if (hipEventSynchronize(ctx->stage_ev[c % GC_SLOTS]) != hipSuccess) { rc = GK_ERR_HIP; break; }
        ready.store(c + 1, std::memory_order_release);
        if (inline_widen) {
            widen_slice(stage + (size_t)(c % GC_SLOTS) * per_chunk, out_host + (size_t)c * per_chunk, chunk_len(c));
            done[(size_t)c].store(1, std::memory_order_release);
        }
        if (c + GC_SLOTS < n_chunks) {          // the slot is refilled once every thread has widened its slice of chunk c
            for (unsigned spins = 0; done[(size_t)c].load(std::memory_order_acquire) < n_thr; ++spins)
                if (spins < 4096) _mm_pause(); else std::this_thread::yield();
            if (!queue_chunk(c + GC_SLOTS)) { rc = GK_ERR_HIP; break; }
        }
    }
    if (rc != GK_OK) stop.store(1);
    for (auto& t : pool) t.join();
    {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_narrow0).count();
        const double st[8] = {2.0, (double)n_thr, (double)n_entries * sizeof(T), ms, ms, 0.0, 0.0, (double)n_chunks};
        memcpy(ctx->copy_stats, st, sizeof st);
    }
    if (rc != GK_OK) {
        (void)hipGetLastError();
        gk_set_error("gk_gram: the compact device-to-host copy failed");
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

// ---------------------------------------------------------------------------------------
// Triangle form of the compact copy (round 5): the WHOLE symmetric matrix of a fit_transform.  Only the 256 x 256 blocks
// on and above the diagonal cross PCIe (half the bytes of the rectangular form: config 3 105 MB instead of 200), block-major
// so that a host thread owns a whole block: it widens the block into its own rows of the caller's matrix and -- off the
// diagonal -- writes the mirrored block from the same 128 KiB (L2-resident) staging data, rows of 256 float64 with
// non-temporal stores.  NORMALISED jobs take the same road: the device matrix stays the exact integer one (the lean,
// unnormalised tile kernel), the 8 N-byte self-similarity vector comes along, and the widening threads multiply every
// entry by rs[i] * rs[j], rs = 1 / sqrt(self similarity) -- the product is formed first, so the result is exactly
// symmetric; the diagonal is exactly 1 (a graph without features: NaN, or 0 under nan_to_num, as the reference's 0/0).
// Normalised host matrices then cost what unnormalised ones do (before: the FULL tile kernel + a plain 8 N^2-byte copy).
// ---------------------------------------------------------------------------------------
#define GC_TB 256
template <typename T>
__global__ __launch_bounds__(256) void gram_pack_tri_kernel(const double* __restrict__ K, i64 N, int nb, T* __restrict__ out) {
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const i64 p = (i64)bi * nb - (i64)bi * (bi - 1) / 2 + (bj - bi);
    T* __restrict__ o = out + p * (i64)(GC_TB * GC_TB);
    const i64 col = (i64)bj * GC_TB + threadIdx.x;
    const bool col_ok = col < N;
    const i64 r0 = (i64)bi * GC_TB;
#pragma unroll 4
    for (int r = 0; r < GC_TB; ++r) {
        const i64 row = r0 + r;
        const double v = (col_ok && row < N) ? __builtin_nontemporal_load(K + row * N + col) : 0.0;
        o[(i64)r * GC_TB + threadIdx.x] = (T)v;
    }
}

// ---- host side of the triangle form.  The widening is CPU work (2-4 cycles per entry in plain SSE2: 16 threads took 5.6 ms
// for the 10^8 entries of config 3 while the 105 MB crossed PCIe in 1.9 ms -- tools/micro/hostwrite.hip: the 800 MB can be
// written in 2.7 ms), so: AVX2 rows (4 entries per convert, 32-byte non-temporal stores) when the CPU has it, the mirrored
// block through eight column buffers (every staging cache line read once per eight columns), blocks handed out one by one
// from an atomic counter (a diagonal block costs half a mirrored one), and a thread pool that outlives the call.
// How many host threads may run at once: cpu_budget.h (shared with ingest.c) -- hardware threads, affinity mask, cgroup quota.
static int host_cpu_budget() { return gk_cpu_budget(); }

struct GkHostPool {
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::vector<std::thread> threads;
    std::function<void(int)> fn;
    u64 gen = 0;
    int active = 0, finished = 0;
    bool quit = false;
    // returns how many threads exist afterwards: std::thread's constructor throws std::system_error when the container's
    // pid / ulimit limits are reached -- that must not cross the extern "C" boundary; the callers work
    // with however many threads there are (the blocks are handed out from a counter), down to none (inline widening)
    int ensure(int n) {
        while ((int)threads.size() < n) {
            const int idx = (int)threads.size();
            try {
            threads.emplace_back([this, idx] {
                u64 seen = 0;
                for (;;) {
                    std::function<void(int)> f;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv_go.wait(lk, [&] { return quit || (gen != seen && idx < active); });
                        if (quit) return;
                        seen = gen;
                        f = fn;
                    }
                    f(idx);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        ++finished;
                    }
                    cv_done.notify_all();
                }
            });
            } catch (...) { break; }
        }
        return (int)threads.size();
    }
    // wakes min(n, threads that exist) workers; returns that number
    int start(int n, std::function<void(int)> f) {
        const int have = ensure(n);
        if (have < n) n = have;
        {
            std::lock_guard<std::mutex> lk(mu);
            fn = std::move(f), active = n, finished = 0, ++gen;
        }
        if (n > 0) cv_go.notify_all();
        return n;
    }
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return finished >= active; });
        active = 0;
    }
    ~GkHostPool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_go.notify_all();
        for (auto& t : threads) t.join();
    }
};
void gk_host_pool_destroy(GkHostPool* p) { delete p; }

// one row: dst[c] = (double)src[c] (* fr * rs[c] when rs); SSE2 baseline
template <typename T>
static void widen_row_sse2(const T* __restrict__ s, double* __restrict__ d, int n, const double* __restrict__ rs, double fr) {
    int c = 0;
    if (rs) {
        for (; c < n && ((uintptr_t)(d + c) & 15); ++c) d[c] = (double)s[c] * (fr * rs[c]);
        for (; c + 2 <= n; c += 2)
            _mm_stream_pd(d + c, _mm_mul_pd(_mm_set_pd((double)s[c + 1], (double)s[c]), _mm_mul_pd(_mm_set1_pd(fr), _mm_loadu_pd(rs + c))));
        for (; c < n; ++c) d[c] = (double)s[c] * (fr * rs[c]);
    } else {
        for (; c < n && ((uintptr_t)(d + c) & 15); ++c) d[c] = (double)s[c];
        for (; c + 2 <= n; c += 2) _mm_stream_pd(d + c, _mm_set_pd((double)s[c + 1], (double)s[c]));
        for (; c < n; ++c) d[c] = (double)s[c];
    }
}
__attribute__((target("avx2"))) static inline __m256d cvt4_avx2(const uint16_t* s) {
    return _mm256_cvtepi32_pd(_mm_cvtepu16_epi32(_mm_loadl_epi64((const __m128i*)s)));
}
__attribute__((target("avx2"))) static inline __m256d cvt4_avx2(const int32_t* s) {
    return _mm256_cvtepi32_pd(_mm_loadu_si128((const __m128i*)s));
}
template <typename T>
__attribute__((target("avx2"))) static void widen_row_avx2(const T* __restrict__ s, double* __restrict__ d, int n,
                                                           const double* __restrict__ rs, double fr) {
    int c = 0;
    if (rs) {
        const __m256d f = _mm256_set1_pd(fr);
        for (; c < n && ((uintptr_t)(d + c) & 31); ++c) d[c] = (double)s[c] * (fr * rs[c]);
        for (; c + 4 <= n; c += 4) _mm256_stream_pd(d + c, _mm256_mul_pd(cvt4_avx2(s + c), _mm256_mul_pd(f, _mm256_loadu_pd(rs + c))));
        for (; c < n; ++c) d[c] = (double)s[c] * (fr * rs[c]);
    } else {
        for (; c < n && ((uintptr_t)(d + c) & 31); ++c) d[c] = (double)s[c];
        for (; c + 4 <= n; c += 4) _mm256_stream_pd(d + c, cvt4_avx2(s + c));
        for (; c < n; ++c) d[c] = (double)s[c];
    }
}

// one block of the triangle: staging data `src` [256][256] -> rows [r0, r0 + nr) x columns [c0, c0 + nc) of `out` and, when
// mirror, rows [c0, c0 + nc) x columns [r0, r0 + nr).  rs == nullptr: plain widening
template <typename T>
static void widen_tri_block(const T* __restrict__ src, double* __restrict__ out, i64 N, i64 r0, int nr, i64 c0, int nc, bool mirror,
                            const double* __restrict__ rs, const double* __restrict__ diag_val, bool avx2) {
    for (int r = 0; r < nr; ++r) {
        double* d = out + (size_t)(r0 + r) * (size_t)N + (size_t)c0;
        const double fr = rs ? rs[r0 + r] : 1.0;
        if (avx2) widen_row_avx2<T>(src + (size_t)r * GC_TB, d, nc, rs ? rs + c0 : nullptr, fr);
        else widen_row_sse2<T>(src + (size_t)r * GC_TB, d, nc, rs ? rs + c0 : nullptr, fr);
        if (rs && !mirror) d[r] = diag_val[r0 + r];                 // a diagonal block: entry (r, r)
    }
    if (!mirror) return;
    alignas(64) T cb[8][GC_TB];
    for (int cg = 0; cg < nc; cg += 8) {
        const int w = nc - cg < 8 ? nc - cg : 8;
        for (int r = 0; r < nr; ++r) {
            const T* __restrict__ s = src + (size_t)r * GC_TB + cg;
            for (int k = 0; k < 8; ++k) cb[k][r] = s[k];            // (the staging block is padded to 256 x 256: k < 8 is in range)
        }
        for (int k = 0; k < w; ++k) {
            double* d = out + (size_t)(c0 + cg + k) * (size_t)N + (size_t)r0;
            const double fc = rs ? rs[c0 + cg + k] : 1.0;
            if (avx2) widen_row_avx2<T>(cb[k], d, nr, rs ? rs + r0 : nullptr, fc);
            else widen_row_sse2<T>(cb[k], d, nr, rs ? rs + r0 : nullptr, fc);
        }
    }
    _mm_sfence();
}

template <typename T>
static int gram_copy_out_tri(gk_ctx* ctx, gk_feat* f, const double* K_dev, i64 N, int normalize, double* out_host) {
    if (!ctx->stage_host) {
        void* h = nullptr;
        GK_HIP_CHECK(hipHostMalloc(&h, GC_CHUNK * GC_SLOTS, hipHostMallocDefault));
        ctx->stage_host = h;
        for (int i = 0; i < GC_SLOTS; ++i) GK_HIP_CHECK(hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming));
    }
    const int nb = (int)cdiv(N, GC_TB);
    const i64 n_blocks = (i64)nb * (nb + 1) / 2;
    const size_t block_elems = (size_t)GC_TB * GC_TB;
    Tmp<T> packed(ctx);
    GK_TRY(packed.alloc((size_t)n_blocks * block_elems));
    gram_pack_tri_kernel<T><<<dim3((unsigned)nb, (unsigned)nb), dim3(256), 0, ctx->stream>>>(K_dev, N, nb, packed.p);
    GK_HIP_CHECK(hipGetLastError());
    // normalised: the factors from the exact self similarities (they come back with the first chunk's synchronisation)
    std::vector<double> rs, dv;
    std::vector<u64> sk;
    if (normalize) {
        sk.resize((size_t)N);
        GK_HIP_CHECK(hipMemcpyAsync(sk.data(), f->selfk, (size_t)N * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    const int per_chunk = (int)(GC_CHUNK / (block_elems * sizeof(T)));            // blocks per chunk: 64 (uint16) / 32 (int32)
    const int n_chunks = (int)cdiv(n_blocks, (i64)per_chunk);
    int n_thr = ctx->opt.gram_copy_threads > 0 ? ctx->opt.gram_copy_threads : host_cpu_budget();
    if (ctx->opt.gram_copy_threads <= 0 && n_thr > 32) n_thr = 32;      // measured (tools/micro/hostwrite.hip): 32-64 threads write at the memory rate
    if (n_thr < 1) n_thr = 1;
    if ((i64)n_thr > n_blocks) n_thr = (int)n_blocks;
    const bool avx2 = __builtin_cpu_supports("avx2") && !ctx->opt.gram_no_avx2;
    // block index -> (bi, bj): first block of every block row
    std::vector<i64> row_first((size_t)nb + 1);
    for (int bi = 0; bi <= nb; ++bi) row_first[(size_t)bi] = (i64)bi * nb - (i64)bi * (bi - 1) / 2;
    std::atomic<int> ready(0), stop(0);
    std::atomic<i64> next_block(0);
    std::vector<std::atomic<int>> done((size_t)n_chunks);              // blocks of the chunk that have been widened
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    const T* stage = (const T*)ctx->stage_host;
    const double* rsp = nullptr;
    const double* dvp = nullptr;
    auto chunk_blocks = [&](int c) { return (int)std::min<i64>((i64)per_chunk, n_blocks - (i64)c * per_chunk); };
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    std::vector<double> busy((size_t)n_thr + 1, 0.0);                  // per widening thread (last slot: the calling thread, inline mode)
    auto widen_one = [&](i64 p) {
            const int c = (int)(p / per_chunk), k = (int)(p - (i64)c * per_chunk);
            int bi = (int)(std::upper_bound(row_first.begin(), row_first.end(), p) - row_first.begin()) - 1;
            const int bj = bi + (int)(p - row_first[(size_t)bi]);
            const i64 r0 = (i64)bi * GC_TB, c0 = (i64)bj * GC_TB;
            widen_tri_block<T>(stage + ((size_t)(c % GC_SLOTS) * per_chunk + (size_t)k) * block_elems, out_host, N, r0,
                               (int)std::min<i64>(GC_TB, N - r0), c0, (int)std::min<i64>(GC_TB, N - c0), bj != bi, rsp, dvp, avx2);
            _mm_sfence();
            done[(size_t)c].fetch_add(1, std::memory_order_release);
    };
    auto worker = [&](int w) {
        for (;;) {
            const i64 p = next_block.fetch_add(1, std::memory_order_relaxed);
            if (p >= n_blocks) return;
            const int c = (int)(p / per_chunk);
            for (unsigned spins = 0; ready.load(std::memory_order_acquire) <= c; ++spins) {
                if (stop.load(std::memory_order_relaxed)) return;
                if (spins < 4096) _mm_pause(); else std::this_thread::yield();
            }
            const auto t = std::chrono::steady_clock::now();
            widen_one(p);
            if (w >= 0 && w < n_thr) busy[(size_t)w] += ms_since(t);
        }
    };
    int rc = GK_OK;
    double landed_ms = 0.0;
    int n_workers = 0;                      // pool threads at work; 0 = none could be created: the caller widens inline
    auto queue_chunk = [&](int c) -> bool {
        const int slot = c % GC_SLOTS;
        return hipMemcpyAsync((char*)ctx->stage_host + (size_t)slot * GC_CHUNK, packed.p + (size_t)c * per_chunk * block_elems,
                              (size_t)chunk_blocks(c) * block_elems * sizeof(T), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
               hipEventRecord(ctx->stage_ev[slot], ctx->stream) == hipSuccess;
    };
    for (int c = 0; c < n_chunks && c < GC_SLOTS; ++c)
        if (!queue_chunk(c)) { rc = GK_ERR_HIP; break; }
    if (!ctx->host_pool) ctx->host_pool = new GkHostPool();
    bool started = false;
    for (int c = 0; c < n_chunks && rc == GK_OK; ++c) {
        if (hipEventSynchronize(ctx->stage_ev[c % GC_SLOTS]) != hipSuccess) { rc = GK_ERR_HIP; break; }
        if (c == 0) {          // the self similarities are on the host now (same stream, queued before chunk 0)
            if (normalize) {
                rs.resize((size_t)N), dv.resize((size_t)N);
                const double nan = std::numeric_limits<double>::quiet_NaN();
                for (i64 i = 0; i < N; ++i) {
                    const u64 s = sk[(size_t)i];
                    // a graph without features: the reference's 0 / 0 (NaN; 0 after numpy.nan_to_num, mode 2)
                    rs[(size_t)i] = s ? 1.0 / std::sqrt((double)s) : (normalize == 2 ? 0.0 : nan);
                    dv[(size_t)i] = s ? 1.0 : (normalize == 2 ? 0.0 : nan);
                }
                rsp = rs.data(), dvp = dv.data();
            }
            n_workers = ctx->host_pool->start(n_thr, worker);
            started = n_workers > 0;
        }
        ready.store(c + 1, std::memory_order_release);
        if (n_workers == 0) {               // no host thread to be had: this chunk's blocks on the calling thread
            const i64 hi = std::min<i64>(n_blocks, (i64)(c + 1) * per_chunk);
            const auto t = std::chrono::steady_clock::now();
            for (i64 p = (i64)c * per_chunk; p < hi; ++p) widen_one(p);
            busy[(size_t)n_thr] += ms_since(t);
        }
        if (c + 1 == n_chunks) landed_ms = ms_since(t_start);
        if (c + GC_SLOTS < n_chunks) {          // the slot is refilled once every block of chunk c is done
            const int need = chunk_blocks(c);
            for (unsigned spins = 0; done[(size_t)c].load(std::memory_order_acquire) < need; ++spins)
                if (spins < 4096) _mm_pause(); else std::this_thread::yield();
            if (!queue_chunk(c + GC_SLOTS)) { rc = GK_ERR_HIP; break; }
        }
    }
    if (rc != GK_OK) stop.store(1);
    if (started) ctx->host_pool->wait();
    {
        double sum = 0.0, mx = 0.0;
        const int used = n_workers > 0 ? n_workers : 1;
        for (double b : busy) sum += b, mx = std::max(mx, b);
        const double st[8] = {1.0, (double)used, (double)n_blocks * (double)block_elems * sizeof(T), landed_ms, ms_since(t_start),
                              sum / used, mx, (double)n_chunks};
        memcpy(ctx->copy_stats, st, sizeof st);
    }
    if (rc != GK_OK) {
        (void)hipGetLastError();
        gk_set_error("gk_gram: the compact device-to-host copy failed");
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

// does a host-output job of these rows take the triangle form?  (gk_gram_rows then leaves a NORMALISED job's device
// matrix unnormalised: the factors are applied by the host threads)
static bool gram_copy_is_tri(const gk_ctx* ctx, const gk_feat* f, i64 row_lo, i64 M) {
    const i64 N = f->n_graphs;
    return f->symmetric && row_lo == 0 && M == N && !ctx->opt.gram_no_compact && !ctx->opt.gram_no_tri && N >= 2048 &&
           f->k_bound > 0.0 && f->k_bound < 2147483647.0;
}

// K_dev [n_entries] float64 on the device -> out_host; integer-valued matrices below the job's bound go compact
static int gram_copy_out(gk_ctx* ctx, gk_feat* f, const double* K_dev, i64 n_entries, int normalize, double* out_host, bool tri) {
    if (tri) {
        if (f->k_bound < 65536.0) return gram_copy_out_tri<uint16_t>(ctx, f, K_dev, f->n_graphs, normalize, out_host);
        return gram_copy_out_tri<int32_t>(ctx, f, K_dev, f->n_graphs, normalize, out_host);
    }
    const bool compact = normalize == 0 && !ctx->opt.gram_no_compact && n_entries >= ((i64)4 << 20) && f->k_bound > 0.0 &&
                         f->k_bound < 2147483647.0;
    if (compact && f->k_bound < 65536.0) return gram_copy_out_narrow<uint16_t>(ctx, K_dev, n_entries, out_host);
    if (compact) return gram_copy_out_narrow<int32_t>(ctx, K_dev, n_entries, out_host);
    const auto t0 = std::chrono::steady_clock::now();
    GK_HIP_CHECK(hipMemcpyAsync(out_host, K_dev, (size_t)n_entries * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const double st[8] = {0.0, 0.0, (double)n_entries * 8.0, ms, ms, 0.0, 0.0, 1.0};
    memcpy(ctx->copy_stats, st, sizeof st);
    return GK_OK;
}

extern "C" int gk_host_copy_stats(gk_ctx* ctx, int* out_cpu, double* out_copy) {
    GK_ARG(ctx, "gk_host_copy_stats: null argument");
    if (out_cpu) {
        gk_cpu_info_t o;
        gk_cpu_info(&o);
        out_cpu[0] = o.online, out_cpu[1] = o.affinity, out_cpu[2] = o.quota_cpus, out_cpu[3] = o.budget;
    }
    if (out_copy) memcpy(out_copy, ctx->copy_stats, sizeof ctx->copy_stats);
    return GK_OK;
}

extern "C" int gk_gram_rows(gk_ctx* ctx, gk_feat* f, int64_t row_lo, int64_t row_hi, int normalize,
                            double* out_host) {
    GK_ARG(ctx && f, "gk_gram: null argument");
    const i64 n_rows = f->symmetric ? f->n_graphs : f->n_graphs - f->n_fit;
    const i64 n_cols = f->symmetric ? f->n_graphs : f->n_fit;
    GK_ARG(row_lo >= 0 && row_hi <= n_rows && row_lo <= row_hi, "gk_gram: bad row range");
    GK_ARG(normalize >= 0 && normalize <= 2, "gk_gram: bad normalize");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ProfScope prof(ctx, "gram");
    const i64 M = row_hi - row_lo;
    if (f->K) { gk_dev_free(ctx, f->K); f->K = nullptr; }
    void* q = nullptr;
    GK_TRY(gk_dev_alloc(ctx, &q, (size_t)(M > 0 ? M : 1) * n_cols * 8));
    f->K = (double*)q, f->K_rows = M, f->K_cols = n_cols;
    // a whole symmetric matrix bound for the host: the triangle form of the compact copy, which also normalises -- the
    // device matrix of such a job stays the exact integer one (gk_gram_checksum / gk_gram_dev_ptr then see THAT matrix)
    const bool tri = out_host && M > 0 && gram_copy_is_tri(ctx, f, row_lo, M);
    GK_TRY(gk_gram_launch(ctx, f, row_lo, row_hi, tri ? 0 : normalize, f->K));
    if (out_host && M > 0) GK_TRY(gram_copy_out(ctx, f, f->K, M * n_cols, normalize, out_host, tri));
    return GK_OK;
}

extern "C" int gk_gram(gk_ctx* ctx, gk_feat* f, int normalize, double* out_host) {
    GK_ARG(f, "gk_gram: null features");
    const i64 n_rows = f->symmetric ? f->n_graphs : f->n_graphs - f->n_fit;
    return gk_gram_rows(ctx, f, 0, n_rows, normalize, out_host);
}

extern "C" int gk_gram_dev_ptr(gk_feat* f, void** out_dev_ptr, int64_t* n_rows, int64_t* n_cols) {
    GK_ARG(f && out_dev_ptr, "gk_gram_dev_ptr: null argument");
    *out_dev_ptr = f->K;
    if (n_rows) *n_rows = f->K_rows;
    if (n_cols) *n_cols = f->K_cols;
    return GK_OK;
}

// ---------------------------------------------------------------------------------------
// Checksums of the Gram output that is still on the device (bench.py asserts the timed matrix without the
// 8 N^2-byte copy; the 50 000-graph parity test checks symmetry of a 20 GB matrix in place):
// sum of all entries, trace and max |K_ij - K_ji| (square outputs), 32x32 tiles transposed through LDS.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_checksum_kernel(const double* __restrict__ K, i64 M, i64 N, int square,
                                                            double* __restrict__ out /* sum, trace, max asym */) {
    __shared__ double ta[32][33], tb[32][33];
    __shared__ double red[3][4];
    const i64 tiles_n = (N + 31) / 32;
    const i64 ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
    double s = 0.0, tr = 0.0, as = 0.0;
    if (!square || tj >= ti) {
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 8 rows per pass
        for (int r = ty; r < 32; r += 8) {
            const i64 i = ti * 32 + r, j = tj * 32 + tx;
            ta[r][tx] = (i < M && j < N) ? K[i * N + j] : 0.0;
            if (square) {
                const i64 i2 = tj * 32 + r, j2 = ti * 32 + tx;
                tb[r][tx] = (i2 < M && j2 < N) ? K[i2 * N + j2] : 0.0;
            }
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const i64 i = ti * 32 + r, j = tj * 32 + tx;
            if (i < M && j < N) {
                const double a = ta[r][tx];
                if (!square) s += a;
                else {
                    const double b = tb[tx][r];            // K[j][i]
                    if (tj > ti) s += a + b;
                    else s += a;                           // diagonal tile: every entry once
                    const double d = a > b ? a - b : b - a;
                    as = d > as ? d : as;
                    if (i == j) tr += a;
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64), tr += __shfl_down(tr, off, 64);
        const double o = __shfl_down(as, off, 64);
        as = o > as ? o : as;
    }
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s, red[1][threadIdx.x >> 6] = tr, red[2][threadIdx.x >> 6] = as;
    __syncthreads();
    if (threadIdx.x == 0) {
        double S = 0, T = 0, A = 0;
        for (int q = 0; q < 4; ++q) { S += red[0][q], T += red[1][q]; A = red[2][q] > A ? red[2][q] : A; }
        if (S != 0.0) atomicAdd(&out[0], S);
        if (T != 0.0) atomicAdd(&out[1], T);
        if (A > 0.0) atomicMax((unsigned long long*)&out[2], (unsigned long long)__double_as_longlong(A));   // A >= 0: bit order == value order
    }
}

extern "C" int gk_gram_checksum(gk_ctx* ctx, gk_feat* f, double* out_sum, double* out_trace, double* out_max_asym) {
    GK_ARG(ctx && f && f->K, "gk_gram_checksum: no Gram output on the device (call gk_gram first)");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    const i64 M = f->K_rows, N = f->K_cols;
    const int square = M == N ? 1 : 0;
    Tmp<double> out(ctx);
    GK_TRY(out.alloc(3));
    GK_TRY(gk_zero_async(ctx, out.p, 24));
    const i64 blocks = ((M + 31) / 32) * ((N + 31) / 32);
    GK_ARG(blocks < (1ll << 31), "gk_gram_checksum: matrix too large");
    if (blocks > 0) gram_checksum_kernel<<<dim3((unsigned)blocks), 256, 0, ctx->stream>>>(f->K, M, N, square, out.p);
    double h[3] = {0, 0, 0};
    GK_HIP_CHECK(hipMemcpyAsync(h, out.p, 24, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (out_sum) *out_sum = h[0];
    if (out_trace) *out_trace = square ? h[1] : 0.0;
    if (out_max_asym) *out_max_asym = square ? h[2] : 0.0;
    return GK_OK;
}

extern "C" int gk_gram_last_stats(gk_feat* f, double* out_flops, double* out_ms_event) {
    GK_ARG(f, "gk_gram_last_stats: null");
    if (f->last_ms < 0 && f->ev0) {
        GK_HIP_CHECK(hipEventSynchronize(f->ev1));
        float ms = 0;
        GK_HIP_CHECK(hipEventElapsedTime(&ms, f->ev0, f->ev1));
        f->last_ms = ms;
    }
    if (out_flops) *out_flops = f->last_flops;
    if (out_ms_event) *out_ms_event = f->last_ms;
    return GK_OK;
}

#ifdef GK_ABLATION
// tools' build only: cycle stamps of workgroup 0 of the last gram_ws_kernel launch ([role][total, barriers, body, tile change])
extern "C" int gk_debug_ws_times(gk_ctx* ctx, unsigned long long* out12) {
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    GK_HIP_CHECK(hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_ws_dbg), sizeof(unsigned long long) * 12));
    return GK_OK;
}
#endif

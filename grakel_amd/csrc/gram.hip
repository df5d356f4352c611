// Gram matrix  K = Phi_rows . Phi_cols^T  on the MFMA units of gfx950.
//
// Label counts are small non-negative integers, so the product is EXACT integer arithmetic.
// A launch sequence (gk_gram_launch) is:
//   1. gram_i8_glds_kernel : dense columns with counts <= 127 as int8 operands,
//                            v_mfma_i32_32x32x32_i8, int32 accumulate (K < 2^31 is checked when the
//                            features are built): 2x the bf16 MFMA rate, bit-exact versus the
//                            reference's float64 result.  Writes every entry of K (float64).
//   2. gram_f64_kernel     : dense columns holding a count > 127 (typical for ShortestPath
//                            histograms) form a narrow float64 side operand,
//                            v_mfma_f64_16x16x4_f64, accumulated onto K (exact while K < 2^53).
//   3. gram_low_kernel     : useful columns present in < 24 graphs (GK_LOW_DF) never enter a dense operand;
//                            their df*(df-1) pair products are added as float64 atomics.
//   4. gram_normalize_kernel, only when 2. or 3. ran and normalisation was requested.
// Histogram-intersection features (kind 1) arrive unary-expanded (features.hip), so step 1 computes
// sum_l min(c_il, c_jl) exactly; step 2 never applies and step 3 adds min(c_a, c_b) per pair.
// The int8 epilogue fuses what the reference does in three extra N^2 passes: the per-level sum
// (all levels are concatenated along K), the diagonal (graph-unique label columns are not in
// Phi_s; K_ii is written from the exact selfk vector instead) and -- when no extra term
// follows -- the normalisation K_ij / sqrt(K_ii K_jj) (weisfeiler_lehman.py:323-328,
// kernel.py:195-204).
#include "common.h"
#include <stdlib.h>
#include <string.h>

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

#define GI_BK 64     // K-step in bytes (two 32-deep MFMA slices)

// Block -> output tile.  Workgroup b is observed to run on XCD b % 8 (speed assumption only),
// so consecutive ids of ONE XCD (b>>3) walk 8x8-tile patches: the ~128 tiles resident on an
// XCD then share 2x(8+8) operand panels instead of ~90, and its 4 MiB L2 serves the re-reads.
// Symmetric jobs only visit patches/tiles on or above the diagonal.
#define GI_PATCH 8
__device__ __forceinline__ bool gram_map_tile(int b, int tiles_m, int tiles_n, int sym, int patch,
                                              int& bm, int& bn) {
    if (!patch) {
        bm = b / tiles_n, bn = b % tiles_n;
        return !(sym && bn < bm);
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
    const int P = patch;
    const i64 pm = (tiles_m + P - 1) / P, pn = (tiles_n + P - 1) / P;
    const i64 np = sym ? (pm * pn - pm * (pm - 1) / 2) : pm * pn;   // sym: pm == pn
    return ((np + 7) / 8) * 8 * P * P;
}

// ---------------------------------------------------------------------------------------
// int8 path, pipelined.  Operand tiles go L2 -> LDS directly (global_load_lds_dwordx4: no VGPR
// round trip, no ds_write) into a ring of NS stages, NS-1 K-steps in flight across the per-step
// barrier (counted s_waitcnt vmcnt, raw s_barrier).  The LDS image is linear per wave
// instruction (16 rows x 64 B = 1 KiB), so the bank-conflict-free layout comes from
// XOR-swizzling the 16-byte chunk index on the SOURCE address and on the fragment read:
//      physical chunk = logical chunk ^ ((row >> 2) & 3)
// which spreads every 16-lane ds_read_b128 group over all 16 slots of the 256-B bank row.
// Measured: the L2->LDS path sustains ~11-14 TB/s chip-wide, so the operand bytes per MAC set
// the ceiling -> the 256x256 tile (8 waves, 128x64 per wave) halves them versus 128x128.
//   <WM,WN,TM,TN>: waves_m x waves_n, MFMA 32x32 tiles per wave; BM = WM*TM*32, BN = WN*TN*32.
// Each wave stages (BM+BN)/(16*waves) = 4 sixteen-row pieces per K-step in both shapes.
// ---------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int WM, int WN, int TM, int TN, int NS>
__global__ __launch_bounds__(64 * WM * WN) void gram_i8_glds_kernel(
    const int8_t* __restrict__ A, const int8_t* __restrict__ B, i64 ld, int k_tiles, int k4_tiles,
    const u64* __restrict__ selfk, double* __restrict__ K, i64 M, i64 N, i64 row_base,
    int symmetric, i64 n_fit, int normalize, int tiles_m, int tiles_n, int tri, int patch) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NW = WM * WN;
    constexpr int STAGE = (BM + BN) * 64;
    constexpr int PPW = (BM + BN) / 16 / NW;                 // 1-KiB pieces per wave per stage
    static_assert((BM + BN) % (16 * NW) == 0, "pieces must divide evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) int8_t smem[];   // NS * STAGE, ONE array
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int bm, bn;
    if (!gram_map_tile(blockIdx.x, tiles_m, tiles_n, tri, patch, bm, bn)) return;

    // staging: piece q of this wave covers rows [16*(wave*PPW+q), +16) of the (A rows, B rows) list
    const int srow = lane >> 2;                              // row inside the piece
    const int schunk = (lane & 3) ^ ((srow >> 2) & 3);       // logical chunk this lane fetches
    const int8_t* gsrc[PPW];
    int sdst[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int r0 = (wave * PPW + q) * 16;
        if (r0 < BM) gsrc[q] = A + ((i64)bm * BM + r0 + srow) * ld + schunk * 16;
        else gsrc[q] = B + ((i64)bn * BN + (r0 - BM) + srow) * ld + schunk * 16;
        sdst[q] = r0 * 64;
    }

    v16i acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

#define GL_ISSUE(KT)                                                                        \
    {                                                                                       \
        const i64 go = (i64)(KT) * GI_BK;                                                   \
        int8_t* st = smem + ((KT) % NS) * STAGE;                                            \
        _Pragma("unroll") for (int q = 0; q < PPW; ++q)                                     \
            __builtin_amdgcn_global_load_lds((glb_void_t*)(gsrc[q] + go), (lds_void_t*)(st + sdst[q]), 16, 0, 0); \
    }

    // Software pipeline inside the K-step: while the MFMAs of one 32-deep K-slice run, the
    // fragments of the next slice are already being read from LDS, so the LDS phase and the
    // MFMA phase of the 8 barrier-synchronised waves overlap instead of alternating.
    for (int p = 0; p < NS; ++p)
        if (p < k_tiles) GL_ISSUE(p);

    const int fr = lane & 31, fh = lane >> 5;
    int offa[TM][2], offb[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rr = wm * TM * 32 + i * 32 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offa[i][ks] = rr * 64 + (((2 * ks + fh) ^ ((rr >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int rr = wn * TN * 32 + j * 32 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offb[j][ks] = BM * 64 + rr * 64 + (((2 * ks + fh) ^ ((rr >> 2) & 3)) << 4);
    }

    // The first k4_tiles K-steps hold 4-bit counts, two columns per byte (features.hip): a 16-byte
    // fragment then feeds TWO MFMA slices (low nibbles, high nibbles; A and B are unpacked alike,
    // so the column order inside the dot product does not matter).  Half the bytes per column
    // through L2 -> LDS -> VGPR, which is what bounds this kernel.
#define GL_MFMA(FA, FB, PACKED)                                                              \
    if (PACKED) {                                                                            \
        v4i al[TM], ah[TM], bl[TN], bh[TN];                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) { al[i] = FA[i] & 0x0f0f0f0f; ah[i] = (FA[i] >> 4) & 0x0f0f0f0f; } \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) { bl[j] = FB[j] & 0x0f0f0f0f; bh[j] = (FB[j] >> 4) & 0x0f0f0f0f; } \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                       \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                   \
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(al[i], bl[j], acc[i][j], 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                       \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                   \
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ah[i], bh[j], acc[i][j], 0, 0, 0); \
    } else {                                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                       \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                   \
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(FA[i], FB[j], acc[i][j], 0, 0, 0); \
    }

    v4i fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    {   // own pieces of stage 0 landed: up to NS-1 later stages may stay in flight
        const int after = k_tiles - 1;
        const int fly = after < NS - 1 ? after : NS - 1;
        if (fly >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PPW) : "memory");
        else if (fly == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PPW) : "memory");
        else if (fly == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
        else if (fly == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    {
        const int8_t* st = smem;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa0[i] = *(const v4i*)(st + offa[i][0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb0[j] = *(const v4i*)(st + offb[j][0]);
    }
    // two copies of the K-step, specialised at compile time (a run-time "packed" flag inside one
    // loop makes the compiler keep two accumulator sets: 128 AGPRs, one wave per SIMD)
#define GL_STEP(PACKED)                                                                      \
    {                                                                                        \
        const int8_t* st = smem + (kt % NS) * STAGE;                                         \
        /* ---- phase A: read slice 1 of stage kt, multiply slice 0 */                        \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa1[i] = *(const v4i*)(st + offa[i][1]); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb1[j] = *(const v4i*)(st + offb[j][1]); \
        GL_MFMA(fa0, fb0, PACKED)                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        /* ---- stage hand-over: stage kt+1 must be complete, stage kt is fully read */       \
        if (kt + 1 < k_tiles) {                                                              \
            const int after = k_tiles - 1 - (kt + 1);   /* stages after kt+1 may stay in flight */ \
            const int fly = after < NS - 2 ? after : NS - 2;                                 \
            if (fly >= 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * PPW) : "memory"); \
            else if (fly == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PPW) : "memory"); \
            else if (fly == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPW) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                 \
        } else {                                                                             \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        }                                                                                    \
        __builtin_amdgcn_s_barrier();                                                        \
        if (kt + NS < k_tiles) GL_ISSUE(kt + NS);       /* overwrites the buffer of stage kt */ \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        /* ---- phase B: read slice 0 of stage kt+1, multiply slice 1 */                      \
        if (kt + 1 < k_tiles) {                                                              \
            const int8_t* sn = smem + ((kt + 1) % NS) * STAGE;                               \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) fa0[i] = *(const v4i*)(sn + offa[i][0]); \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) fb0[j] = *(const v4i*)(sn + offb[j][0]); \
        }                                                                                    \
        GL_MFMA(fa1, fb1, PACKED)                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    }
    int kt = 0;
    for (; kt < k4_tiles; ++kt) GL_STEP(true)
    for (; kt < k_tiles; ++kt) GL_STEP(false)
#undef GL_STEP
#undef GL_ISSUE
#undef GL_MFMA

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool mirror = tri && bm != bn;     // off-diagonal tile of a symmetric job: also write K^T
    const bool even = (N & 1) == 0;
    // The transposed copy of a 128x128 tile goes through LDS (the operand ring is free after the last
    // K-step): the int32 accumulators are written column-major with a 4-word pad per column
    // (conflict-free 16-byte writes), then every wave streams whole columns back -- 128 consecutive
    // K^T entries, 1 KiB of float64 per store instruction -- instead of 16-byte pieces scattered over
    // 32 rows.  Plain counts only (normalised epilogues keep the register path).
    constexpr bool LDS_MIRROR = (BM == 128 && BN == 128);
    constexpr int LDT = BM + 4;
    const bool lds_mirror = LDS_MIRROR && mirror && normalize == 0 && even;
    int* tsm = (int*)smem;
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
                        v[j] = finish_entry((double)acc[mt][nt][4 * q + j], row_base + row, col,
                                            symmetric != 0, selfk, n_fit, normalize);
                        K[row * N + col] = v[j];           // 32 lanes -> 256 contiguous bytes
                    }
                }
                if (lds_mirror) {
                    v4i t;
                    t[0] = acc[mt][nt][4 * q], t[1] = acc[mt][nt][4 * q + 1];
                    t[2] = acc[mt][nt][4 * q + 2], t[3] = acc[mt][nt][4 * q + 3];
                    *(v4i*)(tsm + ctile * LDT + rtile) = t;
                } else if (mirror && col < N) {             // K[col][row0..row0+3]: 32 B per lane
                    double* dst = K + col * N + row0;
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
    if (LDS_MIRROR && lds_mirror) {          // block-uniform
        __syncthreads();
        const i64 r = (i64)bm * BM + 2 * lane;              // two consecutive entries of K^T's row per lane
        for (int c = wave; c < BN; c += NW) {
            const i64 krow = (i64)bn * BN + c;
            if (krow >= N) break;
            const int2 t = *(const int2*)(tsm + c * LDT + 2 * lane);
            double* dst = K + krow * N + r;
            if (r + 1 < M) *(double2*)dst = make_double2((double)t.x, (double)t.y);
            else if (r < M) dst[0] = (double)t.x;
        }
    }
}

template <int WM, int WN, int TM, int TN, int NS>
static int launch_glds(gk_ctx* ctx, gk_feat* f, const int8_t* a, const int8_t* b, i64 M, i64 n_cols,
                       i64 row_lo, int normalize, double* K, int tri, int patch, int patch_sz, double* tiles_done) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int LDS_RING = NS * (BM + BN) * 64;
    constexpr int LDS_T = (BM == 128 && BN == 128) ? BN * (BM + 4) * 4 : 0;     // transposed tile of the epilogue
    constexpr int LDS = LDS_RING > LDS_T ? LDS_RING : LDS_T;
    const int tiles_m = (int)cdiv(M, BM), tiles_n = (int)cdiv(n_cols, BN);
    const i64 blocks = gram_grid_blocks(tiles_m, tiles_n, tri, patch ? patch_sz : 0);
    auto kern = gram_i8_glds_kernel<WM, WN, TM, TN, NS>;
    GK_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    int kt_all = f->k4_tiles + f->k8_tiles, kt4 = f->k4_tiles;
    i64 M_store = M;
    if (const char* abl = getenv("GK_GRAM_ABL")) {     // timing ablations (tools/gram_only.py): WRONG results
        if (!strcmp(abl, "nostore")) M_store = 0;       // K loop only: every store is predicated off
        if (!strcmp(abl, "nok")) kt_all = 0, kt4 = 0;   // epilogue only
    }
    kern<<<dim3((unsigned)blocks), dim3(64 * WM * WN), LDS, ctx->stream>>>(
        a, b, f->n_cols_pad, kt_all, kt4, f->selfk, K, M_store, n_cols, row_lo,
        f->symmetric ? 1 : 0, f->n_fit, normalize, tiles_m, tiles_n, tri, patch ? patch_sz : 0);
    *tiles_done = tri ? (double)tiles_m * (tiles_m + 1) / 2 * BM * BN : (double)M * n_cols;
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

__global__ __launch_bounds__(256) void gram_f64_kernel(
    const double* __restrict__ A, const double* __restrict__ B, i64 ld, int k_tiles,
    const u64* __restrict__ selfk, double* __restrict__ K, i64 M, i64 N, i64 row_base,
    int symmetric, i64 n_fit, int normalize, int tiles_n, int accumulate) {
    __shared__ double sA[GD_BM * GD_LD];
    __shared__ double sB[GD_BN * GD_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bm = blockIdx.x / tiles_n, bn = blockIdx.x % tiles_n;
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
    for (int kt = 0; kt < k_tiles; ++kt) {
        const i64 go = (i64)kt * GD_BK;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sA[lrow * GD_LD + lk + q] = gA[go + q];
            sB[lrow * GD_LD + lk + q] = gB[go + q];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kk = ks * 4 + (lane >> 4);
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
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const i64 col = (i64)bn * GD_BN + wn * 32 + nt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const i64 row = (i64)bm * GD_BM + wm * 32 + mt * 16 + (lane >> 4) + 4 * r;
                if (row < M && col < N) {
                    if (accumulate) {       // K already holds the int8 product (and selfk on the diagonal)
                        if (!(symmetric && row_base + row == col)) K[row * N + col] += acc[mt][nt][r];
                    } else {
                        K[row * N + col] = finish_entry(acc[mt][nt][r], row_base + row, col, symmetric != 0,
                                                        selfk, n_fit, normalize);
                    }
                }
            }
        }
}

// Rare columns (colid == -2): K[g_a][g_b] += c_a * c_b for every ordered pair of graphs that
// share the label.  One wave per rare label run (df < GK_LOW_DF triples): lanes walk the df*df pairs.
// Integer-valued float64 atomics: exact and order independent.
__global__ void gram_low_kernel(const LevelPack P, double* __restrict__ K, i64 n_cols, i64 row_lo, i64 row_hi,
                                int symmetric, i64 n_fit, int minsum) {
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
        if (row < row_lo || row >= row_hi) continue;
        if (symmetric ? (ia == ib) : (gb >= n_fit)) continue;
        const i32 ca = tri_pos[a + 1] - tri_pos[a], cb = tri_pos[b + 1] - tri_pos[b];
        atomicAdd(&K[(row - row_lo) * n_cols + gb], minsum ? (double)(ca < cb ? ca : cb) : (double)ca * (double)cb);
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

// rows [row_lo,row_hi) of the job's Gram matrix into K ([row_hi-row_lo] x n_cols, row major)
int gk_gram_launch(gk_ctx* ctx, gk_feat* f, i64 row_lo, i64 row_hi, int normalize, double* K) {
    const i64 n_cols = f->symmetric ? f->n_graphs : f->n_fit;
    const i64 M = row_hi - row_lo;
    if (M <= 0) return GK_OK;
    const i64 first_row_graph = (f->symmetric ? 0 : f->n_fit) + row_lo;   // row of Phi
    // the MFMA kernel is bracketed by two events that are only READ in gk_gram_last_stats: the call
    // returns as soon as everything is queued, so the host can already queue the next job
    if (!f->ev0) {
        GK_HIP_CHECK(hipEventCreate(&f->ev0));
        GK_HIP_CHECK(hipEventCreate(&f->ev1));
    }
    hipEvent_t e0 = f->ev0, e1 = f->ev1;
    GK_HIP_CHECK(hipEventRecord(e0, ctx->stream));
    double tiles_done = (double)M * n_cols;
    const int normalize_req = normalize;
    const bool has_low = f->n_low_cols > 0, has_wide = f->n_cols_wide > 0;
    if (has_low || has_wide) normalize = 0;   // normalise after the extra terms instead of in the epilogue
    {
        const int8_t* phi = (const int8_t*)f->phi;
        const int8_t* pa = phi + first_row_graph * f->n_cols_pad;
        // full symmetric job: only tiles on/above the diagonal are computed, each written twice
        const int tri = (f->symmetric && row_lo == 0 && M == n_cols && !getenv("GK_GRAM_NO_SYM")) ? 1 : 0;
        const int patch = getenv("GK_GRAM_NO_PATCH") ? 0 : 1;
        const char* shape = getenv("GK_GRAM_TILE");
        if (shape && !strcmp(shape, "256")) {
            GK_TRY((launch_glds<2, 4, 4, 2, 4>(ctx, f, pa, phi, M, n_cols, row_lo, normalize, K, tri, patch, 4, &tiles_done)));
        } else if (shape && !strcmp(shape, "128ns3")) {
            GK_TRY((launch_glds<2, 2, 2, 2, 3>(ctx, f, pa, phi, M, n_cols, row_lo, normalize, K, tri, patch, 8, &tiles_done)));
        } else if (shape && !strcmp(shape, "128ns2")) {
            GK_TRY((launch_glds<2, 2, 2, 2, 2>(ctx, f, pa, phi, M, n_cols, row_lo, normalize, K, tri, patch, 8, &tiles_done)));
        } else if ((shape && !strcmp(shape, "128")) || (!shape && f->n_cols_pad < 8192)) {
            // short K: the 128x128 tile runs two workgroups per CU, so one tile's float64 store
            // epilogue overlaps the other's MFMA loop (measured 0.36 vs 0.41 ms at K = 3968)
            GK_TRY((launch_glds<2, 2, 2, 2, 4>(ctx, f, pa, phi, M, n_cols, row_lo, normalize, K, tri, patch, 8, &tiles_done)));
        } else {
            GK_TRY((launch_glds<2, 4, 4, 2, 4>(ctx, f, pa, phi, M, n_cols, row_lo, normalize, K, tri, patch, 4, &tiles_done)));
        }
    }
    GK_HIP_CHECK(hipGetLastError());
    GK_HIP_CHECK(hipEventRecord(e1, ctx->stream));
    if (has_wide) {   // float64 side operand: K += Phi_w . Phi_w^T (diagonal excluded: it is selfk)
        const double* pw = f->phi_w;
        const int tiles_m = (int)cdiv(M, GD_BM), tiles_n = (int)cdiv(n_cols, GD_BN);
        gram_f64_kernel<<<dim3((unsigned)(tiles_m * (i64)tiles_n)), dim3(256), 0, ctx->stream>>>(
            pw + first_row_graph * f->n_cols_wide_pad, pw, f->n_cols_wide_pad, (int)(f->n_cols_wide_pad / GD_BK),
            f->selfk, K, M, n_cols, row_lo, f->symmetric ? 1 : 0, f->n_fit, 0, tiles_n, 1);
    }
    if (has_low) {
        for (int l0 = 0; l0 < f->n_levels; l0 += GK_PACK_LEVELS) {     // one launch per 16 levels
            LevelPack P;
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
                P, K, n_cols, row_lo, row_hi, f->symmetric ? 1 : 0, f->n_fit, f->kind == GK_FEAT_MINSUM ? 1 : 0);
        }
    }
    if (has_low || has_wide) {
        if (normalize_req)
            gram_normalize_kernel<<<dim3((unsigned)cdiv(M * n_cols, 256)), dim3(256), 0, ctx->stream>>>(
                K, f->selfk, M, n_cols, row_lo, f->symmetric ? 1 : 0, f->n_fit, normalize_req);
        GK_HIP_CHECK(hipGetLastError());
    }
    f->last_ms = -1.0;      // not read yet
    // work actually executed: symmetric jobs only run the tiles on/above the diagonal
    f->last_flops = 2.0 * tiles_done * (double)f->n_cols;     // columns actually holding a label (padding excluded)
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
    GK_TRY(gk_gram_launch(ctx, f, row_lo, row_hi, normalize, f->K));
    if (out_host && M > 0) {
        GK_HIP_CHECK(hipMemcpyAsync(out_host, f->K, (size_t)M * n_cols * 8, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
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

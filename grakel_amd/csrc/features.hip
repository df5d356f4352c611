// Label-count features (the VertexHistogram of every WL level) on gfx950 -- the LABEL-MAJOR builder: pair batches
// (ShortestPath), operand rows wider than features_gm.hip holds in LDS, option feat.no_gm.  Graph batches take the
// graph-major builder (features_gm.hip), which this file dispatches to first.
//
// Input per level: labels[v] and perm[] = nodes grouped by label, ascending node (hence
// ascending graph) inside a group -- exactly what the relabel sort leaves behind.  One pass
// over perm order finds label-run heads and (label,graph) sub-run heads; a packed 64-bit
// scan turns them into run ids and triple ids, so the sparse feature matrix falls out in
// column-major (label-major) order without any per-graph sort:
//      triple t = (label run r, graph g, count c),   c = tri_pos[t+1] - tri_pos[t]
//      df[r]    = tstart[r+1] - tstart[r]            (# graphs containing the label)
// selfk[g] += c^2 over ALL triples (the exact Gram diagonal); only columns that can touch an
// off-diagonal entry are kept for the dense Phi_s that feeds the MFMA Gram:
//      symmetric job   : df >= 2
//      rectangular job : present in a fit graph (< n_fit) AND a target graph (>= n_fit)
// HBM-bound integer work: ~16 bytes per node per level.
//
// ALL LEVELS GO THROUGH EACH KERNEL TOGETHER: the items of level l (the first n_sorted[l]
// positions of its perm: the nodes that can share a label, wl.hip) occupy the index range
// [off[l], off[l] + n_l) of one concatenated item space, off[l] a multiple of the scan tile.  A
// 6-level job is then 9 launches instead of ~45 -- at 1 M nodes every kernel of this file is
// launch-latency bound (5-20 us), not bandwidth bound.  The triple scan restarts per level
// (segmented), the column-id scan runs across the levels so that its prefix IS the column id.
//
// kind 1 (histogram intersection, K_ij = sum_l min(c_il, c_jl); WL-OA,
// weisfeiler_lehman_optimal_assignment.py:268-279) stays on the same integer GEMM through the
// UNARY expansion  min(a, b) = sum_{t>=1} [a >= t][b >= t]:  a dense label column whose largest
// count is m becomes m 0/1 columns ([c>=1], [c>=2], ...), so Phi_s . Phi_s^T IS the min-sum,
// exactly, on the int8 MFMA path; selfk[g] = sum of counts (= nodes x levels).
#include "common.h"
#include <math.h>
#include "scan_fn.h"
#include "features.h"
#include <stdlib.h>

static inline dim3 grid_for(i64 n, int t) { return dim3((unsigned)(n > 0 ? cdiv(n, t) : 1)); }

// the levels of one job, by value: level slot j covers items [off[j], off[j] + n[j])
struct FeatLevels {
    const i32* perm[FEAT_MAX_LEVELS];
    const i32* lab[FEAT_MAX_LEVELS];
    i64 n[FEAT_MAX_LEVELS];
    i64 off[FEAT_MAX_LEVELS + 1];     // multiples of SF_TILE; off[L] = size of the item space
    int level[FEAT_MAX_LEVELS];       // WL level of the slot (levels without shared labels are skipped)
    int synth;                        // slot whose triple / run arrays come from the level-0 histogram (-1: none):
                                      // the per-item kernels skip it
    int L;
    __device__ __forceinline__ int slot_of(i64 i) const {
        int j = 0;
        while (i >= off[j + 1]) ++j;
        return j;
    }
};

// per-level views into the concatenated arrays: level slot j starts at element off[j] + j (one
// sentinel entry of slack per level)
struct FeatArrays {
    i32 *tri_pos, *tri_graph, *tri_run, *tstart, *colid, *wide, *low_all;
    __device__ __forceinline__ i64 base(const FeatLevels& P, int j) const { return P.off[j] + j; }
};

__global__ void feat_flags_kernel(const FeatLevels P, const FeatArrays A, const i32* __restrict__ node_graph,
                                  u64* __restrict__ flag) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.off[P.L]) return;
    const int j = P.slot_of(i);
    const i64 k = i - P.off[j];
    u64 f = 0;
    if (k < P.n[j]) {
        const i32* perm = P.perm[j];
        const i32* lab = P.lab[j];
        const i32 v = j == P.synth ? 0 : perm[k];
        f = 0x100000001ull;
        if (j == P.synth) f = 0;
        else if (k > 0) {
            const i32 p = perm[k - 1];
            const bool lh = lab[v] != lab[p];
            const bool sh = lh || node_graph[v] != node_graph[p];
            f = ((u64)lh << 32) | (u64)sh;
        }
        A.wide[A.base(P, j) + k] = 0;     // per-run flag / maximum filled in by the count kernels (runs <= items)
    }
    flag[i] = f;
}

// packed (label-head, subrun-head) flags -> triple ids / run ids, triples emitted in the scan
struct TripleEmit {
    FeatLevels P; FeatArrays A;
    const i32* node_graph; const u64* flag; u32* tri_of;   // tri_of[i] = triple of item i
    u32* meta;
    __device__ __forceinline__ u64 value(i64 i) const { return flag[i]; }
    __device__ __forceinline__ void emit(i64 i, u64 f, u64 s) const {
        const int j = P.slot_of(i);
        const i64 k = i - P.off[j];
        if (k >= P.n[j] || j == P.synth) return;
        const i64 b = A.base(P, j);
        const i32 t = (i32)(u32)(s & 0xffffffffull) - 1;
        const i32 r = (i32)(u32)(s >> 32) - 1;
        tri_of[i] = (u32)t;
        if (f & 1ull) {
            A.tri_pos[b + t] = (i32)k;
            A.tri_graph[b + t] = node_graph[P.perm[j][k]];
            A.tri_run[b + t] = r;
        }
        if (f >> 32) A.tstart[b + r] = t;
        if (k == P.n[j] - 1) {   // sentinels + counts
            A.tri_pos[b + t + 1] = (i32)P.n[j];
            A.tstart[b + r + 1] = t + 1;
            meta[META_T(P.level[j])] = (u32)(t + 1);
            meta[META_R(P.level[j])] = (u32)(r + 1);
        }
    }
    __device__ __forceinline__ void finish(u64) const {}
    __device__ __forceinline__ i64 seg_first_tile(i64 tile) const {
        return P.off[P.slot_of(tile * SF_TILE)] / SF_TILE;
    }
};

// per item: the count of its (label,graph) triple -> exact self similarity (see below), largest count,
// operand class of the run.
// Also tracks the largest count and flags runs whose counts leave the primary / the int8 range.
__global__ void feat_count_kernel(const FeatLevels P, const FeatArrays A, const u32* __restrict__ tri_of,
                                  int prim_max, int wide_above, u64* __restrict__ selfk, u32* __restrict__ meta,
                                  int n_levels, int kind) {
    __shared__ u32 wmax[4];
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = 0;
    if (i < P.off[P.L]) {
        const int j = P.slot_of(i);
        const i64 k = i - P.off[j];
        if (k < P.n[j] && j != P.synth) {
            const i64 b = A.base(P, j);
            const i32 t = (i32)tri_of[i];
            c = (u32)(A.tri_pos[b + t + 1] - A.tri_pos[b + t]);
            if (!kind) {
                // operand class of the run: byte 0 set = some count exceeds what the primary region holds
                // (fp4: 4), byte 1 set = needs the float64 side.  Plain byte stores of the same value from
                // every such item: a hot run (level 0: a few labels, a million items) must not serialise
                // on an atomic or a read
                char* w = (char*)&A.wide[b + A.tri_run[b + t]];
                if ((int)c > prim_max) w[0] = 1;
                if ((int)c > wide_above) w[1] = 1;
            }
            // exact self similarity: every node counts 1 per level (feat_selfk_init_kernel), a triple of
            // count c adds the remaining c^2 - c once (integer atomics: exact, order independent; kind 1
            // sums min(c, c) = c per triple, i.e. nothing beyond the baseline)
            if (!kind && c >= 2u && k == (i64)A.tri_pos[b + t])
                atomicAdd((unsigned long long*)&selfk[A.tri_graph[b + t]], (unsigned long long)c * c - c);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        u32 o = __shfl_down(c, off, 64);
        c = o > c ? o : c;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 m = wmax[0];
        for (int q = 1; q < (int)(blockDim.x >> 6); ++q) m = wmax[q] > m ? wmax[q] : m;
        // 64 slots instead of one counter (the host takes their maximum): tens of thousands of
        // blocks reading or updating ONE address serialise on a single L2 channel
        if (m) atomicMax(&meta[4 * n_levels + 5 + (blockIdx.x & 63)], m);
    }
}

// kind 1: wide[r] = largest count of label run r (the number of unary columns it expands to).
// Triples are run-major, so a wave usually sees one run: one atomic per wave then.
__global__ void feat_runmax_kernel(const FeatLevels P, const FeatArrays A, const u32* __restrict__ meta) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i32 r = -1, c = 0;
    i64 b = 0;
    if (i < P.off[P.L]) {
        const int j = P.slot_of(i);
        const i64 t = i - P.off[j];
        b = A.base(P, j);
        if (t < (i64)meta[META_T(P.level[j])]) r = A.tri_run[b + t], c = A.tri_pos[b + t + 1] - A.tri_pos[b + t];
    }
    const i32 r0 = __shfl(r, 0, 64);
    const i64 b0 = __shfl(b, 0, 64);
    if (__all((r == r0 && b == b0) || r < 0)) {
        if (r0 < 0) return;
        for (int off = 32; off > 0; off >>= 1) {
            i32 o = __shfl_down(c, off, 64);
            c = o > c ? o : c;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(&A.wide[b0 + r0], c);
    } else if (r >= 0) {
        atomicMax(&A.wide[b + r], c);
    }
}

// selfk[g] = nodes of g x (levels whose triples go through feat_count_kernel or that list nothing) + the
// level-0 histogram's own sum (extra, may be null); feat_count_kernel adds c^2 - c per triple.
__global__ void feat_selfk_init_kernel(const i32* __restrict__ graph_ptr, u64* __restrict__ selfk, i64 n_graphs,
                                       int n_levels_plain, const u64* __restrict__ extra) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_graphs) return;
    selfk[g] = (u64)(graph_ptr[g + 1] - graph_ptr[g]) * (u64)n_levels_plain + (extra ? extra[g] : 0ull);
}

// ---- level 0 with a small label alphabet (wl.hip: gk_batch::level0_hist): no label-grouped order exists.
// hist0: one wave per graph counts its labels in LDS -> C0T[label][graph] and the graph's sum of
// squared counts (kind 1: its node count).  hist0_rows: one workgroup per label reduces its row to
// (graphs containing it, nodes carrying it).  hist0_synth: one workgroup per label writes that label's
// triples (tri_graph, tri_pos as the running node count, tri_run), its run entry (tstart, operand-class
// flags) and workgroup 0 the sentinels and the level's triple / run counts -- exactly the arrays
// TripleEmit and feat_count_kernel produce for a sorted level, at L0 * N items instead of V.
__global__ __launch_bounds__(256) void feat_hist0_kernel(const i32* __restrict__ graph_ptr, const i32* __restrict__ lab0,
                                                         i64 n_graphs, int L0, u32* __restrict__ c0t,
                                                         u64* __restrict__ extra, int kind) {
    __shared__ u32 h[4][GK_HIST0_MAX_LABELS];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const i64 g = (i64)blockIdx.x * 4 + w;
    for (int l = lane; l < L0; l += 64) h[w][l] = 0;
    __syncthreads();
    if (g < n_graphs) {
        const i32 v0 = graph_ptr[g], v1 = graph_ptr[g + 1];
        for (i32 v = v0 + lane; v < v1; v += 64) atomicAdd(&h[w][lab0[v]], 1u);
    }
    __syncthreads();
    if (g >= n_graphs) return;
    u64 s = 0;
    for (int l = lane; l < L0; l += 64) {
        const u32 c = h[w][l];
        c0t[(i64)l * n_graphs + g] = c;
        s += kind ? (u64)c : (u64)c * c;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) extra[g] = s;
}

__device__ __forceinline__ u32 hist0_block_sum(u32 x, u32* red) {       // 1024 threads
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    u32 t = 0;
    for (int q = 0; q < 16; ++q) t += red[q];
    return t;
}

// chunk = 1024 consecutive graphs of one label's row; rows[(l * n_chunks + ch) * 2 + {0,1}] = graphs containing
// the label / nodes carrying it inside the chunk
__global__ __launch_bounds__(1024) void feat_hist0_rows_kernel(const u32* __restrict__ c0t, i64 n_graphs, int n_chunks,
                                                               u32* __restrict__ rows) {
    __shared__ u32 red[16];
    const int l = blockIdx.x / n_chunks, ch = blockIdx.x % n_chunks;
    const i64 g = (i64)ch * 1024 + threadIdx.x;
    const u32 c = g < n_graphs ? c0t[(i64)l * n_graphs + g] : 0u;
    const u32 df = hist0_block_sum(c ? 1u : 0u, red);
    const u32 sz = hist0_block_sum(c, red);
    if (threadIdx.x == 0) rows[2 * (i64)blockIdx.x] = df, rows[2 * (i64)blockIdx.x + 1] = sz;
}

__global__ __launch_bounds__(1024) void feat_hist0_synth_kernel(const u32* __restrict__ c0t, const u32* __restrict__ rows,
                                                                i64 n_graphs, int n_chunks, int L0, const FeatArrays A, i64 base,
                                                                u32* __restrict__ meta, int level, int n_levels, int prim_max,
                                                                int wide_above, int kind) {
    __shared__ u32 red[16];
    __shared__ u32 wsum_t[16], wsum_p[16];
    __shared__ u32 sh[8];
    const int l = blockIdx.x / n_chunks, ch = blockIdx.x % n_chunks, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // triples / nodes before this chunk (earlier labels, earlier chunks of this label), non-empty labels before
    // this label, the label's own totals, and the totals of the level: sums over the (label, chunk) cells
    u32 tb = 0, pb = 0, tt = 0, pt = 0, mine = 0;
    const int cells = L0 * n_chunks;
    for (int q = tid; q < cells; q += 1024) {
        const u32 d = rows[2 * q], z = rows[2 * q + 1];
        if (q < (int)blockIdx.x) tb += d, pb += z;
        if (q / n_chunks == l) mine += d;
        tt += d, pt += z;
    }
    tb = hist0_block_sum(tb, red), pb = hist0_block_sum(pb, red), tt = hist0_block_sum(tt, red);
    pt = hist0_block_sum(pt, red), mine = hist0_block_sum(mine, red);
    // non-empty labels before l / in total: thread q < L0 sums label q's chunks
    u32 ne_before = 0, ne_all = 0;
    if (tid < L0) {
        u32 d = 0;
        for (int c2 = 0; c2 < n_chunks; ++c2) d += rows[2 * (tid * n_chunks + c2)];
        ne_all = d ? 1u : 0u;
        ne_before = (d && tid < l) ? 1u : 0u;
    }
    const u32 rb = hist0_block_sum(ne_before, red), rt = hist0_block_sum(ne_all, red);
    if (blockIdx.x == 0 && tid == 0) {   // sentinels and counts of the level
        A.tri_pos[base + tt] = (i32)pt;
        A.tstart[base + rt] = (i32)tt;
        meta[META_T(level)] = tt;
        meta[META_R(level)] = rt;
    }
    if (mine == 0) return;
    if (ch == 0 && tid == 0) A.tstart[base + rb] = (i32)tb;
    const i64 g = (i64)ch * 1024 + tid;
    const u32 c = g < n_graphs ? c0t[(i64)l * n_graphs + g] : 0u;
    const u32 has = c ? 1u : 0u;
    u32 it = has, ip = c;                         // inclusive scans over the 1024 graphs of the chunk
    for (int off = 1; off < 64; off <<= 1) {
        const u32 yt = __shfl_up(it, off, 64), yp = __shfl_up(ip, off, 64);
        if (lane >= off) it += yt, ip += yp;
    }
    __syncthreads();
    if (lane == 63) wsum_t[w] = it, wsum_p[w] = ip;
    __syncthreads();
    u32 bt = 0, bp = 0;
    for (int q = 0; q < w; ++q) bt += wsum_t[q], bp += wsum_p[q];
    if (has) {
        const i64 t = base + tb + bt + it - 1;
        A.tri_graph[t] = (i32)g;
        A.tri_pos[t] = (i32)(pb + bp + ip - c);
        A.tri_run[t] = (i32)rb;
    }
    // largest count of the chunk: operand class of the label's run (OR / max from every chunk), the job's largest count
    u32 cmax = c;
    for (int off = 32; off > 0; off >>= 1) { const u32 o = __shfl_down(cmax, off, 64); cmax = o > cmax ? o : cmax; }
    __syncthreads();
    if (lane == 0) red[w] = cmax;
    __syncthreads();
    if (tid == 0) {
        u32 m = 0;
        for (int q = 0; q < 16; ++q) m = red[q] > m ? red[q] : m;
        if (kind) atomicMax(&A.wide[base + rb], (i32)m);        // unary width of the run
        else {
            const i32 fl = ((int)m > prim_max ? 1 : 0) | ((int)m > wide_above ? 0x100 : 0);
            if (fl) atomicOr(&A.wide[base + rb], fl);
        }
        atomicMax(&meta[4 * n_levels + 5 + (blockIdx.x & 63)], m);
    }
    (void)sh;
}

// Column classes per label run, fused into ONE prefix sum over the runs of all levels:
//   dense (colid >= 0) : occurs in >= low_df graphs -> a column of the MFMA operand Phi_s (primary
//                        region when every count fits it, else numbered by ColumnIdsByteWide)
//   low   (colid = -2) : useful but rare (df < low_df): its df*(df-1) pair products are added
//                        to K by gram_low_kernel after the GEMM -- a column with df graphs
//                        costs N^2 MACs in the dense product but only df^2 updates here
//   dead  (colid = -1) : cannot touch an off-diagonal entry (graph-unique / one-sided)
// scan value packs (low << 32 | dense width) so one pass yields both running counts; the running
// dense count is the column id itself (levels are concatenated along K).
struct ColumnIds {
    FeatLevels P; FeatArrays A;
    u32* meta; int symmetric; i32 n_fit; i32 low_df; int kind; int n_levels;
    __device__ __forceinline__ u64 value(i64 i) const {
        const int j = P.slot_of(i);
        const i64 r = i - P.off[j];
        if (r >= (i64)meta[META_R(P.level[j])]) return 0ull;
        const i64 b = A.base(P, j);
        const i32 t0 = A.tstart[b + r], t1 = A.tstart[b + r + 1];
        bool useful;
        if (symmetric) useful = (t1 - t0) >= 2;
        else useful = A.tri_graph[b + t0] < n_fit && A.tri_graph[b + t1 - 1] >= n_fit;
        if (!useful) return 0ull;
        if ((t1 - t0) < low_df) return 1ull << 32;
        if (kind) return (u64)(u32)A.wide[b + r];       // unary expansion: one 0/1 column per count level
        return A.wide[b + r] ? (1ull << 63) : 1ull;     // bit 63: dense but not primary (toggles only itself)
    }
    __device__ __forceinline__ void emit(i64 i, u64 v, u64 incl) const {
        const int j = P.slot_of(i);
        const i64 r = i - P.off[j];
        if (r >= P.n[j]) return;
        const i64 b = A.base(P, j);
        const u32 rare = (u32)(v >> 32) & 0x7fffffffu;
        const u32 width = (u32)(v & 0xffffffffull);     // 1 (kind 0) or the run's largest count (kind 1)
        const u32 dense_incl = (u32)(incl & 0xffffffffull), rare_incl = (u32)(incl >> 32) & 0x7fffffffu;
        A.colid[b + r] = width ? (i32)(dense_incl - width) : ((v >> 63) ? -3 : (rare ? -2 : -1));
        if (rare) A.low_all[rare_incl - 1] = (i32)r;    // compact list for gram_low_kernel (level-local run id)
        if (r == P.n[j] - 1) {                          // running totals at the end of the level
            meta[META_C(P.level[j])] = dense_incl;
            meta[3 * n_levels + 4 + P.level[j]] = rare_incl;
        }
    }
    __device__ __forceinline__ void finish(u64 t) const {
        meta[3 * n_levels + 3] = (u32)(t & 0xffffffffull);
        meta[3 * n_levels + 1] = (u32)(t >> 32) & 0x7fffffffu;
    }
    __device__ __forceinline__ i64 seg_first_tile(i64) const { return 0; }
};

// second pass: dense columns that do not fit the primary region -> ids in the secondary int8 region
// (low half of the packed sum) or in the float64 side operand (high half)
struct ColumnIdsByteWide {
    FeatLevels P; FeatArrays A; u32* meta; int n_levels;
    __device__ __forceinline__ u64 value(i64 i) const {
        const int j = P.slot_of(i);
        const i64 r = i - P.off[j];
        const i64 b = A.base(P, j);
        if (r >= P.n[j] || A.colid[b + r] != -3) return 0ull;
        return (A.wide[b + r] & 0xff00) ? (1ull << 32) : 1ull;
    }
    __device__ __forceinline__ void emit(i64 i, u64 w, u64 incl) const {
        if (!w) return;
        const int j = P.slot_of(i);
        i32* c = &A.colid[A.base(P, j) + (i - P.off[j])];
        if (w >> 32) *c = -4 - (i32)((u32)(incl >> 32) - 1);
        else *c = COL_BYTE_BASE + (i32)((u32)(incl & 0xffffffffull) - 1);
    }
    __device__ __forceinline__ void finish(u64 t) const {
        meta[3 * n_levels + 2] = (u32)(t >> 32);
        meta[4 * n_levels + 4] = (u32)(t & 0xffffffffull);
    }
    __device__ __forceinline__ i64 seg_first_tile(i64) const { return 0; }
};

// all levels in one launch: P.first = prefix of the per-level triple counts
// phi: byte staging image [rows][ld], primary-class columns first, secondary int8 columns from byte_col0.
// fp4: the primary region receives the MX fp4 (e2m1) CODE of the count (0,1,2,3,4 -> 0,2,4,5,6), which
// feat_pack_kernel then only has to pack two per byte.
__device__ __forceinline__ int8_t fp4_code(i32 c) { return (int8_t)((0x65420 >> (4 * c)) & 15); }
__global__ void feat_scatter_mixed_kernel(const LevelPack P, int8_t* __restrict__ phi, i64 ld, i64 byte_col0,
                                          double* __restrict__ phi_w, i64 ldw, int kind, int fp4) {
    i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.first[P.n]) return;
    int l = 0;
    while (t >= P.first[l + 1]) ++l;
    t -= P.first[l];
    const i32* __restrict__ tri_pos = P.tri_pos[l];
    const i32* __restrict__ tri_graph = P.tri_graph[l];
    const i32 c = P.colid[l][P.tri_run[l][t]];
    const i32 cnt = tri_pos[t + 1] - tri_pos[t];
    if (c >= 0 && kind) {
        int8_t* row = phi + (i64)tri_graph[t] * ld + c;
        const int8_t one = fp4 ? 2 : 1;
        for (i32 q = 0; q < cnt; ++q) row[q] = one;         // [count >= q+1]
    } else if (c >= COL_BYTE_BASE) phi[(i64)tri_graph[t] * ld + byte_col0 + (c - COL_BYTE_BASE)] = (int8_t)cnt;
    else if (c >= 0) phi[(i64)tri_graph[t] * ld + c] = fp4 ? fp4_code(cnt) : (int8_t)cnt;
    else if (c <= -4) phi_w[(i64)tri_graph[t] * ldw + (-4 - c)] = (double)cnt;
}

// staging image -> GEMM operand row [n8p secondary bytes | primary]: fp4 primary = the n1p staged codes
// packed two per byte (column 2q low, 2q+1 high nibble of byte q), else the n1p bytes as they are.
// One thread per 4 output bytes.
__global__ void feat_pack_kernel(const int8_t* __restrict__ stage, i64 ld_stage, i64 n1p, i64 n8p, int fp4,
                                 int8_t* __restrict__ out, i64 ld_out, i64 n_rows) {
    const i64 words = ld_out >> 2;
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * words) return;
    const i64 row = idx / words, w = idx - row * words;
    const int8_t* src = stage + row * ld_stage;
    u32 o;
    if (w * 4 < n8p) {
        o = *(const u32*)(src + n1p + w * 4);
    } else if (fp4) {
        const u64 x = *(const u64*)(src + (w * 4 - n8p) * 2);
        const u64 lo = x & 0x000f000f000f000full, hi = (x >> 4) & 0x00f000f000f000f0ull;
        const u64 m = lo | hi;                       // byte pairs (b0 | b1<<4) at bits 0, 16, 32, 48
        o = (u32)(m & 0xffull) | (u32)((m >> 8) & 0xff00ull) | (u32)((m >> 16) & 0xff0000ull) |
            (u32)((m >> 24) & 0xff000000ull);
    } else {
        o = *(const u32*)(src + (w * 4 - n8p));
    }
    *(u32*)(out + row * ld_out + w * 4) = o;
}

extern "C" int gk_features_destroy(gk_feat* f) {
    if (!f) return GK_OK;
    gk_ctx* ctx = f->ctx;
    if (f->ev0) (void)hipEventDestroy(f->ev0);
    if (f->ev1) (void)hipEventDestroy(f->ev1);
    for (void* p : f->arena)
        if (p) gk_dev_free(ctx, p);
    void* ptrs[] = {f->meta, f->selfk, f->phi, f->phi_r, f->phi_w, f->K};
    for (void* p : ptrs)
        if (p) gk_dev_free(ctx, p);
    delete f;
    return GK_OK;
}

extern "C" int gk_features_build(gk_ctx* ctx, gk_batch* b, int n_levels, int64_t n_fit, gk_feat** out) {
    return gk_features_build_ex(ctx, b, n_levels, n_fit, GK_FEAT_DOT, out);
}

extern "C" int gk_features_build_ex(gk_ctx* ctx, gk_batch* b, int n_levels, int64_t n_fit, int kind,
                                    gk_feat** out) {
    return gk_features_build_range(ctx, b, 0, n_levels, n_fit, kind, out);
}

extern "C" int gk_features_build_range(gk_ctx* ctx, gk_batch* b, int level_lo, int level_hi, int64_t n_fit, int kind,
                                       gk_feat** out) {
    GK_ARG(ctx && b && out, "gk_features_build: null argument");
    GK_ARG(kind == GK_FEAT_DOT || kind == GK_FEAT_MINSUM, "gk_features_build_ex: unknown kind");
    GK_ARG(level_lo >= 0 && level_hi > level_lo && level_hi <= (b->n_levels > 0 ? b->n_levels : 1),
           "gk_features_build: levels not computed (call gk_wl_relabel first)");
    const int n_levels = level_hi - level_lo;        // local level l of this job = level level_lo + l of the batch
    GK_ARG(n_levels <= FEAT_MAX_LEVELS, "gk_features_build: at most 48 levels per feature job (gk_features_build_range "
                                        "takes a deep hierarchy in chunks; the matrices of the chunks add up)");
    GK_ARG(n_fit >= 1 && n_fit <= b->n_graphs, "gk_features_build: bad n_fit");
    GK_ARG(b->n_levels > 0, "gk_features_build: batch has no label-grouped order (call gk_wl_relabel with n_iter>=0)");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ProfScope prof(ctx, "features");
    const i64 V = b->n_nodes, N = b->n_graphs;
    gk_feat* f = new gk_feat();
    f->ctx = ctx, f->batch = b, f->n_levels = n_levels, f->n_graphs = N, f->n_fit = n_fit, f->n_nodes = V;
    f->symmetric = (n_fit == N);
    f->kind = kind;
    f->level0 = level_lo;
    f->lev.resize(n_levels);
    auto fail = [&](int r) { gk_features_destroy(f); return r; };
    int r;
    void* q = nullptr;
    const size_t n_meta = std::max<size_t>(4 * (size_t)n_levels + 5 + 64, GM_META_WORDS);      // ... + 64 partial maxima of the counts
    if ((r = gk_dev_alloc(ctx, &q, n_meta * 4))) return fail(r);
    f->meta = (u32*)q;
    if ((r = gk_dev_alloc(ctx, &q, (size_t)N * 8))) return fail(r);
    f->selfk = (u64*)q;
    if (gk_zero_async(ctx, f->meta, n_meta * 4) != GK_OK ||
        (V == 0 && gk_zero_async(ctx, f->selfk, (size_t)N * 8) != GK_OK)) {   // V > 0: feat_selfk_kernel writes it all
        gk_set_error("gk_features_build: memset failed");
        return fail(GK_ERR_HIP);
    }
    {
        // df threshold below which a column leaves the dense operand and becomes pair updates.  A dense column costs ~N^2
        // (operand traffic and MFMA time of every tile), a rare one ~df^2 float64 atomics, so the break-even grows with the
        // job: measured optima 12 at 4 000 graphs, 24 at 10 000, 36 at 20 000 (config-3-like), 72-128 at 50 000 (config 5:
        // step 7.7 -> 5.9 ms) -- 24 (N / 10 000)^0.75, kept within [8, 128].  Option feat.low_df fixes it.
        const double scaled_df = 24.0 * pow((double)std::max<i64>(N, 1) / 10000.0, 0.75);
        f->low_df = ctx->opt.low_df > 0 ? ctx->opt.low_df : (int)std::min(128.0, std::max(8.0, floor(scaled_df + 0.5)));
        if (f->low_df < 2) f->low_df = 2;        // 2 == everything useful is dense
    }
    if (!b->graph_ptr) {
        gk_set_error("gk_features_build: batch has no graph_ptr");
        return fail(GK_ERR_STATE);
    }
    // int8 operands need counts <= 127 and every Gram entry < 2^31:
    // K_ij <= sqrt(K_ii K_jj) <= n_levels * max_graph_nodes^2.  If the bound fails everything
    // dense goes to the float64 operand (wide_above = -1 flags every run).  Below 2^24 float32
    // accumulation is exact as well, so counts 0..4 may travel as MX fp4 (e2m1) operands: half the
    // operand bytes and twice the MFMA rate of int8 (gram.hip); counts 5..127 then form a secondary
    // int8 region.
    // kind 1: operands are 0/1 and K_ij <= n_levels * max_graph_nodes.
    const double bound = kind == GK_FEAT_MINSUM
                             ? (double)n_levels * (double)b->max_graph_nodes
                             : (double)n_levels * (double)b->max_graph_nodes * (double)b->max_graph_nodes;
    f->dtype = (kind == GK_FEAT_MINSUM || bound < 2147483647.0) ? 0 : 1;
    f->k_bound = bound;
    f->phi_fp4 = f->dtype == 0 && bound < 16777216.0 && !ctx->opt.gram_no_fp4;
    const int wide_above = kind == GK_FEAT_MINSUM ? 0x7fffffff : (f->dtype == 0 ? 127 : -1);
    const int prim_max = f->dtype != 0 ? -1 : (f->phi_fp4 ? 4 : 127);
    // ---- ShortestPath pair batch in histogram form: per-graph histograms of the distance matrices (features_gm.hip);
    // on a decline the pair items are materialised and the label-major builder below takes over
    if (b->is_pair_batch && b->sp_hist) {
        const bool hist_ok = n_levels == 1 && level_lo == 0 && kind == GK_FEAT_DOT && !ctx->opt.sp_no_hist;
        // Round 6: the operand type of a histogram job is decided ON THE DEVICE from the largest self similarity the histograms
        // leave (K_ij <= sqrt(K_ii K_jj) <= max K_ii: exact, where the a-priori bound is (pairs of the largest graph)^2 --
        // 1.4 x 10^8 at BASELINE config 4, whose largest K_ii is a few 10^5: fp4 + int8 instead of int8 only; 5.8 x 10^10 on the
        // COLLAB-like set: int8 instead of float64).  The counts are classified against 4 / 127 whatever the type turns out to be.
        const bool dyn = hist_ok && !(ctx->opt.sp_static_type & 1);
        const int static_dtype = f->dtype;
        const bool static_fp4 = f->phi_fp4;
        for (int attempt = 0; attempt < 2; ++attempt) {
            // attempt 1: a table of the one-workgroup-per-graph kernel overflowed in a job that had skipped the counter rows
            f->dyn_type = dyn;
            r = hist_ok ? gk_features_build_sp(ctx, b, f, dyn ? (ctx->opt.gram_no_fp4 ? 127 : 4) : prim_max, dyn ? 127 : wide_above, attempt == 1)
                        : GK_ERR_UNSUPPORTED;
            if (r == GK_OK) { *out = f; return GK_OK; }
            f->dyn_type = false, f->dtype = static_dtype, f->phi_fp4 = static_fp4, f->k_bound = bound;
            if (r != GK_ERR_UNSUPPORTED) return fail(r);
            for (void* p : f->arena)
                if (p) gk_dev_free(ctx, p);
            f->arena.clear();
            f->gm = false;
            if (gk_zero_async(ctx, f->meta, n_meta * 4) != GK_OK) return fail(GK_ERR_HIP);
            if (!hist_ok || ctx->opt.sp_no_rows || ctx->opt.sp_rows_all || b->sp_max_nodes > 128) break;   // the rows were in already
        }
        if ((r = gk_sp_materialise(ctx, b))) return fail(r);
    }
    // ---- graph batches with small graphs: the graph-major builder (features_gm.hip); it declines (row wider
    // than its LDS image) with GK_ERR_UNSUPPORTED and this builder takes over
    if (!b->is_pair_batch && V > 0 && b->max_graph_nodes <= (ctx->opt.gm_no_huge ? GM_MAX_NODES : GM_HUGE_MAX_NODES) && !ctx->opt.feat_no_gm) {
        r = gk_features_build_gm(ctx, b, f, n_levels, prim_max, wide_above);
        if (r == GK_OK) { *out = f; return GK_OK; }
        if (r != GK_ERR_UNSUPPORTED) return fail(r);      // incl. GK_ERR_RETRY: the queued relabel was unusable
        for (void* p : f->arena)
            if (p) gk_dev_free(ctx, p);
        f->arena.clear();
        f->gm = false;
        if (gk_zero_async(ctx, f->meta, n_meta * 4) != GK_OK) return fail(GK_ERR_HIP);
    }
    if (b->sr_pending > 0) {          // a queued stream relabel nobody has looked at yet (the graph-major builder did not run)
        std::vector<u32> hw((size_t)b->sr_pending * SR_CTL);
        if ((r = gk_readback(ctx, b->sr_ctl, hw.data(), b->sr_pending * SR_CTL))) return fail(r);
        if (gk_sr_collect(ctx, b, hw.data()) != GK_OK) return fail(GK_ERR_RETRY);
    }
    // ---- the level slots: a level only lists the nodes that can share a label (wl.hip: active-set
    // levels), a level that lists nothing adds one per node to the diagonal and nothing else
    FeatLevels P;
    P.L = 0, P.off[0] = 0, P.synth = -1;
    std::vector<int> slot_of_level(n_levels, -1);
    const bool hist0 = b->level0_hist && !b->is_pair_batch && V > 0 && level_lo == 0;
    const i64 L0 = b->n_labels0;
    for (int l = 0; l < n_levels && V > 0; ++l) {
        const int bl = level_lo + l;                 // the batch's level
        i64 nl = (size_t)bl < b->n_sorted.size() ? b->n_sorted[bl] : V;
        if (l == 0 && hist0) nl = L0 * N;           // items of the synthesized slot: (label, graph) cells
        if (nl == 0) continue;
        // the relabel skipped this level's label-grouped order (sort-free dictionary, wl.hip): build it now
        if (!(l == 0 && hist0) && (size_t)bl < b->perm_valid.size() && !b->perm_valid[bl] && (r = gk_batch_rebuild_order(ctx, b, bl)))
            return fail(r);
        const int j = P.L++;
        slot_of_level[l] = j;
        P.perm[j] = b->perm + (size_t)bl * V, P.lab[j] = b->labels + (size_t)bl * V;
        P.n[j] = nl, P.level[j] = l;
        if (l == 0 && hist0) P.synth = j;
        P.off[j + 1] = P.off[j] + round_up(nl, SF_TILE);
    }
    const i64 total = P.off[P.L];
    std::vector<u32> h(n_meta, 0);
    FeatArrays A{};
    if (total > 0) {
        i32** arrs[] = {&A.tri_pos, &A.tri_graph, &A.tri_run, &A.tstart, &A.colid, &A.wide, &A.low_all};
        for (i32** a : arrs) {
            if ((r = gk_dev_alloc(ctx, &q, (size_t)(total + P.L + 1) * 4))) return fail(r);
            *a = (i32*)q;
            f->arena.push_back(q);
        }
        Tmp<u64> flag(ctx);
        Tmp<u32> tri_of(ctx);
        if ((r = flag.alloc(total)) || (r = tri_of.alloc(total))) return fail(r);
        feat_flags_kernel<<<grid_for(total, 256), 256, 0, ctx->stream>>>(P, A, b->node_graph, flag.p);
        Tmp<u32> c0t(ctx), rows0(ctx);
        Tmp<u64> extra0(ctx);
        if (P.synth >= 0) {
            const int n_chunks = (int)cdiv(N, 1024);
            if ((r = c0t.alloc((size_t)(L0 * N))) || (r = rows0.alloc((size_t)(2 * L0 * n_chunks))) || (r = extra0.alloc((size_t)N))) return fail(r);
            feat_hist0_kernel<<<grid_for(N, 4), 256, 0, ctx->stream>>>(b->graph_ptr, b->labels, N, (int)L0, c0t.p, extra0.p, kind);
            feat_hist0_rows_kernel<<<dim3((unsigned)(L0 * n_chunks)), 1024, 0, ctx->stream>>>(c0t.p, N, n_chunks, rows0.p);
            feat_hist0_synth_kernel<<<dim3((unsigned)(L0 * n_chunks)), 1024, 0, ctx->stream>>>(
                c0t.p, rows0.p, N, n_chunks, (int)L0, A, P.off[P.synth] + P.synth, f->meta, P.level[P.synth], n_levels, prim_max,
                wide_above, kind);
        }
        feat_selfk_init_kernel<<<grid_for(N, 256), 256, 0, ctx->stream>>>(b->graph_ptr, f->selfk, N,
                                                                          n_levels - (P.synth >= 0 ? 1 : 0),
                                                                          P.synth >= 0 ? extra0.p : nullptr);
        TripleEmit te{P, A, b->node_graph, flag.p, tri_of.p, f->meta};
        if ((r = gk_scan_fn<u64, TripleEmit>(ctx, te, total, nullptr))) return fail(r);
        feat_count_kernel<<<grid_for(total, 256), 256, 0, ctx->stream>>>(P, A, tri_of.p, prim_max, wide_above, f->selfk, f->meta,
                                                                         n_levels, kind);
        if (kind == GK_FEAT_MINSUM) feat_runmax_kernel<<<grid_for(total, 256), 256, 0, ctx->stream>>>(P, A, f->meta);
        ColumnIds ci{P, A, f->meta, f->symmetric ? 1 : 0, (i32)n_fit, (i32)f->low_df, kind, n_levels};
        if ((r = gk_scan_fn<u64, ColumnIds>(ctx, ci, total, nullptr))) return fail(r);
        if (kind == GK_FEAT_DOT) {   // always queued (two launches) so that ONE read-back below sizes everything
            ColumnIdsByteWide cw{P, A, f->meta, n_levels};
            if ((r = gk_scan_fn<u64, ColumnIdsByteWide>(ctx, cw, total, nullptr))) return fail(r);
        }
        // one host sync: sizes of the dense operand
        if ((r = gk_readback(ctx, f->meta, h.data(), (int)n_meta))) return fail(r);
    } else if (V > 0) {      // no two nodes share a label at any level: K is its diagonal, n_levels per node
        feat_selfk_init_kernel<<<grid_for(N, 256), 256, 0, ctx->stream>>>(b->graph_ptr, f->selfk, N, n_levels, nullptr);
    }
    const int G = 3 * n_levels;
    f->nnz = 0;
    for (int l = 0; l < n_levels; ++l) f->nnz += h[META_T(l)];
    f->n_cols1 = h[G + 3];
    f->n_cols8 = h[4 * n_levels + 4];
    f->n_cols = f->n_cols1 + f->n_cols8;
    f->n_cols_wide = h[G + 2];
    f->max_count = 0;
    for (int q = 0; q < 64; ++q) f->max_count = std::max<i64>(f->max_count, h[4 * n_levels + 5 + q]);
    f->n_low_cols = h[G + 1];
    {   // per-level views for the Gram kernels
        u32 low_before = 0;
        for (int l = 0; l < n_levels; ++l) {
            LevelTriples& L = f->lev[l];
            const int j = slot_of_level[l];
            if (j < 0) continue;
            const i64 base = P.off[j] + j;
            L.tri_pos = A.tri_pos + base, L.tri_graph = A.tri_graph + base, L.tri_run = A.tri_run + base;
            L.tstart = A.tstart + base, L.colid = A.colid + base, L.wide = A.wide + base;
            L.low_runs = A.low_all + low_before;
            L.n_low = (i64)h[G + 4 + l] - (i64)low_before;
            low_before = h[G + 4 + l];
        }
    }
    // ---- operands: byte staging image (plain byte stores, no two writers per byte), then packed into
    // K-steps of 128 B per row: the Gram kernel is bound by the operand bytes it pulls through
    // L2 -> LDS (gram.hip), whole 128-byte lines per row move ~1.6x faster than half lines
    // (tools/micro/l2lds.hip), and fp4 halves the bytes per column
    const i64 n1p = round_up(f->n_cols1, f->phi_fp4 ? 256 : 128);
    i64 n8p = round_up(f->n_cols8, 128);
    if (n1p + n8p == 0) n8p = 128;            // at least one (all-zero) K-step
    f->k1_steps = (int)(n1p / (f->phi_fp4 ? 256 : 128)), f->k8_steps = (int)(n8p / 128);
    f->n_cols_pad = (f->phi_fp4 ? n1p / 2 : n1p) + n8p;     // BYTES per operand row
    f->n_rows_pad = round_up(N, 256) + 256;   // slack so that tile loads never need row guards
    const i64 ld_stage = n1p + n8p;
    Tmp<int8_t> stage(ctx);
    if ((r = stage.alloc((size_t)f->n_rows_pad * ld_stage))) return fail(r);
    if (gk_zero_async(ctx, stage.p, (size_t)f->n_rows_pad * ld_stage) != GK_OK) return fail(GK_ERR_HIP);
    if ((r = gk_dev_alloc(ctx, &q, (size_t)f->n_rows_pad * f->n_cols_pad))) return fail(r);
    f->phi = q;
    if (f->n_cols_wide > 0) {
        f->n_cols_wide_pad = round_up(f->n_cols_wide, 16);
        const size_t wb = (size_t)f->n_rows_pad * f->n_cols_wide_pad * 8;
        if ((r = gk_dev_alloc(ctx, &q, wb))) return fail(r);
        f->phi_w = (double*)q;
        if (gk_zero_async(ctx, f->phi_w, wb) != GK_OK) return fail(GK_ERR_HIP);
    }
    for (int l0 = 0; l0 < n_levels && V > 0; l0 += GK_PACK_LEVELS) {
        LevelPack S;
        S.n = 0, S.first[0] = 0;
        for (int l = l0; l < n_levels && l < l0 + GK_PACK_LEVELS; ++l) {
            LevelTriples& L = f->lev[l];
            if (h[META_T(l)] == 0 || !L.tri_pos) continue;
            S.tri_pos[S.n] = L.tri_pos, S.tri_graph[S.n] = L.tri_graph, S.tri_run[S.n] = L.tri_run;
            S.colid[S.n] = L.colid;
            S.first[S.n + 1] = S.first[S.n] + h[META_T(l)];
            ++S.n;
        }
        if (S.n == 0) continue;
        feat_scatter_mixed_kernel<<<grid_for(S.first[S.n], 256), 256, 0, ctx->stream>>>(
            S, stage.p, ld_stage, n1p, f->phi_w, f->n_cols_wide_pad, kind, f->phi_fp4 ? 1 : 0);
    }
    feat_pack_kernel<<<grid_for(f->n_rows_pad * (f->n_cols_pad / 4), 256), 256, 0, ctx->stream>>>(
        stage.p, ld_stage, n1p, n8p, f->phi_fp4 ? 1 : 0, (int8_t*)f->phi, f->n_cols_pad, f->n_rows_pad);
    if (hipGetLastError() != hipSuccess) {
        gk_set_error("gk_features_build: kernel launch failed");
        return fail(GK_ERR_HIP);
    }
    *out = f;
    return GK_OK;
}

extern "C" int gk_features_info(gk_feat* f, int64_t* n_cols_kept, int64_t* n_cols_low, int64_t* nnz,
                                int64_t* max_count, int* dtype) {
    GK_ARG(f, "gk_features_info: null");
    if (n_cols_kept) *n_cols_kept = f->n_cols + f->n_cols_wide;
    if (n_cols_low) *n_cols_low = f->n_low_cols;
    if (nnz) *nnz = f->nnz;
    if (max_count) *max_count = f->max_count;
    if (dtype) *dtype = (f->n_cols_wide > 0 && f->n_cols == 0) ? 1 : 0;   // 1: only the float64 operand is in use
    return GK_OK;
}

extern "C" int gk_features_operand(gk_feat* f, int* fp4, int* k_steps_primary, int* k_steps_i8_secondary, int64_t* n_cols_f64) {
    GK_ARG(f, "gk_features_operand: null");
    if (fp4) *fp4 = f->phi_fp4 ? 1 : 0;
    if (k_steps_primary) *k_steps_primary = f->k1_steps;
    if (k_steps_i8_secondary) *k_steps_i8_secondary = f->k8_steps;
    if (n_cols_f64) *n_cols_f64 = f->n_cols_wide;
    return GK_OK;
}

extern "C" int gk_features_operand_rows(gk_feat* f, void** out_phi, void** out_phi_right, int64_t* out_row_bytes, int64_t* out_n_rows,
                                        int64_t* out_own_lo, int64_t* out_own_hi) {
    GK_ARG(f, "gk_features_operand_rows: null");
    if (out_phi) *out_phi = f->phi;
    if (out_phi_right) *out_phi_right = f->phi_r;
    if (out_row_bytes) *out_row_bytes = f->n_cols_pad;
    if (out_n_rows) *out_n_rows = f->n_graphs;
    if (out_own_lo) *out_own_lo = f->own_lo;
    if (out_own_hi) *out_own_hi = f->own_hi;
    return GK_OK;
}

extern "C" int gk_memcpy_dev(gk_ctx* ctx, void* dst_dev, const void* src_dev, uint64_t bytes) {
    GK_ARG(ctx && (bytes == 0 || (dst_dev && src_dev)), "gk_memcpy_dev: null argument");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    if (bytes) GK_HIP_CHECK(hipMemcpyAsync(dst_dev, src_dev, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return GK_OK;
}

extern "C" int gk_features_selfk(gk_ctx* ctx, gk_feat* f, double* out_selfk) {
    GK_ARG(ctx && f && out_selfk, "gk_features_selfk: null argument");
    std::vector<u64> h(f->n_graphs);
    GK_HIP_CHECK(hipMemcpyAsync(h.data(), f->selfk, (size_t)f->n_graphs * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (i64 i = 0; i < f->n_graphs; ++i) out_selfk[i] = (double)h[i];
    return GK_OK;
}

static int debug_operand(gk_ctx* ctx, gk_feat* f, const void* operand, double* out_phi) {
    const i64 N = f->n_graphs, D = f->n_cols, ld = f->n_cols_pad, n1 = f->n_cols1;
    const i64 prim0 = (i64)f->k8_steps * 128;      // first byte of the primary region
    static const double fp4_value[16] = {0, 0.5, 1, 1.5, 2, 3, 4, 6, -0.0, -0.5, -1, -1.5, -2, -3, -4, -6};
    std::vector<unsigned char> h((size_t)N * ld);
    GK_HIP_CHECK(hipMemcpyAsync(h.data(), operand, h.size(), hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (i64 i = 0; i < N; ++i)
        for (i64 j = 0; j < D; ++j) {       // columns: primary class first, then the secondary int8 class
            const unsigned char* row = h.data() + i * ld;
            if (j >= n1) out_phi[i * D + j] = (double)((const int8_t*)row)[j - n1];
            else if (f->phi_fp4) out_phi[i * D + j] = fp4_value[(row[prim0 + (j >> 1)] >> (4 * (j & 1))) & 15];
            else out_phi[i * D + j] = (double)((const int8_t*)row)[prim0 + j];
        }
    return GK_OK;
}

extern "C" int gk_features_debug_phi(gk_ctx* ctx, gk_feat* f, double* out_phi) {
    GK_ARG(ctx && f && out_phi, "gk_features_debug_phi: null argument");
    return debug_operand(ctx, f, f->phi, out_phi);
}

extern "C" int gk_features_debug_phi_right(gk_ctx* ctx, gk_feat* f, double* out_phi, int* split_parts) {
    GK_ARG(ctx && f && out_phi, "gk_features_debug_phi_right: null argument");
    if (split_parts) *split_parts = f->split_parts;
    return debug_operand(ctx, f, f->phi_r ? f->phi_r : f->phi, out_phi);
}

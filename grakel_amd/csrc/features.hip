// Label-count features (the VertexHistogram of every WL level) on gfx950.
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
// kind 1 (histogram intersection, K_ij = sum_l min(c_il, c_jl); WL-OA,
// weisfeiler_lehman_optimal_assignment.py:268-279) stays on the same integer GEMM through the
// UNARY expansion  min(a, b) = sum_{t>=1} [a >= t][b >= t]:  a dense label column whose largest
// count is m becomes m 0/1 columns ([c>=1], [c>=2], ...), so Phi_s . Phi_s^T IS the min-sum,
// exactly, on the int8 MFMA path; selfk[g] = sum of counts (= nodes x levels).
#include "common.h"
#include "scan_fn.h"
#include <stdlib.h>

static inline dim3 grid_for(i64 n, int t) { return dim3((unsigned)(n > 0 ? cdiv(n, t) : 1)); }

// meta layout (u32): per level l: [3l+0]=T (triples) [3l+1]=R (label runs) [3l+2]=cols kept so far
// (cumulative INCLUDING level l); globals at [3*n_levels + 0]=max count
#define META_T(l) (3 * (l) + 0)
#define META_R(l) (3 * (l) + 1)
#define META_C(l) (3 * (l) + 2)

__global__ void feat_flags_kernel(const i32* __restrict__ perm, const i32* __restrict__ lab,
                                  const i32* __restrict__ node_graph, u64* __restrict__ flag, i64 n,
                                  i32* __restrict__ wide) {
    i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    wide[k] = 0;      // per-run flag / maximum filled in by the count kernels (runs <= items)
    i32 v = perm[k];
    u64 f = 0x100000001ull;
    if (k > 0) {
        i32 p = perm[k - 1];
        bool lh = lab[v] != lab[p];
        bool sh = lh || node_graph[v] != node_graph[p];
        f = ((u64)lh << 32) | (u64)sh;
    }
    flag[k] = f;
}

// packed (label-head, subrun-head) flags -> triple ids / run ids, triples emitted in the scan
struct TripleEmit {
    const i32* perm; const i32* node_graph; const u64* flag;
    i32* tri_pos; i32* tri_graph; i32* tri_run; i32* tstart; u32* tri_of;   // tri_of[k] = triple of position k
    u32* meta; int level; i64 n;
    __device__ __forceinline__ u64 value(i64 k) const { return flag[k]; }
    __device__ __forceinline__ void emit(i64 k, u64 f, u64 s) const {
        const i32 t = (i32)(u32)(s & 0xffffffffull) - 1;
        const i32 r = (i32)(u32)(s >> 32) - 1;
        tri_of[k] = (u32)t;
        if (f & 1ull) {
            tri_pos[t] = (i32)k;
            tri_graph[t] = node_graph[perm[k]];
            tri_run[t] = r;
        }
        if (f >> 32) tstart[r] = t;
        if (k == n - 1) {   // sentinels + counts
            tri_pos[t + 1] = (i32)n;
            tstart[r + 1] = t + 1;
            meta[META_T(level)] = (u32)(t + 1);
            meta[META_R(level)] = (u32)(r + 1);
        }
    }
    __device__ __forceinline__ void finish(u64) const {}
};

// per node: add the count of its (label,graph) triple to node_acc[v]; sum over the nodes of a
// graph of these counts == sum over its triples of count^2, so the exact self similarity needs
// no atomics (each node is written once per level).  Also tracks the largest count.
__global__ void feat_count_kernel(const i32* __restrict__ perm, const u32* __restrict__ tri_of,
                                  const i32* __restrict__ tri_pos, const i32* __restrict__ tri_run,
                                  i32* __restrict__ wide, int wide_above, u32* __restrict__ node_acc,
                                  u32* __restrict__ meta, int level, int n_levels, i64 n, int kind,
                                  const i32* __restrict__ node_graph, u32* __restrict__ covered) {
    __shared__ u32 wmax[4];
    i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = 0;
    if (k < n) {
        const i32 t = (i32)tri_of[k];
        c = (u32)(tri_pos[t + 1] - tri_pos[t]);
        if ((int)c > wide_above) wide[tri_run[t]] = 1;       // benign race: all writers store 1
        i32 v = perm[k];
        node_acc[v] = (level == 0 ? 0u : node_acc[v]) + (kind ? 1u : c);   // min(c,c) summed == #nodes
        // partial level (only the nodes that may share a label are listed): count them per graph,
        // every other node of the graph owns its label and adds exactly 1 to the self similarity
        if (covered) atomicAdd(&covered[node_graph[v]], 1u);
    }
    for (int off = 32; off > 0; off >>= 1) {
        u32 o = __shfl_down(c, off, 64);
        c = o > c ? o : c;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 m = wmax[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = wmax[i] > m ? wmax[i] : m;
        // plain read as a filter: once the maximum is established almost no block issues the atomic
        if (m > meta[3 * n_levels]) atomicMax(&meta[3 * n_levels], m);
    }
}

// kind 1: wide[r] = largest count of label run r (the number of unary columns it expands to).
// Triples are run-major, so a wave usually sees one run: one atomic per wave then.
__global__ void feat_runmax_kernel(const i32* __restrict__ tri_pos, const i32* __restrict__ tri_run,
                                   const u32* __restrict__ meta, int level, i32* __restrict__ runmax) {
    const u32 Tn = meta[META_T(level)];
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    i32 r = -1, c = 0;
    if (t < Tn) r = tri_run[t], c = tri_pos[t + 1] - tri_pos[t];
    const i32 r0 = __shfl(r, 0, 64);
    if (__all(r == r0 || r < 0)) {
        if (r0 < 0) return;
        for (int off = 32; off > 0; off >>= 1) {
            i32 o = __shfl_down(c, off, 64);
            c = o > c ? o : c;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(&runmax[r0], c);
    } else if (r >= 0) {
        atomicMax(&runmax[r], c);
    }
}

// one wave per graph: selfk[g] = sum of node_acc over the graph's (contiguous) nodes
__global__ void feat_selfk_kernel(const i32* __restrict__ graph_ptr, const u32* __restrict__ node_acc,
                                  u64* __restrict__ selfk, i64 n_graphs, const u32* __restrict__ covered,
                                  int n_partial) {
    const i64 g = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (g >= n_graphs) return;
    u64 s = 0;
    for (i32 v = graph_ptr[g] + lane; v < graph_ptr[g + 1]; v += 64) s += node_acc[v];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    // n_partial levels listed only `covered[g]` (summed) of the graph's nodes: the others add 1 each
    if (lane == 0)
        selfk[g] = s + (u64)n_partial * (u64)(graph_ptr[g + 1] - graph_ptr[g]) - (u64)(covered ? covered[g] : 0u);
}

// Column classes per label run, fused into the prefix sum:
//   dense (colid >= 0) : occurs in >= low_df graphs -> a column of the MFMA operand Phi_s
//   low   (colid = -2) : useful but rare (df < low_df): its df*(df-1) pair products are added
//                        to K by gram_low_kernel after the GEMM -- a column with df graphs
//                        costs N^2 MACs in the dense product but only df^2 updates here
//   dead  (colid = -1) : cannot touch an off-diagonal entry (graph-unique / one-sided)
// scan value packs (low << 32 | dense) so one pass yields both running counts.
struct ColumnIds {
    const i32* tstart; const i32* tri_graph; i32* colid; u32* meta; int level; int symmetric; i32 n_fit;
    i32 low_df; i32* low_runs; const i32* wide; int kind; int n_levels;
    __device__ __forceinline__ u64 value(i64 r) const {
        if (r >= (i64)meta[META_R(level)]) return 0ull;
        const i32 t0 = tstart[r], t1 = tstart[r + 1];
        bool useful;
        if (symmetric) useful = (t1 - t0) >= 2;
        else useful = tri_graph[t0] < n_fit && tri_graph[t1 - 1] >= n_fit;
        if (!useful) return 0ull;
        if ((t1 - t0) < low_df) return 1ull << 32;
        if (kind) return (u64)(u32)wide[r];       // unary expansion: one 0/1 column per count level
        return wide[r] ? (1ull << 63) : 1ull;     // bit 63: dense but not int8-able (toggles only itself)
    }
    __device__ __forceinline__ void emit(i64 r, u64 v, u64 incl) const {
        const u32 base = level > 0 ? meta[META_C(level - 1)] : 0u;
        const u32 rare = (u32)(v >> 32) & 0x7fffffffu;
        const u32 width = (u32)(v & 0xffffffffull);     // 1 (kind 0) or the run's largest count (kind 1)
        colid[r] = width ? (i32)(base + (u32)(incl & 0xffffffffull) - width)
                         : ((v >> 63) ? -3 : (rare ? -2 : -1));
        if (rare) low_runs[((u32)(incl >> 32) & 0x7fffffffu) - 1] = (i32)r;   // compact list for gram_low_kernel
    }
    // running column totals (emit only reads the PREVIOUS level's entry, so no block races with this)
    __device__ __forceinline__ void finish(u64 t) const {
        meta[META_C(level)] = (level > 0 ? meta[META_C(level - 1)] : 0u) + (u32)(t & 0xffffffffull);
        meta[3 * n_levels + 1] += (u32)(t >> 32) & 0x7fffffffu;           // low columns over all levels
        meta[3 * n_levels + 4 + level] = (u32)(t >> 32) & 0x7fffffffu;     // ... and of this level
    }
};

__global__ void feat_colbase_kernel(u32* __restrict__ meta, const u64* __restrict__ total, int level, int n_levels) {
    const u64 t = total ? *total : 0ull;      // null: a level without any shared label
    meta[META_C(level)] = (level > 0 ? meta[META_C(level - 1)] : 0u) + (u32)(t & 0xffffffffull);
    meta[3 * n_levels + 1] += (u32)(t >> 32) & 0x7fffffffu;     // low columns over all levels
    meta[3 * n_levels + 4 + level] = (u32)(t >> 32) & 0x7fffffffu;   // ... and of this level
}

// second pass (only when some count exceeded 127): ids for the float64 side operand
struct ColumnIdsWide {
    i32* colid; u32* meta; int n_levels;
    __device__ __forceinline__ u32 value(i64 r) const { return colid[r] == -3 ? 1u : 0u; }
    __device__ __forceinline__ void emit(i64 r, u32 w, u32 incl) const {
        if (w) colid[r] = -4 - (i32)(meta[3 * n_levels + 2] + incl - 1);
    }
    __device__ __forceinline__ void finish(u32) const {}     // feat_widebase_kernel: emit reads the base it would update
};

__global__ void feat_widebase_kernel(u32* __restrict__ meta, const u32* __restrict__ total, int n_levels) {
    meta[3 * n_levels + 2] += *total;
}

// all levels in one launch: P.first = prefix of the per-level triple counts
__global__ void feat_scatter_mixed_kernel(const LevelPack P, int8_t* __restrict__ phi, i64 ld,
                                          double* __restrict__ phi_w, i64 ldw, int kind) {
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
        for (i32 q = 0; q < cnt; ++q) row[q] = 1;           // [count >= q+1]
    } else if (c >= 0) phi[(i64)tri_graph[t] * ld + c] = (int8_t)cnt;
    else if (c <= -4) phi_w[(i64)tri_graph[t] * ldw + (-4 - c)] = (double)cnt;
}

template <typename T>
__global__ void feat_scatter_kernel(const i32* __restrict__ tri_pos, const i32* __restrict__ tri_graph,
                                    const i32* __restrict__ tri_run, const i32* __restrict__ colid,
                                    const u32* __restrict__ meta, int level, T* __restrict__ phi, i64 ld) {
    const u32 Tn = meta[META_T(level)];
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tn) return;
    i32 c = colid[tri_run[t]];
    if (c < 0) return;
    phi[(i64)tri_graph[t] * ld + c] = (T)(tri_pos[t + 1] - tri_pos[t]);
}

extern "C" int gk_features_destroy(gk_feat* f) {
    if (!f) return GK_OK;
    gk_ctx* ctx = f->ctx;
    for (auto& L : f->lev) {
        void* ptrs[] = {L.tri_pos, L.tri_graph, L.tri_run, L.tstart, L.colid, L.low_runs, L.wide};
        for (void* p : ptrs)
            if (p) gk_dev_free(ctx, p);
    }
    void* ptrs[] = {f->meta, f->selfk, f->phi, f->phi_w, f->K};
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
    GK_ARG(ctx && b && out, "gk_features_build: null argument");
    GK_ARG(kind == GK_FEAT_DOT || kind == GK_FEAT_MINSUM, "gk_features_build_ex: unknown kind");
    GK_ARG(n_levels >= 1 && n_levels <= (b->n_levels > 0 ? b->n_levels : 1),
           "gk_features_build: levels not computed (call gk_wl_relabel first)");
    GK_ARG(n_fit >= 1 && n_fit <= b->n_graphs, "gk_features_build: bad n_fit");
    GK_ARG(b->n_levels > 0, "gk_features_build: batch has no label-grouped order (call gk_wl_relabel with n_iter>=0)");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ProfScope prof(ctx, "features");
    const i64 V = b->n_nodes, N = b->n_graphs;
    gk_feat* f = new gk_feat();
    f->ctx = ctx, f->batch = b, f->n_levels = n_levels, f->n_graphs = N, f->n_fit = n_fit, f->n_nodes = V;
    f->symmetric = (n_fit == N);
    f->kind = kind;
    f->lev.resize(n_levels);
    auto fail = [&](int r) { gk_features_destroy(f); return r; };
    int r;
    void* q = nullptr;
    const size_t n_meta = 4 * (size_t)n_levels + 4;
    if ((r = gk_dev_alloc(ctx, &q, n_meta * 4))) return fail(r);
    f->meta = (u32*)q;
    if ((r = gk_dev_alloc(ctx, &q, (size_t)N * 8))) return fail(r);
    f->selfk = (u64*)q;
    if (gk_zero_async(ctx, f->meta, n_meta * 4) != GK_OK ||
        (V == 0 && gk_zero_async(ctx, f->selfk, (size_t)N * 8) != GK_OK)) {   // V > 0: feat_selfk_kernel writes it all
        gk_set_error("gk_features_build: memset failed");
        return fail(GK_ERR_HIP);
    }
    Tmp<u64> flag(ctx);
    Tmp<u32> cflag(ctx), node_acc(ctx), covered(ctx);
    Tmp<u64> ctotal64(ctx);
    if ((r = flag.alloc(V)) || (r = cflag.alloc(V)) || (r = ctotal64.alloc(1)) || (r = node_acc.alloc(V)))
        return fail(r);
    // levels whose label-grouped order only lists the nodes that can share a label (active-set
    // relabelling, wl.hip): the feature pass runs over that prefix alone
    int n_partial = 0;
    auto level_items = [&](int l) -> i64 { return (size_t)l < b->n_sorted.size() ? b->n_sorted[l] : V; };
    for (int l = 0; l < n_levels; ++l) n_partial += level_items(l) < V ? 1 : 0;
    if (n_partial > 0) {
        if ((r = covered.alloc(N))) return fail(r);
        if (gk_zero_async(ctx, covered.p, (size_t)N * 4) != GK_OK) return fail(GK_ERR_HIP);
    }
    {
        const char* e = getenv("GK_LOW_DF");     // df threshold below which a column leaves the dense operand
        f->low_df = e ? atoi(e) : 32;
        if (f->low_df < 2) f->low_df = 2;        // 2 == everything useful is dense
    }
    if (!b->graph_ptr) {
        gk_set_error("gk_features_build: batch has no graph_ptr");
        return fail(GK_ERR_STATE);
    }
    // int8 operands need counts <= 127 and every Gram entry < 2^31:
    // K_ij <= sqrt(K_ii K_jj) <= n_levels * max_graph_nodes^2.  If the bound fails everything
    // dense goes to the float64 operand (wide_above = -1 flags every run).
    const double bound = (double)n_levels * (double)b->max_graph_nodes * (double)b->max_graph_nodes;
    // kind 1: operands are 0/1 and K_ij <= n_levels * max_graph_nodes, always int8-able.
    f->dtype = (kind == GK_FEAT_MINSUM || bound < 2147483647.0) ? 0 : 1;
    const int wide_above = kind == GK_FEAT_MINSUM ? 0x7fffffff : (f->dtype == 0 ? 127 : -1);
    for (int l = 0; l < n_levels && V > 0; ++l) {
        LevelTriples& L = f->lev[l];
        const i64 nl = level_items(l);       // items of this level's label-grouped order that matter
        if (nl == 0) {                        // every node owns its label: no triples, no columns
            feat_colbase_kernel<<<1, 1, 0, ctx->stream>>>(f->meta, nullptr, l, n_levels);
            continue;
        }
        i32** arrs[] = {&L.tri_pos, &L.tri_graph, &L.tri_run, &L.tstart, &L.colid, &L.low_runs, &L.wide};
        for (i32** a : arrs) {
            if ((r = gk_dev_alloc(ctx, &q, (size_t)(nl + 1) * 4))) return fail(r);
            *a = (i32*)q;
        }
        const i32* lab = b->labels + (size_t)l * V;
        const i32* perm = b->perm + (size_t)l * V;
        feat_flags_kernel<<<grid_for(nl, 256), 256, 0, ctx->stream>>>(perm, lab, b->node_graph, flag.p, nl, L.wide);
        TripleEmit te{perm, b->node_graph, flag.p, L.tri_pos, L.tri_graph, L.tri_run, L.tstart, cflag.p,
                      f->meta, l, nl};
        if ((r = gk_scan_fn<u64, TripleEmit>(ctx, te, nl, nullptr))) return fail(r);
        feat_count_kernel<<<grid_for(nl, 256), 256, 0, ctx->stream>>>(perm, cflag.p, L.tri_pos, L.tri_run, L.wide, wide_above,
                                                                      node_acc.p, f->meta, l, n_levels, nl, kind,
                                                                      b->node_graph, nl < V ? covered.p : nullptr);
        if (kind == GK_FEAT_MINSUM)
            feat_runmax_kernel<<<grid_for(nl, 256), 256, 0, ctx->stream>>>(L.tri_pos, L.tri_run, f->meta, l, L.wide);
        ColumnIds ci{L.tstart, L.tri_graph, L.colid, f->meta, l, f->symmetric ? 1 : 0, (i32)n_fit, (i32)f->low_df,
                     L.low_runs, L.wide, kind, n_levels};
        if ((r = gk_scan_fn<u64, ColumnIds>(ctx, ci, nl, nullptr))) return fail(r);
    }
    if (V > 0)
        feat_selfk_kernel<<<grid_for(N * 64, 256), 256, 0, ctx->stream>>>(b->graph_ptr, node_acc.p, f->selfk, N,
                                                                          n_partial > 0 ? covered.p : nullptr, n_partial);
    // one host sync: sizes of the dense operand
    std::vector<u32> h(n_meta);
    if (hipMemcpyAsync(h.data(), f->meta, n_meta * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        gk_set_error("gk_features_build: %s", hipGetErrorString(hipGetLastError()));
        return fail(GK_ERR_HIP);
    }
    f->nnz = 0;
    for (int l = 0; l < n_levels; ++l) f->nnz += h[META_T(l)];
    f->n_cols = V > 0 ? h[META_C(n_levels - 1)] : 0;
    f->max_count = h[3 * n_levels];
    f->n_low_cols = h[3 * n_levels + 1];
    for (int l = 0; l < n_levels; ++l) f->lev[l].n_low = h[3 * n_levels + 4 + l];
    // ids of the float64 side operand (second pass, only when a dense column is too wide for int8)
    f->n_cols_wide = 0;
    if (V > 0 && kind == GK_FEAT_DOT && (f->max_count > 127 || f->dtype == 1)) {
        Tmp<u32> wtotal(ctx);
        if ((r = wtotal.alloc(1))) return fail(r);
        for (int l = 0; l < n_levels; ++l) {
            if (level_items(l) == 0) continue;
            ColumnIdsWide cw{f->lev[l].colid, f->meta, n_levels};
            if ((r = gk_scan_fn<u32, ColumnIdsWide>(ctx, cw, level_items(l), wtotal.p))) return fail(r);
            feat_widebase_kernel<<<1, 1, 0, ctx->stream>>>(f->meta, wtotal.p, n_levels);
        }
        u32 hw = 0;
        if (hipMemcpyAsync(&hw, f->meta + 3 * n_levels + 2, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            gk_set_error("gk_features_build: %s", hipGetErrorString(hipGetLastError()));
            return fail(GK_ERR_HIP);
        }
        f->n_cols_wide = hw;
    }
    f->n_cols_pad = round_up(f->n_cols > 0 ? f->n_cols : 1, 128);
    f->n_rows_pad = round_up(N, 256) + 256;   // slack so that tile loads never need row guards
    const size_t phi_bytes = (size_t)f->n_rows_pad * f->n_cols_pad;
    if ((r = gk_dev_alloc(ctx, &q, phi_bytes))) return fail(r);
    f->phi = q;
    if (gk_zero_async(ctx, f->phi, phi_bytes) != GK_OK) return fail(GK_ERR_HIP);
    if (f->n_cols_wide > 0) {
        f->n_cols_wide_pad = round_up(f->n_cols_wide, 16);
        const size_t wb = (size_t)f->n_rows_pad * f->n_cols_wide_pad * 8;
        if ((r = gk_dev_alloc(ctx, &q, wb))) return fail(r);
        f->phi_w = (double*)q;
        if (gk_zero_async(ctx, f->phi_w, wb) != GK_OK) return fail(GK_ERR_HIP);
    }
    for (int l0 = 0; l0 < n_levels && V > 0; l0 += GK_PACK_LEVELS) {
        LevelPack P;
        P.n = 0, P.first[0] = 0;
        for (int l = l0; l < n_levels && l < l0 + GK_PACK_LEVELS; ++l) {
            LevelTriples& L = f->lev[l];
            if (h[META_T(l)] == 0 || !L.tri_pos) continue;
            P.tri_pos[P.n] = L.tri_pos, P.tri_graph[P.n] = L.tri_graph, P.tri_run[P.n] = L.tri_run;
            P.colid[P.n] = L.colid;
            P.first[P.n + 1] = P.first[P.n] + h[META_T(l)];
            ++P.n;
        }
        if (P.n == 0) continue;
        feat_scatter_mixed_kernel<<<grid_for(P.first[P.n], 256), 256, 0, ctx->stream>>>(
            P, (int8_t*)f->phi, f->n_cols_pad, f->phi_w, f->n_cols_wide_pad, kind);
    }
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

extern "C" int gk_features_selfk(gk_ctx* ctx, gk_feat* f, double* out_selfk) {
    GK_ARG(ctx && f && out_selfk, "gk_features_selfk: null argument");
    std::vector<u64> h(f->n_graphs);
    GK_HIP_CHECK(hipMemcpyAsync(h.data(), f->selfk, (size_t)f->n_graphs * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (i64 i = 0; i < f->n_graphs; ++i) out_selfk[i] = (double)h[i];
    return GK_OK;
}

extern "C" int gk_features_debug_phi(gk_ctx* ctx, gk_feat* f, double* out_phi) {
    GK_ARG(ctx && f && out_phi, "gk_features_debug_phi: null argument");
    const i64 N = f->n_graphs, D = f->n_cols, ld = f->n_cols_pad;
    std::vector<unsigned char> h((size_t)N * ld);
    GK_HIP_CHECK(hipMemcpyAsync(h.data(), f->phi, h.size(), hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (i64 i = 0; i < N; ++i)
        for (i64 j = 0; j < D; ++j)
            out_phi[i * D + j] = (double)((const int8_t*)h.data())[i * ld + j];
    return GK_OK;
}

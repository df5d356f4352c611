// ShortestPath kernel on gfx950: batched all-pairs shortest paths and the (l_u, l_v, d)
// pair dictionary (reference: grakel/graph.py:588-687,1767-1794 and
// grakel/kernels/shortest_path.py:412-499,510-511).
//
//   sp_fw_kernel     one workgroup per graph, the n x n distance matrix lives in LDS
//                    (up to 160 KiB: n <= 200), n barrier-separated Floyd-Warshall sweeps.
//                    LDS/VALU-bound (sum n^3 min-plus ops), not HBM-bound.
//   sp_relax_kernel  graphs that do not fit LDS: one workgroup per (graph, source), the
//                    distance ROW lives in LDS, edge relaxation sweeps until fixpoint.
//   sp_emit_kernel   every ordered pair u != v with finite d becomes an item with the exact
//                    64-bit key (l_u, l_v, d); items are graph-major, so the stable key sort
//                    of the shared dictionary code leaves (key, graph) runs = Phi triples.
// Distances are exact int32 sums of positive integer edge weights (unit by default).
#include "common.h"
#include <stdlib.h>

#define SP_INF 0x3f000000
#define SP_THREADS 256
#define SP_FW_MAX_N 200          // 200*201*4 B = 160.8 KB > 160 KiB? -> see sp_fw_cap()
#define SP_ROW_MAX_N 32768

int gk_dictionary_from_keys(gk_ctx* ctx, const u64* keys, i64 n, int key_bits, i32* lab, i32* perm, u32* count_dev);

static inline dim3 grid_for(i64 n, int t) { return dim3((unsigned)(n > 0 ? cdiv(n, t) : 1)); }

static int sp_fw_cap() {
    // largest n with n*(n|1)*4 bytes <= 160 KiB
    int n = 1;
    while ((i64)(n + 1) * ((n + 1) | 1) * 4 <= 160 * 1024) ++n;
    return n;
}

__global__ void sp_sq_kernel(const i32* __restrict__ graph_ptr, u64* __restrict__ sq, i64 n_graphs) {
    i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_graphs) {
        u64 n = (u64)(graph_ptr[g + 1] - graph_ptr[g]);
        sq[g] = n * n;
    }
}

__device__ __forceinline__ void block_count_max(u32 cnt, u32 mx, u32* pair_count_g, u32* maxd) {
    __shared__ u32 rc[16], rm[16];
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off, 64);
        u32 o = __shfl_down(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) { rc[threadIdx.x >> 6] = cnt; rm[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 c = 0, m = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { c += rc[i]; m = rm[i] > m ? rm[i] : m; }
        atomicAdd(pair_count_g, c);
        if (m) atomicMax(maxd, m);
    }
}

// One workgroup per graph with n in (n_lo, n_hi]; blockDim = 256 for small graphs, 1024 for the
// large ones: a single 4-wave workgroup leaves one wave per SIMD and cannot hide the ~100-cycle
// LDS latency of the relaxation (measured 3.3 us per pivot at n = 110), 16 waves can.
__global__ __launch_bounds__(1024) void sp_fw_kernel(
    const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx,
    const i32* __restrict__ w, const u64* __restrict__ dist_ptr, i32* __restrict__ dist,
    u32* __restrict__ pair_count, u32* __restrict__ maxd, int n_lo, int n_hi) {
    extern __shared__ __attribute__((aligned(16))) i32 d[];
    const int g = blockIdx.x, tid = threadIdx.x, NT = blockDim.x, NW = blockDim.x >> 6;
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0;
    if (n <= n_lo || n > n_hi) return;
    const int ld = n | 1;
    for (int idx = tid; idx < n * ld; idx += NT) d[idx] = SP_INF;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const i32 e0 = row_ptr[v0 + i], e1 = row_ptr[v0 + i + 1];
        for (i32 e = e0; e < e1; ++e) {
            int j = col_idx[e] - v0;
            i32 wt = w ? w[e] : 1;
            if (wt < d[i * ld + j]) d[i * ld + j] = wt;
        }
        d[i * ld + i] = 0;                      // np.fill_diagonal(dist, 0): graph.py:1786
    }
    __syncthreads();
    // Thread (tx, ty): columns j = tx, tx+64, ...; rows i = ty, ty+NW, ...  Per pivot k the row
    // d[k][j] is read once into a register and four rows are relaxed at a time with independent
    // LDS loads (a one-row-at-a-time loop is a chain of dependent LDS round trips).
    const int tx = tid & 63, ty = tid >> 6;
    for (int k = 0; k < n; ++k) {
        for (int j = tx; j < n; j += 64) {
            const i32 dkj = d[k * ld + j];
            if (dkj < SP_INF) {
                int i = ty;
                for (; i + 3 * NW < n; i += 4 * NW) {
                    const int i1 = i + NW, i2 = i + 2 * NW, i3 = i + 3 * NW;
                    const i32 a0 = d[i * ld + k], a1 = d[i1 * ld + k], a2 = d[i2 * ld + k], a3 = d[i3 * ld + k];
                    const i32 c0 = d[i * ld + j], c1 = d[i1 * ld + j], c2 = d[i2 * ld + j], c3 = d[i3 * ld + j];
                    const i32 u0 = a0 + dkj, u1 = a1 + dkj, u2 = a2 + dkj, u3 = a3 + dkj;
                    if (u0 < c0) d[i * ld + j] = u0;
                    if (u1 < c1) d[i1 * ld + j] = u1;
                    if (u2 < c2) d[i2 * ld + j] = u2;
                    if (u3 < c3) d[i3 * ld + j] = u3;
                }
                for (; i < n; i += NW) {
                    const i32 u = d[i * ld + k] + dkj;
                    if (u < d[i * ld + j]) d[i * ld + j] = u;
                }
            }
        }
        __syncthreads();
    }
    u32 cnt = 0, mx = 0;
    i32* out = dist + dist_ptr[g];
    for (int i = ty; i < n; i += NW)
        for (int j = tx; j < n; j += 64) {
            i32 x = d[i * ld + j];
            out[i * n + j] = x;
            if (i != j && x < SP_INF) { ++cnt; mx = (u32)x > mx ? (u32)x : mx; }
        }
    block_count_max(cnt, mx, &pair_count[g], maxd);
}

// grid (n_graphs, max_n): block (g, src) for graphs larger than the Floyd-Warshall LDS cap
__global__ __launch_bounds__(SP_THREADS) void sp_relax_kernel(
    const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx,
    const i32* __restrict__ w, const u64* __restrict__ dist_ptr, i32* __restrict__ dist,
    u32* __restrict__ pair_count, u32* __restrict__ maxd, int cap) {
    extern __shared__ __attribute__((aligned(16))) i32 row[];
    __shared__ int changed;
    const int g = blockIdx.x, src = blockIdx.y, tid = threadIdx.x;
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0;
    if (n <= cap || src >= n) return;
    for (int i = tid; i < n; i += SP_THREADS) row[i] = i == src ? 0 : SP_INF;
    __syncthreads();
    for (int sweep = 0; sweep < n; ++sweep) {
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int u = tid; u < n; u += SP_THREADS) {
            const i32 du = row[u];
            if (du < SP_INF) {
                const i32 e0 = row_ptr[v0 + u], e1 = row_ptr[v0 + u + 1];
                for (i32 e = e0; e < e1; ++e) {
                    int v = col_idx[e] - v0;
                    i32 nd = du + (w ? w[e] : 1);
                    if (nd < row[v]) { atomicMin(&row[v], nd); changed = 1; }
                }
            }
        }
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    u32 cnt = 0, mx = 0;
    i32* out = dist + dist_ptr[g] + (u64)src * n;
    for (int j = tid; j < n; j += SP_THREADS) {
        i32 x = row[j];
        out[j] = x;
        if (j != src && x < SP_INF) { ++cnt; mx = (u32)x > mx ? (u32)x : mx; }
    }
    block_count_max(cnt, mx, &pair_count[g], maxd);
}

__global__ __launch_bounds__(SP_THREADS) void sp_emit_kernel(
    const i32* __restrict__ graph_ptr, const i32* __restrict__ node_label, const u64* __restrict__ dist_ptr,
    const i32* __restrict__ dist, const u32* __restrict__ pair_base, u64* __restrict__ keys,
    i32* __restrict__ item_graph, u64 n_labels, u64 d1, int with_labels) {
    __shared__ u32 cursor;
    const int g = blockIdx.x, tid = threadIdx.x;
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0;
    if (tid == 0) cursor = 0;
    __syncthreads();
    const i32* dg = dist + dist_ptr[g];
    const u32 base = pair_base[g];
    for (int idx = tid; idx < n * n; idx += SP_THREADS) {
        const int i = idx / n, j = idx - i * n;
        const i32 x = dg[idx];
        if (i != j && x < SP_INF) {
            u64 key = (u64)x;
            if (with_labels)
                key += d1 * ((u64)(u32)node_label[v0 + i] * n_labels + (u64)(u32)node_label[v0 + j]);
            u32 slot = base + atomicAdd(&cursor, 1u);
            keys[slot] = key;
            item_graph[slot] = g;
        }
    }
}

static int bits_for64(u64 v) {
    int b = 0;
    while (b < 64 && (v >> b)) ++b;
    return b;
}

struct SpDist {
    Tmp<u64> sq, dist_ptr, total;
    Tmp<i32> dist, wdev;
    Tmp<u32> pair_count, maxd;
    explicit SpDist(gk_ctx* c) : sq(c), dist_ptr(c), total(c), dist(c), wdev(c), pair_count(c), maxd(c) {}
};

static int sp_compute_dist(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, SpDist& s, u64* total_sq) {
    const i64 N = b->n_graphs;
    GK_TRY(s.sq.alloc(N)); GK_TRY(s.dist_ptr.alloc(N)); GK_TRY(s.total.alloc(1));
    GK_TRY(s.pair_count.alloc(N)); GK_TRY(s.maxd.alloc(1));
    GK_TRY(gk_zero_async(ctx, s.pair_count.p, (size_t)N * 4));
    GK_TRY(gk_zero_async(ctx, s.maxd.p, 4));
    sp_sq_kernel<<<grid_for(N, 256), 256, 0, ctx->stream>>>(b->graph_ptr, s.sq.p, N);
    GK_TRY(gk_scan_u64(ctx, s.sq.p, s.dist_ptr.p, N, true, s.total.p));
    GK_HIP_CHECK(hipMemcpyAsync(total_sq, s.total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    GK_ARG(*total_sq < (1ull << 31), "ShortestPath: sum of n^2 exceeds int32 item indexing");
    GK_TRY(s.dist.alloc(*total_sq));
    const i32* w = nullptr;
    if (edge_weight && b->n_edges > 0) {
        GK_TRY(s.wdev.alloc(b->n_edges));
        GK_HIP_CHECK(hipMemcpyAsync(s.wdev.p, edge_weight, (size_t)b->n_edges * 4, hipMemcpyHostToDevice, ctx->stream));
        w = s.wdev.p;
    }
    const int cap = sp_fw_cap();
    const int nmax = b->max_graph_nodes;
    ProfScope prof_fw(ctx, "sp_fw", 2);       // the all-pairs kernels alone (bench.py: min-plus rate)
    {
        // two launches over all graphs: small graphs (n <= 48) with 256 threads, the rest up to
        // the LDS cap with 1024 threads; a launch's workgroups exit at once for the other class
        const int split = 48;
        const int nsmall = nmax < split ? nmax : split;
        size_t lds = (size_t)nsmall * (nsmall | 1) * 4;
        sp_fw_kernel<<<dim3((unsigned)N), 256, lds, ctx->stream>>>(
            b->graph_ptr, b->row_ptr, b->col_idx, w, s.dist_ptr.p, s.dist.p, s.pair_count.p, s.maxd.p, 0, split);
        if (nmax > split) {
            const int nfw = nmax < cap ? nmax : cap;
            lds = (size_t)nfw * (nfw | 1) * 4;
            GK_TRY(gk_func_lds(ctx, (const void*)sp_fw_kernel, (int)lds));
            sp_fw_kernel<<<dim3((unsigned)N), 1024, lds, ctx->stream>>>(
                b->graph_ptr, b->row_ptr, b->col_idx, w, s.dist_ptr.p, s.dist.p, s.pair_count.p, s.maxd.p, split, cap);
        }
    }
    if (nmax > cap) {
        GK_ARG(nmax <= SP_ROW_MAX_N, "ShortestPath: graphs above 32768 vertices are not supported");
        size_t lds = (size_t)nmax * 4;
        GK_TRY(gk_func_lds(ctx, (const void*)sp_relax_kernel, (int)lds));
        GK_ARG(nmax <= 65535, "ShortestPath: grid.y overflow");
        sp_relax_kernel<<<dim3((unsigned)N, (unsigned)nmax), SP_THREADS, lds, ctx->stream>>>(
            b->graph_ptr, b->row_ptr, b->col_idx, w, s.dist_ptr.p, s.dist.p, s.pair_count.p, s.maxd.p, cap);
    }
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

extern "C" int gk_sp_build(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, int with_labels,
                           gk_batch** out_pair_batch, int64_t* out_n_pairs, int64_t* out_n_keys) {
    return gk_sp_build_levels(ctx, b, edge_weight, with_labels, 1, out_pair_batch, out_n_pairs, out_n_keys);
}

// One pair batch with n_levels levels: level l holds the dictionary ids of the keys
// (l_u, l_v, d) built from the WL labels of level l (level 0 = the input labels), so that
// gk_features_build(pair_batch, n_levels, ...) + gk_gram give sum_l K_SP(level l) -- the WL
// framework over the ShortestPath base kernel (weisfeiler_lehman.py:260-270).  Distances are
// computed once.
extern "C" int gk_sp_build_levels(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, int with_labels,
                                  int n_levels, gk_batch** out_pair_batch, int64_t* out_n_pairs,
                                  int64_t* out_n_keys) {
    GK_ARG(ctx && b && out_pair_batch, "gk_sp_build: null argument");
    GK_ARG(!b->is_pair_batch, "gk_sp_build: needs a graph batch");
    GK_ARG(n_levels >= 1, "gk_sp_build_levels: n_levels must be >= 1");
    GK_ARG(n_levels == 1 || n_levels <= b->n_levels,
           "gk_sp_build_levels: levels not computed (call gk_wl_relabel first)");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ProfScope prof(ctx, "sp");
    const i64 N = b->n_graphs, V = b->n_nodes;
    {   // distances are int32 sums below SP_INF: the longest simple path must stay under it, otherwise a
        // finite distance would silently count as "unreachable"
        i64 wmax = 1;
        if (edge_weight)
            for (i64 e = 0; e < b->n_edges; ++e) {
                GK_ARG(edge_weight[e] > 0, "ShortestPath: edge weights must be positive integers");
                if (edge_weight[e] > wmax) wmax = edge_weight[e];
            }
        const i64 longest = (i64)(b->max_graph_nodes > 1 ? b->max_graph_nodes - 1 : 0) * wmax;
        if (longest >= (i64)SP_INF) {
            gk_set_error("ShortestPath: a path of %d vertices with edge weight %lld can exceed the int32 distance range",
                         b->max_graph_nodes, (long long)wmax);
            return GK_ERR_UNSUPPORTED;
        }
    }
    SpDist s(ctx);
    u64 total_sq = 0;
    GK_TRY(sp_compute_dist(ctx, b, edge_weight, s, &total_sq));
    // pair offsets double as the pair batch's graph_ptr[N+1]
    Tmp<u32> ptotal(ctx);
    void* gpq = nullptr;
    GK_TRY(gk_dev_alloc(ctx, &gpq, (size_t)(N + 1) * 4));
    struct PtrGuard { gk_ctx* c; void* p; ~PtrGuard() { if (p) gk_dev_free(c, p); } } gp_guard{ctx, gpq};
    u32* pair_base = (u32*)gpq;
    GK_TRY(ptotal.alloc(1));
    GK_TRY(gk_scan_u32(ctx, s.pair_count.p, pair_base, N, true, ptotal.p));
    GK_HIP_CHECK(hipMemcpyAsync(pair_base + N, ptotal.p, 4, hipMemcpyDeviceToDevice, ctx->stream));
    u32 h_pairs = 0, h_maxd = 0;
    GK_HIP_CHECK(hipMemcpyAsync(&h_pairs, ptotal.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipMemcpyAsync(&h_maxd, s.maxd.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const u64 d1 = (u64)h_maxd + 1;
    gk_batch* pb = new gk_batch();
    pb->ctx = ctx, pb->is_pair_batch = true;
    pb->n_graphs = N, pb->n_nodes = h_pairs, pb->n_edges = 0, pb->n_labels0 = 0;
    pb->graph_ptr = (i32*)pair_base;
    gp_guard.p = nullptr;   // owned by the pair batch from here on
    i64 nm = b->max_graph_nodes;
    pb->max_graph_nodes = (i32)((nm * (nm - 1) < 2147483647ll) ? nm * (nm - 1) : 2147483647ll);
    auto fail = [&](int r) { gk_batch_destroy(pb); return r; };
    void* q = nullptr;
    int r;
    const size_t np = h_pairs > 0 ? h_pairs : 1;
    if ((r = gk_dev_alloc(ctx, &q, np * 4))) return fail(r);
    pb->node_graph = (i32*)q;
    if ((r = gk_dev_alloc(ctx, &q, np * 4 * (size_t)n_levels))) return fail(r);
    pb->labels = (i32*)q;
    if ((r = gk_dev_alloc(ctx, &q, np * 4 * (size_t)n_levels))) return fail(r);
    pb->perm = (i32*)q;
    pb->cap_levels = n_levels;
    Tmp<u64> keys(ctx);
    Tmp<u32> nkeys(ctx);
    if ((r = keys.alloc(np)) || (r = nkeys.alloc((size_t)n_levels))) return fail(r);
    for (int l = 0; l < n_levels; ++l) {
        u64 L = 1;
        if (with_labels) {
            const i64 cnt = l == 0 ? (i64)b->n_labels0 : (i64)b->label_counts[l];
            L = (u64)(cnt > 0 ? cnt : 1);
            if (l == 0 && b->n_levels > 0 && (u64)b->label_counts[0] > L) L = (u64)b->label_counts[0];
        }
        if (2 * bits_for64(L) + bits_for64(d1) > 63) {
            gk_set_error("ShortestPath: (label,label,distance) key exceeds 64 bits");
            return fail(GK_ERR_ARG);
        }
        const int key_bits = bits_for64(d1 * L * L - 1);
        sp_emit_kernel<<<dim3((unsigned)N), SP_THREADS, 0, ctx->stream>>>(
            b->graph_ptr, b->labels + (size_t)l * V, s.dist_ptr.p, s.dist.p, pair_base, keys.p, pb->node_graph,
            L, d1, with_labels ? 1 : 0);
        if ((r = gk_dictionary_from_keys(ctx, keys.p, h_pairs, key_bits, pb->labels + (size_t)l * np,
                                         pb->perm + (size_t)l * np, nkeys.p + l)))
            return fail(r);
    }
    std::vector<u32> h_keys((size_t)n_levels, 0);
    if (hipMemcpyAsync(h_keys.data(), nkeys.p, 4 * (size_t)n_levels, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        gk_set_error("gk_sp_build: %s", hipGetErrorString(hipGetLastError()));
        return fail(GK_ERR_HIP);
    }
    pb->n_levels = n_levels;
    pb->label_counts.assign(h_keys.begin(), h_keys.end());
    *out_pair_batch = pb;
    if (out_n_pairs) *out_n_pairs = h_pairs;
    if (out_n_keys)
        for (int l = 0; l < n_levels; ++l) out_n_keys[l] = h_keys[l];
    return GK_OK;
}

extern "C" int gk_sp_debug_apsp(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, int64_t graph,
                                int32_t* out_dist) {
    GK_ARG(ctx && b && out_dist, "gk_sp_debug_apsp: null argument");
    GK_ARG(graph >= 0 && graph < b->n_graphs && !b->is_pair_batch, "gk_sp_debug_apsp: bad graph index");
    SpDist s(ctx);
    u64 total_sq = 0;
    GK_TRY(sp_compute_dist(ctx, b, edge_weight, s, &total_sq));
    std::vector<i32> gp(2);
    std::vector<u64> dp(1);
    GK_HIP_CHECK(hipMemcpyAsync(gp.data(), b->graph_ptr + graph, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipMemcpyAsync(dp.data(), s.dist_ptr.p + graph, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const i64 n = gp[1] - gp[0];
    std::vector<i32> h((size_t)(n * n > 0 ? n * n : 1));
    if (n > 0) {
        GK_HIP_CHECK(hipMemcpyAsync(h.data(), s.dist.p + dp[0], (size_t)n * n * 4, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    for (i64 i = 0; i < n * n; ++i) out_dist[i] = h[i] >= SP_INF ? -1 : h[i];
    return GK_OK;
}

// WeisfeilerLehman.transform as a LOOK-UP against the fitted dictionaries (reference: grakel/kernels/weisfeiler_lehman.py:
// 435-476 relabels only the targets and looks their credentials up in the fitted `_inv_labels[i]`, :493-498 sums the base
// kernels' rectangular matrices; vertex_histogram.py:138-184 counts the targets' labels in the fitted columns).
//
// The joint route (gk_batch_concat + relabel of fitted graphs and targets together) costs O(fitted nodes) per call.  Here a fit
// leaves on the device, per level l >= 1:
//   * a table  hash of the full signature (own fitted label at l-1, multiset of the neighbours' fitted labels)  ->  fitted class,
//     one entry per fitted class, computed from one representative node of the class (open addressing, 64-bit keys);
//   * the inverted index  fitted class -> (graph, count) entries  (nodes stably sorted by class: a class's nodes then come
//     graph after graph), also for level 0; the fitted self similarities.
// A transform relabels the TARGETS ALONE (any route: that fixes their own partition exactly, hash collisions included), then
// level by level matches every target class to a fitted class or to none: a class is matched iff the class of its
// representative at the level before and those of all its neighbours are matched and the signature written in FITTED ids is in
// the table -- verified against the fitted representative's full signature, the hash only proposes.  K[t, f] = sum over levels
// and matched classes of count_t * count_f is then accumulated per target graph in LDS by walking the index lists: work
// proportional to the targets, not to the fit.  Everything is exact integer arithmetic.
// Declines (GK_ERR_UNSUPPORTED; the caller takes the joint route): two fitted classes with one 64-bit hash, a node of more than
// TF_MAXDEG neighbours among the representatives, a target graph above GM_MAX_NODES nodes.
#include "common.h"
#include <string.h>
#include "features.h"
#include "scan_fn.h"
#include "wl_sig.h"
#include <memory>
#include <vector>

#define TF_MAXDEG 64
#define TA_COLS 32768            // fitted graphs per accumulator block (u32 in LDS)
#define TT_SPLIT_MAX_TARGETS 16  // up to here the accumulation is spread over the chip (tt_items / tt_walk / tt_finish)
#define TA_SLOTS 2048            // label-count table of one target graph (n <= GM_MAX_NODES)

static inline dim3 grid_for(i64 n, int t) { return dim3((unsigned)(n > 0 ? cdiv(n, t) : 1)); }

struct gk_wl_fitted {
    gk_ctx* ctx = nullptr;
    gk_batch* fit = nullptr;
    u64 fit_gen = 0;
    int n_levels = 0;
    i64 N = 0, V = 0;
    i32 L0 = 0;
    std::vector<u64*> table;      // [level] hash (0 = empty)
    std::vector<i32*> tval;       // [level] class of the slot
    std::vector<u64> mask;        // [level] capacity - 1
    std::vector<i32*> rep;        // [level] one node per class
    std::vector<i32*> cls_ptr;    // [level] first entry of each class (+ end)
    std::vector<i32*> ent_graph;  // [level] entries: graph (build only)
    std::vector<i32*> ent_pos;    // [level] entries: start in the class-sorted node order (count = next start - start; build only)
    std::vector<uint2*> ent;      // [level] entries as the accumulation reads them: (graph, count)
    std::vector<i64> count;       // classes per level
    u64* selfk = nullptr;         // [N] fitted self similarities
    u32* flags = nullptr;         // bit 0: two fitted classes share a hash, bit 1: degree above TF_MAXDEG
};

static inline u64 tf_seed(int level) { return 0x6a09e667f3bcc908ULL + (u64)level * 0x9E3779B97F4A7C15ULL; }

// one representative per class (any writer wins)
__global__ void tf_rep_kernel(const i32* __restrict__ lab, i32* __restrict__ rep, i64 V) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V) rep[lab[v]] = (i32)v;
}

// fitted class c -> table: hash of its representative's signature in the labels of the level before
__global__ void tf_insert_kernel(const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx, const i32* __restrict__ lab_prev,
                                 const i32* __restrict__ rep, i64 count, u64* __restrict__ table, i32* __restrict__ tval, u64 mask,
                                 u64 seed, u32* __restrict__ flags) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    const i32 r = rep[c];
    const i32 s = row_ptr[r];
    const int d = row_ptr[r + 1] - s;
    if (d > TF_MAXDEG) { atomicOr(flags, 2u); return; }
    u64 acc = sig_head((u32)lab_prev[r], (u32)d, seed);
    for (int k = 0; k < d; ++k) acc += sig_elem((u32)lab_prev[col_idx[s + k]], seed);
    u64 h = mix64(acc);
    if (h == 0) h = 1;
    u64 slot = h & mask;
    for (;;) {
        const unsigned long long old = atomicCAS((unsigned long long*)&table[slot], 0ull, (unsigned long long)h);
        if (old == 0ull) { tval[slot] = (i32)c; return; }
        if (old == h) { atomicOr(flags, 1u); return; }       // two classes, one hash: the look-up cannot tell them apart
        slot = (slot + 1) & mask;
    }
}

// inverted index of one level: nodes sorted by class (stable: ascending node = ascending graph inside a class); an entry
// starts where the class or the graph changes
struct TfIndex {
    const u64* ks; const u32* perm; const i32* node_graph;
    i32* ent_graph; i32* ent_pos; u32* cls_cnt; u32* n_ent; i64 V;
    __device__ __forceinline__ bool chead(i64 k) const { return k == 0 || ks[k] != ks[k - 1]; }
    __device__ __forceinline__ u32 value(i64 k) const {
        return (chead(k) || node_graph[perm[k]] != node_graph[perm[k - 1]]) ? 1u : 0u;
    }
    __device__ __forceinline__ void emit(i64 k, u32 head, u32 incl) const {
        if (!head) return;
        const u32 e = incl - 1u;
        ent_graph[e] = node_graph[perm[k]];
        ent_pos[e] = (i32)k;
        atomicAdd(&cls_cnt[ks[k]], 1u);
    }
    __device__ __forceinline__ void finish(u32 total) const { *n_ent = total; ent_pos[total] = (i32)V; }
    __device__ __forceinline__ i64 seg_first_tile(i64) const { return 0; }
};

__global__ void tf_labels_to_keys_kernel(const i32* __restrict__ lab, u64* __restrict__ keys, i64 n) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = (u64)(u32)lab[i];
}

// fitted self similarity: sum over levels and entries of count^2
__global__ void tf_selfk_kernel(const i32* __restrict__ ent_graph, const i32* __restrict__ ent_pos, const u32* __restrict__ n_ent,
                                unsigned long long* __restrict__ selfk, uint2* __restrict__ ent) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (i64)*n_ent) return;
    const u64 c = (u64)(ent_pos[e + 1] - ent_pos[e]);
    ent[e] = make_uint2((u32)ent_graph[e], (u32)c);
    atomicAdd(&selfk[ent_graph[e]], c * c);
}

extern "C" int gk_wl_fitted_destroy(gk_wl_fitted* w) {
    if (!w) return GK_OK;
    gk_ctx* ctx = w->ctx;
    auto rel = [&](void* p) { if (p) gk_dev_free(ctx, p); };
    for (auto p : w->table) rel(p);
    for (auto p : w->tval) rel(p);
    for (auto p : w->rep) rel(p);
    for (auto p : w->cls_ptr) rel(p);
    for (auto p : w->ent_graph) rel(p);
    for (auto p : w->ent_pos) rel(p);
    for (auto p : w->ent) rel(p);
    rel(w->selfk);
    rel(w->flags);
    delete w;
    return GK_OK;
}

extern "C" int gk_wl_fitted_create(gk_ctx* ctx, gk_batch* fit, int n_iter, gk_wl_fitted** out) {
    GK_ARG(ctx && fit && out, "gk_wl_fitted_create: null argument");
    GK_ARG(!fit->is_pair_batch && fit->ctx == ctx, "gk_wl_fitted_create: needs a graph batch of this context");
    GK_ARG(n_iter >= 0 && fit->n_levels == n_iter + 1, "gk_wl_fitted_create: the batch is not relabelled for n_iter levels (gk_wl_relabel first)");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    const i64 V = fit->n_nodes, N = fit->n_graphs;
    const int L = n_iter + 1;
    if (V <= 0) return GK_ERR_UNSUPPORTED;
    // tt_accumulate_kernel adds count_target * count_fitted per level in 32-bit LDS words: K[t, f] <= levels * 1024 (the
    // largest target graph of the look-up route) * the largest fitted graph must stay below 2^32, else the joint route
    // (whose Gram picks its accumulator from the job's bound) takes the job
    if ((double)L * (double)GM_MAX_NODES * (double)fit->max_graph_nodes >= 4294967296.0) return GK_ERR_UNSUPPORTED;
    gk_wl_fitted* w = new gk_wl_fitted();
    w->ctx = ctx, w->fit = fit, w->fit_gen = fit->relabel_gen, w->n_levels = L, w->N = N, w->V = V, w->L0 = fit->n_labels0;
    w->table.assign(L, nullptr), w->tval.assign(L, nullptr), w->mask.assign(L, 0), w->rep.assign(L, nullptr);
    w->cls_ptr.assign(L, nullptr), w->ent_graph.assign(L, nullptr), w->ent_pos.assign(L, nullptr), w->count.assign(L, 0);
    w->ent.assign(L, nullptr);
    auto fail = [&](int r) { gk_wl_fitted_destroy(w); return r; };
    int r;
    void* q = nullptr;
#define W_ALLOC(dst, type, n) do { if ((r = gk_dev_alloc(ctx, &q, (size_t)(n) * sizeof(type)))) return fail(r); dst = (type*)q; } while (0)
    W_ALLOC(w->selfk, u64, N);
    W_ALLOC(w->flags, u32, 4);
    if ((r = gk_zero_async(ctx, w->selfk, (size_t)N * 8)) || (r = gk_zero_async(ctx, w->flags, 16))) return fail(r);
    Tmp<u64> keys(ctx), ks(ctx);
    Tmp<u32> perm(ctx);
    if ((r = keys.alloc(V)) || (r = ks.alloc(V)) || (r = perm.alloc(V))) return fail(r);
    for (int l = 0; l < L; ++l) {
        Tmp<u32> cls_cnt(ctx);
        const i64 count = l == 0 ? (i64)fit->n_labels0 : fit->label_counts[l];
        w->count[l] = count;
        const i32* lab = fit->labels + (size_t)l * V;
        // ---- inverted index
        W_ALLOC(w->cls_ptr[l], i32, count + 1);
        W_ALLOC(w->ent_graph[l], i32, V + 1);
        W_ALLOC(w->ent_pos[l], i32, V + 2);
        W_ALLOC(w->ent[l], uint2, V + 1);
        if ((r = cls_cnt.alloc((size_t)count + 2)) || (r = gk_zero_async(ctx, cls_cnt.p, ((size_t)count + 2) * 4))) return fail(r);
        tf_labels_to_keys_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(lab, keys.p, V);
        int bits = 1;
        while (bits < 64 && ((u64)(count > 0 ? count - 1 : 0) >> bits)) ++bits;
        if ((r = gk_radix_sort_pairs(ctx, keys.p, nullptr, ks.p, perm.p, V, bits))) return fail(r);
        TfIndex ti{ks.p, perm.p, fit->node_graph, w->ent_graph[l], w->ent_pos[l], cls_cnt.p, cls_cnt.p + count + 1, V};
        if ((r = gk_scan_fn<u32, TfIndex>(ctx, ti, V, nullptr))) return fail(r);
        if ((r = gk_scan_u32(ctx, cls_cnt.p, (u32*)w->cls_ptr[l], count + 1, true, nullptr))) return fail(r);     // entry count + 1 slots: [count] = all entries
        tf_selfk_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(w->ent_graph[l], w->ent_pos[l], cls_cnt.p + count + 1,
                                                                    (unsigned long long*)w->selfk, w->ent[l]);
        // the two build arrays are done with (stream order): only the packed entries stay
        gk_dev_free(ctx, w->ent_graph[l]), gk_dev_free(ctx, w->ent_pos[l]);
        w->ent_graph[l] = nullptr, w->ent_pos[l] = nullptr;
        if (l == 0) continue;
        // ---- signature table
        u64 cap = 64;
        while (cap < 2 * (u64)count) cap <<= 1;
        w->mask[l] = cap - 1;
        W_ALLOC(w->table[l], u64, cap);
        W_ALLOC(w->tval[l], i32, cap);
        W_ALLOC(w->rep[l], i32, count);
        if ((r = gk_zero_async(ctx, w->table[l], (size_t)cap * 8))) return fail(r);
        tf_rep_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(lab, w->rep[l], V);
        tf_insert_kernel<<<grid_for(count, 256), 256, 0, ctx->stream>>>(fit->row_ptr, fit->col_idx, fit->labels + (size_t)(l - 1) * V, w->rep[l], count,
                                                                         w->table[l], w->tval[l], w->mask[l], tf_seed(l),
                                                                         w->flags);
    }
#undef W_ALLOC
    if (hipGetLastError() != hipSuccess) { gk_set_error("gk_wl_fitted_create: kernel launch failed"); return fail(GK_ERR_HIP); }
    u32 hf[4] = {0, 0, 0, 0};
    if ((r = gk_readback(ctx, w->flags, hf, 4))) return fail(r);
    if (hf[0]) {
        gk_set_error("gk_wl_fitted_create: %s -- transform takes the joint route", (hf[0] & 1u) ? "two fitted classes share a 64-bit signature hash"
                                                                                              : "a class representative has more than 64 neighbours");
        return fail(GK_ERR_UNSUPPORTED);
    }
    *out = w;
    return GK_OK;
}

// ---- transform ----------------------------------------------------------------------------------------------------------
// target class k of a level >= 1 -> fitted class or -1.  xlat of the level before: level 0 is the identity below L0 (the
// targets' input labels are ids of the fit's label map, unseen ones above it), later levels go through map_prev.
// Round 5: ONE WAVE per target class, lane i holds neighbour i (at most TF_MAXDEG = 64): the translated labels, the hash
// (wave sum) and -- when the hash is in the table -- both sorted neighbour lists live in one register per lane
// (wave_bitonic_sort<1>); the thread-per-class form of round 4 kept two int[64] arrays in scratch (528 bytes per lane) and
// sorted them by insertion.
struct TmLevel {
    const i32* t_lab_prev; const i32* t_lab; i32* t_rep; i64 t_count; const i32* map_prev; i32* map_cur; int prev_is_level0;
    const u64* table; const i32* tval; u64 mask, seed; const i32* f_lab_prev; const i32* f_rep;
};
__device__ __forceinline__ void tt_match_class(const TmLevel& Q, i64 k, int lane, const i32* __restrict__ t_row_ptr,
                                               const i32* __restrict__ t_col_idx, i32 L0, const i32* __restrict__ f_row_ptr,
                                               const i32* __restrict__ f_col_idx, u32* __restrict__ flags) {
    const i32 r = Q.t_rep[k];
    const i32 s = t_row_ptr[r];
    const int d = t_row_ptr[r + 1] - s;
    if (d > TF_MAXDEG) {                                   // wave-uniform
        if (lane == 0) { atomicOr(flags, 2u); Q.map_cur[k] = -1; }
        return;
    }
    const i32 own_raw = Q.t_lab_prev[r];
    const i32 own = Q.prev_is_level0 ? (own_raw < L0 ? own_raw : -1) : Q.map_prev[own_raw];
    i32 x[1] = {0x7fffffff};
    bool mine_ok = true;
    u64 part = 0;
    if (lane < d) {
        const i32 raw = Q.t_lab_prev[t_col_idx[s + lane]];
        const i32 y = Q.prev_is_level0 ? (raw < L0 ? raw : -1) : Q.map_prev[raw];
        mine_ok = y >= 0;
        x[0] = y;
        part = sig_elem((u32)y, Q.seed);
    }
    const bool all = own >= 0 && __builtin_amdgcn_ballot_w64(!mine_ok) == 0ull;
    i32 res = -1;
    if (all) {                                             // wave-uniform
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        u64 h = mix64(sig_head((u32)own, (u32)d, Q.seed) + part);
        if (h == 0) h = 1;
        u64 slot = h & Q.mask;
        for (;;) {
            const u64 t = Q.table[slot];
            if (t == 0) break;
            if (t == h) {
                // the hash proposes, the full signature of the fitted representative decides
                const i32 c = Q.tval[slot];
                const i32 rf = Q.f_rep[c];
                const i32 sf = f_row_ptr[rf];
                bool ok = Q.f_lab_prev[rf] == own && f_row_ptr[rf + 1] - sf == d;
                if (ok) {
                    i32 y[1] = {0x7fffffff};
                    if (lane < d) y[0] = Q.f_lab_prev[f_col_idx[sf + lane]];
                    wave_bitonic_sort<1>(x, lane);
                    wave_bitonic_sort<1>(y, lane);
                    ok = __builtin_amdgcn_ballot_w64(x[0] != y[0]) == 0ull;
                }
                if (ok) res = c;
                break;
            }
            slot = (slot + 1) & Q.mask;
        }
    }
    if (lane == 0) Q.map_cur[k] = res;
}

__global__ __launch_bounds__(256) void tt_match_kernel(const TmLevel Q, const i32* __restrict__ t_row_ptr, const i32* __restrict__ t_col_idx,
                                                       i32 L0, const i32* __restrict__ f_row_ptr, const i32* __restrict__ f_col_idx,
                                                       u32* __restrict__ flags) {
    const i64 k = ((i64)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (k >= Q.t_count) return;
    tt_match_class(Q, k, threadIdx.x & 63, t_row_ptr, t_col_idx, L0, f_row_ptr, f_col_idx, flags);
}

// A handful of targets (one graph to classify: the serving case): ALL levels in one single-workgroup launch -- per level the
// representatives (any writer wins), a workgroup barrier, the classes dealt to the 16 waves, a barrier -- instead of two
// launches per level (ten dependent launches of ~5 us for h = 5).
#define TM_MAX_LEVELS 16
#define TM_MAX_NODES 8192
struct TmLevels { TmLevel lv[TM_MAX_LEVELS]; int L; };
__global__ __launch_bounds__(1024) void tt_match_levels_kernel(const TmLevels P, i64 Vt, const i32* __restrict__ t_row_ptr,
                                                               const i32* __restrict__ t_col_idx, i32 L0, const i32* __restrict__ f_row_ptr,
                                                               const i32* __restrict__ f_col_idx, u32* __restrict__ flags) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int l = 1; l < P.L; ++l) {
        const TmLevel& Q = P.lv[l];
        for (i64 v = tid; v < Vt; v += 1024) Q.t_rep[Q.t_lab[v]] = (i32)v;
        __threadfence_block();
        __syncthreads();
        for (i64 k = w; k < Q.t_count; k += 16) tt_match_class(Q, k, lane, t_row_ptr, t_col_idx, L0, f_row_ptr, f_col_idx, flags);
        __threadfence_block();
        __syncthreads();
    }
}

struct TaLevels {
    const i32* t_lab[FEAT_MAX_LEVELS];       // target labels of the level
    const i32* map[FEAT_MAX_LEVELS];         // target class -> fitted class (null: level 0, identity below L0)
    const i32* cls_ptr[FEAT_MAX_LEVELS];
    const uint2* ent[FEAT_MAX_LEVELS];       // fitted (graph, count) entries, grouped by class
    int L;
};

// one workgroup per target graph: per level the graph's label counts (LDS table), then for every label that has a fitted class the
// fitted (graph, count) entries of that class: acc[f] += count_t * count_f (LDS atomics), a wave per label, lanes over its entries
__global__ __launch_bounds__(1024) void tt_accumulate_kernel(const TaLevels P, const i32* __restrict__ t_graph_ptr, i32 L0, i64 n_fit,
                                                             const u64* __restrict__ x_selfk, int normalize, double* __restrict__ K,
                                                             u64* __restrict__ y_selfk) {
    extern __shared__ __attribute__((aligned(16))) u32 ta_lds[];
    u32* acc = ta_lds;                                   // [TA_COLS]
    i32* keys = (i32*)(ta_lds + TA_COLS);                // [TA_SLOTS]
    u32* cnt = (u32*)(keys + TA_SLOTS);                  // [TA_SLOTS]
    i32* item_c = (i32*)(cnt + TA_SLOTS);                // [GM_MAX_NODES] work list: fitted class
    u32* item_n = (u32*)(item_c + GM_MAX_NODES);         // [GM_MAX_NODES]            count in the target
    __shared__ u32 n_items;
    __shared__ unsigned long long self_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const i64 t = blockIdx.x;
    const i32 v0 = t_graph_ptr[t];
    const int n = t_graph_ptr[t + 1] - v0;
    u32 T = 64;
    while (T < 2u * (u32)n) T <<= 1;
    const u32 tmask = T - 1u;
    if (tid == 0) self_s = 0ull;
    for (i64 cb = 0; cb < n_fit; cb += TA_COLS) {
        const i64 ce = cb + TA_COLS < n_fit ? cb + TA_COLS : n_fit;
        for (int i = tid; i < (int)(ce - cb); i += 1024) acc[i] = 0u;
        for (int l = 0; l < P.L; ++l) {
            __syncthreads();                              // the level before is done with the tables
            for (u32 i = tid; i < T; i += 1024) keys[i] = -1, cnt[i] = 0u;
            if (tid == 0) n_items = 0u;
            __syncthreads();
            const i32* __restrict__ lab = P.t_lab[l];
            for (int i = tid; i < n; i += 1024) {
                const i32 x = lab[v0 + i];
                u32 h = ((u32)x * 2654435761u) >> 8 & tmask;
                for (;;) {
                    const i32 old = atomicCAS(&keys[h], -1, x);
                    if (old == -1 || old == x) { atomicAdd(&cnt[h], 1u); break; }
                    h = (h + 1u) & tmask;
                }
            }
            __syncthreads();
            unsigned long long sq = 0ull;
            for (u32 i = tid; i < T; i += 1024) {
                const i32 x = keys[i];
                if (x < 0) continue;
                const u32 c = cnt[i];
                sq += (unsigned long long)c * c;
                const i32 fc = P.map[l] ? P.map[l][x] : (x < L0 ? x : -1);
                if (fc >= 0) {
                    const u32 e = atomicAdd(&n_items, 1u);
                    item_c[e] = fc, item_n[e] = c;
                }
            }
            if (cb == 0 && sq) atomicAdd(&self_s, sq);
            __syncthreads();
            const u32 ni = n_items;
            const i32* __restrict__ cp = P.cls_ptr[l];
            const uint2* __restrict__ en = P.ent[l];
            for (u32 it = w; it < ni; it += 16) {
                const i32 fc = item_c[it];
                const u32 ct = item_n[it];
                const i32 lo = cp[fc], hi = cp[fc + 1];
                // four entries per lane in flight (one dependent load per trip made the walk a chain of memory latencies)
                for (i32 e0 = lo + lane; e0 < hi; e0 += 256) {
                    uint2 x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[u] = e0 + 64 * u < hi ? en[e0 + 64 * u] : make_uint2(0xffffffffu, 0u);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const i64 f = (i64)x[u].x;
                        if (x[u].x != 0xffffffffu && f >= cb && f < ce) atomicAdd(&acc[f - cb], ct * x[u].y);
                    }
                }
            }
        }
        __syncthreads();
        const double ys = (double)self_s;
        for (i64 f = cb + tid; f < ce; f += 1024) {
            double v = (double)acc[f - cb];
            if (normalize) {
                const double den = sqrt(ys * (double)x_selfk[f]);
                v = v / den;
                if (normalize == 2) {                     // numpy.nan_to_num: nan -> 0, +inf -> the largest double
                    if (v != v) v = 0.0;
                    else if (v > 1.7976931348623157e308) v = 1.7976931348623157e308;
                }
            }
            K[t * n_fit + f] = v;
        }
        __syncthreads();
    }
    if (tid == 0) y_selfk[t] = self_s;
}

// ---- a handful of targets: the accumulation spread over the chip.  tt_accumulate_kernel gives a target graph ONE workgroup,
// which walks every fitted (graph, count) entry of every matched label alone (~0.1 ms for one config-3 target against 10 000
// fitted graphs: the serving case).  Here: (1) a workgroup per target lists its items (level, fitted class, count in the
// target) and its self similarity, (2) waves from all over the chip take the items one by one and add count_t * count_f to a
// 32-bit accumulator row in HBM (the fitted entries of a class are distinct graphs: no contention inside a wave),
// (3) one pass turns the accumulators into the float64 / normalised row.
__global__ __launch_bounds__(1024) void tt_items_kernel(const TaLevels P, const i32* __restrict__ t_graph_ptr, i32 L0, int3* __restrict__ items,
                                                        u32* __restrict__ n_items, i64 item_cap, u64* __restrict__ y_selfk) {
    extern __shared__ __attribute__((aligned(16))) u32 ti_lds[];
    i32* keys = (i32*)ti_lds;                            // [TA_SLOTS]
    u32* cnt = (u32*)(keys + TA_SLOTS);                  // [TA_SLOTS]
    __shared__ unsigned long long self_s;
    const int tid = threadIdx.x;
    const i64 t = blockIdx.x;
    const i32 v0 = t_graph_ptr[t];
    const int n = t_graph_ptr[t + 1] - v0;
    u32 T = 64;
    while (T < 2u * (u32)n) T <<= 1;
    const u32 tmask = T - 1u;
    if (tid == 0) self_s = 0ull;
    for (int l = 0; l < P.L; ++l) {
        __syncthreads();
        for (u32 i = tid; i < T; i += 1024) keys[i] = -1, cnt[i] = 0u;
        __syncthreads();
        const i32* __restrict__ lab = P.t_lab[l];
        for (int i = tid; i < n; i += 1024) {
            const i32 x = lab[v0 + i];
            u32 h = ((u32)x * 2654435761u) >> 8 & tmask;
            for (;;) {
                const i32 old = atomicCAS(&keys[h], -1, x);
                if (old == -1 || old == x) { atomicAdd(&cnt[h], 1u); break; }
                h = (h + 1u) & tmask;
            }
        }
        __syncthreads();
        unsigned long long sq = 0ull;
        for (u32 i = tid; i < T; i += 1024) {
            const i32 x = keys[i];
            if (x < 0) continue;
            const u32 c = cnt[i];
            sq += (unsigned long long)c * c;
            const i32 fc = P.map[l] ? P.map[l][x] : (x < L0 ? x : -1);
            if (fc >= 0) {
                const u32 e = atomicAdd(&n_items[t], 1u);
                if ((i64)e < item_cap) items[t * item_cap + e] = make_int3(l, fc, (int)c);
            }
        }
        if (sq) atomicAdd(&self_s, sq);
    }
    __syncthreads();
    if (tid == 0) y_selfk[t] = self_s;
}

__global__ __launch_bounds__(256) void tt_walk_kernel(const TaLevels P, const int3* __restrict__ items, const u32* __restrict__ n_items,
                                                      i64 item_cap, i64 n_fit, u32* __restrict__ acc) {
    const i64 t = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const u32 ni = n_items[t];
    const u32 waves = gridDim.x * 4u;
    for (u32 it = blockIdx.x * 4u + (threadIdx.x >> 6); it < ni; it += waves) {
        const int3 q = items[t * item_cap + it];
        const i32 lo = P.cls_ptr[q.x][q.y], hi = P.cls_ptr[q.x][q.y + 1];
        const uint2* __restrict__ en = P.ent[q.x];
        for (i32 e0 = lo + lane; e0 < hi; e0 += 256) {
            uint2 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = e0 + 64 * u < hi ? en[e0 + 64 * u] : make_uint2(0xffffffffu, 0u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (x[u].x != 0xffffffffu) atomicAdd(&acc[t * n_fit + (i64)x[u].x], (u32)q.z * x[u].y);
        }
    }
}

__global__ void tt_finish_kernel(const u32* __restrict__ acc, i64 n_targets, i64 n_fit, const u64* __restrict__ x_selfk,
                                 const u64* __restrict__ y_selfk, int normalize, double* __restrict__ K) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_targets * n_fit) return;
    const i64 t = i / n_fit, f = i - t * n_fit;
    double v = (double)acc[i];
    if (normalize) {
        v = v / sqrt((double)y_selfk[t] * (double)x_selfk[f]);
        if (normalize == 2) {
            if (v != v) v = 0.0;
            else if (v > 1.7976931348623157e308) v = 1.7976931348623157e308;
        }
    }
    K[i] = v;
}

extern "C" int gk_wl_fitted_selfk(gk_ctx* ctx, gk_wl_fitted* w, double* out_selfk) {
    GK_ARG(ctx && w && out_selfk, "gk_wl_fitted_selfk: null argument");
    std::vector<u64> h((size_t)w->N);
    GK_HIP_CHECK(hipMemcpyAsync(h.data(), w->selfk, (size_t)w->N * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (i64 i = 0; i < w->N; ++i) out_selfk[i] = (double)h[i];
    return GK_OK;
}

// targets: relabelled ALONE (gk_wl_relabel with the fit's n_iter), level-0 ids in the fit's id space (unseen input labels >= the
// fit's n_labels0).  out_K: host, [n_targets x n_fitted] float64; out_y_selfk: host, [n_targets].
extern "C" int gk_wl_transform(gk_ctx* ctx, gk_wl_fitted* w, gk_batch* tb, int normalize, double* out_K, double* out_y_selfk) {
    GK_ARG(ctx && w && tb && out_K && out_y_selfk, "gk_wl_transform: null argument");
    GK_ARG(!tb->is_pair_batch && tb->ctx == ctx && w->ctx == ctx, "gk_wl_transform: needs graph batches of this context");
    GK_ARG(tb->n_levels == w->n_levels, "gk_wl_transform: the targets are not relabelled for the fit's n_iter (gk_wl_relabel first)");
    if (w->fit_gen != w->fit->relabel_gen) {
        gk_set_error("gk_wl_transform: the fitted batch was relabelled again: its label ids are not the ones of this fitted state");
        return GK_ERR_STATE;
    }
    if (tb->max_graph_nodes > GM_MAX_NODES || w->n_levels > FEAT_MAX_LEVELS) return GK_ERR_UNSUPPORTED;
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ProfScope prof(ctx, "transform");
    const i64 Vt = tb->n_nodes, Nt = tb->n_graphs, Nf = w->N;
    const int L = w->n_levels;
    gk_batch* fb = w->fit;
    // small results (one graph to classify: an 80 KB row) come back in ONE copy through the context's pinned block: K, the
    // targets' self similarities and the flag words sit in one device block.  (Three hipMemcpyAsync into pageable memory
    // cost ~0.1 ms of the 0.17 ms the call took for one target.)
    const size_t k_bytes = (size_t)Nt * (size_t)Nf * 8, ys_bytes = (size_t)Nt * 8;
    // (larger outputs are pinned blocks of the caller's pool already -- engine.PinnedPool from 1 MB up -- and take the direct copy)
    const bool one_copy = k_bytes + ys_bytes + 16 <= ((size_t)1 << 20);
    Tmp<char> block(ctx);
    Tmp<u32> flags_own(ctx);
    u32* flags_p = nullptr;
    if (one_copy) {
        GK_TRY(block.alloc(k_bytes + ys_bytes + 16));
        flags_p = (u32*)(block.p + k_bytes + ys_bytes);
    } else {
        GK_TRY(flags_own.alloc(4));
        flags_p = flags_own.p;
    }
    struct { u32* p; } flags = {flags_p};
    GK_TRY(gk_zero_async(ctx, flags.p, 16));
    std::vector<std::unique_ptr<Tmp<i32>>> maps((size_t)L), reps((size_t)L);
    TaLevels P = {};
    P.L = L;
    for (int l = 0; l < L; ++l) {
        P.t_lab[l] = tb->labels + (size_t)l * Vt;
        P.cls_ptr[l] = w->cls_ptr[l], P.ent[l] = w->ent[l];
        P.map[l] = nullptr;
        if (l == 0 || Vt == 0) continue;
        const i64 tc = tb->label_counts[l];
        maps[l].reset(new Tmp<i32>(ctx)), reps[l].reset(new Tmp<i32>(ctx));
        GK_TRY(maps[l]->alloc(tc)); GK_TRY(reps[l]->alloc(tc));
        P.map[l] = maps[l]->p;
    }
    {
        auto level = [&](int l) {
            TmLevel Q;
            Q.t_lab_prev = P.t_lab[l - 1], Q.t_lab = P.t_lab[l], Q.t_rep = reps[l]->p, Q.t_count = tb->label_counts[l];
            Q.map_prev = l >= 2 ? maps[l - 1]->p : nullptr, Q.map_cur = maps[l]->p, Q.prev_is_level0 = l == 1 ? 1 : 0;
            Q.table = w->table[l], Q.tval = w->tval[l], Q.mask = w->mask[l], Q.seed = tf_seed(l);
            Q.f_lab_prev = fb->labels + (size_t)(l - 1) * fb->n_nodes, Q.f_rep = w->rep[l];
            return Q;
        };
        if (Vt > 0 && L >= 2 && L <= TM_MAX_LEVELS && Vt <= TM_MAX_NODES && !ctx->opt.tt_no_fused) {
            TmLevels A = {};
            A.L = L;
            for (int l = 1; l < L; ++l) A.lv[l] = level(l);
            tt_match_levels_kernel<<<dim3(1), 1024, 0, ctx->stream>>>(A, Vt, tb->row_ptr, tb->col_idx, w->L0, fb->row_ptr, fb->col_idx, flags.p);
        } else if (Vt > 0) {
            for (int l = 1; l < L; ++l) {
                const TmLevel Q = level(l);
                tf_rep_kernel<<<grid_for(Vt, 256), 256, 0, ctx->stream>>>(Q.t_lab, Q.t_rep, Vt);
                tt_match_kernel<<<grid_for(Q.t_count * 64, 256), 256, 0, ctx->stream>>>(Q, tb->row_ptr, tb->col_idx, w->L0, fb->row_ptr, fb->col_idx,
                                                                                      flags.p);
            }
        }
    }
    Tmp<double> K_own(ctx);
    Tmp<u64> ys_own(ctx);
    struct { double* p; } K = {nullptr};
    struct { u64* p; } ys = {nullptr};
    if (one_copy) K.p = (double*)block.p, ys.p = (u64*)(block.p + k_bytes);
    else {
        GK_TRY(K_own.alloc((size_t)Nt * (size_t)Nf)); GK_TRY(ys_own.alloc(Nt));
        K.p = K_own.p, ys.p = ys_own.p;
    }
    if (Nt <= TT_SPLIT_MAX_TARGETS && !ctx->opt.tt_no_fused) {
        const i64 item_cap = (i64)tb->max_graph_nodes * L;           // distinct labels of a graph per level <= its vertices
        Tmp<int3> items(ctx);
        Tmp<u32> n_items(ctx), acc(ctx);
        GK_TRY(items.alloc((size_t)Nt * (size_t)item_cap)); GK_TRY(n_items.alloc(Nt)); GK_TRY(acc.alloc((size_t)Nt * (size_t)Nf));
        GK_TRY(gk_zero_async(ctx, n_items.p, (size_t)Nt * 4));
        GK_TRY(gk_zero_async(ctx, acc.p, (size_t)Nt * (size_t)Nf * 4));
        tt_items_kernel<<<dim3((unsigned)Nt), 1024, TA_SLOTS * 8, ctx->stream>>>(P, tb->graph_ptr, w->L0, items.p, n_items.p, item_cap, ys.p);
        const unsigned per_target = (unsigned)std::max<i64>(1, 256 / Nt);
        tt_walk_kernel<<<dim3(per_target, (unsigned)Nt), 256, 0, ctx->stream>>>(P, items.p, n_items.p, item_cap, Nf, acc.p);
        tt_finish_kernel<<<grid_for(Nt * Nf, 256), 256, 0, ctx->stream>>>(acc.p, Nt, Nf, w->selfk, ys.p, normalize, K.p);
    } else {
        const int lds = (TA_COLS + 2 * TA_SLOTS + 2 * GM_MAX_NODES) * 4;
        GK_TRY(gk_func_lds(ctx, (const void*)tt_accumulate_kernel, lds));
        tt_accumulate_kernel<<<dim3((unsigned)Nt), 1024, lds, ctx->stream>>>(P, tb->graph_ptr, w->L0, Nf, w->selfk, normalize, K.p, ys.p);
    }
    GK_HIP_CHECK(hipGetLastError());
    u32 hf[4] = {0, 0, 0, 0};
    std::vector<u64> hy((size_t)Nt);
    if (one_copy) {
        if (!ctx->xfer_host) GK_HIP_CHECK(hipHostMalloc(&ctx->xfer_host, GK_XFER_BYTES, hipHostMallocDefault));
        GK_HIP_CHECK(hipMemcpyAsync(ctx->xfer_host, block.p, k_bytes + ys_bytes + 16, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const char* h = (const char*)ctx->xfer_host;
        memcpy(out_K, h, k_bytes);
        memcpy(hy.data(), h + k_bytes, ys_bytes);
        memcpy(hf, h + k_bytes + ys_bytes, 16);
    } else {
        GK_HIP_CHECK(hipMemcpyAsync(hf, flags.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipMemcpyAsync(out_K, K.p, k_bytes, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipMemcpyAsync(hy.data(), ys.p, ys_bytes, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    if (hf[0]) return GK_ERR_UNSUPPORTED;                  // a target representative above TF_MAXDEG neighbours
    for (i64 i = 0; i < Nt; ++i) out_y_selfk[i] = (double)hy[i];
    return GK_OK;
}

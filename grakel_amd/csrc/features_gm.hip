// Label-count features, GRAPH-MAJOR form (the default for graph batches): no label-grouped node order is
// needed, so the relabel sort's by-product (perm[]) is not read at all and the per-item passes of
// features.hip (flags, triple scan, counts: ~16 bytes x items x 4 passes) disappear.
//
// A graph is small (n <= 111 in every config), so one WAVE owns a graph:
//   gm_pairs_kernel  per level, the wave holds the graph's shared labels in registers (two per lane, n <= 128) and
//                    turns them into (label, count) entries without LDS atomics: a handful of distinct labels
//                    (deep levels, level 0) by one ballot round per label, otherwise an in-register bitonic
//                    sort (28 shuffle steps) + run lengths from ballots; graphs above 128 nodes use an
//                    open-addressing table in LDS.  Entries are stored compactly (ent_lab / ent_cnt / ent_n per
//                    level and graph); df[label] += 1 / count class go to workgroup-private LDS histograms
//                    (small label spaces) or guarded global atomics; exact self similarity
//                    selfk[g] = n_g x levels + sum over entries of (c^2 - c).
//                    Only nodes whose class has at least two members take part (ids of an active-set level
//                    say so themselves, wl.hip; full levels pass a flag array).
//   GmColumns scan   over the labels that can be shared (all levels concatenated): df / cmax -> column
//                    class exactly as features.hip (dense primary / int8 / float64 / rare / dead), dense
//                    column ids and the offsets of the rare labels' (graph, count) lists in ONE prefix sum.
//   gm_rows_kernel   the wave assembles its graph's whole operand row in LDS (fp4 codes OR-ed into place,
//                    int8 counts stored) and writes it once, coalesced: no zeroed staging image, no pack
//                    pass; rare entries go to their label's list (cursor atomics, order irrelevant).
// Results are identical to features.hip's (same classes, same K); column ORDER differs, which no consumer
// can observe.  Falls back to features.hip for pair batches (ShortestPath) and graphs above GM_MAX_NODES.
#include "common.h"
#include "scan_fn.h"
#include "features.h"
#include "wl_sig.h"
#include <stdlib.h>

static inline dim3 grid_for(i64 n, int t) { return dim3((unsigned)(n > 0 ? cdiv(n, t) : 1)); }

struct GmLevels {
    const i32* lab[FEAT_MAX_LEVELS];      // level labels
    const unsigned char* flag[FEAT_MAX_LEVELS];   // 1 = the node's class has >= 2 members (null: decided by id_base alone)
    i32 id_base[FEAT_MAX_LEVELS];         // ids below it are singletons for good (active-set layout), else 0
    i64 off[FEAT_MAX_LEVELS + 1];         // start of the level's labels [id_base, count) in the concatenated label space
    int level[FEAT_MAX_LEVELS];
    int L;
    __device__ __forceinline__ int slot_of(i64 q) const {
        int j = 0;
        while (q >= off[j + 1]) ++j;
        return j;
    }
};

// per shareable label q: df (graphs containing it), cmax (largest count), side flags (rectangular jobs)
struct GmLabelArrays {
    u32* df; u32* cmax; unsigned char* side;      // side: bit 0 = occurs in a fitted graph, bit 1 = in a target graph
    i32* colid; u32* roff; u32* cursor; i32* low_q;
};

#define GM_WAVES 8
#define GM_FEW 24           // at most this many distinct labels: counted by ballots instead of the sort
#define GM_FEW1 10          // ... for graphs of at most 64 nodes (one label per lane, 21 sorting steps)

// One wave per graph (persistent workgroups: a wave walks graphs w, w + stride, ...), all levels.  Per level the
// wave counts its graph's labels in a small open-addressing table in LDS (T slots, T >= 2 x the largest
// graph): a node inserts its label (compare-and-swap on the key, add one to the slot's counter); the lane
// whose insertion claimed the slot owns the (label, graph) entry.  O(n) LDS atomics per level.
//
// df / count class per label: a million (label, graph) entries are a million RANDOM L2 transactions when they
// go straight to global counters (measured 230 us, thousands of them piling onto the few hot labels).  Levels
// with a small label space (level 0, level 1, the active-set levels: P.priv_off[j] >= 0) therefore count in
// a workgroup-private LDS histogram -- 16 bits per label: graphs seen by this workgroup (12 bits), fitted /
// target side (2 bits), "a count exceeds the primary / the int8 range" (2 bits) -- written out once per
// workgroup and summed by gm_reduce_kernel.  The other levels (few entries: most of their nodes own their
// label) keep the guarded global atomics.
#define GM_PRIV_COUNT_MASK 0x0fffu
#define GM_PRIV_SIDE_SHIFT 12
#define GM_PRIV_BIG1 0x4000u          // some count > prim_max
#define GM_PRIV_BIG2 0x8000u          // some count > wide_above

struct GmPriv {
    i32 off[FEAT_MAX_LEVELS];         // first private bin of the slot, -1: global atomics
    int bins;                         // private bins in total (even)
};

// The level slots as the kernels of the graph batches read them -- from DEVICE memory, so that a job whose label counts
// only exist on the device (stream layout, wl_stream.hip) is queued without a host round trip.  Labels of slot j that can be
// shared: [0, ncc) and [lo, lo + S); their index in the concatenated label space is off + x resp. off + ncc + (x - lo).
// (host-driven layouts: ncc = 0, lo = first shareable id, S = labels from there on.)
struct GmTable {
    u32 lo[FEAT_MAX_LEVELS], S[FEAT_MAX_LEVELS], ncc[FEAT_MAX_LEVELS];
    u32 off[FEAT_MAX_LEVELS + 1];
    i32 poff[FEAT_MAX_LEVELS];        // first private bin of the slot, -1: global atomics (GmPriv)
    u32 bins;                         // private bins in total (even)
    u32 Q;                            // shareable labels of all slots = off[L]
    u32 L;
};

// First launch of a graph-major feature job: the zero fill of the per-label counters -- only the Q labels there are, not
// the bound the arrays were allocated for -- and, for the stream layout, the level table computed from the control words
// (ctl != null; host-known layouts upload their table before the launch and pass Q).
__global__ __launch_bounds__(256) void gm_prep_kernel(const u32* __restrict__ ctl, int level0, int L, u32 Q_host, int priv_ok,
                                                       u32 priv_budget, GmTable* __restrict__ Td, u32* __restrict__ df,
                                                       u32* __restrict__ cmax, u32* __restrict__ cursor, u32* __restrict__ side_words) {
    __shared__ u32 Qs;
    __shared__ u32 cw[FEAT_MAX_LEVELS][3];
    if (ctl && (int)threadIdx.x < L) {            // one round trip for all levels' words, not one per level
        const u32* c = ctl + (size_t)(level0 + (int)threadIdx.x) * SR_CTL;
        cw[threadIdx.x][0] = c[SR_NCC], cw[threadIdx.x][1] = c[SR_F], cw[threadIdx.x][2] = c[SR_S];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!ctl) Qs = Q_host;
        else {
            const bool w = blockIdx.x == 0;
            u32 off = 0, bins = 0;
            for (int j = 0; j < L; ++j) {
                const u32 ncc = cw[j][0], lo = ncc + cw[j][1], S = cw[j][2], space = ncc + S;
                i32 po = -1;
                if (priv_ok && space > 0 && bins + space <= priv_budget) po = (i32)bins, bins += space;
                if (w) Td->lo[j] = lo, Td->S[j] = S, Td->ncc[j] = ncc, Td->off[j] = off, Td->poff[j] = po;
                off += space;
            }
            if (w) Td->off[L] = off, Td->bins = (bins + 1u) & ~1u, Td->Q = off, Td->L = (u32)L;
            Qs = off;
        }
    }
    __syncthreads();
    const u32 Q = Qs;
    const u32 stride = gridDim.x * 256u;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < Q; i += stride) df[i] = 0u, cmax[i] = 0u, cursor[i] = 0u;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < (Q + 3u) / 4u; i += stride) side_words[i] = 0u;
}

// The same for a layout the HOST knows (the ShortestPath histogram form): the table travels as a kernel argument and block 0
// stores it -- no staged host-to-device copy (a launch of its own) in front of the job's first kernel.
__global__ __launch_bounds__(256) void gm_prep_table_kernel(const GmTable Tv, GmTable* __restrict__ Td, u32* __restrict__ df,
                                                             u32* __restrict__ cmax, u32* __restrict__ cursor, u32* __restrict__ side_words) {
    if (blockIdx.x == 0) {
        const u32* src = (const u32*)&Tv;
        u32* dst = (u32*)Td;
        for (int i = threadIdx.x; i < (int)(sizeof(GmTable) / 4); i += 256) dst[i] = src[i];
    }
    const u32 Q = Tv.Q;
    const u32 stride = gridDim.x * 256u;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < Q; i += stride) df[i] = 0u, cmax[i] = 0u, cursor[i] = 0u;
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < (Q + 3u) / 4u; i += stride) side_words[i] = 0u;
}

#ifdef GK_ABLATION
// tools' build only (make abl; tools/dev/feat_ablate.py): timing ablations of gm_pairs_kernel, WRONG results by construction.
// GK_GM_ABL bit 1: no df / class counting (emit), 2: no sort network, 4: no entry stores, 8: label loads only, 16 / 32: no
// counting in the private histograms / by global atomics, 64: global counting without its read-modify-writes, 128: without its guard reads
__device__ int g_gm_abl;
#define GM_ABL(bit) (g_gm_abl & (bit))
#else
#define GM_ABL(bit) 0
#endif

template <int WAVES>
__device__ __forceinline__ void gm_pairs_body(const GmLevels P, const GmLabelArrays A, const GmTable* __restrict__ Tb, int priv_cap_words,
                                                                const i32* __restrict__ graph_ptr, i64 n_graphs, i64 V,
                                                                i32* __restrict__ ent_lab, u32* __restrict__ ent_cnt, u32* __restrict__ ent_n, u64* __restrict__ selfk,
                                                                u32* __restrict__ meta, int n_levels, int kind, i64 n_fit,
                                                                int rectangular, u32 df_cap, int T, int prim_max,
                                                                int wide_above, u32* __restrict__ part, u32* __restrict__ wgmeta, int wave_max) {
    extern __shared__ __attribute__((aligned(16))) i32 gm_lds[];      // private histogram | per wave: keys[T] | count + owner << 16 [T]
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32* priv = (u32*)gm_lds;                                          // two 16-bit bins per word
    const int priv_words = (int)(Tb->bins / 2u);                       // <= priv_cap_words (the LDS image is laid out for the cap)
    for (int t = threadIdx.x; t < priv_words; t += blockDim.x) priv[t] = 0;
    __syncthreads();
    i32* keys = gm_lds + priv_cap_words + (size_t)w * 2 * T;
    u32* co = (u32*)(keys + T);
    u32 maxc = 0, entries = 0;
    for (i64 g = (i64)blockIdx.x * WAVES + w; g < n_graphs; g += (i64)gridDim.x * WAVES) {
        const i32 v0 = graph_ptr[g], v1 = graph_ptr[g + 1];
        const int n = v1 - v0;
        u64 extra = 0;
        const u32 side_bit = g < n_fit ? 1u : 2u;
        const i32 BIG = 0x7fffffff;
        const bool small = n <= 128;
        if (n == 0) {
            for (int j = lane; j < P.L; j += 64) ent_n[(i64)j * n_graphs + g] = 0;
            if (lane == 0) selfk[g] = 0;
            continue;
        }
        if (n > wave_max) continue;                           // a whole workgroup's (gm_pairs_huge_kernel)
        // the next level's labels (and flags) are fetched while the current level is counted
        i32 ra = 0, rb = 0;
        u32 fa = 0, fb = 0;                                   // 0: no node at this position / not shared
        auto fetch = [&](int j) __attribute__((always_inline)) {
            const i32* __restrict__ lab = P.lab[j];
            const unsigned char* __restrict__ fl = P.flag[j];
            ra = rb = 0, fa = fb = 0;
            if (lane < n) { ra = lab[v0 + lane]; fa = fl ? fl[v0 + lane] : 1u; }
            if (lane + 64 < n) { rb = lab[v0 + lane + 64]; fb = fl ? fl[v0 + lane + 64] : 1u; }
        };
        if (small && P.L > 0) fetch(0);
        for (int j = 0; j < P.L; ++j) {
            const i32* __restrict__ lab = P.lab[j];
            const unsigned char* __restrict__ fl = P.flag[j];
            const u32 lo = Tb->lo[j], nsh = Tb->S[j], ncc = Tb->ncc[j], qoff = Tb->off[j];
            const u32 space = Tb->off[j + 1] - qoff;
            const i32 poff = Tb->poff[j];
            // can the label be shared (see GmTable), and its index in the concatenated label space
            auto shareable = [&](i32 x) __attribute__((always_inline)) { return (u32)x - lo < nsh || (u32)x < ncc; };
            auto qof = [&](i32 x) __attribute__((always_inline)) { return (i32)(qoff + ((u32)x < ncc ? (u32)x : ncc + ((u32)x - lo))); };
            // one (label, graph, count) entry: df / class flags of the label, the graph's self similarity
            auto emit = [&](i32 x, u32 c) __attribute__((always_inline)) {
                if (GM_ABL(1) || (GM_ABL(16) && poff >= 0) || (GM_ABL(32) && poff < 0)) { extra += c; return; }
                if (poff >= 0) {
                    const u32 bin = (u32)poff + ((u32)qof(x) - qoff);
                    const int sh = 16 * (bin & 1u);
                    u32 add = 1u;
                    if (rectangular) add |= side_bit << GM_PRIV_SIDE_SHIFT;
                    if ((int)c > prim_max) add |= GM_PRIV_BIG1;
                    if ((int)c > wide_above) add |= GM_PRIV_BIG2;
                    const u32 cur = (priv[bin >> 1] >> sh) & 0xffffu;
                    const u32 flags = add & ~cur & 0xf000u;                 // flag bits not set yet
                    if (flags) atomicOr(&priv[bin >> 1], flags << sh);
                    atomicAdd(&priv[bin >> 1], 1u << sh);
                } else {
                    const i64 q = qof(x);
                    // guarded global atomics (device-scope atomics execute memory-side: slow, and a label present
                    // in thousands of graphs would queue thousands of them on one address): add only while df counts
                    // (the cost is the read-modify-writes themselves, ~0.2 ns each at config 5's half a million entries of low-df
                    // classes; the guard reads are free next to them and unguarded adds on the hot labels cost three times more)
                    if ((GM_ABL(128) ? 0u : __hip_atomic_load(&A.df[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < df_cap && !GM_ABL(64)) atomicAdd(&A.df[q], 1u);
                    if (c >= 2u && (GM_ABL(128) ? 0u : __hip_atomic_load(&A.cmax[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < c && !GM_ABL(64)) atomicMax(&A.cmax[q], c);
                    if (rectangular && !(A.side[q] & side_bit)) atomicOr((u32*)(A.side + (q & ~3ll)), side_bit << (8 * (q & 3)));
                }
                if (!kind) extra += (u64)c * c - c;
                maxc = c > maxc ? c : maxc;
                ++entries;
            };
            i32* __restrict__ el = ent_lab + (i64)j * V + v0;
            u32* __restrict__ ec = ent_cnt + (i64)j * V + v0;
            u32* __restrict__ ne = ent_n + (i64)j * n_graphs + g;
            if (small) {
                i32 a = (fa && shareable(ra)) ? ra : BIG, b2 = (fb && shareable(rb)) ? rb : BIG;
                if (j + 1 < P.L) fetch(j + 1);
                u64 Ma = __ballot(a != BIG), Mb = __ballot(b2 != BIG);
                if (GM_ABL(8)) { extra += (u64)(a ^ b2); continue; }
                if (!(Ma | Mb)) {                                    // nothing shared in this graph at this level
                    if (lane == 0) *ne = 0;
                    continue;
                }
                // (the one-register sort of a small graph is ~200 instructions: cheaper than a ballot round per label beyond ~10)
                if (__builtin_popcountll(Ma) + __builtin_popcountll(Mb) <= (n <= 64 ? GM_FEW1 : GM_FEW) || space <= (u32)(n <= 64 ? GM_FEW1 : GM_FEW)) {
                    // few distinct labels (deep levels: a handful of shared nodes per graph; level 0: a handful of
                    // labels): one wave-uniform round per distinct label -- take the first remaining node's label,
                    // ballot its equals; entry k lands in lane k
                    i32 mx = 0;
                    u32 mc = 0;
                    int k = 0;
                    while (Ma | Mb) {
                        const int src = Ma ? __builtin_ctzll(Ma) : __builtin_ctzll(Mb);
                        const i32 x = Ma ? __builtin_amdgcn_readlane(a, src) : __builtin_amdgcn_readlane(b2, src);
                        const u64 ea = __ballot(a == x), eb = __ballot(b2 == x);
                        if (lane == k) mx = x, mc = (u32)(__builtin_popcountll(ea) + __builtin_popcountll(eb));
                        Ma &= ~ea, Mb &= ~eb, ++k;
                    }
                    if (lane < k) el[lane] = qof(mx), ec[lane] = mc;
                    if (lane == 0) *ne = (u32)k;
                    if (lane < k) emit(mx, mc);
                    continue;
                }
                if (n <= 64) {
                    // ---- at most 64 nodes: ONE label per lane, 21 compare-exchange steps, none of the second register's work
                    // (config 5: 50 000 graphs of 30 nodes spent most of this kernel sorting 64 empty positions each)
#pragma unroll
                    for (int k = 2; k <= 64; k <<= 1) {
                        if (GM_ABL(2)) break;
#pragma unroll
                        for (int jj = k >> 1; jj > 0; jj >>= 1) {
                            const i32 pa = lane_xor_by(a, jj);               // (DPP / swizzle / permlane: wl_sig.h; no LDS crossbar pass)
                            const bool lower = (lane & jj) == 0;
                            const bool asc = k == 64 ? true : (lane & k) == 0;
                            const i32 mna = a < pa ? a : pa, mxa = a < pa ? pa : a;
                            a = (lower == asc) ? mna : mxa;
                        }
                    }
                    const i32 up_a = __shfl_up(a, 1, 64);
                    const bool ha = lane == 0 || a != up_a;
                    const u64 Ha = __ballot(ha);
                    const u64 above = lane == 63 ? 0ull : (~0ull << (lane + 1));
                    const u64 ma = Ha & above;
                    const int next_a = ma ? __builtin_ctzll(ma) : 64;
                    const u32 ca = (ha && a != BIG) ? (u32)(next_a - lane) : 0u;
                    const u64 Va = __ballot(ca != 0);
                    if (ca) { const int k = __builtin_popcountll(Va & ((1ull << lane) - 1ull)); el[k] = qof(a), ec[k] = ca; }
                    if (lane == 0) *ne = (u32)__builtin_popcountll(Va);
                    if (ca) emit(a, ca);
                    continue;
                }
                // ---- the graph's shared labels sorted in registers (two per lane: positions lane and lane + 64), a
                // bitonic network of 28 compare-exchange steps on wave shuffles; equal labels are then neighbours
#pragma unroll
                for (int k = 2; k <= 128; k <<= 1) {
                    if (GM_ABL(2)) break;
#pragma unroll
                    for (int jj = k >> 1; jj > 0; jj >>= 1) {
                        if (jj == 64) {                      // partner = the lane's other register (only k == 128: ascending)
                            const i32 mn = a < b2 ? a : b2, mx2 = a < b2 ? b2 : a;
                            a = mn, b2 = mx2;
                        } else {
                            const i32 pa = lane_xor_by(a, jj), pb = lane_xor_by(b2, jj);
                            const bool lower = (lane & jj) == 0;
                            // direction: ascending iff (position & k) == 0; position = lane (+ 64 for the second register)
                            const bool asc_a = k == 128 ? true : (k == 64 ? true : (lane & k) == 0);
                            const bool asc_b = k == 128 ? true : (k == 64 ? false : (lane & k) == 0);
                            const i32 mna = a < pa ? a : pa, mxa = a < pa ? pa : a;
                            const i32 mnb = b2 < pb ? b2 : pb, mxb = b2 < pb ? pb : b2;
                            a = (lower == asc_a) ? mna : mxa;
                            b2 = (lower == asc_b) ? mnb : mxb;
                        }
                    }
                }
                // ---- run heads and run lengths over the 128 sorted positions
                const i32 up_a = __shfl_up(a, 1, 64), up_b = __shfl_up(b2, 1, 64), a63 = __shfl(a, 63, 64);
                const bool ha = lane == 0 || a != up_a;
                const bool hb = b2 != (lane == 0 ? a63 : up_b);
                const u64 Ha = __ballot(ha), Hb = __ballot(hb);
                const u64 above = lane == 63 ? 0ull : (~0ull << (lane + 1));
                const u64 ma = Ha & above, mb = Hb & above;
                const int next_a = ma ? __builtin_ctzll(ma) : (Hb ? 64 + __builtin_ctzll(Hb) : 128);
                const int next_b = mb ? 64 + __builtin_ctzll(mb) : 128;
                const u32 ca = (ha && a != BIG) ? (u32)(next_a - lane) : 0u;
                const u32 cb = (hb && b2 != BIG) ? (u32)(next_b - 64 - lane) : 0u;
                // entries stored compactly: slots 0 .. ne - 1 of the graph's node range
                const u64 Va = __ballot(ca != 0), Vb = __ballot(cb != 0);
                const u64 below = (1ull << lane) - 1ull;
                const int na = __builtin_popcountll(Va);
                if (ca && !GM_ABL(4)) { const int k = __builtin_popcountll(Va & below); el[k] = qof(a), ec[k] = ca; }
                if (cb && !GM_ABL(4)) { const int k = na + __builtin_popcountll(Vb & below); el[k] = qof(b2), ec[k] = cb; }
                if (lane == 0) *ne = (u32)(na + __builtin_popcountll(Vb));
                if (ca) emit(a, ca);
                if (cb) emit(b2, cb);
            } else {
                // ---- larger graphs: open-addressing table in LDS (compare-and-swap insertion, the claimer owns the entry).
                // Round 6: the table is sized to THIS graph (2n slots, not twice the job's largest graph: a job with one
                // graph of 1 000 vertices cleared 2 048 slots per level for every graph of 130), and the entries are stored
                // compactly (the rows kernel walked a slot per node and level: 2 M slots for the 341 k vertices of the D&D-like set)
                u32 Tg = 256;
                while (Tg < 2u * (u32)n && Tg < (u32)T) Tg <<= 1;
                const u32 gmask = Tg - 1u;
                for (u32 t = lane; t < Tg; t += 64) keys[t] = -1, co[t] = 0;
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < n; i += 64) {
                    const i32 x = lab[v0 + i];
                    if (!shareable(x) || (fl && !fl[v0 + i])) continue;
                    u32 h = ((u32)x * 2654435761u) >> 8 & gmask;
                    for (;;) {
                        const i32 old = atomicCAS(&keys[h], -1, x);
                        if (old == -1) { atomicAdd(&co[h], ((u32)i << 16) | 1u); break; }
                        if (old == x) { atomicAdd(&co[h], 1u); break; }
                        h = (h + 1u) & gmask;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                u32 n_ent = 0;                                    // wave-uniform: entries stored so far
                for (int i0 = 0; i0 < n; i0 += 64) {
                    const int i = i0 + lane;
                    i32 x = 0;
                    u32 c = 0;
                    if (i < n) {
                        x = lab[v0 + i];
                        if (shareable(x) && (!fl || fl[v0 + i])) {
                            u32 h = ((u32)x * 2654435761u) >> 8 & gmask;
                            while (keys[h] != x) h = (h + 1u) & gmask;
                            const u32 e = co[h];
                            if ((e >> 16) == (u32)i) c = e & 0xffffu;
                        }
                    }
                    const u64 m = __ballot(c != 0);
                    if (c) {
                        const u32 k = n_ent + (u32)__builtin_popcountll(m & ((1ull << lane) - 1ull));     // k <= i: behind every position still to be read
                        el[k] = qof(x), ec[k] = c;
                        emit(x, c);
                    }
                    n_ent += (u32)__builtin_popcountll(m);
                }
                if (lane == 0) *ne = n_ent;
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        for (int off = 32; off > 0; off >>= 1) extra += __shfl_down(extra, off, 64);
        if (lane == 0) selfk[g] = (u64)n * (u64)n_levels + extra;
    }
    for (int off = 32; off > 0; off >>= 1) {
        entries += __shfl_down(entries, off, 64);
        const u32 o = __shfl_down(maxc, off, 64);
        maxc = o > maxc ? o : maxc;
    }
    // largest count / entries of this workgroup: one plain store per workgroup (thousands of atomics on two cache
    // lines of meta[] serialise in one L2 channel); block 0 of gm_scan_apply_kernel folds them into meta[]
    __syncthreads();                                   // the counting tables are done with: reuse their first words
    u32* red = (u32*)(gm_lds + priv_cap_words);
    if (lane == 0) red[2 * w] = maxc, red[2 * w + 1] = entries;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 m = 0, e = 0;
        for (int k = 0; k < WAVES; ++k) m = red[2 * k] > m ? red[2 * k] : m, e += red[2 * k + 1];
        wgmeta[2 * blockIdx.x] = m, wgmeta[2 * blockIdx.x + 1] = e;
    }
    u32* mine = part + (size_t)blockIdx.x * priv_words;
    for (int t = threadIdx.x; t < priv_words; t += blockDim.x) mine[t] = priv[t];
}

// Occupancy: the body needs ~106 SGPRs, and the 800-SGPR file of a SIMD then holds 6 waves.  16-wave workgroups would
// run ONE per CU (the other half of a 2-per-CU grid queues behind it: measured waves live 30 us, the kernel 73 us);
// capping the SGPRs at 80 to fit two spills to scratch and is no faster (75 us).  So: 8-wave workgroups, three per CU
// (24 waves / CU), 61 us.
__global__ __launch_bounds__(64 * GM_WAVES) void gm_pairs_kernel(const GmLevels P, const GmLabelArrays A, const GmTable* __restrict__ Tb, int priv_cap_words,
                                                                const i32* __restrict__ graph_ptr, i64 n_graphs, i64 V,
                                                                i32* __restrict__ ent_lab, u32* __restrict__ ent_cnt, u32* __restrict__ ent_n, u64* __restrict__ selfk,
                                                                u32* __restrict__ meta, int n_levels, int kind, i64 n_fit,
                                                                int rectangular, u32 df_cap, int T, int prim_max,
                                                                int wide_above, u32* __restrict__ part, u32* __restrict__ wgmeta, int wave_max) {
    gm_pairs_body<GM_WAVES>(P, A, Tb, priv_cap_words, graph_ptr, n_graphs, V, ent_lab, ent_cnt, ent_n, selfk, meta, n_levels, kind, n_fit, rectangular, df_cap, T, prim_max, wide_above, part, wgmeta, wave_max);
}
// Graphs above GM_MAX_NODES vertices (round 6: protein contact graphs of thousands of residues, discussion threads of
// thousands of posts -- the D&D- and REDDIT-like sets): ONE WORKGROUP counts such a graph, level by level, in a 16 384-slot
// table that fills its LDS -- the wave form's large-graph path with workgroup barriers.  Rounds 1-5 sent every job with one
// such graph to the label-major builder, and with it to the relabel route with full sorts (its builder reads the
// label-grouped node order): the twelve large graphs of the D&D-like set decided the route of the other 1 166.
// Runs AFTER gm_reduce: its per-label statistics go onto the reduced ones with the guarded global atomics.
#define GMH_T 16384
#define GMH_THREADS 1024
__global__ __launch_bounds__(GMH_THREADS) void gm_pairs_huge_kernel(const GmLevels P, const GmLabelArrays A, const GmTable* __restrict__ Tb,
                                                                    const i32* __restrict__ graph_ptr, i64 n_graphs, i64 V,
                                                                    i32* __restrict__ ent_lab, u32* __restrict__ ent_cnt, u32* __restrict__ ent_n,
                                                                    u64* __restrict__ selfk, int n_levels, int kind, i64 n_fit, int rectangular,
                                                                    u32 df_cap, u32* __restrict__ wgmeta, int wg_base, int wave_max) {
    extern __shared__ __attribute__((aligned(16))) i32 gm_lds[];      // keys[GMH_T] | count + owner << 16 [GMH_T]
    __shared__ u64 red_x[GMH_THREADS / 64];
    __shared__ u32 red_m[GMH_THREADS / 64], red_e[GMH_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // (every LDS access indexes the one dynamic array: with derived pointers the compiler loses the address space of the
    // atomics -- flat instructions on an LDS offset, a fault at address 0 on gfx950)
    u32* const lds_u = (u32*)gm_lds;
#define co(h) lds_u[GMH_T + (int)(h)]
    u32 maxc = 0, entries = 0;
    for (i64 g = blockIdx.x; g < n_graphs; g += gridDim.x) {
        const i32 v0 = graph_ptr[g];
        const int n = graph_ptr[g + 1] - v0;
        if (n <= wave_max) continue;                          // (workgroup-uniform) a wave's
        const u32 side_bit = g < n_fit ? 1u : 2u;
        u64 extra = 0;
        u32 Tg = 4096;                                        // this graph's table: 2n slots
        while (Tg < 2u * (u32)n && Tg < (u32)GMH_T) Tg <<= 1;
        const u32 tmask = Tg - 1u;
        for (int j = 0; j < P.L; ++j) {
            const i32* __restrict__ lab = P.lab[j];
            const unsigned char* __restrict__ fl = P.flag[j];
            const u32 lo = Tb->lo[j], nsh = Tb->S[j], ncc = Tb->ncc[j], qoff = Tb->off[j];
            auto shareable = [&](i32 x) __attribute__((always_inline)) { return (u32)x - lo < nsh || (u32)x < ncc; };
            auto qof = [&](i32 x) __attribute__((always_inline)) { return (i32)(qoff + ((u32)x < ncc ? (u32)x : ncc + ((u32)x - lo))); };
            i32* __restrict__ el = ent_lab + (i64)j * V + v0;
            u32* __restrict__ ec = ent_cnt + (i64)j * V + v0;
            __syncthreads();                                  // the level before is done with the table
            for (u32 t = tid; t < Tg; t += GMH_THREADS) gm_lds[t] = -1, co(t) = 0;
            if (tid == 0) lds_u[2 * GMH_T] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += GMH_THREADS) {
                const i32 x = lab[v0 + i];
                if (!shareable(x) || (fl && !fl[v0 + i])) continue;
                u32 h = ((u32)x * 2654435761u) >> 8 & tmask;
                for (;;) {
                    const i32 old = atomicCAS(&gm_lds[h], -1, x);
                    if (old == -1) { atomicAdd(&co(h), ((u32)i << 16) | 1u); break; }       // the claimer owns the entry
                    if (old == x) { atomicAdd(&co(h), 1u); break; }
                    h = (h + 1u) & tmask;
                }
            }
            __syncthreads();
            // the entries are stored compactly (any order): a wave reserves its slots with ONE LDS atomic from lane 0 and places them
            // by ballot -- the per-lane atomic on one counter that this replaced was aggregated by the compiler into a sequence
            // that faulted on gfx950 (round 6; tools/dev/huge_dbg.py reproduces the job)
            for (int i0 = w * 64; i0 < n; i0 += GMH_THREADS) {
                const int i = i0 + lane;
                i32 x = 0;
                u32 c = 0;
                if (i < n) {
                    x = lab[v0 + i];
                    if (shareable(x) && (!fl || fl[v0 + i])) {
                        u32 h = ((u32)x * 2654435761u) >> 8 & tmask;
                        while (gm_lds[h] != x) h = (h + 1u) & tmask;
                        const u32 e = co(h);
                        if ((e >> 16) == (u32)i) c = e & 0xffffu;
                    }
                }
                const u64 m = __ballot(c != 0);
                u32 base = 0;
                if (lane == 0 && m) base = atomicAdd(&lds_u[2 * GMH_T], (u32)__builtin_popcountll(m));
                base = __shfl(base, 0, 64);
                if (c) {
                    const u32 k = base + (u32)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    const i64 q = qof(x);
                    el[k] = (i32)q, ec[k] = c;
                    if (__hip_atomic_load(&A.df[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < df_cap) atomicAdd(&A.df[q], 1u);
                    if (c >= 2u && __hip_atomic_load(&A.cmax[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c) atomicMax(&A.cmax[q], c);
                    if (rectangular && !(A.side[q] & side_bit)) atomicOr((u32*)(A.side + (q & ~3ll)), side_bit << (8 * (q & 3)));
                    if (!kind) extra += (u64)c * c - c;
                    maxc = c > maxc ? c : maxc;
                    ++entries;
                }
            }
            __syncthreads();
            if (tid == 0) ent_n[(i64)j * n_graphs + g] = lds_u[2 * GMH_T];
        }
        for (int off = 32; off > 0; off >>= 1) extra += __shfl_down(extra, off, 64);
        __syncthreads();
        if (lane == 0) red_x[w] = extra;
        __syncthreads();
        if (tid == 0) {
            u64 x = 0;
            for (int k = 0; k < GMH_THREADS / 64; ++k) x += red_x[k];
            selfk[g] = (u64)n * (u64)n_levels + x;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        entries += __shfl_down(entries, off, 64);
        const u32 o = __shfl_down(maxc, off, 64);
        maxc = o > maxc ? o : maxc;
    }
    __syncthreads();
    if (lane == 0) red_m[w] = maxc, red_e[w] = entries;
    __syncthreads();
    if (tid == 0) {
        u32 m = 0, e = 0;
        for (int k = 0; k < GMH_THREADS / 64; ++k) m = red_m[k] > m ? red_m[k] : m, e += red_e[k];
        wgmeta[2 * (wg_base + (int)blockIdx.x)] = m, wgmeta[2 * (wg_base + (int)blockIdx.x) + 1] = e;
    }
}
#undef co

// sum the workgroups' private histograms: df (saturating at what the column scan distinguishes), the count class as
// a representative cmax (1, prim_max + 1 or wide_above + 1), the side bits
__global__ __launch_bounds__(1024) void gm_reduce_kernel(const GmLabelArrays A, const GmTable* __restrict__ Tb,
                                                         const u32* __restrict__ part, int n_wg, int prim_max, int wide_above,
                                                         int rectangular) {
    // block = 64 consecutive words x 16 row groups (wave w sums the workgroups w, w + 16, ...), combined in LDS
    __shared__ u32 sc0[16][64], sc1[16][64], sf[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int words = (int)(Tb->bins / 2u);
    if ((int)blockIdx.x * 64 >= words) return;                 // the grid covers the cap of the private bins
    const int t = blockIdx.x * 64 + lane;
    u32 c0 = 0, c1 = 0, fl = 0;
    if (t < words)
        for (int g = w; g < n_wg; g += 64) {       // four rows in flight per wave (one load per trip was a chain of L2 latencies)
            u32 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = g + 16 * u < n_wg ? part[(size_t)(g + 16 * u) * words + t] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 += x[u] & GM_PRIV_COUNT_MASK, c1 += (x[u] >> 16) & GM_PRIV_COUNT_MASK;
                fl |= (x[u] & 0xf000u) | ((x[u] >> 16 & 0xf000u) << 16);
            }
        }
    sc0[w][lane] = c0, sc1[w][lane] = c1, sf[w][lane] = fl;
    __syncthreads();
    if (w != 0 || t >= words) return;
    for (int k = 1; k < 16; ++k) c0 += sc0[k][lane], c1 += sc1[k][lane], fl |= sf[k][lane];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const u32 bin = 2u * (u32)t + (u32)k, c = k ? c1 : c0, f = k ? (fl >> 16) : (fl & 0xffffu);
        int j = -1;
        for (int s = 0; s < (int)Tb->L; ++s)
            if (Tb->poff[s] >= 0 && (i64)bin >= Tb->poff[s] && (i64)bin < (i64)Tb->poff[s] + (i64)(Tb->off[s + 1] - Tb->off[s])) j = s;
        if (j < 0) continue;                                   // padding bin
        const i64 q = (i64)Tb->off[j] + (bin - (u32)Tb->poff[j]);
        A.df[q] = c;
        A.cmax[q] = (f & GM_PRIV_BIG2) ? (u32)wide_above + 1u : ((f & GM_PRIV_BIG1) ? (u32)prim_max + 1u : (c ? 1u : 0u));
        if (rectangular) A.side[q] = (unsigned char)((f >> GM_PRIV_SIDE_SHIFT) & 3u);
    }
}

// Column classes per shareable label, one prefix sum for everything:
//   a = primary dense columns (low 32) | secondary int8 columns (high 32)
//   b = float64 columns (low 32) | rare labels (high 32)
//   c = rare (graph, count) entries before this label
struct Gm3 {
    u64 a, b, c;
    __device__ __forceinline__ Gm3& operator+=(const Gm3& o) { a += o.a, b += o.b, c += o.c; return *this; }
};
__device__ __forceinline__ Gm3 gm3_shfl_up(const Gm3& x, int off) {
    Gm3 y;
    y.a = __shfl_up(x.a, off, 64), y.b = __shfl_up(x.b, off, 64), y.c = __shfl_up(x.c, off, 64);
    return y;
}
__device__ __forceinline__ Gm3 gm3_shfl_down(const Gm3& x, int off) {
    Gm3 y;
    y.a = __shfl_down(x.a, off, 64), y.b = __shfl_down(x.b, off, 64), y.c = __shfl_down(x.c, off, 64);
    return y;
}

struct GmColumns {
    GmLabelArrays A; const GmTable* Tb; int symmetric; int low_df; int kind; int prim_max; int wide_above; u32* meta;
    const u32* wgmeta; int n_wg;                  // gm_pairs_kernel's per-workgroup (largest count, entries)
    int allow_split;                              // labels with counts above wide_above: split int8 columns (features.h) when <= GM_SPLIT_MAX_PARTS parts do
    const u32* dyn;                               // ShortestPath histogram jobs: meta[GM_META_TYPE], 2 = every dense column to the float64 operand (nullptr: the type is the host's)
    // parts of the split columns from the largest count of the job: every workgroup of the scans derives the same from the pair
    // kernel's per-workgroup maxima, all of its threads loading at once (a single wave walking the list was 3 us per workgroup)
    __device__ __forceinline__ int parts_by_block(u32* red /* [G3_THREADS / 64] shared */) const {
        if (!allow_split) return 0;
        u32 m = 0;
        for (int k = threadIdx.x; k < n_wg; k += blockDim.x) m = wgmeta[2 * k] > m ? wgmeta[2 * k] : m;
        for (int off = 32; off > 0; off >>= 1) {
            const u32 o = __shfl_xor(m, off, 64);
            m = o > m ? o : m;
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        m = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = red[w] > m ? red[w] : m;
        __syncthreads();
        const u32 p = (m + 126u) / 127u;
        return (p >= 2u && p <= (u32)GM_SPLIT_MAX_PARTS) ? (int)p : 0;
    }
    __device__ __forceinline__ Gm3 value(i64 q, int parts) const {
        Gm3 v{0, 0, 0};
        const u32 df = A.df[q];
        if (df == 0) return v;
        int prim_max = this->prim_max, wide_above = this->wide_above;
        if (dyn) {
            const u32 type = *dyn;
            if (type == 2u) prim_max = -1, wide_above = -1, parts = 0;
            else if (type == 3u) prim_max = 127, wide_above = 127, parts = 0;       // int8 + float64 side operand
        }
        const bool useful = symmetric ? df >= 2u : A.side[q] == 3;
        if (!useful) return v;
        if ((int)df < low_df) { v.b = 1ull << 32, v.c = df; return v; }
        const u32 m = A.cmax[q] ? A.cmax[q] : 1u;
        if (kind) v.a = m;                                   // unary expansion: one 0/1 column per count level
        else if ((int)m <= prim_max) v.a = 1;
        else if ((int)m <= wide_above) v.a = 1ull << 32;
        else if (parts) v.a = (u64)(parts * parts) << 32, v.b = 1;      // split: parts^2 int8 columns (v.b counts the labels)
        else v.b = 1;
        return v;
    }
    __device__ __forceinline__ void emit(i64 q, const Gm3& v, const Gm3& incl) const {
        i32 c = -1;                                          // dead
        if (v.b >> 32) {                                     // rare
            c = -2;
            A.roff[q] = (u32)(incl.c - v.c);
            A.low_q[(u32)(incl.b >> 32) - 1] = (i32)q;
        } else if (v.a & 0xffffffffull) c = (i32)((u32)(incl.a & 0xffffffffull) - (u32)(v.a & 0xffffffffull));
        else if ((v.a >> 32) > 1) c = COL_SPLIT_BASE + (i32)((u32)(incl.a >> 32) - (u32)(v.a >> 32));
        else if (v.a >> 32) c = COL_BYTE_BASE + (i32)((u32)(incl.a >> 32) - 1);
        else if (v.b & 0xffffffffull) c = -4 - (i32)((u32)(incl.b & 0xffffffffull) - 1);
        A.colid[q] = c;
    }
    __device__ __forceinline__ void finish(const Gm3& t, int parts) const {
        meta[GM_META_SPLIT] = (u32)parts;
        meta[GM_META_PRIM] = (u32)(t.a & 0xffffffffull), meta[GM_META_INT8] = (u32)(t.a >> 32);
        meta[GM_META_F64] = (u32)(t.b & 0xffffffffull), meta[GM_META_RARE] = (u32)(t.b >> 32);
        meta[GM_META_RARE_ENTRIES] = (u32)t.c;
    }
};

// scan_fn.h specialised to the three-word sum (its templates need arithmetic shuffles of T)
#define G3_THREADS 256
#define G3_ITEMS 4
#define G3_TILE (G3_THREADS * G3_ITEMS)
__global__ __launch_bounds__(G3_THREADS) void gm_scan_sums_kernel(const GmColumns f, Gm3* __restrict__ partial) {
    __shared__ Gm3 wsum[G3_THREADS / 64];
    const i64 base = (i64)blockIdx.x * G3_TILE;
    const i64 Q = f.Tb->Q;                      // the grid covers the bound the arrays were allocated for
    if (base >= Q) return;                      // (gm_scan_apply_kernel only reads the sums of the tiles in use)
    __shared__ u32 pred[G3_THREADS / 64];
    const int parts = f.parts_by_block(pred);
    Gm3 s{0, 0, 0};
#pragma unroll
    for (int i = 0; i < G3_ITEMS; ++i) {
        const i64 idx = base + (i64)i * G3_THREADS + threadIdx.x;
        if (idx < Q) s += f.value(idx, parts);
    }
    for (int off = 32; off > 0; off >>= 1) s += gm3_shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        Gm3 t{0, 0, 0};
        for (int i = 0; i < G3_THREADS / 64; ++i) t += wsum[i];
        partial[blockIdx.x] = t;
    }
}

// The operand sizes as soon as they are known (round 6): the tile sums of gm_scan_sums_kernel are all the host waits for --
// column counts per class, rare entries, the largest count -- so one workgroup adds them up, writes meta[] and POSTS the
// mailbox itself (meta[], then the control words of a queued stream relabel); gm_scan_apply_kernel, which only turns the
// same sums into column ids, runs behind it while the host is on its round trip.
__global__ __launch_bounds__(G3_THREADS) void gm_totals_post_kernel(const GmColumns f, const Gm3* __restrict__ partial, int n_tiles_cap,
                                                                    int n_meta, const u32* __restrict__ ctl, int n_ctl,
                                                                    u32* __restrict__ mbox, u32 seq) {
    __shared__ Gm3 bsum[G3_THREADS / 64];
    __shared__ u32 pred[G3_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const i64 Q = f.Tb->Q;
    const int n_tiles = (int)((Q + G3_TILE - 1) / G3_TILE) < n_tiles_cap ? (int)((Q + G3_TILE - 1) / G3_TILE) : n_tiles_cap;
    const int parts = f.parts_by_block(pred);
    if (w == 0) {                                      // the pair kernels' per-workgroup statistics
        u32 m = 0, e = 0;
        for (int k = lane; k < f.n_wg; k += 64) m = f.wgmeta[2 * k] > m ? f.wgmeta[2 * k] : m, e += f.wgmeta[2 * k + 1];
        for (int off = 32; off > 0; off >>= 1) {
            const u32 o = __shfl_down(m, off, 64);
            m = o > m ? o : m, e += __shfl_down(e, off, 64);
        }
        if (lane == 0) f.meta[GM_META_MAXC] = m, f.meta[GM_META_NNZ] = e;
    }
    Gm3 s{0, 0, 0};
    for (int i = threadIdx.x; i < n_tiles; i += G3_THREADS) s += partial[i];
    for (int off = 32; off > 0; off >>= 1) s += gm3_shfl_down(s, off);
    if (lane == 0) bsum[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        Gm3 t{0, 0, 0};
        for (int q = 0; q < G3_THREADS / 64; ++q) t += bsum[q];
        f.finish(t, parts);
    }
    __threadfence();
    __syncthreads();
    for (int i = threadIdx.x; i < n_meta + n_ctl; i += G3_THREADS)
        __hip_atomic_store(&mbox[1 + i], i < n_meta ? __hip_atomic_load(&f.meta[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ctl[i - n_meta],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(&mbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(G3_THREADS) void gm_scan_apply_kernel(const GmColumns f, const Gm3* __restrict__ partial) {
    __shared__ Gm3 wsum[G3_THREADS / 64];
    __shared__ Gm3 bsum[G3_THREADS / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const i64 Q = f.Tb->Q;
    const i64 last_tile = Q > 0 ? (Q - 1) / G3_TILE : 0;
    if ((i64)blockIdx.x > last_tile) return;         // the grid covers the bound the arrays were allocated for
    __shared__ u32 pred[G3_THREADS / 64];
    const int parts = f.parts_by_block(pred);
    if (blockIdx.x == 0 && w == 0) {                  // fold the pair kernel's per-workgroup statistics
        u32 m = 0, e = 0;
        for (int k = lane; k < f.n_wg; k += 64) m = f.wgmeta[2 * k] > m ? f.wgmeta[2 * k] : m, e += f.wgmeta[2 * k + 1];
        for (int off = 32; off > 0; off >>= 1) {
            const u32 o = __shfl_down(m, off, 64);
            m = o > m ? o : m, e += __shfl_down(e, off, 64);
        }
        if (lane == 0) f.meta[GM_META_MAXC] = m, f.meta[GM_META_NNZ] = e;
    }
    Gm3 s{0, 0, 0};
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += G3_THREADS) s += partial[i];
    for (int off = 32; off > 0; off >>= 1) s += gm3_shfl_down(s, off);
    if (lane == 0) bsum[w] = s;
    __syncthreads();
    Gm3 carry{0, 0, 0};
    for (int q = 0; q < G3_THREADS / 64; ++q) carry += bsum[q];
    const i64 tile0 = (i64)blockIdx.x * G3_TILE;
#pragma unroll
    for (int i = 0; i < G3_ITEMS; ++i) {
        const i64 idx = tile0 + (i64)i * G3_THREADS + threadIdx.x;
        Gm3 v{0, 0, 0};
        if (idx < Q) v = f.value(idx, parts);
        Gm3 inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const Gm3 y = gm3_shfl_up(inc, off);
            if (lane >= off) inc += y;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        Gm3 woff{0, 0, 0}, row{0, 0, 0};
#pragma unroll
        for (int q = 0; q < G3_THREADS / 64; ++q) {
            if (q < w) woff += wsum[q];
            row += wsum[q];
        }
        if (idx < Q) {
            Gm3 incl = carry;
            incl += woff;
            incl += inc;
            f.emit(idx, v, incl);
        }
        carry += row;
        __syncthreads();
    }
    if ((i64)blockIdx.x == last_tile && threadIdx.x == 0) f.finish(carry, parts);
}

// digits of a split column group (features.h: COL_SPLIT_BASE): left rows hold digit p at p*parts+r, right rows digit r
__device__ __forceinline__ void gm_split_write(unsigned char* row, i32 base, u32 c, int parts, int right) {
    for (int p = 0; p < parts; ++p) {
        const u32 lo = 127u * (u32)p;
        const unsigned char a = (unsigned char)(c > lo ? (c - lo > 127u ? 127u : c - lo) : 0u);
        for (int r = 0; r < parts; ++r) row[base + (right ? r * parts + p : p * parts + r)] = a;
    }
}

__device__ __forceinline__ u32 gm_wave_incl_scan(u32 x) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off, 64);
        if ((int)(threadIdx.x & 63) >= off) x += y;
    }
    return x;
}

// One entry (label q with column id col, count c) of graph g into the operand row being assembled in LDS (or, for the float64
// side operand and the rare lists, to their places in HBM).  Returns 1 for a split column.
__device__ __forceinline__ int gm_row_entry(unsigned char* row, const GmLabelArrays& A, i64 g, i64 q, i32 col, u32 c, i64 prim0, int fp4,
                                            int kind, int parts, double* __restrict__ phi_w, i64 ldw, i32* __restrict__ low_graph,
                                            i32* __restrict__ low_cnt, i32* __restrict__ low_lab) {
    if (col >= COL_SPLIT_BASE) { gm_split_write(row, col - COL_SPLIT_BASE, c, parts, 0); return 1; }
    if (col >= COL_BYTE_BASE) row[col - COL_BYTE_BASE] = (unsigned char)c;              // secondary int8 region: bytes [0, prim0)
    else if (col >= 0) {
        if (kind) {                                                                     // unary run of ones
            for (u32 x = 0; x < c; ++x) {
                const i64 cc = col + x;
                if (fp4) atomicOr((u32*)(row + prim0 + ((cc >> 1) & ~3ll)), 2u << (8 * ((cc >> 1) & 3) + 4 * (cc & 1)));
                else row[prim0 + cc] = 1;
            }
        } else if (fp4) {
            const u32 code = (0x65420u >> (4 * c)) & 15u;
            atomicOr((u32*)(row + prim0 + ((col >> 1) & ~3)), code << (8 * ((col >> 1) & 3) + 4 * (col & 1)));
        } else row[prim0 + col] = (unsigned char)c;
    } else if (col <= -4) phi_w[g * ldw + (-4 - col)] = (double)c;
    else if (col == -2) {
        const u32 pos = A.roff[q] + atomicAdd(&A.cursor[q], 1u);
        low_graph[pos] = (i32)g, low_cnt[pos] = (i32)c, low_lab[pos] = (i32)q;
    }
    return 0;
}

// The graphs' entries are read in trips of GM_ROW_BATCH per thread: first every (count, label) pair of the trip, then every
// column id, then the writes.  One entry at a time is a chain of three dependent memory latencies (slot -> label -> column id)
// per loop trip, and that chain, not bandwidth, bound these kernels (30 us at config 3 for a 25 MB operand).
// The entry slots of level j are [0, slots[j]) of the graph's node range: entry k of the graph (all levels back to back) is
// slot k - pre[j] of the level with pre[j] <= k < pre[j + 1].
#define GM_ROW_BATCH 4
#define GM_ROWS_GATHER(TID, STRIDE)                                                                                                  \
    for (int k0 = 0; k0 < n_ent; k0 += GM_ROW_BATCH * (STRIDE)) {                                                                    \
        u32 cc[GM_ROW_BATCH];                                                                                                        \
        i64 qq[GM_ROW_BATCH];                                                                                                        \
        i32 col[GM_ROW_BATCH];                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < GM_ROW_BATCH; ++u) {                                                                   \
            const int k = k0 + u * (STRIDE) + (TID);                                                                                 \
            cc[u] = 0u, qq[u] = 0;                                                                                                   \
            if (k < n_ent) {                                                                                                         \
                int j = 0;                                                                                                           \
                while (j + 1 < P.L && (int)pre[j + 1] <= k) ++j;                                                                     \
                const i64 at = (i64)j * V + v0 + (k - (int)pre[j]);                                                                  \
                cc[u] = cnt[at], qq[u] = ent_lab[at];                                                                                \
            }                                                                                                                        \
        }                                                                                                                            \
        _Pragma("unroll") for (int u = 0; u < GM_ROW_BATCH; ++u) col[u] = cc[u] ? A.colid[qq[u]] : -1;                               \
        _Pragma("unroll") for (int u = 0; u < GM_ROW_BATCH; ++u)                                                                     \
            if (cc[u]) { GM_ROWS_BODY(qq[u], col[u], cc[u]) }                                                                        \
    }

// one workgroup per graph: the operand row in LDS (row_bytes <= GM_ROW_LDS_MAX), written once
__global__ __launch_bounds__(1024) void gm_rows_kernel(const GmLevels P, const GmLabelArrays A,
                                                      const i32* __restrict__ graph_ptr, i64 V, const i32* __restrict__ ent_lab,
                                                      const u32* __restrict__ cnt, const u32* __restrict__ ent_n, i64 n_graphs,
                                                      int8_t* __restrict__ phi, i64 ld, i64 prim0 /* first byte of the primary region */,
                                                      int fp4, int kind, double* __restrict__ phi_w, i64 ldw,
                                                      i32* __restrict__ low_graph, i32* __restrict__ low_cnt, i32* __restrict__ low_lab,
                                                      int8_t* __restrict__ phi_r, int parts, i64 own_lo, i64 own_hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char row[];
    const i64 g = blockIdx.x;
    if (g >= n_graphs) {                                         // a padding row (the grid covers n_rows_pad): zeros, both operands
        uint4* z = (uint4*)(phi + g * ld);
        for (i64 i = threadIdx.x; i < ld / 16; i += blockDim.x) z[i] = make_uint4(0, 0, 0, 0);
        if (phi_r) {
            z = (uint4*)(phi_r + g * ld);
            for (i64 i = threadIdx.x; i < ld / 16; i += blockDim.x) z[i] = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    // multi-GPU, the operand-row exchange (grakel_amd/dist.py: exchange="phi"): a rank assembles and stores the rows of ITS
    // graphs [own_lo, own_hi) only -- the others arrive with the all-gather --, but every graph's rare-label entries and
    // float64 side columns are still listed: those stay complete on every rank
    const bool own = g >= own_lo && g < own_hi;
    const i32 v0 = graph_ptr[g];
    __shared__ u32 pre[FEAT_MAX_LEVELS + 1];                     // entries of this graph in the levels before level j
    if (threadIdx.x < 64) {
        const u32 sl = (int)threadIdx.x < P.L ? ent_n[(i64)threadIdx.x * n_graphs + g] : 0u;
        const u32 inc = gm_wave_incl_scan(sl);
        if ((int)threadIdx.x < P.L) pre[threadIdx.x] = inc - sl;
        if ((int)threadIdx.x == P.L - 1) pre[P.L] = inc;
    }
    for (i64 i = threadIdx.x; i < ld / 16; i += blockDim.x) ((uint4*)row)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const int n_ent = P.L > 0 ? (int)pre[P.L] : 0;
    int seen_split = 0;
#define GM_ROWS_BODY(Q_, COL_, C_) if (own || COL_ < 0) seen_split |= gm_row_entry(row, A, g, Q_, COL_, C_, prim0, fp4, kind, parts, phi_w, ldw, low_graph, low_cnt, low_lab);
    GM_ROWS_GATHER((int)threadIdx.x, (int)blockDim.x)
#undef GM_ROWS_BODY
    if (!own) return;                                            // (workgroup-uniform)
    __syncthreads();
    uint4* dst = (uint4*)(phi + g * ld);
    for (i64 i = threadIdx.x; i < ld / 16; i += blockDim.x) dst[i] = ((const uint4*)row)[i];
    if (!phi_r) return;
    // the right operand's row: the same but for the digits of the split columns
    if (__syncthreads_or(seen_split)) {
#define GM_ROWS_BODY(Q_, COL_, C_) if (COL_ >= COL_SPLIT_BASE) gm_split_write(row, COL_ - COL_SPLIT_BASE, C_, parts, 1);
        GM_ROWS_GATHER((int)threadIdx.x, (int)blockDim.x)
#undef GM_ROWS_BODY
        __syncthreads();
    }
    dst = (uint4*)(phi_r + g * ld);
    for (i64 i = threadIdx.x; i < ld / 16; i += blockDim.x) dst[i] = ((const uint4*)row)[i];
}

// The same with one WAVE per graph (four graphs per workgroup) for operand rows of up to GM_ROW_WAVE_MAX bytes: 50 000
// graphs of 30 nodes are 50 000 workgroups of 256 threads for 180 entry slots each in the form above (112 us, bound by
// workgroup dispatch); wave-level barriers only.
#define GM_ROW_WAVE_MAX 8192
__global__ __launch_bounds__(256) void gm_rows_wave_kernel(const GmLevels P, const GmLabelArrays A,
                                                           const i32* __restrict__ graph_ptr, i64 V, const i32* __restrict__ ent_lab,
                                                           const u32* __restrict__ cnt, const u32* __restrict__ ent_n, i64 n_graphs,
                                                           int8_t* __restrict__ phi, i64 ld, i64 prim0, int fp4, int kind,
                                                           double* __restrict__ phi_w, i64 ldw, i32* __restrict__ low_graph,
                                                           i32* __restrict__ low_cnt, i32* __restrict__ low_lab, i64 n_rows_pad,
                                                           int8_t* __restrict__ phi_r, int parts, i64 own_lo, i64 own_hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char row_all[];
    __shared__ u32 pre_all[4][FEAT_MAX_LEVELS + 1];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const i64 g = (i64)blockIdx.x * 4 + w;
    if (g >= n_graphs) {                                          // wave-uniform: no workgroup barrier below
        // the padding rows [n_graphs, n_rows_pad) of the operand are all zero (tile loads need no row guards)
        if (g < n_rows_pad) {
            uint4* dst = (uint4*)(phi + g * ld);
            for (i64 i = lane; i < ld / 16; i += 64) dst[i] = make_uint4(0, 0, 0, 0);
            if (phi_r) {
                dst = (uint4*)(phi_r + g * ld);
                for (i64 i = lane; i < ld / 16; i += 64) dst[i] = make_uint4(0, 0, 0, 0);
            }
        }
        return;
    }
    unsigned char* row = row_all + (size_t)w * ld;
    u32* pre = pre_all[w];
    const i32 v0 = graph_ptr[g];
    {
        const u32 sl = lane < P.L ? ent_n[(i64)lane * n_graphs + g] : 0u;
        const u32 inc = gm_wave_incl_scan(sl);
        if (lane < P.L) pre[lane] = inc - sl;
        if (lane == P.L - 1) pre[P.L] = inc;
    }
    for (i64 i = lane; i < ld / 16; i += 64) ((uint4*)row)[i] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int n_ent = P.L > 0 ? (int)pre[P.L] : 0;
    int seen_split = 0;
    const bool own = g >= own_lo && g < own_hi;                 // see gm_rows_kernel (wave-uniform)
#define GM_ROWS_BODY(Q_, COL_, C_) if (own || COL_ < 0) seen_split |= gm_row_entry(row, A, g, Q_, COL_, C_, prim0, fp4, kind, parts, phi_w, ldw, low_graph, low_cnt, low_lab);
    GM_ROWS_GATHER(lane, 64)
#undef GM_ROWS_BODY
    if (!own) return;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint4* dst = (uint4*)(phi + g * ld);
    for (i64 i = lane; i < ld / 16; i += 64) dst[i] = ((const uint4*)row)[i];
    if (!phi_r) return;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (__any(seen_split)) {
#define GM_ROWS_BODY(Q_, COL_, C_) if (COL_ >= COL_SPLIT_BASE) gm_split_write(row, COL_ - COL_SPLIT_BASE, C_, parts, 1);
        GM_ROWS_GATHER(lane, 64)
#undef GM_ROWS_BODY
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    dst = (uint4*)(phi_r + g * ld);
    for (i64 i = lane; i < ld / 16; i += 64) dst[i] = ((const uint4*)row)[i];
}

// Second half of the graph-major builder, shared with the ShortestPath histogram form (gk_features_build_sp): column classes
// from df / cmax, operand sizes (one host read-back), operand rows, rare lists.  graph_ptr = item ranges of the graphs
// (node ranges, or pair ranges of a pair batch), V = items in all; ent / cnt / ent_n = the graphs' (label, count) entries.
// Q = bound of the label space (the arrays' allocation; the label count itself is Tb->Q on the device)
static int gm_finish(gk_ctx* ctx, gk_feat* f, const GmLevels& P, GmLabelArrays& A, const GmTable* Tb, i64 Q, const i32* graph_ptr, i64 N, i64 V,
                     const i32* ent, const u32* cnt, const u32* ent_n, const u32* wgmeta, int grid, int prim_max, int wide_above,
                     bool huge_graphs = false) {
    const int kind = f->kind;
    void* q = nullptr;
    std::vector<u32> h(GM_META_WORDS, 0);
    u32 early_seq = 0;
    if (Q > 0) {
        // counts above 127 under an int8 operand: split columns unless the option keeps the float64 side operand
        const int allow_split = (kind == GK_FEAT_DOT && (f->dtype == 0 || f->dyn_type) && !ctx->opt.gram_no_split8) ? 1 : 0;
        GmColumns gc{A, Tb, f->symmetric ? 1 : 0, f->low_df, kind, prim_max, wide_above, f->meta, wgmeta, grid, allow_split,
                     f->dyn_type ? f->meta + GM_META_TYPE : nullptr};
        const i64 nblk = cdiv(Q, G3_TILE);
        Tmp<Gm3> partial(ctx);
        GK_TRY(partial.alloc((size_t)nblk));
        // the sizes are posted from the tile sums, the column ids follow behind the post (gm_totals_post_kernel)
        const int nw0 = (f->batch && f->batch->sr_pending > 0) ? f->batch->sr_pending * SR_CTL : 0;
        if (GM_META_WORDS + nw0 <= GK_MBOX_WORDS - 1 && !ctx->opt.gm_no_early_post) early_seq = gk_mbox_begin(ctx);
        if (nblk > 1 || early_seq) gm_scan_sums_kernel<<<dim3((unsigned)nblk), G3_THREADS, 0, ctx->stream>>>(gc, partial.p);
        if (early_seq)
            gm_totals_post_kernel<<<1, G3_THREADS, 0, ctx->stream>>>(gc, partial.p, (int)nblk, GM_META_WORDS, nw0 ? f->batch->sr_ctl : nullptr, nw0,
                                                                     ctx->mbox_dev, early_seq);
        gm_scan_apply_kernel<<<dim3((unsigned)nblk), G3_THREADS, 0, ctx->stream>>>(gc, partial.p);
    }
    gk_batch* fb = f->batch;
    if (early_seq) {
        const int nw = (fb && fb->sr_pending > 0) ? fb->sr_pending * SR_CTL : 0;
        std::vector<u32> h2((size_t)GM_META_WORDS + (size_t)nw);
        GK_TRY(gk_mbox_wait(ctx, early_seq, h2.data(), GM_META_WORDS + nw));
        std::copy(h2.begin(), h2.begin() + GM_META_WORDS, h.begin());
        if (nw && gk_sr_collect(ctx, fb, h2.data() + GM_META_WORDS) != GK_OK) return GK_ERR_RETRY;
    } else if (fb && fb->sr_pending > 0) {
        // the relabel in front of this job was only queued (gk_sr_enqueue): its control words ride on the same round trip
        const int nw = fb->sr_pending * SR_CTL;
        std::vector<u32> h2((size_t)GM_META_WORDS + (size_t)nw);
        GK_TRY(gk_readback2(ctx, f->meta, GM_META_WORDS, fb->sr_ctl, nw, h2.data()));
        std::copy(h2.begin(), h2.begin() + GM_META_WORDS, h.begin());
        if (gk_sr_collect(ctx, fb, h2.data() + GM_META_WORDS) != GK_OK) return GK_ERR_RETRY;      // collision / overflow: the caller starts over
    } else
        GK_TRY(gk_readback(ctx, f->meta, h.data(), GM_META_WORDS));     // one host sync: sizes of the operand
    if (h[GM_META_OVF]) return GK_ERR_UNSUPPORTED;                  // a histogram table of gk_features_build_sp overflowed
    if (f->dyn_type) {                                              // the type the device chose (sp_type_kernel)
        const u32 type = h[GM_META_TYPE];                           // 3: int8 operand + float64 side operand (sp_type8_kernel)
        f->dtype = type == 2u ? 1 : 0;
        f->phi_fp4 = type == 0u && !ctx->opt.gram_no_fp4;
        if (type < 2u) f->k_bound = (double)h[GM_META_SELFMAX] + 1.0;
        else h[GM_META_SPLIT] = 0;                                  // (the column scan ignored the split in that case)
    }
    f->n_cols1 = h[GM_META_PRIM], f->n_cols8 = h[GM_META_INT8], f->n_cols = f->n_cols1 + f->n_cols8;
    f->n_cols_wide = h[GM_META_F64], f->n_low_cols = h[GM_META_RARE];
    f->split_parts = 0, f->n_split_labels = 0;
    if (h[GM_META_SPLIT] >= 2 && f->n_cols_wide > 0)         // those labels live in parts^2 int8 columns each, not in a float64 operand
        f->split_parts = (int)h[GM_META_SPLIT], f->n_split_labels = f->n_cols_wide, f->n_cols_wide = 0;
    f->max_count = 0, f->nnz = 0;
    for (int k = 0; k < 64; ++k) {
        f->max_count = std::max<i64>(f->max_count, h[GM_META_MAXC + k]);
        if (GM_META_NNZ + k != GM_META_OVF) f->nnz += h[GM_META_NNZ + k];
    }
    const i64 rare_entries = h[GM_META_RARE_ENTRIES];
    f->rare_entries = rare_entries;
    // ---- operand: [secondary int8 | primary], 128-byte K-steps (features.hip has the rationale)
    const i64 n1p = round_up(f->n_cols1, f->phi_fp4 ? 256 : 128);
    i64 n8p = round_up(f->n_cols8, 128);
    if (n1p + n8p == 0) n8p = 128;
    f->k1_steps = (int)(n1p / (f->phi_fp4 ? 256 : 128)), f->k8_steps = (int)(n8p / 128);
    f->n_cols_pad = (f->phi_fp4 ? n1p / 2 : n1p) + n8p;
    f->n_rows_pad = round_up(N, 256) + 256;
    const i64 row_lds_max = ctx->opt.gm_row_lds_max > 0 ? (i64)ctx->opt.gm_row_lds_max : (i64)GM_ROW_LDS_MAX;     // option: test hook
    if (f->n_cols_pad > row_lds_max) return GK_ERR_UNSUPPORTED;              // caller falls back to features.hip
    GK_TRY(gk_dev_alloc(ctx, &q, (size_t)f->n_rows_pad * f->n_cols_pad));
    f->phi = q;
    if (f->split_parts) {                                      // the right operand (its rows differ in the split columns only)
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)f->n_rows_pad * f->n_cols_pad));
        f->phi_r = q;
    }
    if (f->n_cols_wide > 0) {
        f->n_cols_wide_pad = round_up(f->n_cols_wide, 16);
        const size_t wb = (size_t)f->n_rows_pad * f->n_cols_wide_pad * 8;
        GK_TRY(gk_dev_alloc(ctx, &q, wb));
        f->phi_w = (double*)q;
        GK_TRY(gk_zero_async(ctx, f->phi_w, wb));
    }
    i32* lg = nullptr;
    i32* lc = nullptr;
    i32* ll = nullptr;
    if (rare_entries > 0) {
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)rare_entries * 4));
        lg = (i32*)q, f->arena.push_back(q);
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)rare_entries * 4));
        lc = (i32*)q, f->arena.push_back(q);
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)rare_entries * 4));
        ll = (i32*)q, f->arena.push_back(q);
    }
    GK_TRY(gk_func_lds(ctx, (const void*)gm_rows_kernel, (int)f->n_cols_pad));
    // whose operand rows: all of them, or -- options feat.rows_lo / feat.rows_hi, the multi-GPU operand-row exchange -- one rank's
    i64 own_lo = 0, own_hi = N;
    if (ctx->opt.feat_rows_hi > ctx->opt.feat_rows_lo) own_lo = ctx->opt.feat_rows_lo, own_hi = std::min<i64>(N, ctx->opt.feat_rows_hi);
    f->own_lo = own_lo, f->own_hi = own_hi;
    // one workgroup per graph (64- and 128-thread workgroups measured the same 30 us: the chain of dependent loads
    // slot -> entry -> column id binds, not the number of workgroups in flight)
    // small rows AND few entries per graph: a wave per graph, four graphs per workgroup (config 5, ~100 entries per graph:
    // 112 -> 85 us; ShortestPath histograms with ~10x the entries per graph: 30 us by workgroups, 76 us by waves)
    if (f->n_cols_pad <= GM_ROW_WAVE_MAX && f->nnz <= 192 * N && !ctx->opt.gm_rows_wg && !huge_graphs)      // (a graph of thousands of vertices: a workgroup's walk)
        gm_rows_wave_kernel<<<dim3((unsigned)cdiv(f->n_rows_pad, 4)), 256, (size_t)f->n_cols_pad * 4, ctx->stream>>>(
            P, A, graph_ptr, V, ent, cnt, ent_n, N, (int8_t*)f->phi, f->n_cols_pad, n8p, f->phi_fp4 ? 1 : 0, kind, f->phi_w,
            f->n_cols_wide_pad, lg, lc, ll, f->n_rows_pad, (int8_t*)f->phi_r, f->split_parts, own_lo, own_hi);        // ... and zeroes the padding rows
    else {
        // (the grid covers the padding rows too: workgroups behind the last graph zero one row each -- a launch less)
        // 1 024 threads per graph where the graphs hold thousands of entries each (the D&D-like ShortestPath job: 13 k on average,
        // 112 k in its largest graph -- 440 trips of dependent loads for 256 threads: 558 -> 217 us), 256 otherwise
        const int rows_threads = (f->nnz > 2048 * N && !ctx->opt.gm_rows_256) ? 1024 : 256;
        gm_rows_kernel<<<dim3((unsigned)f->n_rows_pad), rows_threads, (size_t)f->n_cols_pad, ctx->stream>>>(
            P, A, graph_ptr, V, ent, cnt, ent_n, N, (int8_t*)f->phi, f->n_cols_pad, n8p, f->phi_fp4 ? 1 : 0, kind, f->phi_w,
            f->n_cols_wide_pad, lg, lc, ll, (int8_t*)f->phi_r, f->split_parts, own_lo, own_hi);
    }
    GK_HIP_CHECK(hipGetLastError());
    // ---- what gram.hip needs for the rare labels: their list, per label the start / length of its entries
    f->gm = true;
    f->gm_low_q = A.low_q, f->gm_roff = A.roff, f->gm_low_graph = lg, f->gm_low_cnt = lc, f->gm_low_lab = ll;
    f->gm_df = A.df;            // read by the pair-update kernels of gram.hip (owned by the job's arena)
    return GK_OK;
}

// per-label arrays of a graph-major job (qa entries each): df / colid / roff / low_q stay with the job (arena),
// cmax / cursor / side live as long as the builder runs
static int gm_label_arrays(gk_ctx* ctx, gk_feat* f, size_t qa, GmLabelArrays& A, Tmp<u32>& scratch) {
    void* q = nullptr;
    i32** keep[] = {(i32**)&A.df, &A.colid, (i32**)&A.roff, &A.low_q};
    for (i32** a : keep) {
        GK_TRY(gk_dev_alloc(ctx, &q, qa * 4));
        *a = (i32*)q;
        f->arena.push_back(q);
    }
    GK_TRY(scratch.alloc(2 * qa + qa / 4));             // [cmax | cursor | side (bytes)]
    A.cmax = scratch.p, A.cursor = scratch.p + qa;
    A.side = (unsigned char*)(scratch.p + 2 * qa);
    return GK_OK;
}

// ---------------------------------------------------------------------------------------------------
int gk_features_build_gm(gk_ctx* ctx, gk_batch* b, gk_feat* f, int n_levels, int prim_max, int wide_above) {
#ifdef GK_ABLATION
    {
        const char* e = getenv("GK_GM_ABL");
        const int v = e ? atoi(e) : 0;
        GK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_gm_abl), &v, sizeof v));
    }
#endif
    const i64 V = b->n_nodes, N = b->n_graphs;
    const int kind = f->kind;
    const bool stream = b->stream_layout;                     // the label counts of the levels only exist on the device
    // ---- level slots
    GmLevels P = {};                                    // unused slots stay null (deterministic)
    GmTable Tv = {};
    P.L = 0, P.off[0] = 0;
    i64 Qb = 0;                                         // bound of the label space (= the space itself when the host knows it)
    for (int ll = 0; ll < n_levels; ++ll) {
        const int l = f->level0 + ll;                         // the batch's level (a feature job may cover a range of levels)
        if (stream) {
            // every level is a slot (which ones list nothing is not known here); at most half of the nodes sit in
            // shared classes' worth of labels, plus the carried classes
            const int j = P.L++;
            P.lab[j] = b->labels + (size_t)l * V, P.level[j] = ll, P.flag[j] = nullptr, P.id_base[j] = 0;
            Qb += l == 0 ? (i64)b->n_labels0 : std::min<i64>(b->n_iso, (i64)b->n_labels0) + V / 2;
            P.off[j + 1] = Qb;
            continue;
        }
        const i64 nl = (l == 0 && b->level0_hist) ? V : ((size_t)l < b->n_sorted.size() ? b->n_sorted[l] : V);
        if (nl == 0) continue;                                // nothing shared: baseline only
        const int j = P.L++;
        const bool act = (size_t)l < b->active_layout.size() && b->active_layout[l];
        const i64 count = l == 0 ? (i64)b->n_labels0 : b->label_counts[l];     // level 0 keeps the input ids
        P.lab[j] = b->labels + (size_t)l * V, P.level[j] = ll;
        P.id_base[j] = act ? (i32)(V - nl) : 0;
        P.flag[j] = nullptr;
        P.off[j + 1] = P.off[j] + (count - P.id_base[j]);
        // full level with a listed prefix: the relabel left per-node flags (wl.hip: HeadAssignSplit)
        if (!act && nl < V && b->shared_flag) P.flag[j] = b->shared_flag + (size_t)l * V;
        Tv.lo[j] = (u32)P.id_base[j], Tv.S[j] = (u32)(count - P.id_base[j]), Tv.ncc[j] = 0, Tv.off[j] = (u32)P.off[j];
        Qb = P.off[j + 1];
    }
    const i64 Q = Qb;
    GK_ARG(Q < (1ll << 31), "gk_features_build: label space too large");
    Tv.L = (u32)P.L, Tv.Q = (u32)Q, Tv.off[P.L] = (u32)Q;
    // ---- per-label arrays, the graphs' entries
    const size_t qa = (size_t)round_up(Q > 0 ? Q : 1, 64);
    GmLabelArrays A;
    Tmp<u32> scratch(ctx);
    GK_TRY(gm_label_arrays(ctx, f, qa, A, scratch));
    Tmp<u32> cnt(ctx);           // the graphs' entries: slot v of level j = (label index ent.p[..], count cnt.p[..]), count 0 = no entry
    Tmp<i32> ent(ctx);
    GK_TRY(cnt.alloc((size_t)(P.L > 0 ? P.L : 1) * (size_t)(V > 0 ? V : 1)));
    GK_TRY(ent.alloc((size_t)(P.L > 0 ? P.L : 1) * (size_t)(V > 0 ? V : 1)));
    Tmp<u32> ent_n(ctx);         // entries (slots in use) per level and graph
    GK_TRY(ent_n.alloc((size_t)(P.L > 0 ? P.L : 1) * (size_t)(N > 0 ? N : 1)));
    const int rectangular = f->symmetric ? 0 : 1;
    // counting table per wave: only graphs of more than 128 nodes use it (smaller ones sort in registers), so a job without
    // such graphs leaves the LDS to the private histograms (config 3: level 2's 6 799 shared classes then count in LDS too)
    // graphs above wave_max vertices: a workgroup each (gm_pairs_huge_kernel), persistent over the graph list.  A wave counts
    // a graph of up to GM_MAX_NODES vertices, but as ~ 100 dependent LDS trips per level at a thousand vertices (the tail of
    // the D&D-like set: 121 us); a job that has graphs above GM_MAX_NODES anyway hands everything above GM_WG_NODES over
    const int wave_max = (b->max_graph_nodes > GM_MAX_NODES && !ctx->opt.gm_no_huge) ? GM_WG_NODES : GM_MAX_NODES;
    int T = 64;
    if (b->max_graph_nodes > 128)
        while (T < 2 * std::min<int>(b->max_graph_nodes, wave_max)) T <<= 1;
    const bool huge = b->max_graph_nodes > wave_max;
    const i64 grid_h = huge ? std::min<i64>(N, 1024) : 0;
    // ---- which levels count in workgroup-private histograms (small label spaces first come, 32 K bins in all)
    const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    const int waves = GM_WAVES, per_cu = 3;
    i64 grid = cdiv(N, waves);
    if (grid > per_cu * (i64)n_cu) grid = per_cu * (i64)n_cu;
    // per_cu workgroups share a CU's 160 KiB: private histogram + one counting table per wave
    const i64 priv_budget = std::max<i64>(0, (160 * 1024 / per_cu - 1024 - (i64)waves * 2 * T * 4) / 2) & ~1ll;
    const bool priv_ok = kind == GK_FEAT_DOT && !ctx->opt.gm_no_priv && cdiv(N, grid * waves) + 1 < (i64)GM_PRIV_COUNT_MASK;
    i64 bins = 0;
    for (int j = 0; j < FEAT_MAX_LEVELS; ++j) Tv.poff[j] = -1;
    for (int j = 0; j < P.L && !stream; ++j) {
        const i64 ids = P.off[j + 1] - P.off[j];
        if (priv_ok && ids > 0 && bins + ids <= priv_budget) Tv.poff[j] = (i32)bins, bins += ids;
    }
    Tv.bins = (u32)((bins + 1) & ~1ll);
    // the LDS image is laid out for the cap of the private bins when only the device knows the label spaces
    const int priv_cap_words = stream ? (int)(priv_ok ? priv_budget / 2 : 0) : (int)(Tv.bins / 2);
    const size_t pairs_lds = (size_t)priv_cap_words * 4 + (size_t)waves * 2 * T * 4;
    GK_ARG(pairs_lds <= 160 * 1024, "gk_features_build: graph too large for the graph-major builder");
    GK_TRY(gk_func_lds(ctx, (const void*)gm_pairs_kernel, (int)pairs_lds));
    Tmp<u32> part(ctx), wgmeta(ctx);
    Tmp<GmTable> table(ctx);
    GK_TRY(table.alloc(1));
    GK_TRY(wgmeta.alloc((size_t)(grid + grid_h) * 2));
    GK_TRY(part.alloc((size_t)grid * (size_t)(priv_cap_words > 0 ? priv_cap_words : 1)));
    // host-known layout: the table travels as it is (a by-value kernel argument of this size ends up in per-thread scratch)
    if (!stream) GK_HIP_CHECK(hipMemcpyAsync(table.p, &Tv, sizeof(GmTable), hipMemcpyHostToDevice, ctx->stream));
    gm_prep_kernel<<<dim3((unsigned)std::min<i64>(cdiv(Q > 0 ? Q : 1, 1024), stream ? 64 : 1024)), 256, 0, ctx->stream>>>(
        stream ? b->sr_ctl : nullptr, f->level0, P.L, (u32)Q, priv_ok ? 1 : 0, (u32)priv_budget, table.p, A.df, A.cmax, A.cursor, (u32*)A.side);
    gm_pairs_kernel<<<dim3((unsigned)grid), 64 * waves, pairs_lds, ctx->stream>>>(
        P, A, table.p, priv_cap_words, b->graph_ptr, N, V, ent.p, cnt.p, ent_n.p, f->selfk, f->meta, n_levels, kind, f->n_fit, rectangular,
        (u32)(f->low_df > 2 ? f->low_df : 2), T, prim_max, wide_above, part.p, wgmeta.p, wave_max);
    if (priv_cap_words > 0)
        gm_reduce_kernel<<<grid_for(priv_cap_words, 64), 1024, 0, ctx->stream>>>(A, table.p, part.p, (int)grid, prim_max, wide_above, rectangular);
    if (huge) {
        GK_TRY(gk_func_lds(ctx, (const void*)gm_pairs_huge_kernel, GMH_T * 8 + 16));
        gm_pairs_huge_kernel<<<dim3((unsigned)grid_h), GMH_THREADS, GMH_T * 8 + 16, ctx->stream>>>(
            P, A, table.p, b->graph_ptr, N, V, ent.p, cnt.p, ent_n.p, f->selfk, n_levels, kind, f->n_fit, rectangular,
            (u32)(f->low_df > 2 ? f->low_df : 2), wgmeta.p, (int)grid, wave_max);
    }
    return gm_finish(ctx, f, P, A, table.p, Q, b->graph_ptr, N, V, ent.p, cnt.p, ent_n.p, wgmeta.p, (int)(grid + grid_h), prim_max, wide_above, huge);
}

// ---------------------------------------------------------------------------------------------------
// ShortestPath pair batch in histogram form (sp.hip): the features of graph g are the counts of its keys (l_u, l_v, d)
// over the ordered pairs u != v with a finite distance (shortest_path.py:412-499).  No pair items exist: a workgroup
// walks the graph's n x n distance matrix, maps every key to its dense id (sp_idtab) and counts the ids in an LDS table;
// the slots in use are the graph's (label, count) entries -- the same arrays gm_pairs_kernel leaves, so the column
// classes, the operand rows and the rare lists come from the shared second half (gm_finish).  Replaces the pair-item
// arrays (12 bytes x 4.4 M pairs at BASELINE config 4), a three-pass sort of them and the label-major builder's four
// passes over them.
// ---------------------------------------------------------------------------------------------------
#define SPH_THREADS 1024
#define SPH_T 8192                  // table slots (label id + count): 64 KiB
#define SPH_INF 0x3f000000
#define SPH_LAB 2048                // node labels of a graph staged in LDS up to this many vertices

struct SpSource {
    const i32* node_ptr; const i32* node_label; const u64* dist_ptr; const i32* dist; const u32* idtab; const i32* pair_base;
    u64 L, d1; int with_labels;
};

// Round 6: BATCHES of graphs.  One workgroup per CU walks its graphs one after the other, and every graph is four phases
// (clear, walk, void check, compaction) with a barrier and a chain of dependent round trips each -- 16 graphs x ~7 us per
// workgroup at BASELINE config 4, whatever the 900 entries of a graph cost.  The table is 8 192 slots because ONE graph may need
// them; a 30-vertex graph needs 2 048.  So a workgroup now takes as many of its graphs at a time as fit the table (regions of
// T_k slots, T_k = the power of two >= 2 pairs, up to SPH_BATCH graphs, their labels together within SPH_LAB) and runs every
// phase once for all of them: an entry finds its graph by a scan over at most eight prefix values, a 64-slot chunk of the
// table belongs to one graph (regions are multiples of 64), everything else is as before.  bmax = 1 is the round-5 walk.
#ifdef GK_ABLATION
// tools' build only: workgroup 0's cycles per phase of sp_hist_kernel (batch selection, clear + labels, walk, void check,
// compaction, finish) and its batches / graphs: tools/dev/sph_times.py
__device__ unsigned long long g_sph_dbg[8];
#define SPH_DBG_DECL unsigned long long t_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_x = __builtin_readcyclecounter();
#define SPH_DBG(k) { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_y = __builtin_readcyclecounter(); t_ph[k] += t_y - t_x; t_x = t_y; } }
#define SPH_DBG_CNT(k, v) { if (blockIdx.x == 0 && threadIdx.x == 0) t_ph[k] += (v); }
#define SPH_DBG_OUT() { if (blockIdx.x == 0 && threadIdx.x == 0) for (int q_ = 0; q_ < 8; ++q_) g_sph_dbg[q_] = t_ph[q_]; }
extern "C" int gk_debug_sph_times(gk_ctx* ctx, unsigned long long* out8) {
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    GK_HIP_CHECK(hipDeviceSynchronize());
    GK_HIP_CHECK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_sph_dbg), sizeof(unsigned long long) * 8));
    return GK_OK;
}
#else
#define SPH_DBG_DECL
#define SPH_DBG(k)
#define SPH_DBG_CNT(k, v)
#define SPH_DBG_OUT()
#endif
#define SPH_BATCH 8
#define SPH_CAND 8                  // candidate graphs wave 0 fetches per batch
// A barrier that waits for the wave's LDS traffic only.  The threads of this kernel talk through LDS alone -- nothing it writes
// to HBM is read back -- and __syncthreads() waits for every outstanding global access as well: a round of the kernel was
// five phases of "global latency + barrier" (workgroup 0 at config 4, cycles per phase and round: select 4 900, clear + labels
// 3 400, walk 8 600, void check 3 000, compaction 13 700 of which ~4 000 waiting for the entry stores to be acknowledged).
__device__ __forceinline__ void sph_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(SPH_THREADS) void sp_hist_kernel(const SpSource S, const GmLevels P, const GmLabelArrays A, const GmPriv R,
                                                              i64 n_graphs, i32* __restrict__ ent_lab, u32* __restrict__ ent_cnt,
                                                              u32* __restrict__ ent_n, u64* __restrict__ selfk, i64 n_fit, int rectangular,
                                                              u32 df_cap, int prim_max, int wide_above, u32* __restrict__ part,
                                                              u32* __restrict__ wgmeta, u32* __restrict__ overflow, u32 skip_above, int bmax,
                                                              u64* __restrict__ selfk8) {
    extern __shared__ __attribute__((aligned(16))) i32 gm_lds[];      // private df histogram | keys[SPH_T] | counts[SPH_T]
    __shared__ u32 ovf_s, void_s, red_m[SPH_THREADS / 64], red_e[SPH_THREADS / 64];
    __shared__ int b_g[SPH_BATCH], b_n[SPH_BATCH], b_v0[SPH_BATCH], b_base[SPH_BATCH], b_ns[SPH_BATCH];
    __shared__ u32 b_t0[SPH_BATCH + 1], b_e0[SPH_BATCH + 1], b_np[SPH_BATCH], b_nent[SPH_BATCH];
    __shared__ const i32* b_d32[SPH_BATCH];
    __shared__ const unsigned char* b_d8[SPH_BATCH];
    __shared__ unsigned long long b_extra[SPH_BATCH];
    __shared__ u32 b_sq8[SPH_BATCH];                     // sum of c^2 over the entries with c <= 127 (at most 6 144 x 127^2: 32 bits)
    __shared__ int nb_s, done_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    u32* priv = (u32*)gm_lds;
    const int priv_words = R.bins / 2;
    for (int t = tid; t < priv_words; t += SPH_THREADS) priv[t] = 0;
    i32* keys = gm_lds + priv_words;
    u32* co = (u32*)(keys + SPH_T);
    unsigned short* clist = (unsigned short*)(co + SPH_T);           // claimed slots of every region, in claim order
    const i32 poff = R.off[0];
    u32 maxc = 0, entries = 0;
    // wave 0 selects the batches; the candidates of the NEXT selection are fetched right after the current one (their
    // latency passes under the round's work instead of in front of it)
    i64 next = blockIdx.x;                               // (uniform over the lanes of wave 0)
    int cn = 0;
    i32 cv0 = 0, cbase = 0;
    u32 cnp = 0;
    u64 cd = 0;
    auto fetch = [&]() __attribute__((always_inline)) {
        const i64 g = next + (i64)lane * (i64)gridDim.x;
        cn = 0, cv0 = 0, cbase = 0, cnp = 0, cd = 0;
        if (lane < SPH_CAND && g < n_graphs) {
            cv0 = S.node_ptr[g], cn = S.node_ptr[g + 1] - cv0;
            cbase = S.pair_base[g], cnp = (u32)(S.pair_base[g + 1] - cbase);
            cd = S.dist_ptr[g];
        }
    };
    if (w == 0) fetch();
    SPH_DBG_DECL
    for (;;) {
        // ---- the next batch of this workgroup's graphs (g = blockIdx.x + k gridDim.x): the first sixteen lanes of wave 0 hold a
        // candidate each, every lane of the wave runs the same selection over them
        if (w == 0) {
            // every candidate lane sizes its own graph; inclusive prefixes over the candidates (DPP row shifts) say which of them
            // still fit -- the batch ends in front of the first real candidate that does not (one lane walking the candidates
            // one after the other was 3 500 cycles per batch with fifteen waves waiting)
            const i64 g_mine = next + (i64)lane * (i64)gridDim.x;
            const bool valid = lane < SPH_CAND && g_mine < n_graphs;
            const bool real = valid && cnp != 0 && cnp <= skip_above;
            u32 T = 0, E = 0, one = real ? 1u : 0u;
            if (real) {
                // a table region of 21/16 of the pairs (distinct keys <= pairs: never full), a multiple of 64, capped.  Round 6: NOT a
                // power of two (the slot is hash * T >> 32) -- twice the pairs rounded up to one put four 30-vertex graphs into a
                // batch where seven fit
                T = (cnp + (cnp >> 2) + (cnp >> 4) + 64u) & ~63u;
                if (T > (u32)SPH_T) T = (u32)SPH_T;
                E = (u32)cn * (u32)cn;                    // (a histogram graph has at most 6 144 pairs... but any number of vertices)
            }
            const u32 E_sat = (u64)cn * (u64)cn > 0x0fffffffull ? 0x0fffffffu : E;      // saturating: eight of them stay below 2^31
            u32 pT = T, pE = E_sat, pN = one;
#define SPH_SCAN_STEP(CTRL)                                                                 \
            pT += (u32)__builtin_amdgcn_update_dpp(0, (int)pT, CTRL, 0xF, 0xF, false);          \
            pE += (u32)__builtin_amdgcn_update_dpp(0, (int)pE, CTRL, 0xF, 0xF, false);          \
            pN += (u32)__builtin_amdgcn_update_dpp(0, (int)pN, CTRL, 0xF, 0xF, false);
            SPH_SCAN_STEP(0x111) SPH_SCAN_STEP(0x112) SPH_SCAN_STEP(0x114)                      // row_shr:1 / 2 / 4, zeros shifted in
#undef SPH_SCAN_STEP
            static_assert(SPH_CAND == 8, "the prefix above covers eight candidate lanes");
            // the first real candidate always fits (pN == 1); a graph whose n^2 saturates travels alone
            const bool fits = !real || pN == 1u || (pT <= (u32)SPH_T && pN <= (u32)bmax && pE < 0x0fffffffu && E_sat < 0x0fffffffu);
            const u64 stop = __ballot(lane < SPH_CAND && (!valid || !fits)) | (1ull << SPH_CAND);
            const int used = (int)__builtin_ctzll(stop);
            const bool acc = real && lane < used;
            const int nb = (int)__popcll(__ballot(acc));
            if (acc) {
                const int at = (int)pN - 1;
                const i32* d32 = S.dist + (cd & ~SP_BYTE_FLAG);
                b_g[at] = (int)g_mine, b_n[at] = cn, b_v0[at] = cv0, b_base[at] = cbase, b_np[at] = cnp;
                b_d32[at] = d32, b_ns[at] = (cn + 15) & ~15;
                b_d8[at] = (cd & SP_BYTE_FLAG) ? (const unsigned char*)(((uintptr_t)d32 + 15) & ~(uintptr_t)15) : nullptr;
                b_t0[at] = pT - T, b_e0[at] = pE - E_sat, b_nent[at] = 0, b_extra[at] = 0ull, b_sq8[at] = 0u;
                if (at == nb - 1) b_t0[nb] = pT, b_e0[nb] = E_sat >= 0x0fffffffu ? E : pE;
            }
            if (valid && lane < used && cnp == 0) { ent_n[g_mine] = 0, selfk[g_mine] = 0; if (selfk8) selfk8[g_mine] = 0; }
            next += (i64)used * (i64)gridDim.x;
            if (lane > nb && lane <= SPH_BATCH) b_t0[lane] = 0xffffffffu, b_e0[lane] = 0xffffffffu;
            if (lane == 0) {
                if (nb == 0) b_t0[0] = 0, b_e0[0] = 0;
                nb_s = nb, ovf_s = 0, void_s = 0, done_s = next >= n_graphs ? 1 : 0;
            }
            fetch();
        }
        sph_lds_barrier();
        SPH_DBG(0)
        const int nb = nb_s;
        SPH_DBG_CNT(6, 1) SPH_DBG_CNT(7, nb)
        if (nb == 0) {                                    // workgroup-uniform
            if (done_s) break;
            sph_lds_barrier();                            // (wave 0 rewrites the words just read)
            continue;
        }
        u32 e0r[SPH_BATCH], t0r[SPH_BATCH + 1];           // prefix values of the batch: uniform, kept in scalar registers
#pragma unroll
        for (int q = 0; q < SPH_BATCH; ++q) e0r[q] = (u32)__builtin_amdgcn_readfirstlane((int)b_e0[q]);
#pragma unroll
        for (int q = 0; q <= SPH_BATCH; ++q) t0r[q] = (u32)__builtin_amdgcn_readfirstlane((int)b_t0[q]);
        const u32 T_all = (u32)__builtin_amdgcn_readfirstlane((int)b_t0[nb]), E_all = (u32)__builtin_amdgcn_readfirstlane((int)b_e0[nb]);
        for (u32 t = tid; t < T_all; t += SPH_THREADS) keys[t] = -1, co[t] = 0;
        sph_lds_barrier();
        SPH_DBG(1)
        const bool single_full = nb == 1 && T_all == (u32)SPH_T;     // the one case in which a table can fill up
        const u32 t_cap = (u32)SPH_T - ((u32)SPH_T >> 2);           // three quarters full at most
        // four entries per thread and trip: the four distances and the eight labels are in flight together, then the four id
        // look-ups (the kernel is a chain of dependent round trips otherwise)
        for (u32 idx0 = 0; idx0 < E_all; idx0 += 4 * SPH_THREADS) {
            // an overflowing table -- here or in any other workgroup -- voids the whole job (the caller falls back to pair
            // items): stop walking.  Round 5: the 33 M pairs of a 5 748-vertex graph were walked to the end for nothing (122 ms)
            if (single_full &&
                (*(volatile u32*)&ovf_s || (idx0 % (64 * SPH_THREADS) == 0 && __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))))
                break;
            i32 x[4], id[4];
            u32 li[4], lj[4];
            int kk[4];
            bool diag[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32 e = idx0 + (u32)u * SPH_THREADS + (u32)tid;
                x[u] = SPH_INF, kk[u] = 0, li[u] = 0, lj[u] = 0, diag[u] = true;
                if (e < E_all) {
                    int k = 0;
                    u32 eb = 0;
#pragma unroll
                    for (int q = 1; q < SPH_BATCH; ++q) k += e >= e0r[q] ? 1 : 0, eb = e >= e0r[q] ? e0r[q] : eb;
                    const int n = b_n[k];
                    const u32 loc = e - eb;
                    const int i = (int)(loc / (u32)n), j = (int)(loc - (u32)i * (u32)n);
                    kk[u] = k, diag[u] = i == j;
                    if (b_d8[k]) {
                        const unsigned char b8 = b_d8[k][(size_t)i * b_ns[k] + j];
                        x[u] = b8 == 255 ? SPH_INF : (i32)b8;
                    } else x[u] = b_d32[k][loc];
                    if (S.with_labels) li[u] = (u32)S.node_label[b_v0[k] + i], lj[u] = (u32)S.node_label[b_v0[k] + j];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                id[u] = -1;
                if (!diag[u] && x[u] < SPH_INF) id[u] = (i32)S.idtab[(u64)x[u] + S.d1 * ((u64)li[u] * S.L + (u64)lj[u])];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (id[u] < 0) continue;
                u32 t0 = 0, t1 = t0r[1];
#pragma unroll
                for (int q = 1; q < SPH_BATCH; ++q)
                    if (kk[u] >= q) t0 = t0r[q], t1 = t0r[q + 1];
                const u32 T = t1 - t0;
                u32 h = (u32)(((u64)((u32)id[u] * 2654435761u) * (u64)T) >> 32);
                for (;;) {
                    i32 old = keys[t0 + h];
                    if (old == -1) {
                        if (single_full && *(volatile u32*)&ovf_s) break;
                        old = atomicCAS(&keys[t0 + h], -1, id[u]);
                        if (old == -1) {
                            // the claim takes the key's entry slot and lists the table slot: the compaction walks the claims, not
                            // the table (a 30-vertex graph fills a seventh of its 2 048 slots)
                            const u32 e = atomicAdd(&b_nent[kk[u]], 1u);
                            clist[t0 + e] = (unsigned short)h;
                            if (single_full && e + 1u > t_cap) ovf_s = 1u;
                            atomicAdd(&co[t0 + h], 1u);
                            break;
                        }
                    }
                    if (old == id[u]) { atomicAdd(&co[t0 + h], 1u); break; }
                    h = h + 1u == T ? 0u : h + 1u;
                }
            }
        }
        SPH_DBG(2)
        if (single_full) {
            // ONE decision per workgroup: another workgroup may raise `overflow` between two threads' loads, and
            // the two paths below meet different barriers -- what every thread acts on is the workgroup's word
            if (*(volatile u32*)&ovf_s | __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) void_s = 1u;
            sph_lds_barrier();
            if (*(volatile u32*)&void_s) {                // more distinct keys than the table holds: the caller falls back
                if (tid == 0) { atomicOr(overflow, 1u); ent_n[b_g[0]] = 0, selfk[b_g[0]] = 0; if (selfk8) selfk8[b_g[0]] = 0; }
                sph_lds_barrier();                        // the batch words are rewritten by wave 0 next
                continue;
            }
        }
        sph_lds_barrier();
        SPH_DBG(3)
        u32 n0r[SPH_BATCH + 1];                           // prefix of the graphs' entry counts (uniform)
        n0r[0] = 0;
#pragma unroll
        for (int q = 0; q < SPH_BATCH; ++q) n0r[q + 1] = q < nb ? n0r[q] + (u32)__builtin_amdgcn_readfirstlane((int)b_nent[q]) : 0xffffffffu;
        const u32 N_all = n0r[nb];
        for (u32 i0 = 0; i0 < N_all; i0 += SPH_THREADS) {
            const u32 i = i0 + (u32)tid;
            const bool act = i < N_all;
            int k = 0;
            u32 nb0 = 0, t0 = 0;
#pragma unroll
            for (int q = 1; q < SPH_BATCH; ++q)
                if (act && i >= n0r[q]) k = q, nb0 = n0r[q], t0 = t0r[q];
            u64 extra = 0;
            if (act) {
                const u32 e = i - nb0;
                const u32 t = t0 + (u32)clist[t0 + e];
                const i32 x = keys[t];
                const u32 c = co[t];
                const i32 base = b_base[k];
                ent_lab[base + e] = x, ent_cnt[base + e] = c;
                if (poff >= 0) {                              // df / count class in the workgroup's private histogram (gm_pairs_kernel)
                    const u32 bin = (u32)poff + (u32)x;
                    const int sh = 16 * (bin & 1u);
                    u32 add = 0;
                    if (rectangular) add |= (b_g[k] < n_fit ? 1u : 2u) << GM_PRIV_SIDE_SHIFT;
                    if ((int)c > prim_max) add |= GM_PRIV_BIG1;
                    if ((int)c > wide_above) add |= GM_PRIV_BIG2;
                    const u32 cur = (priv[bin >> 1] >> sh) & 0xffffu;
                    const u32 flags = add & ~cur & 0xf000u;
                    if (flags) atomicOr(&priv[bin >> 1], flags << sh);
                    atomicAdd(&priv[bin >> 1], 1u << sh);
                } else {
                    const u32 side_bit = b_g[k] < n_fit ? 1u : 2u;
                    const i64 q = P.off[0] + x;
                    if (__hip_atomic_load(&A.df[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < df_cap) atomicAdd(&A.df[q], 1u);
                    if (c >= 2u && __hip_atomic_load(&A.cmax[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c) atomicMax(&A.cmax[q], c);
                    if (rectangular && !(A.side[q] & side_bit)) atomicOr((u32*)(A.side + (q & ~3ll)), side_bit << (8 * (q & 3)));
                }
                extra = (u64)c * c - c;
                maxc = c > maxc ? c : maxc;
                ++entries;
                if (selfk8 && c <= 127u) atomicAdd(&b_sq8[k], c * c);
            }
            // sum of c^2 - c per graph: one LDS atomic per wave where the wave's entries belong to one graph, else one per lane
            const int k0 = __builtin_amdgcn_readfirstlane(k);
            if (__ballot(act && k != k0) == 0ull) {
                extra = wave_sum_u64(extra);
                if (lane == 0 && extra) atomicAdd(&b_extra[k0], (unsigned long long)extra);
            } else if (extra) atomicAdd(&b_extra[k], (unsigned long long)extra);
        }
        sph_lds_barrier();
        SPH_DBG(4)
        if (tid < nb) {
            ent_n[b_g[tid]] = b_nent[tid];
            selfk[b_g[tid]] = (u64)b_np[tid] + (u64)b_extra[tid];      // sum of c^2 = sum of c + sum of (c^2 - c)
            if (selfk8) selfk8[b_g[tid]] = (u64)b_sq8[tid];
        }
        sph_lds_barrier();                                // the batch words are rewritten by wave 0 next
        SPH_DBG(5)
    }
    SPH_DBG_OUT()
    for (int off = 32; off > 0; off >>= 1) {
        entries += __shfl_down(entries, off, 64);
        const u32 o = __shfl_down(maxc, off, 64);
        maxc = o > maxc ? o : maxc;
    }
    if (lane == 0) red_m[w] = maxc, red_e[w] = entries;
    __syncthreads();
    if (tid == 0) {
        u32 m = 0, e = 0;
        for (int k = 0; k < SPH_THREADS / 64; ++k) m = red_m[k] > m ? red_m[k] : m, e += red_e[k];
        wgmeta[2 * blockIdx.x] = m, wgmeta[2 * blockIdx.x + 1] = e;
    }
    u32* mine = part + (size_t)blockIdx.x * priv_words;
    for (int t = tid; t < priv_words; t += SPH_THREADS) mine[t] = priv[t];
}

// ---------------------------------------------------------------------------------------------------
// Large graphs of the histogram form (round 5).  One workgroup and one 8 192-slot LDS table per graph was the whole
// route: D&D-like graphs hold up to 112 k distinct keys (880 of 1 178 above the table), a 3 782-vertex discussion thread
// is 14 M matrix entries for ONE workgroup -- both sets fell back to 643 M / 172 M explicit pair items, a three-pass sort
// of them and the label-major builder (81 / 49 ms).  Now every graph above SPH_SMALL_PAIRS pairs owns a COUNTER ROW in
// HBM (one u32 per dense key id, Q of them), its matrix is cut into row units of ~256 k entries, and
//   sp_rows_count_kernel    one workgroup per unit, a wave per matrix row: keys are summed up in an LDS table (hot keys --
//                           14 M pairs of a thread are a few hundred distinct keys -- never leave the CU), a key that no
//                           longer fits goes to the graph's counter row with a global atomic, and so does the table when
//                           the unit ends;
//   sp_rows_compact_kernel  one workgroup per graph: the non-zero counters of its row become the graph's (id, count)
//                           entries, with the same per-label statistics sp_hist_kernel leaves.
// Graphs of at most SPH_SMALL_PAIRS pairs stay with sp_hist_kernel, whose table cannot overflow on them.
// ---------------------------------------------------------------------------------------------------
#define SPH_SMALL_PAIRS 6144u       // three quarters of SPH_T: distinct keys <= pairs
#define SPR_THREADS 1024
#define SPR_T 8192                  // LDS table slots (key + count)
#define SPR_COLS 16384              // column terms d1 * label staged in LDS up to this many vertices
#define SPR_CHECK_MAX 64            // rounds (of one matrix row per wave) between two fill checks of the table, at most
#define SPR_UNIT (128 * 1024)       // matrix entries per unit (measured: 64 k .. 1 M, REDDIT- and D&D-like)

struct SpUnit { i32 g, r0, r1, row; };

// one key, c times: LDS table while it has room (half full at most: short probes also for the keys that miss), else the row
__device__ __forceinline__ void spr_add(i32 key, u32 c, i32* keys, u32* co, u32 tmask, u32 t_cap, u32* n_ent, u32* __restrict__ row,
                                        const u32* __restrict__ idtab) {
    u32 h = ((u32)key * 2654435761u) >> 8 & tmask;
    // the probe bound: claims race past the fill check (up to a thread each), so a small table can fill up completely and a
    // key that is not in it would circle forever; the row takes any key at any time
    for (int probe = 0; probe < 32; ++probe) {
        i32 old = keys[h];
        if (old == -1) {
            if (*(volatile u32*)n_ent >= t_cap) break;
            old = atomicCAS(&keys[h], -1, key);
            if (old == -1) {
                atomicAdd(n_ent, 1u);
                atomicAdd(&co[h], c);
                return;
            }
        }
        if (old == key) { atomicAdd(&co[h], c); return; }
        h = (h + 1u) & tmask;
    }
    atomicAdd(&row[idtab[key]], c);
}

// the counting loop of sp_rows_count_kernel over a unit's matrix rows, for 32-bit and for byte matrices (B8; a branch on
// the form inside the unrolled stages kept the compiler from batching the loads: 2.29 instead of 1.50 ms REDDIT-like)
template <bool B8, int RPW>
__device__ __forceinline__ void spr_count_rows(const SpSource& S, const SpUnit& un, const SpMat& M, int n, i32 v0, u32 d1, bool col_in_lds,
                                               const u32* colterm, i32* keys, u32* co, u32 tmask, u32 t_cap, u32* n_ent,
                                               u32* __restrict__ row, bool merge, const i32* __restrict__ order, bool flush) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const i32* dg = M.d32;
    // Rounds of RPW rows per wave, in the LABEL-SORTED order of the graph's rows (sp_row_order_kernel), and the table is
    // emptied into the counter row whenever it is nearly full (round 6).  A key (l_u, l_v, d) belongs to the rows of ONE label:
    // walked label by label, a table epoch holds the keys of a few row labels and every key leaves it once or twice -- a
    // D&D-like unit in matrix order met more distinct keys than the table holds and most of its counts were memory-side
    // atomics on the row (172 M of them, 2.5 ms).  A job whose tables never fill (REDDIT-like) only pays the barrier.
    // The fill checks are barriers, and a barrier per round costs a job whose tables never fill 14 % (REDDIT-like: 1.41 ->
    // 1.61 ms): the next check is scheduled from the growth the table showed between the last two (every wave computes the
    // same round from the same snapshot), at most SPR_CHECK_MAX rounds ahead.
    // RPW (1, 2 or 4 by the unit's row count): a wave takes RPW neighbouring rows of the sorted order TOGETHER -- eight entries
    // per lane and trip are then 8 / RPW columns of RPW rows: one column term per RPW entries instead of one each, and rows of
    // the same label give the same key wherever their distances agree, which the run merging below folds into one count.
    constexpr int NCS = 8 / RPW;                          // columns per lane and trip
    const u32 soft_cap = t_cap - (t_cap >> 2);
    __shared__ u32 fill_snap;
    int round = 0, next_check = 0, last_round = -1;
    u32 last_fill = 0;
    for (int rr = un.r0; rr < un.r1; rr += RPW * (SPR_THREADS / 64), ++round) {
      const int rbase = rr + RPW * w;
      if (rbase < un.r1) {
        int ri[RPW];
        u32 rowterm[RPW];
        const i32* dr[RPW];
        const unsigned char* dr8[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            ri[q] = -1, rowterm[q] = 0, dr[q] = dg, dr8[q] = M.d8;
            if (rbase + q < un.r1) {
                const int i = order ? order[v0 + rbase + q] : rbase + q;
                ri[q] = i;
                rowterm[q] = S.with_labels ? d1 * (u32)S.L * (u32)S.node_label[v0 + i] : 0u;
                dr[q] = dg + (size_t)i * n;
                dr8[q] = M.d8 + (size_t)i * M.ns;         // byte matrix (round 6): same lane <-> column mapping, a byte per lane
            }
        }
        // eight entries per lane and trip, stage by stage: the distance loads together, then the eight table probes together
        // (one LDS latency instead of eight in a row -- the kernel is bound by dependent round trips, not by bandwidth or by
        // atomic conflicts: merging equal keys of a wave first, by run detection or by ballot rounds, made it slower), then
        // the counts as fire-and-forget atomics; a key that is not where its hash points takes the probing path.
        // Round 6: a lane's consecutive entries mostly carry the SAME key on the rows that cost most -- a hub reaches nearly
        // every vertex in one or two steps and the labels are few -- and the count of a key is an LDS atomic that serialises
        // over the lanes holding it.  A lane therefore adds up runs of equal keys first: inside a trip, and across trips through
        // one PENDING (key, count) pair that is flushed when its key changes and at the end of the rows.
        // (Merging ACROSS lanes -- ballot rounds, run detection by shuffle -- cost more than it saved in round 5.)
        i32 pk = -1;
        u32 pc = 0;
        for (int j0 = 0; j0 < n; j0 += NCS * 64) {
            i32 x[8], key[8], old[8];
            u32 h[8], c[8], ct[NCS];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                 // entry u: row u % RPW, column step u / RPW
                const int q = u % RPW, j = j0 + (u / RPW) * 64 + lane;
                const bool have = ri[q] >= 0 && j < n;
                if (B8) {
                    const unsigned char b8 = have ? dr8[q][j] : (unsigned char)255;
                    x[u] = b8 == 255 ? SPH_INF : (i32)b8;
                } else x[u] = have ? dr[q][j] : SPH_INF;
            }
#pragma unroll
            for (int cs = 0; cs < NCS; ++cs) {
                const int j = j0 + cs * 64 + lane;
                ct[cs] = j < n ? (col_in_lds ? colterm[j] : (S.with_labels ? d1 * (u32)S.node_label[v0 + j] : 0u)) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = u % RPW, j = j0 + (u / RPW) * 64 + lane;
                key[u] = -1, c[u] = 1u;
                if (j != ri[q] && x[u] < SPH_INF) key[u] = (i32)(rowterm[q] + ct[u / RPW] + (u32)x[u]);
            }
            if (merge) {
                if (pk >= 0 && key[0] == pk) c[0] += pc, pk = -1;      // the pending pair joins the trip's first entry ...
#pragma unroll
                for (int u = 1; u < 8; ++u)
                    if (key[u] >= 0 && key[u] == key[u - 1]) c[u] += c[u - 1], key[u - 1] = -1;
                const i32 fk = pk;                                     // ... or is flushed with it (slot 7); the trip's last entry waits
                const u32 fc = pc;
                pk = key[7], pc = c[7];
                key[7] = fk, c[7] = fc;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = ((u32)key[u] * 2654435761u) >> 8 & tmask;
#pragma unroll
            for (int u = 0; u < 8; ++u) old[u] = key[u] >= 0 ? keys[h[u]] : -1;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (key[u] < 0) continue;
                if (old[u] == key[u]) atomicAdd(&co[h[u]], c[u]);
                else spr_add(key[u], c[u], keys, co, tmask, t_cap, n_ent, row, S.idtab);
            }
        }
        if (pk >= 0) spr_add(pk, pc, keys, co, tmask, t_cap, n_ent, row, S.idtab);
      }
      if (flush && round == next_check && rr + RPW * (SPR_THREADS / 64) < un.r1) {        // workgroup-uniform
          __syncthreads();
          if (threadIdx.x == 0) fill_snap = *(volatile u32*)n_ent;
          __syncthreads();
          u32 fill = fill_snap;
          const u32 grown = fill > last_fill ? fill - last_fill : 0u;
          const u32 rate = grown / (u32)(round - last_round) + 1u;               // new keys per round, rounded up
          if (fill >= soft_cap) {
              for (u32 t = threadIdx.x; t <= tmask; t += SPR_THREADS) {
                  const i32 k = keys[t];
                  if (k >= 0) atomicAdd(&row[S.idtab[k]], co[t]), keys[t] = -1, co[t] = 0;
              }
              if (threadIdx.x == 0) *n_ent = 0;
              __syncthreads();
              fill = 0;
          }
          u32 ahead = (soft_cap - fill) / rate / 2u;
          ahead = ahead < 1u ? 1u : (ahead > (u32)SPR_CHECK_MAX ? (u32)SPR_CHECK_MAX : ahead);
          last_fill = fill, last_round = round, next_check = round + (int)ahead;
      }
    }
}

// operand type of a histogram job from its largest self similarity (features.h: GM_META_TYPE); one workgroup.
// Type 3, int8 operand + float64 side operand: where the full self similarities say float64 (2), does the INT8 PART of the
// operand stay exact?  The columns whose largest count is above 127 go to the float64 side operand whatever happens; what the
// int32 accumulators of the MFMA product see is the rest: K8_ij <= sqrt(K8_ii K8_jj), and K8_gg <= selfk8[g] = the sum of c^2 over
// the graph's entries with c <= 127 (the histogram kernels leave it next to selfk).  Below 2^31: exact.  The D&D-like set
// multiplied 96 k columns in float64 because ONE graph of 5 748 vertices has a self similarity above 2^31 -- most of those
// columns never hold a count above 127.
__global__ __launch_bounds__(1024) void sp_type_kernel(const u64* __restrict__ selfk, const u64* __restrict__ selfk8, i64 n, u32* __restrict__ meta,
                                                       int fp4_ok) {
    __shared__ u64 red[16], red8[16];
    u64 m = 0, m8 = 0;
    for (i64 g = threadIdx.x; g < n; g += 1024) {
        m = selfk[g] > m ? selfk[g] : m;
        if (selfk8) m8 = selfk8[g] > m8 ? selfk8[g] : m8;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_down(m, off, 64), o8 = __shfl_down(m8, off, 64);
        m = o > m ? o : m, m8 = o8 > m8 ? o8 : m8;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m, red8[threadIdx.x >> 6] = m8;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 16; ++q) m = red[q] > m ? red[q] : m, m8 = red8[q] > m8 ? red8[q] : m8;
        u32 type = (m < (1ull << 24) && fp4_ok) ? 0u : (m < 0x7fffffffull ? 1u : 2u);
        if (type == 2u && selfk8 && m8 < 0x7fffffffull) type = 3u;
        meta[GM_META_TYPE] = type;
        meta[GM_META_SELFMAX] = m < 0xffffffffull ? (u32)m : 0xffffffffu;
    }
}

// rows of a graph grouped by label (any order inside a label): order[v0 + p] = local row at position p.  One workgroup
// per counter-row graph, a counting sort on an LDS histogram of the job's labels (identity beyond SPO_LABELS labels).
#define SPO_LABELS 8192
__global__ __launch_bounds__(256) void sp_row_order_kernel(const i32* __restrict__ row_graph, const i32* __restrict__ node_ptr,
                                                           const i32* __restrict__ node_label, i32* __restrict__ order, int L, int with_labels) {
    __shared__ u32 hist[SPO_LABELS];
    __shared__ u32 wsum[4];
    const int g = row_graph[blockIdx.x], tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const i32 v0 = node_ptr[g];
    const int n = node_ptr[g + 1] - v0;
    if (!with_labels || L > SPO_LABELS) {
        for (int r = tid; r < n; r += 256) order[v0 + r] = r;
        return;
    }
    for (int t = tid; t < L; t += 256) hist[t] = 0;
    __syncthreads();
    for (int r = tid; r < n; r += 256) atomicAdd(&hist[node_label[v0 + r]], 1u);
    __syncthreads();
    const int per = (L + 255) / 256;                         // exclusive prefix: a contiguous run of bins per thread
    u32 mine = 0;
    for (int q = 0; q < per; ++q) mine += tid * per + q < L ? hist[tid * per + q] : 0u;
    u32 inc = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(inc, off, 64);
        if (lane >= off) inc += y;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    u32 run = inc - mine;
    for (int q = 0; q < w; ++q) run += wsum[q];
    for (int q = 0; q < per; ++q)
        if (tid * per + q < L) { const u32 c = hist[tid * per + q]; hist[tid * per + q] = run; run += c; }
    __syncthreads();
    for (int r = tid; r < n; r += 256) order[v0 + atomicAdd(&hist[node_label[v0 + r]], 1u)] = r;
}

__global__ __launch_bounds__(SPR_THREADS) void sp_rows_count_kernel(const SpSource S, const SpUnit* __restrict__ units, u32* __restrict__ rows,
                                                                    i64 Q, int slots, int merge, const i32* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) i32 gm_lds[];      // keys[slots] | counts[slots] | column terms[min(n, SPR_COLS)]
    __shared__ u32 n_ent_s;
    const SpUnit un = units[blockIdx.x];
    const int tid = threadIdx.x;
    i32* keys = gm_lds;
    u32* co = (u32*)(keys + slots);
    u32* colterm = co + slots;
    const i32 v0 = S.node_ptr[un.g];
    const int n = S.node_ptr[un.g + 1] - v0;
    const u32 tmask = (u32)slots - 1u, t_cap = (u32)slots >> 1;
    for (int t = tid; t < slots; t += SPR_THREADS) keys[t] = -1, co[t] = 0;
    const bool col_in_lds = n <= SPR_COLS;
    const u32 d1 = (u32)S.d1;
    if (col_in_lds)
        for (int j = tid; j < n; j += SPR_THREADS) colterm[j] = S.with_labels ? d1 * (u32)S.node_label[v0 + j] : 0u;
    if (tid == 0) n_ent_s = 0;
    __syncthreads();
    const SpMat M = sp_mat(S.dist, S.dist_ptr, un.g, n);
    u32* row = rows + (size_t)un.row * (size_t)Q;
    // rows per wave by the unit's row count (a unit of a 5 748-vertex graph is 22 rows: sixteen waves of four would leave ten idle)
    const int n_unit_rows = un.r1 - un.r0;
    const int rpw = (merge & 4) ? 1 : (n_unit_rows >= 4 * (SPR_THREADS / 64) ? 4 : (n_unit_rows >= 2 * (SPR_THREADS / 64) ? 2 : 1));
#define SPR_GO(B8_, RPW_) spr_count_rows<B8_, RPW_>(S, un, M, n, v0, d1, col_in_lds, colterm, keys, co, tmask, t_cap, &n_ent_s, row, (merge & 1) != 0, order, (merge & 2) == 0)
    if (M.d8) { if (rpw == 4) SPR_GO(true, 4); else if (rpw == 2) SPR_GO(true, 2); else SPR_GO(true, 1); }
    else { if (rpw == 4) SPR_GO(false, 4); else if (rpw == 2) SPR_GO(false, 2); else SPR_GO(false, 1); }
#undef SPR_GO
    __syncthreads();
    for (int t = tid; t < slots; t += SPR_THREADS) {
        const i32 k = keys[t];
        if (k >= 0) atomicAdd(&row[S.idtab[k]], co[t]);
    }
}

#define SPC_TRIP 16
struct SpRowsOut {
    i32* ent_lab; u32* ent_cnt; u32* ent_n; u64* selfk; u32* part; u32* wgmeta; u64* selfk8;
};

__global__ __launch_bounds__(SPR_THREADS) void sp_rows_compact_kernel(const i32* __restrict__ row_graph, int n_rows, const u32* __restrict__ rows,
                                                                      i64 Q, const i32* __restrict__ pair_base, const GmLevels P,
                                                                      const GmLabelArrays A, const GmPriv R, const SpRowsOut O, i64 n_fit,
                                                                      int rectangular, u32 df_cap, int prim_max, int wide_above, int wg0) {
    extern __shared__ __attribute__((aligned(16))) i32 gm_lds[];      // private df histogram
    __shared__ u32 n_ent_s, red_m[SPR_THREADS / 64], red_e[SPR_THREADS / 64];
    __shared__ u64 red_x[SPR_THREADS / 64], red_8[SPR_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    u32* priv = (u32*)gm_lds;
    const int priv_words = R.bins / 2;
    for (int t = tid; t < priv_words; t += SPR_THREADS) priv[t] = 0;
    const i32 poff = R.off[0];
    u32 maxc = 0, entries = 0;
    for (int k = blockIdx.x; k < n_rows; k += gridDim.x) {
        const int g = row_graph[k];
        const u32* row = rows + (size_t)k * (size_t)Q;
        const i32 base = pair_base[g];
        const u32 np = (u32)(pair_base[g + 1] - base);
        const u32 side_bit = g < n_fit ? 1u : 2u;
        __syncthreads();
        if (tid == 0) n_ent_s = 0;
        __syncthreads();
        u64 extra = 0, sq8 = 0;                            // sq8: sum of c^2 over the entries with c <= 127 (sp_type_kernel)
        for (i64 t0 = 0; t0 < Q; t0 += SPC_TRIP * SPR_THREADS) {      // (sixteen counters per thread in flight: the scan is a chain of latencies)
            u32 cc[SPC_TRIP];
#pragma unroll
            for (int u = 0; u < SPC_TRIP; ++u) {
                const i64 t = t0 + u * SPR_THREADS + tid;
                cc[u] = t < Q ? row[t] : 0u;
            }
#pragma unroll
            for (int u = 0; u < SPC_TRIP; ++u) {
                const i64 t = t0 + u * SPR_THREADS + tid;
                const u32 c = cc[u];
                const u64 m = __ballot(c > 0);
                if (!m) continue;                                  // wave-uniform
                u32 wbase = 0;
                if (lane == 0) wbase = atomicAdd(&n_ent_s, (u32)__popcll(m));
                wbase = __shfl(wbase, 0, 64);
                if (c == 0) continue;
                const i32 x = (i32)t;
                const u32 e = wbase + (u32)__popcll(m & ((1ull << lane) - 1ull));
                O.ent_lab[base + e] = x, O.ent_cnt[base + e] = c;
                if (poff >= 0) {                              // df / count class in the workgroup's private histogram (gm_pairs_kernel)
                    const u32 bin = (u32)poff + (u32)x;
                    const int sh = 16 * (bin & 1u);
                    u32 add = 0;
                    if (rectangular) add |= side_bit << GM_PRIV_SIDE_SHIFT;
                    if ((int)c > prim_max) add |= GM_PRIV_BIG1;
                    if ((int)c > wide_above) add |= GM_PRIV_BIG2;
                    const u32 cur = (priv[bin >> 1] >> sh) & 0xffffu;
                    const u32 flags = add & ~cur & 0xf000u;
                    if (flags) atomicOr(&priv[bin >> 1], flags << sh);
                    atomicAdd(&priv[bin >> 1], 1u << sh);
                } else {
                    const i64 q = P.off[0] + x;
                    if (__hip_atomic_load(&A.df[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < df_cap) atomicAdd(&A.df[q], 1u);
                    if (c >= 2u && __hip_atomic_load(&A.cmax[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c) atomicMax(&A.cmax[q], c);
                    if (rectangular && !(A.side[q] & side_bit)) atomicOr((u32*)(A.side + (q & ~3ll)), side_bit << (8 * (q & 3)));
                }
                extra += (u64)c * c - c;
                if (c <= 127u) sq8 += (u64)c * c;
                maxc = c > maxc ? c : maxc;
                ++entries;
            }
        }
        extra = wave_sum_u64(extra), sq8 = wave_sum_u64(sq8);
        if (lane == 0) red_x[w] = extra, red_8[w] = sq8;
        __syncthreads();
        if (tid == 0) {
            u64 x = 0;
            u64 x8 = 0;
            for (int q = 0; q < SPR_THREADS / 64; ++q) x += red_x[q], x8 += red_8[q];
            O.ent_n[g] = n_ent_s;
            O.selfk[g] = (u64)np + x;                     // sum of c^2 = sum of c + sum of (c^2 - c)
            if (O.selfk8) O.selfk8[g] = x8;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        entries += __shfl_down(entries, off, 64);
        const u32 o = __shfl_down(maxc, off, 64);
        maxc = o > maxc ? o : maxc;
    }
    if (lane == 0) red_m[w] = maxc, red_e[w] = entries;
    __syncthreads();
    if (tid == 0) {
        u32 m = 0, e = 0;
        for (int q = 0; q < SPR_THREADS / 64; ++q) m = red_m[q] > m ? red_m[q] : m, e += red_e[q];
        O.wgmeta[2 * (wg0 + blockIdx.x)] = m, O.wgmeta[2 * (wg0 + blockIdx.x) + 1] = e;
    }
    u32* mine = O.part + (size_t)(wg0 + blockIdx.x) * priv_words;
    for (int t = tid; t < priv_words; t += SPR_THREADS) mine[t] = priv[t];
}

int gk_features_build_sp(gk_ctx* ctx, gk_batch* pb, gk_feat* f, int prim_max, int wide_above, bool force_rows) {
    const i64 N = pb->n_graphs, V = pb->n_nodes;              // V = pairs = entry slots
    const i64 Q = pb->label_counts.empty() ? 0 : pb->label_counts[0];
    if (V <= 0 || Q <= 0) return GK_ERR_UNSUPPORTED;

    // ---- which graphs count through a counter row, and the row units of their matrices
    std::vector<SpUnit> units;
    std::vector<i32> row_graph;
    u32 skip_above = 0xffffffffu;
    // A job without a graph above 128 vertices keeps to sp_hist_kernel alone (BASELINE config 4: every graph fits its table,
    // and the row machinery -- two host copies, three launches -- cost 0.16 of 0.72 ms there); should a table overflow after
    // all, the caller comes back with force_rows instead of leaving for the pair items.
    const bool use_rows = !ctx->opt.sp_no_rows && (force_rows || ctx->opt.sp_rows_all || pb->sp_max_nodes > 128);
    if (use_rows && pb->sp_h_node_ptr.size() != (size_t)N + 1) {       // graph sizes and pair ranges for the host, once per pair batch
        pb->sp_h_node_ptr.assign((size_t)N + 1, 0), pb->sp_h_pair_base.assign((size_t)N + 1, 0);
        GK_HIP_CHECK(hipMemcpyAsync(pb->sp_h_node_ptr.data(), pb->sp_node_ptr, (size_t)(N + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipMemcpyAsync(pb->sp_h_pair_base.data(), pb->graph_ptr, (size_t)(N + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    if (use_rows) {
        skip_above = ctx->opt.sp_rows_all ? 0u : SPH_SMALL_PAIRS;
        const i64 unit = ctx->opt.sp_hist_unit > 0 ? (i64)ctx->opt.sp_hist_unit : (i64)SPR_UNIT;
        for (i64 g = 0; g < N; ++g) {
            const u32 np = pb->sp_h_pair_base[g + 1] - pb->sp_h_pair_base[g];
            if (np == 0 || np <= skip_above) continue;
            const i32 n = pb->sp_h_node_ptr[g + 1] - pb->sp_h_node_ptr[g];
            const i32 row = (i32)row_graph.size();
            row_graph.push_back((i32)g);
            const i32 rpu = (i32)std::max<i64>(1, unit / n);
            for (i32 r0 = 0; r0 < n; r0 += rpu) units.push_back(SpUnit{(i32)g, r0, std::min(n, r0 + rpu), row});
        }
        // the rows are N_rows x Q counters: a job whose rows do not fit a modest share of the HBM leaves for the pair items
        if ((double)row_graph.size() * (double)Q * 4.0 > 16.0 * 1024 * 1024 * 1024) return GK_ERR_UNSUPPORTED;
    }
    const i64 n_rows = (i64)row_graph.size();
    const bool any_small = n_rows < N;

    GmLevels P = {};
    P.L = 1, P.off[0] = 0, P.off[1] = Q, P.lab[0] = nullptr, P.flag[0] = nullptr, P.id_base[0] = 0, P.level[0] = 0;
    const size_t qa = (size_t)round_up(Q, 64);
    GmLabelArrays A;
    Tmp<u32> scratch(ctx);
    GK_TRY(gm_label_arrays(ctx, f, qa, A, scratch));
    Tmp<u32> cnt(ctx), ent_n(ctx);
    Tmp<i32> ent(ctx);
    GK_TRY(cnt.alloc((size_t)V)); GK_TRY(ent.alloc((size_t)V)); GK_TRY(ent_n.alloc((size_t)N));
    const int rectangular = f->symmetric ? 0 : 1;
    const int n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    const i64 grid1 = any_small ? (N < n_cu ? N : n_cu) : 0;  // one workgroup per CU: the table and the private histogram fill its LDS
    const i64 grid2 = n_rows > 0 ? std::min<i64>(n_rows, 2 * (i64)n_cu) : 0;
    const i64 grid = grid1 + grid2;
    const i64 priv_budget = (160 * 1024 - 1024 - (i64)SPH_T * 10) / 2;
    GmPriv R;
    R.bins = 0;
    for (int j = 0; j < FEAT_MAX_LEVELS; ++j) R.off[j] = -1;
    const i64 per_wg = std::max<i64>(grid1 > 0 ? cdiv(N, grid1) : 0, grid2 > 0 ? cdiv(n_rows, grid2) : 0);
    if (!ctx->opt.gm_no_priv && Q <= priv_budget && per_wg + 1 < (i64)GM_PRIV_COUNT_MASK) R.off[0] = 0, R.bins = (int)Q;
    R.bins = (R.bins + 1) & ~1;
    const size_t lds = (size_t)R.bins * 2 + (size_t)SPH_T * 10;
    Tmp<u32> part(ctx), wgmeta(ctx);
    GK_TRY(wgmeta.alloc((size_t)grid * 2));
    GK_TRY(part.alloc((size_t)grid * (size_t)(R.bins / 2 > 0 ? R.bins / 2 : 1)));
    GmTable Tv = {};
    Tv.L = 1, Tv.Q = (u32)Q, Tv.lo[0] = 0, Tv.S[0] = (u32)Q, Tv.ncc[0] = 0, Tv.off[0] = 0, Tv.off[1] = (u32)Q, Tv.bins = (u32)R.bins;
    for (int j = 0; j < FEAT_MAX_LEVELS; ++j) Tv.poff[j] = R.off[j];
    Tmp<GmTable> table(ctx);
    GK_TRY(table.alloc(1));
    static_assert(sizeof(GmTable) % 4 == 0 && sizeof(GmTable) <= 3072, "GmTable travels as a kernel argument");
    gm_prep_table_kernel<<<dim3((unsigned)std::min<i64>(cdiv(Q, 1024), 1024)), 256, 0, ctx->stream>>>(Tv, table.p, A.df, A.cmax, A.cursor, (u32*)A.side);
    SpSource S{pb->sp_node_ptr, pb->sp_node_label, pb->sp_dist_ptr, pb->sp_dist, pb->sp_idtab, pb->graph_ptr,
               (u64)pb->sp_L, (u64)pb->sp_dcap, pb->sp_with_labels};
    Tmp<u64> selfk8(ctx);                                 // per graph: sum of c^2 over its entries with c <= 127 (sp_type_kernel: type 3)
    if (f->dyn_type && f->k_bound >= 2147483647.0 && !(ctx->opt.sp_static_type & 2)) GK_TRY(selfk8.alloc((size_t)N));       // (below that bound the type cannot be 2)
    if (grid1 > 0) {
        GK_TRY(gk_func_lds(ctx, (const void*)sp_hist_kernel, (int)lds));
        sp_hist_kernel<<<dim3((unsigned)grid1), SPH_THREADS, lds, ctx->stream>>>(
            S, P, A, R, N, ent.p, cnt.p, ent_n.p, f->selfk, f->n_fit, rectangular, (u32)(f->low_df > 2 ? f->low_df : 2), prim_max,
            wide_above, part.p, wgmeta.p, f->meta + GM_META_OVF, skip_above, ctx->opt.sp_hist_no_batch ? 1 : SPH_BATCH, selfk8.p);
    }
    Tmp<u32> rows(ctx);
    Tmp<SpUnit> units_dev(ctx);
    Tmp<i32> row_graph_dev(ctx);
    if (n_rows > 0) {
        int slots = SPR_T;
        if (ctx->opt.sp_hist_slots >= 16 && ctx->opt.sp_hist_slots <= SPR_T && !(ctx->opt.sp_hist_slots & (ctx->opt.sp_hist_slots - 1)))
            slots = ctx->opt.sp_hist_slots;
        GK_TRY(rows.alloc((size_t)n_rows * (size_t)Q));
        GK_TRY(units_dev.alloc(units.size()));
        GK_TRY(row_graph_dev.alloc((size_t)n_rows));
        GK_TRY(gk_zero_async(ctx, rows.p, (size_t)n_rows * (size_t)Q * 4));
        // pageable sources: the copies are complete when the calls return, the vectors may go
        GK_HIP_CHECK(hipMemcpyAsync(units_dev.p, units.data(), units.size() * sizeof(SpUnit), hipMemcpyHostToDevice, ctx->stream));
        GK_HIP_CHECK(hipMemcpyAsync(row_graph_dev.p, row_graph.data(), (size_t)n_rows * 4, hipMemcpyHostToDevice, ctx->stream));
        const i64 cols = std::min<i64>(pb->sp_max_nodes > 0 ? pb->sp_max_nodes : 1, SPR_COLS);
        const size_t lds1 = (size_t)slots * 8 + (size_t)cols * 4;
        GK_TRY(gk_func_lds(ctx, (const void*)sp_rows_count_kernel, (int)lds1));
        // rows walked label by label, the table emptied when it fills (option sp.rows_no_merge: 1 = no per-lane runs, 2 = matrix
        // order and no emptying -- the round-5 walk --, 4 = one matrix row per wave and round instead of up to four)
        const int walk = ctx->opt.sp_rows_no_merge;
        Tmp<i32> order(ctx);
        if (!(walk & 2)) {
            GK_TRY(order.alloc((size_t)(pb->sp_src_nodes > 0 ? pb->sp_src_nodes : 1)));
            sp_row_order_kernel<<<dim3((unsigned)n_rows), 256, 0, ctx->stream>>>(row_graph_dev.p, pb->sp_node_ptr, pb->sp_node_label, order.p,
                                                                                (int)pb->sp_L, pb->sp_with_labels);
        }
        sp_rows_count_kernel<<<dim3((unsigned)units.size()), SPR_THREADS, lds1, ctx->stream>>>(S, units_dev.p, rows.p, Q, slots,
                                                                                              (walk & 1 ? 0 : 1) | (walk & 6), (walk & 2) ? nullptr : order.p);
        const size_t lds2 = (size_t)R.bins * 2;
        GK_TRY(gk_func_lds(ctx, (const void*)sp_rows_compact_kernel, (int)lds2));
        SpRowsOut O{ent.p, cnt.p, ent_n.p, f->selfk, part.p, wgmeta.p, selfk8.p};
        sp_rows_compact_kernel<<<dim3((unsigned)grid2), SPR_THREADS, lds2, ctx->stream>>>(
            row_graph_dev.p, (int)n_rows, rows.p, Q, pb->graph_ptr, P, A, R, O, f->n_fit, rectangular, (u32)(f->low_df > 2 ? f->low_df : 2),
            prim_max, wide_above, (int)grid1);
    }
    GK_HIP_CHECK(hipGetLastError());
    if (R.bins > 0)
        gm_reduce_kernel<<<grid_for(R.bins / 2, 64), 1024, 0, ctx->stream>>>(A, table.p, part.p, (int)grid, prim_max, wide_above, rectangular);
    if (f->dyn_type) sp_type_kernel<<<1, 1024, 0, ctx->stream>>>(f->selfk, selfk8.p, N, f->meta, ctx->opt.gram_no_fp4 ? 0 : 1);

    // the overflow word travels with the operand sizes: meta[] is read back once, in gm_finish
    return gm_finish(ctx, f, P, A, table.p, Q, pb->graph_ptr, N, V, ent.p, cnt.p, ent_n.p, wgmeta.p, (int)grid, prim_max, wide_above);
}

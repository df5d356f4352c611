// Weisfeiler-Lehman relabelling WITHOUT host round trips (the default route for graph batches; wl.hip keeps the
// host-driven route for everything this one declines, and as the redo path).
//
// Reference: grakel/kernels/weisfeiler_lehman.py:223-258 (fit) -- one level = signature of every node, a dictionary
// that numbers the distinct signatures, the new labels.
//
// wl.hip decides per level on the HOST which route a level takes (full / active set / single workgroup), which costs a
// device -> host -> device round trip per level (mailbox waits: profiles/r03_step_timeline.txt, 160 us of idle device per
// profiled step).  Here every level is the SAME four launches with static grids; everything data dependent is read from
// device memory (gk_batch::sr_ctl, SR_CTL words per level), so a whole job -- relabel, and the feature builder behind it
// (features_gm.hip reads the same words) -- is queued without the host knowing a single count:
//   sr_sig      signature key of every ACTIVE node (class of two or more members at the level before, not isolated);
//               everybody else gets the sentinel key and costs one label read.  A 256-node chunk with many active nodes
//               stages its neighbour labels through LDS (coalesced col_idx stream), a sparse chunk gathers per thread.
//   sr_part     TILE-LOCAL partition of the active keys by their top digit (sentinels dropped): no global histogram, no
//               prefix over the tiles -- a bucket is the union of its 2048-key tiles' runs; also leaves pos_of[item] =
//               position of the item's key, so that the last pass runs in ITEM order
//   sr_dict     one workgroup per top-digit bucket: LDS table of the bucket's DISTINCT keys (as scan_sort.hip's
//               bucket_dict_kernel), slots ranked separately for classes of two or more members and singletons;
//               the claimer of a shared class records itself as its representative
//   sr_finish   node order: new label, exact verification of the full signature against the class representative
//               (counts mismatches = hash collisions), and for a node whose class just became a singleton its FINAL id
//               at every later level.
// Label ids of a level l >= 1 (dense):
//   [carried classes 0 .. n_cc) [frozen nodes .. + F_l) [shared classes .. + S_l) [new singletons .. + T_l)
// carried = isolated vertices grouped by input label (they never split and never merge: gk_batch::car_class), frozen =
// nodes whose class was a singleton at an earlier level (classes only split, so they keep a class of their own for ever:
// their id is final from the level after the one they froze at: id - S), F_{l+1} = F_l + T_l.  A label can be shared iff
// id < n_cc or n_cc + F_l <= id < n_cc + F_l + S_l -- the only labels the feature builder looks at.
// DENSE and LIST levels.  A level whose active nodes are more than a quarter of the batch walks the nodes (item j = node j,
// inactive nodes carry the sentinel key); otherwise it walks the LIST of active nodes the level before left behind: every
// workgroup of sr_finish compacts the members of shared classes among its 1024 items into its own segment (seg[1024 b ..],
// wg_cnt[b]; no atomics), and the next level's sr_sig turns the segments into compact item arrays (key[j], node_of[j],
// j < SR_LISTED of the level before).  Which of the two a level is, is decided ON THE DEVICE by every kernel from the same
// control word; grids are static (sized for the dense case), workgroups beyond the item count leave at once.
// The host reads the control words back ONCE at the end; a hash collision (SR_UNRES) or a table overflow (SR_OVF) sends
// the whole job to wl.hip's route (which re-seeds / sorts as before).
#include "common.h"
#include "features.h"
#include "wl_sig.h"
#include <vector>

#define SR_SENT (~0ull)
#define SR_TILE 2048            // keys per partition tile
#define SR_DENSE_MIN 48         // active nodes of a 256-node chunk from which the chunk stages its neighbour lists through LDS

static inline dim3 grid_for(i64 n, int t) { return dim3((unsigned)(n > 0 ? cdiv(n, t) : 1)); }

template <typename T>
__device__ __forceinline__ T sr_wave_incl_scan(T x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    return x;
}

// items of a level: all nodes (dense) or the previous level's listed nodes; ctl_prev == null: level 1, dense
__device__ __forceinline__ bool sr_dense(const u32* __restrict__ ctl_prev, i64 V, i64& n_items) {
    if (!ctl_prev) { n_items = V; return true; }
    const u32 listed = ctl_prev[SR_LISTED];
    const bool dense = (u64)listed * 4ull > (u64)V;
    n_items = dense ? V : (i64)listed;
    return dense;
}

// ---- prologue: control words, level 0's row, the carried classes' labels at every level ------------------------------------------
__global__ __launch_bounds__(256) void sr_init_kernel(u32* __restrict__ ctl, int n_levels, u32 n_labels0, u32 n_present0,
                                                       const i32* __restrict__ car_nodes, const i32* __restrict__ car_class,
                                                       u32 n_car, i32* __restrict__ labels, i64 V) {
    const u32 ncc = n_car ? (u32)car_class[n_car] : 0u;
    if (blockIdx.x == 0) {
        for (int t = threadIdx.x; t < n_levels * SR_CTL; t += 256) {
            const int l = t / SR_CTL, w = t - l * SR_CTL;
            u32 x = 0;
            if (l == 0) x = w == SR_S ? n_labels0 : (w == SR_COUNT ? n_present0 : 0u);     // level 0: every input label can be shared
            else if (w == SR_NCC) x = ncc;
            ctl[t] = x;
        }
        return;
    }
    const u32 k = (blockIdx.x - 1u) * 256u + threadIdx.x;
    if (k >= n_car) return;
    const i32 v = car_nodes[k], id = car_class[k];
    for (int l = 1; l < n_levels; ++l) labels[(size_t)l * V + v] = id;
}

#ifdef GK_ABLATION
// tools' build only: timing ablations of the stream relabel's kernels (WRONG results by construction, the job falls back to the
// host-driven route or fails its checks).  GK_SR_ABL bit 1: signatures without the neighbour-list write-back, 2: without
// the per-node sort, 4: without the gather (labels = column indices), 8: finish without the verification, 16: finish without the
// frozen nodes' later-level writes
__device__ int g_sr_abl;
#define SR_ABL(bit) (g_sr_abl & (bit))
#else
#define SR_ABL(bit) 0
#endif

// ---- signatures -------------------------------------------------------------------------------------------------------
template <bool LEVEL1>
__global__ __launch_bounds__(SIG_THREADS) void sr_sig_kernel(
    const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx, const i32* __restrict__ lab_prev,
    i32* __restrict__ nbr_sorted, u64* __restrict__ key_out, i64 V, u64 seed, u64 mask, int sig_regs,
    const u32* __restrict__ ctl_prev, const u32* __restrict__ seg, const u32* __restrict__ wg_cnt, u32* __restrict__ node_of) {
    __shared__ i32 buf[SIG_LDS_CAP];
    __shared__ int wcnt[SIG_THREADS / 64];
    const int tid = threadIdx.x;
    if (!LEVEL1) {
        i64 n_items;
        if (!sr_dense(ctl_prev, V, n_items)) {
            // ---- list level: this workgroup takes a quarter of one segment of the previous level's sr_finish
            const u32 b = blockIdx.x >> 2, q0 = (blockIdx.x & 3u) * SIG_THREADS;
            const u32 cnt = wg_cnt[b];
            if (q0 >= cnt) return;
            u32 before = 0;
            for (u32 i = tid; i < b; i += SIG_THREADS) before += wg_cnt[i];
            for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off, 64);
            if ((tid & 63) == 0) wcnt[tid >> 6] = (int)before;
            __syncthreads();
            const u32 base = (u32)(wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
            const u32 k = q0 + (u32)tid;
            if (k >= cnt) return;
            const i32 v = (i32)seg[(size_t)b * 1024u + k];
            const i32 s = row_ptr[v];
            const int d = row_ptr[v + 1] - s;
            const i32 own = lab_prev[v];
            i32* x = nbr_sorted + s;
            u64 acc;
            if (d <= 16) acc = node_key_regs(col_idx, lab_prev, x, s, d, (u32)own, seed);
            else {
                for (int kk = 0; kk < d; ++kk) x[kk] = lab_prev[col_idx[s + kk]];
                insertion_sort(x, d);
                acc = sig_head((u32)own, (u32)d, seed);
                for (int kk = 0; kk < d; ++kk) acc += sig_elem((u32)x[kk], seed);
            }
            key_out[base + k] = mix64(acc) & mask;
            node_of[base + k] = (u32)v;
            return;
        }
    }
    const i64 v0 = (i64)blockIdx.x * SIG_THREADS;
    const i64 v1 = (v0 + SIG_THREADS < V) ? v0 + SIG_THREADS : V;
    const i64 v = v0 + tid;
    bool act = false;
    i32 s = 0, own = 0;
    int d = 0;
    if (v < v1) {
        s = row_ptr[v];
        d = row_ptr[v + 1] - s;
        own = lab_prev[v];
        if (LEVEL1) act = d > 0;
        else {
            const u32 lo = ctl_prev[SR_NCC] + ctl_prev[SR_F];
            act = (u32)own - lo < ctl_prev[SR_S];
        }
    }
    {
        const u64 m = __ballot(act);
        if ((tid & 63) == 0) wcnt[tid >> 6] = __builtin_popcountll(m);
    }
    __syncthreads();
    const int n_act = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];        // block-uniform
    if (n_act == 0) {
        if (v < v1) key_out[v] = SR_SENT;
        return;
    }
    u64 acc = 0;
    if (n_act >= SR_DENSE_MIN) {
        // dense chunk: coalesced stream over the chunk's col_idx, neighbour labels staged in LDS (wl.hip: wl_signature_small_kernel)
        const i32 e0 = row_ptr[v0], e1 = row_ptr[v1];
        const int cnt = e1 - e0;
        const bool use_lds = cnt <= SIG_LDS_CAP;
        // four neighbours per thread and trip: the four column indices first, then the four labels they point at (one
        // neighbour per trip is a chain of two dependent memory latencies per trip)
        for (int i0 = tid; i0 < cnt; i0 += 4 * SIG_THREADS) {
            i32 cc[4], ll[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cc[u] = i0 + u * SIG_THREADS < cnt ? col_idx[e0 + i0 + u * SIG_THREADS] : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) ll[u] = i0 + u * SIG_THREADS < cnt ? (SR_ABL(4) ? cc[u] : lab_prev[cc[u]]) : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * SIG_THREADS;
                if (i < cnt) {
                    if (use_lds) buf[i] = ll[u];
                    else nbr_sorted[e0 + i] = ll[u];
                }
            }
        }
        __syncthreads();
        int dwave = act ? d : 0;
        for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(dwave, off, 64); dwave = o > dwave ? o : dwave; }
        if (act) {
            acc = sig_head((u32)own, (u32)d, seed);
            if (use_lds && dwave <= 16 && sig_regs) {
                i32* x = buf + (s - e0);
                i32 r[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) r[k] = (k < d && (k < 8 || dwave > 8)) ? x[k] : 0x7fffffff;
                if (SR_ABL(2)) {}
                else if (dwave <= 8) sort_regs<8>(r);
                else sort_regs<16>(r);
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < d) { x[k] = r[k]; acc += sig_elem((u32)r[k], seed); }
            } else if (use_lds) {
                i32* x = buf + (s - e0);
                insertion_sort(x, d);
                for (int k = 0; k < d; ++k) acc += sig_elem((u32)x[k], seed);
            } else {
                i32* x = nbr_sorted + s;
                insertion_sort(x, d);
                for (int k = 0; k < d; ++k) acc += sig_elem((u32)x[k], seed);
            }
        }
        __syncthreads();
        if (use_lds && !SR_ABL(1))
            for (int i = tid; i < cnt; i += SIG_THREADS) nbr_sorted[e0 + i] = buf[i];
    } else if (act) {
        // sparse chunk: the few active nodes gather their own lists
        i32* x = nbr_sorted + s;
        if (d <= 16) acc = node_key_regs(col_idx, lab_prev, x, s, d, (u32)own, seed);
        else {
            for (int k = 0; k < d; ++k) x[k] = lab_prev[col_idx[s + k]];
            insertion_sort(x, d);
            acc = sig_head((u32)own, (u32)d, seed);
            for (int k = 0; k < d; ++k) acc += sig_elem((u32)x[k], seed);
        }
    }
    if (v < v1) key_out[v] = act ? (mix64(acc) & mask) : SR_SENT;
}

// level 1 with few input labels and small degrees: exact 32-bit codes (wl.hip: wl_signature_exact_kernel); isolated
// vertices are carried, not coded
__global__ __launch_bounds__(256) void sr_sig_exact_kernel(
    const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx, const i32* __restrict__ lab_prev,
    u64* __restrict__ key_out, i64 n, int L, u64 R, u32* __restrict__ unresolved) {
    __shared__ u64 pw[20];
    if ((int)threadIdx.x <= L) {
        u64 p = 1;
        for (int i = 0; i < (int)threadIdx.x; ++i) p *= R;
        pw[threadIdx.x] = p;
    }
    __syncthreads();
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const i32 s = row_ptr[v], e = row_ptr[v + 1];
    if (e == s) { key_out[v] = SR_SENT; return; }
    const u32 own = (u32)lab_prev[v];
    bool in_range = own < (u32)L;
    u64 key = (u64)own * pw[L];
    for (i32 k0 = s; k0 < e; k0 += 8) {              // eight neighbours per trip: column indices, then labels, then the code
        i32 cc[8];
        u32 ll[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cc[u] = k0 + u < e ? col_idx[k0 + u] : -1;
#pragma unroll
        for (int u = 0; u < 8; ++u) ll[u] = cc[u] >= 0 ? (u32)lab_prev[cc[u]] : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (cc[u] < 0) continue;
            in_range = in_range && ll[u] < (u32)L;
            key += pw[ll[u] < (u32)L ? ll[u] : 0];
        }
    }
    if (!in_range) atomicAdd(unresolved, 1u);
    u32 x = (u32)key;                    // bijective scramble: the partition's top digit stays balanced
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    key_out[v] = (u64)x;
}

// ---- partition by the top digit, TILE-LOCAL: a tile's 2048 keys are grouped by digit inside the tile's own range, and the
// (start, count) of every digit run goes to runs[digit][tile].  Nothing global is needed to place a key -- no histogram pass,
// no prefix pass over the tiles' counts (two launches of ~5 us each per level, whatever the level's size) -- and the owner of
// bucket d (sr_dict) walks its 64-byte runs tile by tile instead of one contiguous range.  Sentinel keys take no part.
// pos_of[item] = position of its key (0xffffffff for a sentinel); values = nodes.
#define SR_MAX_TILES 1088       // tiles a bucket owner indexes in LDS (2.2 M items; gk_bucket_dictionary_fits binds earlier)
__global__ __launch_bounds__(1024) void sr_part_kernel(const u64* __restrict__ kin, u64* __restrict__ kout, u32* __restrict__ vout,
                                                       u32* __restrict__ pos_of, i64 V, int shift, u32* __restrict__ runs, int nblk,
                                                       const u32* __restrict__ ctl_prev, const u32* __restrict__ node_of) {
    constexpr int THREADS = 1024, NWAVE = THREADS / 64, ROUNDS = SR_TILE / THREADS, NQ = ROUNDS * NWAVE;   // NQ == 32
    __shared__ __attribute__((aligned(16))) u32 cnt[NQ * 256];     // 32 KiB
    __shared__ u32 dsum[4];
    __shared__ u32 n_act;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const i64 tile0 = (i64)blockIdx.x * SR_TILE;
    i64 n;
    const bool dense = sr_dense(ctl_prev, V, n);
    if (tile0 >= n) return;                       // the grid covers the dense case
    u64 key[ROUNDS];
    u32 rank[ROUNDS];
    bool act[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const i64 idx = tile0 + r * THREADS + tid;
        key[r] = idx < n ? kin[idx] : SR_SENT;
        act[r] = key[r] != SR_SENT;
    }
#pragma unroll
    for (int q = 0; q < NQ * 256 / THREADS; ++q) cnt[q * THREADS + tid] = 0;
    __syncthreads();
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const u32 d = (u32)(key[r] >> shift) & 255u;
        u64 m = __ballot(act[r]);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const u64 bb = __ballot(act[r] && bit);
            m &= bit ? bb : ~bb;
        }
        rank[r] = (u32)__popcll(m & lt);
        if (act[r] && rank[r] == 0) cnt[(r * NWAVE + w) * 256 + d] = (u32)__popcll(m);
    }
    __syncthreads();
    u32 total = 0;
    if (tid < 256) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) total += cnt[q * 256 + tid];
    }
    const u32 dincl = sr_wave_incl_scan(total);
    if (lane == 63 && w < 4) dsum[w] = dincl;
    __syncthreads();
    if (tid < 256) {
        u32 run = dincl - total;                 // keys of the tile with smaller digits
        for (int q = 0; q < w; ++q) run += dsum[q];
        runs[(i64)tid * nblk + blockIdx.x] = (run << 16) | total;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const u32 c = cnt[q * 256 + tid];
            cnt[q * 256 + tid] = run;
            run += c;
        }
    }
    if (tid == 0) n_act = dsum[0] + dsum[1] + dsum[2] + dsum[3];          // active keys of the tile
    __syncthreads();
    // the grouped tile is assembled in LDS (over the counters, once every thread has its position) and leaves as coalesced
    // rows: a lane's keys go to 256 different runs, i.e. 64 separate 8-byte stores per wave instruction when written directly
    u32 local[ROUNDS], val[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const i64 idx = tile0 + r * THREADS + tid;
        local[r] = 0xffffffffu, val[r] = 0u;
        if (act[r]) {
            const u32 d = (u32)(key[r] >> shift) & 255u;
            local[r] = cnt[(r * NWAVE + w) * 256 + d] + rank[r];
            val[r] = dense ? (u32)idx : node_of[idx];
        }
    }
    __syncthreads();
    u64* stage_k = (u64*)cnt;                      // [SR_TILE] 16 KiB
    u32* stage_v = cnt + 2 * SR_TILE;              // [SR_TILE]  8 KiB
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const i64 idx = tile0 + r * THREADS + tid;
        if (act[r]) stage_k[local[r]] = key[r], stage_v[local[r]] = val[r];
        if (idx < n) pos_of[idx] = act[r] ? (u32)tile0 + local[r] : 0xffffffffu;
    }
    __syncthreads();
    const u32 na = n_act;
    for (u32 i = tid; i < na; i += THREADS) kout[tile0 + i] = stage_k[i], vout[tile0 + i] = stage_v[i];
}

// ---- dictionary of one top-digit bucket (scan_sort.hip: bucket_dict_kernel, ranked for the stream layout) ------------
#define SRD_SLOTS 12288          // 12 B per slot: 144 KiB of LDS
#define SRD_MAX_DISTINCT 9216    // load factor 0.75
#define SRD_MAX_CHUNKS 64        // claim flags of a thread: one bit per 1024-item chunk

// item_out[pos] = rank of the item's class among the bucket's shared / singleton classes (bits 0-13) | bucket << 14 | singleton << 31;
// nd_sh / nd_si / listed [bucket] = shared classes, singleton classes, items of shared classes; rep2[bucket * SRD_SLOTS + rank] =
// representative (the claiming item's node) of each shared class
__global__ __launch_bounds__(1024) void sr_dict_kernel(const u64* __restrict__ kx, const u32* __restrict__ vx, const u32* __restrict__ runs,
                                                       int nblk, i64 V, const u32* __restrict__ ctl_prev, int shift,
                                                       u32* __restrict__ item_out, i32* __restrict__ rep2,
                                                       u32* __restrict__ nd_sh, u32* __restrict__ nd_si, u32* __restrict__ listed,
                                                       u32* __restrict__ overflow, u32 max_distinct, u64* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char srd_lds[];
#define SRD_STAMP(k) do { if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
    SRD_STAMP(0);
    unsigned long long* key_s = (unsigned long long*)srd_lds;                  // [SRD_SLOTS] key + 1, 0 = empty
    u32* word_s = (u32*)(srd_lds + (size_t)SRD_SLOTS * 8);                     // [SRD_SLOTS] 1 / 2 = one / several members, later rank | singleton << 31
    __shared__ u32 tpre[SR_MAX_TILES + 1];      // items of this bucket in the tiles before tile t
    __shared__ u32 toff[SR_MAX_TILES];          // position of the bucket's run in tile t
    __shared__ u32 wsum[16];
    __shared__ u32 n_claimed, ovf;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const u32 bucket = blockIdx.x;
    if (tid == 0) n_claimed = 0, ovf = 0;
    i64 n_items;
    sr_dense(ctl_prev, V, n_items);
    const int nt = (int)((n_items + SR_TILE - 1) / SR_TILE);        // tiles in use
    u32 size = 0;
    {   // the bucket's runs: exclusive prefix of their lengths over the tiles
        u32 carry = 0;
        for (int t0 = 0; t0 < nt; t0 += 1024) {
            const int t = t0 + tid;
            const u32 rw = t < nt ? runs[(i64)bucket * nblk + t] : 0u;
            const u32 c = rw & 0xffffu;
            const u32 inc = sr_wave_incl_scan(c);
            __syncthreads();                     // wsum of the round before is read below
            if (lane == 63) wsum[w] = inc;
            __syncthreads();
            u32 before = 0, all = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q < w) before += wsum[q];
                all += wsum[q];
            }
            if (t < nt) tpre[t] = carry + before + inc - c, toff[t] = (u32)t * SR_TILE + (rw >> 16);
            carry += all;
        }
        size = carry;
        if (tid == 0) tpre[nt] = size;
        __syncthreads();
    }
    if (size == 0) {
        if (tid == 0) nd_sh[bucket] = 0, nd_si[bucket] = 0, listed[bucket] = 0;
        return;
    }
    SRD_STAMP(1);
    // bucket-local item i -> its position: the run of the tile with tpre[t] <= i < tpre[t + 1]
    // (runs are ~size / nt items each: start at the proportional guess and gallop -- a few LDS reads instead of the 11
    // dependent ones of a bisection over all tiles, which cost more than the table look-up they serve)
    auto locate = [&](u32 i) __attribute__((always_inline)) {
        int lo = (int)(((u64)i * (u64)nt) / (u64)size), hi;
        if (tpre[lo] <= i) {
            int step = 1;
            hi = lo + 1;
            while (hi < nt && tpre[hi] <= i) { lo = hi; hi += step; step <<= 1; }
            if (hi > nt) hi = nt;
        } else {
            int step = 1;
            hi = lo;
            lo = hi - 1;
            while (lo > 0 && tpre[lo] > i) { hi = lo; lo -= step; step <<= 1; }
            if (lo < 0) lo = 0;
        }
        while (hi - lo > 1) {                    // tpre[lo] <= i < tpre[hi]
            const int mid = (lo + hi) >> 1;
            if (tpre[mid] <= i) lo = mid; else hi = mid;
        }
        return toff[lo] + (i - tpre[lo]);
    };
    const u64 kmask = (1ull << shift) - 1ull;       // shift <= 48: key + 1 never wraps to 0
    // a table of four slots per item is as good as the full one (distinct keys <= items) and much quicker to clear and to rank
    const u32 nslots = size >= (u32)SRD_SLOTS / 4u ? (u32)SRD_SLOTS : (size < 64u ? 256u : 4u * size);
    for (u32 t = tid; t < nslots; t += 1024) key_s[t] = 0ull, word_s[t] = 0u;
    __syncthreads();
    SRD_STAMP(2);
    const bool too_long = size > (u32)SRD_MAX_CHUNKS * 1024u;
    u64 claimed = 0;
    // four items per thread and trip: their positions first, then the four key loads in flight together, then the table
    // (one load in flight per thread made the pass a chain of memory latencies: 16 waves per CU cannot hide one each)
    u32 pp0[4] = {0u, 0u, 0u, 0u}, hh0[4] = {0u, 0u, 0u, 0u};     // first trip: positions and table slots, kept for the look-up pass
    if (!too_long) {
        for (u32 i0 = tid; i0 < size; i0 += 4096u) {
            u32 pp[4];
            u64 kk[4];
#pragma unroll
            for (u32 u = 0; u < 4u; ++u) pp[u] = i0 + u * 1024u < size ? locate(i0 + u * 1024u) : 0u;
#pragma unroll
            for (u32 u = 0; u < 4u; ++u) kk[u] = i0 + u * 1024u < size ? kx[pp[u]] : 0ull;
#pragma unroll
            for (u32 u = 0; u < 4u; ++u) {
                const bool have = i0 + u * 1024u < size;
                const int c = (int)((i0 + u * 1024u) >> 10);
                bool won = false;
                // the table is declared full at max_distinct claims; checked once per item, not per probe (a dependent LDS read
                // on the path to every claim): every thread claims at most once in between, 9216 + 1024 < 12288 slots
                if (have && !*(volatile u32*)&ovf) {
                    const u64 k1 = (kk[u] & kmask) + 1ull;
                    u32 h = (u32)((((k1 * 0x9E3779B97F4A7C15ull) >> 32) * (u64)nslots) >> 32);
                    for (;;) {
                        unsigned long long v = key_s[h];
                        if (v == 0ull) {
                            v = atomicCAS(&key_s[h], 0ull, (unsigned long long)k1);
                            if (v == 0ull) {                                      // claimed: this item owns the class
                                won = true;
                                atomicMax(&word_s[h], 1u);
                                break;
                            }
                        }
                        if (v == k1) { if (*(volatile u32*)&word_s[h] != 2u) atomicMax(&word_s[h], 2u); break; }
                        h = h + 1u == nslots ? 0u : h + 1u;
                    }
                    if (i0 < 1024u) pp0[u] = pp[u], hh0[u] = h;
                }
                // distinct keys are counted once per wave (a thousand lanes adding to ONE LDS word serialise: half of this
                // pass at a level where every key is new); a wave can overshoot max_distinct by 63 slots, the table has room
                if (won) claimed |= 1ull << c;
                const u64 wm = __ballot(won);
                if (wm && lane == 0 && atomicAdd(&n_claimed, (u32)__popcll(wm)) + (u32)__popcll(wm) > max_distinct) ovf = 1u;
            }
        }
    }
    __syncthreads();
    SRD_STAMP(3);
    if (too_long || ovf) {              // not handled here: every item a "singleton" of rank 0 (memory-safe), the job is redone by wl.hip
        for (u32 i = tid; i < size; i += 1024) item_out[locate(i)] = 0x80000000u | (bucket << 14);
        if (tid == 0) {
            nd_sh[bucket] = 0, nd_si[bucket] = 1, listed[bucket] = 0;
            atomicOr(overflow, 1u);
        }
        return;
    }
    // ---- rank the slots in use, shared classes (low half) and singletons (high half) apart: thread t owns slots [12t, 12t + 12),
    // read and written as three 16-byte words (twelve strided 4-byte LDS accesses per thread and direction were ~4 us per bucket)
    constexpr int PER = SRD_SLOTS / 1024;
    u32 mw[PER];
    const bool any_mine = (u32)(PER * tid) < nslots;
    {
        const uint4* src = (const uint4*)(word_s + PER * tid);
#pragma unroll
        for (int q4 = 0; q4 < PER / 4; ++q4) {
            const uint4 x = any_mine ? src[q4] : make_uint4(0, 0, 0, 0);
            mw[4 * q4] = x.x, mw[4 * q4 + 1] = x.y, mw[4 * q4 + 2] = x.z, mw[4 * q4 + 3] = x.w;
        }
    }
    u32 mine = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if ((u32)(PER * tid + q) >= nslots) mw[q] = 0u;
        mine += mw[q] == 2u ? 1u : (mw[q] == 1u ? 0x10000u : 0u);
    }
    const u32 inc = sr_wave_incl_scan(mine);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    u32 before = inc - mine, all = 0;
    for (int q = 0; q < 16; ++q) {
        if (q < w) before += wsum[q];
        all += wsum[q];
    }
    u32 b_sh = before & 0xffffu, b_si = before >> 16;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (mw[q] == 2u) mw[q] = b_sh++;
        else if (mw[q] == 1u) mw[q] = (b_si++) | 0x80000000u;
    }
    if (any_mine) {
        uint4* dst = (uint4*)(word_s + PER * tid);
#pragma unroll
        for (int q4 = 0; q4 < PER / 4; ++q4) dst[q4] = make_uint4(mw[4 * q4], mw[4 * q4 + 1], mw[4 * q4 + 2], mw[4 * q4 + 3]);
    }
    if (tid == 0) nd_sh[bucket] = all & 0xffffu, nd_si[bucket] = all >> 16;
    __syncthreads();
    SRD_STAMP(4);
    // ---- every item looks its class up again
    u32 my_listed = 0;
    // the first 4096 items (all of them, in a bucket of a balanced partition): the slot is known from the insert pass, no key
    // reload, no probing
#pragma unroll
    for (u32 u = 0; u < 4u; ++u) {
        if ((u32)tid + u * 1024u >= size) continue;
        const u32 v = word_s[hh0[u]];
        item_out[pp0[u]] = (v & 0x80003fffu) | (bucket << 14);
        if (!(v >> 31)) {
            ++my_listed;
            if ((claimed >> u) & 1ull) rep2[(size_t)bucket * SRD_SLOTS + (v & 0x3fffu)] = (i32)vx[pp0[u]];
        }
    }
    for (u32 i0 = tid + 4096u; i0 < size; i0 += 4096u) {
        u32 pp[4];
        u64 kk[4];
#pragma unroll
        for (u32 u = 0; u < 4u; ++u) pp[u] = i0 + u * 1024u < size ? locate(i0 + u * 1024u) : 0u;
#pragma unroll
        for (u32 u = 0; u < 4u; ++u) kk[u] = i0 + u * 1024u < size ? kx[pp[u]] : 0ull;
#pragma unroll
        for (u32 u = 0; u < 4u; ++u) {
            if (i0 + u * 1024u >= size) continue;
            const int c = (int)((i0 + u * 1024u) >> 10);
            const u64 k1 = (kk[u] & kmask) + 1ull;
            u32 h = (u32)((((k1 * 0x9E3779B97F4A7C15ull) >> 32) * (u64)nslots) >> 32);
            while (key_s[h] != k1) h = h + 1u == nslots ? 0u : h + 1u;
            const u32 v = word_s[h];
            item_out[pp[u]] = (v & 0x80003fffu) | (bucket << 14);
            if (!(v >> 31)) {
                ++my_listed;
                if ((claimed >> c) & 1ull) rep2[(size_t)bucket * SRD_SLOTS + (v & 0x3fffu)] = (i32)vx[pp[u]];
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) my_listed += __shfl_down(my_listed, off, 64);
    __syncthreads();                    // wsum is read above by every thread
    if (lane == 0) wsum[w] = my_listed;
    __syncthreads();
    if (tid == 0) {
        u32 t = 0;
        for (int q = 0; q < 16; ++q) t += wsum[q];
        listed[bucket] = t;
    }
    SRD_STAMP(5);
#undef SRD_STAMP
}

// ---- item order: labels, verification, final ids of the nodes that just froze, the next level's active list ----------
__global__ __launch_bounds__(1024) void sr_finish_kernel(const u32* __restrict__ pos_of, const u32* __restrict__ item,
                                                         const i32* __restrict__ rep2,
                                                         const u32* __restrict__ nd_sh, const u32* __restrict__ nd_si,
                                                         const u32* __restrict__ listed, u32* __restrict__ ctl_cur, u32* __restrict__ ctl_next,
                                                         const u32* __restrict__ ctl_prev, i32* __restrict__ labels, i64 V, int level,
                                                         int n_levels, int verify, const i32* __restrict__ row_ptr,
                                                         const i32* __restrict__ nbr_sorted, const u32* __restrict__ node_of,
                                                         u32* __restrict__ seg, u32* __restrict__ wg_cnt) {
    __shared__ u32 bsh[257], bsi[257];
    __shared__ u32 d1[4], d2[4], d3[4];
    __shared__ u32 wsh[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    {   // class ids are dense across the buckets: prefixes of their shared / singleton class counts
        const u32 a = tid < 256 ? nd_sh[tid] : 0u, b = tid < 256 ? nd_si[tid] : 0u;
        const u32 li = tid < 256 ? listed[tid] : 0u;
        const u32 ia = sr_wave_incl_scan(a), ib = sr_wave_incl_scan(b), il = sr_wave_incl_scan(li);
        if (lane == 63 && w < 4) d1[w] = ia, d2[w] = ib, d3[w] = il;
        __syncthreads();
        if (tid < 256) {
            u32 y = ia, z = ib;
            for (int q = 0; q < w; ++q) y += d1[q], z += d2[q];
            bsh[tid + 1] = y, bsi[tid + 1] = z;            // inclusive -> entry tid + 1
        }
        if (tid == 0) bsh[0] = 0, bsi[0] = 0;
        __syncthreads();
    }
    const u32 S = bsh[256], T = bsi[256];
    const u32 listed_all = d3[0] + d3[1] + d3[2] + d3[3];             // items of shared classes = the next level's active nodes
    const u32 F = ctl_cur[SR_F], ncc = ctl_cur[SR_NCC];
    if (blockIdx.x == 0 && tid == 0) {
        ctl_cur[SR_S] = S, ctl_cur[SR_T] = T, ctl_cur[SR_COUNT] = ncc + F + S + T;
        ctl_cur[SR_LISTED] = listed_all;
        if (ctl_next) ctl_next[SR_F] = F + T;
    }
    i64 n_items;
    const bool dense = sr_dense(ctl_prev, V, n_items);
    // the next level walks a list only when these are at most a quarter of the batch (sr_dense): no list otherwise
    const bool want_list = ctl_next != nullptr && (u64)listed_all * 4ull <= (u64)V;
    const i64 j = (i64)blockIdx.x * 1024 + tid;
    u32 pos = 0xffffffffu;
    i32 v = 0;
    if (j < n_items) {
        pos = pos_of[j];
        v = dense ? (i32)j : (i32)node_of[j];
    }
    bool shared = false;
    if (pos != 0xffffffffu) {                    // else: not active -- the label of this level was written when the node froze (or is carried)
        const u32 t = item[pos];
        const int lo = (int)((t >> 14) & 255u);  // the item's bucket
        const u32 rank = t & 0x3fffu;
        i32* lab = labels + (size_t)level * V;
        if (t >> 31) {
            const u32 id = ncc + F + S + bsi[lo] + rank;
            lab[v] = (i32)id;
            const i32 fin = (i32)(id - S);       // = n_cc + F_{l+1} slot: the node's id at every later level
            if (!SR_ABL(16))
                for (int l2 = level + 1; l2 < n_levels; ++l2) labels[(size_t)l2 * V + v] = fin;
        } else {
            shared = true;
            lab[v] = (i32)(ncc + F + bsh[lo] + rank);
            if (verify && !SR_ABL(8)) {
                const i32 r = rep2[(size_t)lo * SRD_SLOTS + rank];
                if (r != v) {
                    const i32* lab_prev = labels + (size_t)(level - 1) * V;
                    bool ok = lab_prev[v] == lab_prev[r];
                    const i32 s = row_ptr[v], sr = row_ptr[r];
                    const int d = row_ptr[v + 1] - s;
                    ok = ok && (d == row_ptr[r + 1] - sr);
                    if (ok)
                        for (int k = 0; k < d; ++k)
                            if (nbr_sorted[s + k] != nbr_sorted[sr + k]) { ok = false; break; }
                    if (!ok) atomicAdd(&ctl_cur[SR_UNRES], 1u);
                }
            }
        }
    }
    if (!want_list) return;                      // block-uniform
    // ---- this workgroup's segment of the next level's active list (ballot compaction: no atomics, ascending items)
    const u64 m = __ballot(shared);
    if (lane == 0) wsh[w] = (u32)__popcll(m);
    __syncthreads();
    u32 before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const u32 c = wsh[q];
        if (q < w) before += c;
        all += c;
    }
    if (shared) seg[(size_t)blockIdx.x * 1024u + before + (u32)__popcll(m & ((1ull << lane) - 1ull))] = (u32)v;
    if (tid == 0) wg_cnt[blockIdx.x] = all;
}

// label-grouped node order of a stream-layout level, on demand (the label-major feature builder, features.hip): key =
// compact shareable label, everything else behind
__global__ void sr_order_keys_kernel(const i32* __restrict__ lab, u64* __restrict__ keys, i64 n, u32 ncc, u32 lo, u32 S) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const u32 x = (u32)lab[v];
    keys[v] = x < ncc ? (u64)x : (x - lo < S ? (u64)(ncc + (x - lo)) : (u64)ncc + (u64)S);
}

int gk_sr_rebuild_order(gk_ctx* ctx, gk_batch* b, int level) {
    const i64 V = b->n_nodes;
    const u32 ncc = b->sr_ncc, F = b->sr_F[level], S = b->sr_S[level];
    Tmp<u64> keys(ctx), ks(ctx);
    GK_TRY(keys.alloc(V)); GK_TRY(ks.alloc(V));
    sr_order_keys_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(b->labels + (size_t)level * V, keys.p, V, ncc, ncc + F, S);
    GK_TRY(gk_radix_sort_pairs(ctx, keys.p, nullptr, ks.p, (u32*)(b->perm + (size_t)level * V), V, bits_for((u64)ncc + (u64)S)));
    GK_HIP_CHECK(hipGetLastError());
    b->perm_valid[level] = 1;
    return GK_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Queues the whole relabel (no host read-back).  GK_OK: queued, b->sr_pending = n_levels until gk_sr_collect has seen the
// control words; GK_ERR_UNSUPPORTED: not a job for this route (nothing was queued).
int gk_sr_enqueue(gk_ctx* ctx, gk_batch* b, int n_levels, int hash_bits, bool default_bits) {
#ifdef GK_ABLATION
    {
        const char* e = getenv("GK_SR_ABL");
        const int v = e ? atoi(e) : 0;
        GK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_sr_abl), &v, sizeof v));
    }
#endif
    const i64 V = b->n_nodes;
    const i64 n_car = b->n_iso;
    // ---- is this the job this route is built for?  (graph batches of small graphs whose features the graph-major builder
    // takes, a handful of input labels; everything else keeps wl.hip's route)
    if (ctx->opt.wl_no_stream || b->is_pair_batch || V <= 0 || n_levels < 2 || b->n_big > 0) return GK_ERR_UNSUPPORTED;
    if (b->max_graph_nodes > (ctx->opt.gm_no_huge ? GM_MAX_NODES : GM_HUGE_MAX_NODES) || ctx->opt.feat_no_gm || ctx->opt.wl_no_bucket_dict || ctx->opt.wl_no_hist0) return GK_ERR_UNSUPPORTED;
    if (!(b->n_labels0 >= 1 && b->n_labels0 <= GK_HIST0_MAX_LABELS)) return GK_ERR_UNSUPPORTED;
    if (hash_bits < 32 || hash_bits > 56 || !gk_bucket_dictionary_fits(ctx, V) || cdiv(V, SR_TILE) > SR_MAX_TILES) return GK_ERR_UNSUPPORTED;
    if (b->n_isolated != n_car || (n_car > 0 && !(b->car_class && b->car_nodes))) return GK_ERR_UNSUPPORTED;    // option wl.no_iso: nothing carried
    if (ctx->opt.wl_no_active_set || ctx->opt.wl_no_split) return GK_ERR_UNSUPPORTED;                 // route options of wl.hip: run that route
    if (b->sr_ctl_levels < n_levels) {
        if (b->sr_ctl) gk_dev_free(ctx, b->sr_ctl);
        b->sr_ctl = nullptr, b->sr_ctl_levels = 0;
        void* q = nullptr;
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)n_levels * SR_CTL * 4));
        b->sr_ctl = (u32*)q, b->sr_ctl_levels = n_levels;
    }
    u32* ctl = b->sr_ctl;
    Tmp<u64> dbg(ctx);                                  // option wl.debug = 2: cycle stamps of sr_dict's phases
    if (ctx->opt.wl_debug == 2) { GK_TRY(dbg.alloc((size_t)n_levels * 2048)); GK_TRY(gk_zero_async(ctx, dbg.p, (size_t)n_levels * 2048 * 8)); }
    const int nblk = (int)cdiv(V, SR_TILE);
    Tmp<u64> key(ctx), kx(ctx);
    Tmp<u32> vx(ctx), pos_of(ctx), item(ctx), hist(ctx), small(ctx), node_of(ctx), seg(ctx), wg_cnt(ctx);
    Tmp<i32> rep2(ctx);
    GK_TRY(key.alloc(V)); GK_TRY(kx.alloc(V)); GK_TRY(vx.alloc(V)); GK_TRY(pos_of.alloc(V)); GK_TRY(item.alloc(V));
    GK_TRY(rep2.alloc((size_t)256 * SRD_SLOTS)); GK_TRY(hist.alloc((size_t)256 * nblk)); GK_TRY(small.alloc(4 * 256));
    const i64 n_fin = cdiv(V, 1024);                      // workgroups of sr_finish = segments of the active list
    GK_TRY(node_of.alloc(V / 4 + 1)); GK_TRY(seg.alloc((size_t)n_fin * 1024)); GK_TRY(wg_cnt.alloc((size_t)n_fin));
    u32* nd_sh = small.p;
    u32* nd_si = small.p + 256;
    u32* listed = small.p + 512;
    sr_init_kernel<<<dim3(1u + (unsigned)cdiv(n_car, 256)), 256, 0, ctx->stream>>>(
        ctl, n_levels, (u32)b->n_labels0, (u32)b->n_labels0_present, b->car_nodes, b->car_class, (u32)n_car, b->labels, V);
    const u64 full_mask = (1ull << hash_bits) - 1ull;
    const int sig_regs = ctx->opt.wl_sig_no_regs ? 0 : 1;
    const int lds = SRD_SLOTS * 12;
    GK_TRY(gk_func_lds(ctx, (const void*)sr_dict_kernel, lds));
    u32 max_distinct = SRD_MAX_DISTINCT;
    if (ctx->opt.bd_slots > 0 && (u32)ctx->opt.bd_slots < max_distinct) max_distinct = (u32)ctx->opt.bd_slots;     // test hook
    for (int lvl = 1; lvl < n_levels; ++lvl) {
        const i32* prev = b->labels + (size_t)(lvl - 1) * V;
        u32* cc = ctl + (size_t)lvl * SR_CTL;
        u32* cn = lvl + 1 < n_levels ? cc + SR_CTL : nullptr;
        int bits = hash_bits, verify = 1;
        // level 1 of a job with few input labels: exact 32-bit signature codes (wl.hip: relabel_level)
        bool exact_code = false;
        const u64 code_R = (u64)b->max_degree + 1;
        if (lvl == 1 && default_bits && b->n_labels0 <= 16 && !ctx->opt.wl_no_exact1) {
            double span = (double)b->n_labels0;
            for (int i = 0; i < b->n_labels0; ++i) span *= (double)code_R;
            exact_code = span < 4294967296.0;
        }
        if (exact_code) {
            bits = 32, verify = 0;
            sr_sig_exact_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(b->row_ptr, b->col_idx, prev, key.p, V, (int)b->n_labels0,
                                                                           code_R, cc + SR_UNRES);
        } else if (lvl == 1) {
            sr_sig_kernel<true><<<grid_for(V, SIG_THREADS), SIG_THREADS, 0, ctx->stream>>>(
                b->row_ptr, b->col_idx, prev, b->nbr_sorted, key.p, V, level_seed(lvl, 0), full_mask, sig_regs, nullptr, nullptr, nullptr,
                nullptr);
        } else {
            sr_sig_kernel<false><<<grid_for(V, SIG_THREADS), SIG_THREADS, 0, ctx->stream>>>(
                b->row_ptr, b->col_idx, prev, b->nbr_sorted, key.p, V, level_seed(lvl, 0), full_mask, sig_regs, cc - SR_CTL, seg.p,
                wg_cnt.p, node_of.p);
        }
        const u32* cp = lvl >= 2 ? cc - SR_CTL : nullptr;       // the level before: how many items this level has (null: level 1, all nodes)
        const int shift = 8 * ((bits + 7) / 8 - 1);
        sr_part_kernel<<<dim3(nblk), 1024, 0, ctx->stream>>>(key.p, kx.p, vx.p, pos_of.p, V, shift, hist.p, nblk, cp, node_of.p);
        sr_dict_kernel<<<dim3(256), 1024, lds, ctx->stream>>>(kx.p, vx.p, hist.p, nblk, V, cp, shift, item.p, rep2.p, nd_sh, nd_si, listed,
                                                              cc + SR_OVF, max_distinct, dbg.p ? dbg.p + (size_t)lvl * 2048 : nullptr);
        sr_finish_kernel<<<dim3((unsigned)n_fin), 1024, 0, ctx->stream>>>(pos_of.p, item.p, rep2.p, nd_sh, nd_si, listed, cc, cn, cp,
                                                                          b->labels, V, lvl, n_levels, verify, b->row_ptr, b->nbr_sorted,
                                                                          node_of.p, seg.p, wg_cnt.p);
    }
    GK_HIP_CHECK(hipGetLastError());
    // provisionally a stream-layout batch: the feature builder can be queued behind this without the host knowing a count
    b->sr_pending = n_levels;
    b->stream_layout = true;
    b->level0_hist = true;
    b->n_levels = n_levels;
    if (dbg.p) {
        std::vector<u64> hd((size_t)n_levels * 2048);
        GK_HIP_CHECK(hipMemcpyAsync(hd.data(), dbg.p, hd.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int lvl = 1; lvl < n_levels; ++lvl) {
            double ph[5] = {0, 0, 0, 0, 0}, mx = 0;
            int nb = 0;
            for (int wg = 0; wg < 256; ++wg) {
                const u64* d = hd.data() + (size_t)lvl * 2048 + wg * 8;
                if (!d[5]) continue;
                ++nb;
                for (int k = 0; k < 5; ++k) ph[k] += (double)(d[k + 1] - d[k]);
                mx = std::max(mx, (double)(d[5] - d[0]));
            }
            if (nb) fprintf(stderr, "[gk] sr_dict level %d cycles (avg of %d buckets): runs %.0f clear %.0f insert %.0f rank %.0f lookup %.0f | slowest bucket %.0f\n",
                            lvl, nb, ph[0] / nb, ph[1] / nb, ph[2] / nb, ph[3] / nb, ph[4] / nb, mx);
        }
    }
    return GK_OK;
}

// The control words have reached the host (h = n_levels * SR_CTL words): a hash collision or a table overflow hands the job to
// wl.hip's route (GK_ERR_UNSUPPORTED, the batch is no stream-layout batch any more); otherwise the host-side description of
// the levels is filled in.
int gk_sr_collect(gk_ctx* ctx, gk_batch* b, const u32* h) {
    const int n_levels = b->sr_pending;
    const i64 V = b->n_nodes, n_car = b->n_iso;
    b->sr_pending = 0;
    for (int lvl = 1; lvl < n_levels; ++lvl)
        if (h[(size_t)lvl * SR_CTL + SR_UNRES] || h[(size_t)lvl * SR_CTL + SR_OVF]) {
            if (ctx->opt.wl_debug)
                fprintf(stderr, "[gk] stream relabel: level %d unresolved %u overflow %u -> host-driven route\n", lvl,
                        h[(size_t)lvl * SR_CTL + SR_UNRES], h[(size_t)lvl * SR_CTL + SR_OVF]);
            b->stream_layout = false;
            b->n_levels = 0;
            return GK_ERR_UNSUPPORTED;
        }
    // ---- what the consumers on the host need
    b->stream_layout = true;
    b->level0_hist = true;
    b->sr_ncc = n_levels > 1 ? h[SR_CTL + SR_NCC] : 0u;
    b->sr_F.assign((size_t)n_levels, 0), b->sr_S.assign((size_t)n_levels, 0);
    b->n_sorted.assign((size_t)n_levels, V);
    b->active_layout.assign((size_t)n_levels, 0);
    b->perm_valid.assign((size_t)n_levels, 0);
    b->n_levels = n_levels;
    b->label_counts.assign((size_t)n_levels, 0);
    for (int lvl = 0; lvl < n_levels; ++lvl) {
        const u32* c = h + (size_t)lvl * SR_CTL;
        b->label_counts[lvl] = c[SR_COUNT];
        b->sr_F[lvl] = c[SR_F], b->sr_S[lvl] = c[SR_S];
        if (lvl > 0) b->n_sorted[lvl] = (i64)c[SR_LISTED] + n_car;       // nodes that can share their label
        if (ctx->opt.wl_debug)
            fprintf(stderr, "[gk] stream level %d: frozen %u shared classes %u new singletons %u labels %u listed %u\n", lvl, c[SR_F],
                    c[SR_S], c[SR_T], c[SR_COUNT], c[SR_LISTED]);
    }
    return GK_OK;
}

int gk_wl_relabel_stream(gk_ctx* ctx, gk_batch* b, int n_levels, int hash_bits, bool default_bits, std::vector<u32>& counts) {
    GK_TRY(gk_sr_enqueue(ctx, b, n_levels, hash_bits, default_bits));
    std::vector<u32> h((size_t)n_levels * SR_CTL);
    GK_TRY(gk_readback(ctx, b->sr_ctl, h.data(), n_levels * SR_CTL));
    GK_TRY(gk_sr_collect(ctx, b, h.data()));
    counts.assign((size_t)n_levels, 0);
    for (int lvl = 0; lvl < n_levels; ++lvl) counts[lvl] = (u32)b->label_counts[lvl];
    return GK_OK;
}


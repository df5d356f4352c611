// Internal declarations shared by the translation units of libgk_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/gk_hip.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t i64;
typedef int32_t i32;

void gk_set_error(const char* fmt, ...);

#define GK_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            gk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                     \
            return GK_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define GK_TRY(expr)                 \
    do {                             \
        int _r = (expr);             \
        if (_r != GK_OK) return _r;  \
    } while (0)

#define GK_ARG(cond, msg)            \
    do {                             \
        if (!(cond)) {               \
            gk_set_error("%s", msg); \
            return GK_ERR_ARG;       \
        }                            \
    } while (0)

static inline i64 cdiv(i64 a, i64 b) { return (a + b - 1) / b; }
static inline i64 round_up(i64 a, i64 b) { return cdiv(a, b) * b; }

struct ProfSlot {
    double ms = 0;
    i64 launches = 0;
};

// Per-context caching allocator: hipMalloc'ed blocks are kept in a size-ordered free list and
// handed out again.  All work of a context is on ONE stream, so reusing a block as soon as
// it is released is ordered correctly by the stream itself (same contract as a stream-ordered
// pool, but without hipMallocAsync -- see DESIGN.md "allocator").
struct BlockCache {
    std::multimap<size_t, void*> free_blocks;   // capacity -> block
    std::map<void*, size_t> live;               // block -> capacity
    size_t bytes_total = 0;
    // option debug.guard: blocks with a red zone on either side (GK_GUARD_BYTES before the block, and from the end of the
    // REQUESTED size to GK_GUARD_BYTES behind the capacity), filled with a pattern when the block is handed out and checked
    // when it comes back and at gk_synchronize
    std::map<void*, size_t> guarded;            // block (user pointer) -> requested bytes; in `live` or `free_blocks` as well
    u32* guard_faults = nullptr;                // device: [0] overwritten words, [1] requested size (KiB) of the first block hit
};
#define GK_GUARD_BYTES 4096
#define GK_GUARD_TAIL_MAX (64u << 10)           // bytes of the slack behind the requested size that are checked

// Route / tuning options of a context (gk_set_option, include/gk_hip.h).  Every option leaves the results
// unchanged: each one removes or forces one of several equivalent routes (the parity tests run the job through
// every one of them) or resizes a capacity so that a fallback is taken.  All zero = production behaviour.  The
// library reads NO environment variables.
struct gk_opts {
    // relabel routes (wl.hip)
    int wl_no_tiny = 0, wl_no_listscan = 0, wl_no_iso = 0, wl_no_split = 0, wl_no_exact1 = 0, wl_no_active_set = 0;
    int wl_no_bucket_dict = 0, wl_no_hist0 = 0, wl_frozen_words = 0, wl_flag_bytes = 0, wl_sig_no_regs = 0, wl_debug = 0;
    int sort_buckets = 0;        // 0 decide per level, 1 never, 2 always the bucket finish of the sort
    int bd_slots = 0;            // > 0: capacity (distinct keys per bucket) of the sort-free dictionary, to force its overflow
    // features
    int gm_rows_wg = 0;          // operand rows always by the workgroup-per-graph kernel (small rows: a wave per graph otherwise)
    int feat_no_gm = 0, gm_no_priv = 0, low_df = 0 /* 0 = 24 */, gm_row_lds_max = 0 /* 0 = GM_ROW_LDS_MAX */;
    // Gram
    int gram_dd = 0;             // Gram kernel form with two accumulator sets and direct stores (no parked tile, five-stage ring):
                                 // 0 chosen per job (small fp4 jobs), 1 always, 2 never
    int gram_no_fp4 = 0, gram_no_ws = 0, gram_no_sym = 0, gram_no_patch = 0, gram_xcc = 0;
    int sp_no_prep = 0;          // 1: the set-up of a ShortestPath job (clears, squares, prefix, size classes) as the seven launches of rounds 3-5 instead of sp_prep_small_kernel
    int feat_rows_lo = 0, feat_rows_hi = 0;   // hi > lo: the graph-major builder assembles the operand rows of the graphs [lo, hi) only (multi-GPU operand-row exchange; the others are expected through gk_features_operand)
    int sp_bfs_one_stream = 0;   // 1: the size classes of the bit-parallel breadth-first search one after the other on the context's stream (round 5) instead of alternating over two streams
    int gm_no_early_post = 0;    // 1: the graph-major builder's operand sizes are read back after the column scan (rounds 2-5) instead of being posted from its tile sums
    int gm_rows_256 = 0;         // 1: the workgroup-per-graph operand-row kernel always with 256 threads (1 024 where the graphs hold thousands of entries each, round 6)
    int gm_no_huge = 0;          // 1: graphs above 1 024 vertices send the job to the label-major feature builder (and off the relabel route without host round trips), as in rounds 1-5
    int gram_strip = 0;          // tile order of the tile kernels: 0 = the rule (gram.hip: launch_tiles), 1 = 8 x 8 patches (rounds 1-5), 2 / 4 / 8 / 16 / 32 = strip walk with strips of that many tile columns
    int gram_pair_cap = 0;       // test hook: capacity of the per-tile pair buckets (0: four times the mean load + 128)
    int gram_fold = 0;           // rare labels' pair updates INSIDE the tile kernel (which then normalises in its epilogue as well):
                                 // 0 = when it pays (normalised jobs whose separate normalisation pass costs more than the binning), 1 = whenever legal, 2 = never
    int gram_no_split8 = 0;      // counts above 127: 1 = float64 side operand (gram_f64_kernel) instead of split int8 columns
    int gram_no_split64 = 0;     // float64 side product: one workgroup per tile for the whole K loop (no split, no atomics)
    int gram_no_compact = 0;     // host copies of integer-valued matrices travel as uint16 / int32 and are widened by host threads: 1 = plain float64 copy
    int gram_copy_threads = 0;   // host threads of that widening (0: min(hardware threads, 16))
    int wl_no_frozen_skip = 0;   // 1: the wave / workgroup signature kernels of a full level gather and sort the neighbours of nodes that are alone in their class already
    int wl_no_converge = 0;      // 1: the host-driven relabel computes every level even when two consecutive levels have the same number of labels (a converged partition; round 6 copies the rest)
    int wl_no_wave_sig = 0;      // 1: nodes of degree 33..1024 keep the workgroup signature kernel and the one-thread verifier (rounds 1-4) instead of the wave-per-node kernels
    int scan_direct_max = 0;     // test hook: tiles up to which the fused scans (scan_fn.h) sum their predecessors per block (0: 4096); 1 forces the prefixed form
    int tt_no_fused = 0;         // look-up transform: the target classes matched level by level (two launches per level) even for a handful of targets
    int gram_no_avx2 = 0;        // 1: the widening threads keep to SSE2 (what a CPU without AVX2 runs)
    int gram_no_tri = 0;         // 1: a full symmetric matrix does NOT take the triangle form of the compact copy (blocks on / above the diagonal over PCIe, mirrored -- and, for normalised jobs, scaled -- by the host threads)
    // ShortestPath
    int sp_no_hist = 0;          // pair features through explicit pair items + the sorting dictionary instead of per-graph histograms
    int sp_no_bfs = 0;           // large unit-weight graphs by the row relaxation kernel instead of the bit-parallel breadth-first search
    int sp_bfs_no_lds_cols = 0;  // test hook: the breadth-first search reads the adjacency entries from HBM / L2, not from LDS
    int sp_bfs_no_bytes = 0;     // 1: the breadth-first search stores its distance matrices as 32-bit entries (round 5) instead of BYTES (round 6)
    int sp_no_rows = 0;          // histogram form: no per-graph counter rows (a graph whose LDS table overflows sends the job to the pair items)
    int sp_rows_all = 0;         // test hook: every graph with a pair counts through counter rows (default: graphs above 6 144 pairs)
    int sp_no_fused_mark = 0;    // 1: the key marks of a small-graph job by sp_mark_kernel over the stored matrices (rounds 4-5) instead of inside the packed all-pairs kernels
    int sp_hist_no_batch = 0;    // 1: the one-workgroup-per-CU histogram kernel takes its graphs one at a time (rounds 4-5) instead of as many as fit its table
    int sp_static_type = 0;      // 1: the operand type of a ShortestPath histogram job from the a-priori bound pairs^2 (rounds 3-5) instead of the job's largest self similarity;
                                 // 2: the device decision without the mixed type (int8 operand + float64 side operand when only the int8 columns' part stays below 2^31)
    int sp_rows_no_merge = 0;    // bit 0: the counting workgroups add every matrix entry to the LDS table on its own (round 5) instead of per-lane runs of equal keys;
                                 // bit 1: they walk a graph's rows in matrix order and never empty the table (round 5) instead of label by label, emptying it when it fills;
                                 // bit 2: a wave takes one matrix row at a time instead of up to four neighbouring rows of the sorted order together
    int sp_hist_unit = 0;        // test hook: distance-matrix entries per counting workgroup (0: 131 072)
    int sp_hist_slots = 0;       // test hook: slots of the counting workgroups' LDS table (0: 8 192; a power of two)
    int sp_no_pk = 0;            // never the 16-bit packed register kernel (32-bit registers up to 64 vertices, LDS beyond)
    int sp_no_reg = 0;           // all-pairs distances of small graphs by the LDS workgroup kernel instead of wave-per-graph registers
    // plumbing
    int wl_no_stream = 0;        // never the relabel route without host round trips (wl_stream.hip)
    int no_mailbox = 0;
    int guard = 0;               // debug: red zones around every block of the allocator, checked at release and at gk_synchronize (GK_ERR_STATE)
    int poison = 0;              // debug: fill every block handed out by the allocator with this byte pattern (| 0x100)
};

struct GkHostPool;
void gk_host_pool_destroy(GkHostPool* p);
struct gk_ctx {
    int device = 0;
    gk_opts opt;
    BlockCache cache;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    // second stream for kernels that are independent of what the main stream runs next and too small to fill the chip alone
    // (sp.hip: the two register Floyd-Warshall kernels side by side); fork / join by events, created on first use
    hipStream_t side_stream = nullptr;
    hipEvent_t side_fork = nullptr, side_join = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // user timer
    hipEvent_t pv0 = nullptr, pv1 = nullptr;   // profile timer
    bool profile = false;
    std::map<std::string, ProfSlot> prof;
    // small device -> host read-backs (see gk_readback): mapped pinned host memory the device
    // writes into, then a sequence word the host spins on -- no staging copy, no stream drain
    u32* mbox_host = nullptr;
    u32* mbox_dev = nullptr;
    // pinned staging ring + events of the compact device -> host copy of a Gram matrix (gram.hip: gram_copy_out)
    void* stage_host = nullptr;
    hipEvent_t stage_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    void* xfer_host = nullptr;                 // pinned block for small results that come back in one copy (wl_transform.hip)
    double copy_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the last host copy of a Gram matrix (gk_host_copy_stats)
    struct GkHostPool* host_pool = nullptr;    // host threads of that copy's widening (created on first use, joined by gk_destroy)
    u32 mbox_seq = 0;
    int n_cu = 0;                              // compute units of the device (persistent-kernel grids)
    std::map<const void*, int> func_lds;       // kernel -> dynamic LDS limit already set for THIS context's device (gk_func_lds)
};
#define GK_XFER_BYTES ((size_t)8 << 20)
#define GK_MAX_RANKS 64          // shards of gk_batch_from_shards / ranks of a gk_comm
#define GK_MBOX_WORDS 512
#define GK_HIST0_MAX_LABELS 256

// Read n_words (<= GK_MBOX_WORDS - 1) u32 values at device address src back to dst_host, ordered after
// everything queued on the context's stream so far.  Returns when the values have arrived.
int gk_readback(gk_ctx* ctx, const u32* src_dev, u32* dst_host, int n_words);
// fork: the side stream waits for everything queued on the main stream so far; join: the main stream waits for the side stream
int gk_side_fork(gk_ctx* ctx, hipStream_t* side);
int gk_side_join(gk_ctx* ctx);
u32 gk_mbox_begin(gk_ctx* ctx);                                     // 0: no mailbox, use gk_readback
int gk_mbox_wait(gk_ctx* ctx, u32 seq, u32* dst_host, int n_words);

// Raise a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) to at least `bytes`; remembered per
// context, i.e. per device (contexts of several devices may live in one process)
int gk_func_lds(gk_ctx* ctx, const void* func, int bytes);

// Device allocation through the context's block cache (stream-ordered reuse on ctx->stream).
int gk_dev_alloc(gk_ctx* ctx, void** p, size_t bytes);
void gk_dev_free(gk_ctx* ctx, void* p);

// Zero-fill on the context's stream with our own kernel (not hipMemsetAsync: see DESIGN.md).
int gk_zero_async(gk_ctx* ctx, void* p, size_t bytes);

// RAII temp buffer (freed stream-ordered at scope exit).
template <typename T>
struct Tmp {
    gk_ctx* ctx;
    T* p = nullptr;
    explicit Tmp(gk_ctx* c) : ctx(c) {}
    int alloc(size_t n) {
        void* q = nullptr;
        if (p) gk_dev_free(ctx, p), p = nullptr;      // a second alloc replaces the block
        int r = gk_dev_alloc(ctx, &q, (n ? n : 1) * sizeof(T));
        p = (T*)q;
        return r;
    }
    ~Tmp() {
        if (p) gk_dev_free(ctx, p);
    }
    Tmp(const Tmp&) = delete;
    Tmp& operator=(const Tmp&) = delete;
};

struct ProfScope {          // scopes may nest ("sp" around "sp_fw"): each owns its pair of events
    gk_ctx* ctx;
    const char* name;
    i64 launches;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(gk_ctx* c, const char* n, i64 l = 1) : ctx(c), name(n), launches(l) {
        if (ctx->profile && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess)
            (void)hipEventRecord(e0, ctx->stream);
    }
    ~ProfScope() {
        if (ctx->profile && e0 && e1) {
            (void)hipEventRecord(e1, ctx->stream);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            ProfSlot& s = ctx->prof[name];
            s.ms += ms;
            s.launches += launches;
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

// ---------------------------------------------------------------------------------------
// Batch: CSR graphs + per-level WL labels and label-grouped node orders.
// ---------------------------------------------------------------------------------------
struct gk_batch {
    gk_ctx* ctx = nullptr;
    i64 n_graphs = 0, n_nodes = 0, n_edges = 0;
    i32 n_labels0 = 0;
    i32 max_graph_nodes = 0;
    i32* graph_ptr = nullptr;   // [n_graphs+1]
    i32* row_ptr = nullptr;     // [n_nodes+1]
    i32* col_idx = nullptr;     // [n_edges]
    i32* node_graph = nullptr;  // [n_nodes]
    i32* big_nodes = nullptr;   // [n_big] nodes with degree > deg_small
    int wave_sig = 1;           // the batch's route for vertices above deg_small, fixed when the batch is created (option wl.no_wave_sig then): wave-per-vertex kernels (1) or the workgroup kernel + one-thread verifier of rounds 1-4 (0)
    int deg_small = 32;         // WL_DEG_SMALL, or 16 once the batch has any vertex above WL_DEG_SMALL (wl.hip: batch_finish)
    i64 n_big = 0;
    i32 max_degree = 0;
    // isolated vertices (degree 0): their WL class is "all isolated vertices with the same input
    // label" at every level >= 1 and never splits, so active-set levels carry these classes along
    // instead of hashing and sorting them again (wl.hip).  iso_info[v] >= 0: number of isolated
    // vertices before v; < 0: v is isolated and -1 - iso_info[v] is its slot in the carried list
    // (grouped by input label, ascending node inside a group); car_class[slot] = dense class id,
    // car_class[n_iso] = number of carried classes.
    i32* iso_info = nullptr;    // [n_nodes], null when the batch has no isolated vertex
    i32* car_class = nullptr;   // [n_iso + 1]
    i32* car_nodes = nullptr;   // [n_iso] the carried list itself: vertex at each slot
    i64 n_iso = 0;
    i64 n_isolated = 0;         // isolated vertices of the batch (n_iso stays 0 under option wl.no_iso: nothing is carried then)
    // levels
    int n_levels = 0;                  // levels currently valid (0 = only level-0 labels)
    int cap_levels = 0;
    i32* labels = nullptr;             // [cap_levels][n_nodes]
    i32* perm = nullptr;               // [cap_levels][n_nodes] nodes grouped by label (stable)
    unsigned char* shared_flag = nullptr;   // [cap_levels][n_nodes] full levels: 1 = the node's class has >= 2 members
    std::vector<i64> label_counts;     // per level
    // per level: perm[level][0 .. n_sorted) holds every node that can share its label with another
    // node (grouped by label); the nodes behind it carry labels of their own.  Empty = n_nodes.
    std::vector<i64> n_sorted;
    // per level: 1 when the ids follow the active-set layout [frozen singletons | carried classes | active classes]:
    // a node can share its label iff its id >= n_nodes - n_sorted[level] (features.hip, graph-major path)
    std::vector<char> active_layout;
    std::vector<char> perm_valid;      // per level: 0 when the relabel skipped the label-grouped order (sort-free dictionary)
    // scratch kept between levels
    i32* nbr_sorted = nullptr;         // [n_edges] sorted neighbour labels of the level being built
    bool is_pair_batch = false;        // ShortestPath items: no CSR, level 0 only
    // ShortestPath pair batch in HISTOGRAM form (sp.hip, features_gm.hip: gk_features_build_sp): no item arrays -- the
    // (l_u, l_v, d) counts of a graph are taken straight from its distance matrix; graph_ptr = pair ranges (entry slots).
    // The item arrays (labels / perm / node_graph) are materialised on demand (gk_sp_materialise) for the label-major builder.
    bool sp_hist = false;
    i32* sp_node_ptr = nullptr;        // [n_graphs + 1] node ranges of the source graphs
    i32* sp_node_label = nullptr;      // [source nodes] level-0 labels
    u64* sp_dist_ptr = nullptr;        // [n_graphs] start of each distance matrix
    i32* sp_dist = nullptr;            // the n x n matrices (SP_INF = unreachable)
    u32* sp_idtab = nullptr;           // [sp_keyspace] key -> dense feature id
    i64 sp_L = 0, sp_dcap = 0, sp_keyspace = 0, sp_src_nodes = 0;
    i32 sp_max_nodes = 0;       // largest source graph (sp_emit_kernel's slab grid)
    std::vector<i32> sp_h_node_ptr;    // host copies of sp_node_ptr / graph_ptr (pair ranges): the feature builder cuts the
    std::vector<u32> sp_h_pair_base;   // matrices of the large graphs into row units on the host
    int sp_with_labels = 0;
    // level 0 of a batch with at most GK_HIST0_MAX_LABELS input labels is never sorted: the label-count
    // features of that level come from one LDS histogram per graph (features.hip), perm[0] stays unused
    bool level0_hist = false;
    i32 n_labels0_present = 0;         // distinct level-0 label ids that occur (valid when n_labels0 <= GK_HIST0_MAX_LABELS)
    // ---- stream layout (wl_stream.hip: the relabel route without host round trips).  ids of a level l >= 1 are
    //   [carried classes 0 .. n_cc) | frozen nodes [n_cc, n_cc + F_l) | shared classes [.., + S_l) | new singletons [.., + T_l)
    // a label can be shared iff id < n_cc or n_cc + F_l <= id < n_cc + F_l + S_l; F_l / S_l / T_l live on the device
    // (sr_ctl, SR_CTL words per level) so that the feature builder can be queued without the host knowing them
    u64 relabel_gen = 0;               // bumped by every gk_wl_relabel: label ids of an earlier call are gone (wl_transform.hip)
    bool stream_layout = false;
    int sr_pending = 0;                // > 0: a stream relabel of that many levels is queued, its control words not yet seen by the host
    u32* sr_ctl = nullptr;             // [cap of sr_ctl_levels][SR_CTL]
    int sr_ctl_levels = 0;
    std::vector<u32> sr_F, sr_S;       // host copies (after the job's one read-back)
    u32 sr_ncc = 0;
};

// per-level control words of the stream layout (device)
#define SR_CTL 16
#define SR_F 0          // frozen nodes before this level (ids [n_cc, n_cc + F))
#define SR_S 1          // classes with two or more members among the level's active nodes
#define SR_T 2          // singleton classes among them
#define SR_COUNT 3      // labels of the level = n_cc + F + S + T
#define SR_LISTED 4     // active nodes whose class is shared (= the next level's active nodes)
#define SR_UNRES 5      // nodes whose full signature differs from their class representative's (hash collision)
#define SR_NCC 6        // carried classes (isolated vertices by input label)
#define SR_OVF 7        // a bucket of the dictionary overflowed

// ---------------------------------------------------------------------------------------
// ShortestPath distance matrices: 32-bit entries (SP_INF = unreachable), or -- round 6, the graphs the bit-parallel
// breadth-first search takes (unit weights, every distance below 255) -- BYTES (255 = unreachable).  A byte matrix lives at
// the start of the graph's own 32-bit region (n > 128: n * round_up(n, 16) + 15 <= 4 n^2), rows padded to 16 bytes so that a
// lane fetches eight entries with one load; bit 63 of the graph's dist_ptr word says which form it is (set by the search
// kernel itself, cleared whenever the offsets are computed again).  The matrices of the REDDIT-like set are 2.6 GB as
// 32-bit entries, written once and read twice.
// ---------------------------------------------------------------------------------------
#define SP_BYTE_FLAG (1ull << 63)
struct SpMat {
    const i32* d32; const unsigned char* d8; int ns;      // d8 != nullptr: byte form, row stride ns
};
__device__ __forceinline__ SpMat sp_mat(const i32* __restrict__ dist, const u64* __restrict__ dist_ptr, i64 g, int n) {
    const u64 o = dist_ptr[g];
    SpMat m;
    m.d32 = dist + (o & ~SP_BYTE_FLAG);
    m.d8 = (o & SP_BYTE_FLAG) ? (const unsigned char*)(((uintptr_t)m.d32 + 15) & ~(uintptr_t)15) : nullptr;
    m.ns = (n + 15) & ~15;
    return m;
}
// entry (i, j) as a 32-bit distance with `inf` for an unreachable pair
__device__ __forceinline__ i32 sp_mat_at(const SpMat& m, int n, int i, int j, i32 inf) {
    if (m.d8) {
        const unsigned char x = m.d8[(size_t)i * m.ns + j];
        return x == 255 ? inf : (i32)x;
    }
    return m.d32[(size_t)i * n + j];
}

// ---------------------------------------------------------------------------------------
// Features
// ---------------------------------------------------------------------------------------
struct LevelTriples {
    i32* tri_pos = nullptr;    // [n_nodes+1] start position (in perm order) of each triple
    i32* tri_graph = nullptr;  // [n_nodes+1]
    i32* tri_run = nullptr;    // [n_nodes+1] label run index of each triple
    i32* tstart = nullptr;     // [n_nodes+1] first triple of each label run
    i32* colid = nullptr;      // [n_nodes]   dense column id per label run, -2 rare, -1 dead
    i32* low_runs = nullptr;   // [n_nodes]   compact list of the rare label runs
    i32* wide = nullptr;       // [n_nodes+1] kind 0: 1 when some count of the run exceeds the int8 range;
                               //             kind 1: the largest count of the run (its unary width)
    i64 n_low = 0;
};

// Up to GK_PACK_LEVELS levels handed to ONE kernel launch by value (per-level launches of
// 5-20 us kernels cost more in launch latency than in work).
#define GK_PACK_LEVELS 16
struct LevelPack {
    const i32* tri_pos[GK_PACK_LEVELS];
    const i32* tri_graph[GK_PACK_LEVELS];
    const i32* tri_run[GK_PACK_LEVELS];
    const i32* tstart[GK_PACK_LEVELS];
    const i32* colid[GK_PACK_LEVELS];
    const i32* low_runs[GK_PACK_LEVELS];
    i64 first[GK_PACK_LEVELS + 1];    // prefix of the per-level item counts
    int n;
};

struct gk_feat {
    int kind = 0;               // GK_FEAT_DOT (0) | GK_FEAT_MINSUM (1), see gk_features_build_ex
    gk_ctx* ctx = nullptr;
    gk_batch* batch = nullptr;
    int n_levels = 0, level0 = 0;     // levels [level0, level0 + n_levels) of the batch
    i64 n_graphs = 0, n_fit = 0, n_nodes = 0;
    bool symmetric = true;
    std::vector<LevelTriples> lev;   // per-level VIEWS into the arena arrays below
    std::vector<void*> arena;        // concatenated triple / run arrays of all levels (owned)
    u32* meta = nullptr;        // device: per level {T, R, ncols_cum}, then globals
    u64* selfk = nullptr;       // [n_graphs] exact integer self similarity
    i64 n_cols = 0, n_cols_pad = 0, nnz = 0, max_count = 0, n_low_cols = 0;
    i64 own_lo = 0, own_hi = 0; // graphs whose operand rows THIS job assembled (all of them unless options feat.rows_lo / feat.rows_hi said otherwise)
    int low_df = 24;            // columns occurring in fewer graphs are applied as pair updates
    i64 n_rows_pad = 0;
    int dtype = 0;              // 0: int8 Phi, 1: f64 Phi
    bool dyn_type = false;      // the operand type of this job was decided on the device from the exact self similarities (features.h: GM_META_TYPE)
    // Dense MFMA operand, [n_rows_pad][n_cols_pad BYTES], K-steps of 128 B per row:
    //   [secondary: n_cols8 int8 columns, k8_steps steps]  -- only when the primary region is fp4: counts 5..127
    //   [primary: n_cols1 columns, k1_steps steps]          -- phi_fp4: two columns per byte as MX fp4 (e2m1) codes of
    //                                                          the counts 0..4 (256 columns per step); else int8 (128 per step)
    void* phi = nullptr;
    i64 n_cols1 = 0, n_cols8 = 0;
    int k1_steps = 0, k8_steps = 0;
    double k_bound = 0.0;       // upper bound of every entry of the job's matrix (n_levels * max_graph_nodes^2; * nodes for min-sum)
    bool phi_fp4 = false;       // counts <= 4 travel as fp4 (needs every Gram entry < 2^24: f32 accumulation stays exact)
    // dense columns holding a count > 127 cannot be int8 operands: they form a (usually
    // narrow) float64 side operand whose product is accumulated onto K after the int8 GEMM
    i64 n_cols_wide = 0, n_cols_wide_pad = 0;
    double* phi_w = nullptr;    // [n_rows_pad][n_cols_wide_pad]
    // labels with counts above 127 as split int8 columns (features.h: COL_SPLIT_BASE): phi is the LEFT operand of the
    // Gram product, phi_r the RIGHT one (same layout; the rows differ in the split columns only).  nullptr: phi on both sides
    void* phi_r = nullptr;
    int split_parts = 0;
    i64 n_split_labels = 0;
    // graph-major builder (features_gm.hip): the rare labels as lists instead of label-major triples
    bool gm = false;
    i32* gm_low_q = nullptr;    // [n_low_cols] label index of each rare label
    u32* gm_roff = nullptr;     // per label index: first entry of its list
    u32* gm_df = nullptr;       // per label index: entries (graphs) of its list
    i32* gm_low_graph = nullptr, *gm_low_cnt = nullptr;    // the lists: graph, count
    i32* gm_low_lab = nullptr;  // ... and the label index of every entry (the pair binning of gram.hip runs one thread per entry)
    // the rare labels' pair updates binned by 128x128 output tile (gram.hip: built on the first full symmetric Gram job,
    // applied to the parked tile inside gram_ws_kernel instead of as float64 atomics afterwards)
    u32* pair_cnt = nullptr;    // [pair_T * pair_T] pairs of tile (bm, bn), bm <= bn (may exceed pair_cap: the rest is in pair_ovf)
    uint2* pairs = nullptr;     // bucket of tile t = pairs[t * pair_cap ...]: .x = row in tile | col in tile << 7, .y = value
    int pair_T = 0;             // tiles per side; 0: not built
    int pair_cap = 0;           // bucket capacity
    uint4* pair_ovf = nullptr;  // pairs that found their bucket full: (row, col, value, -), applied as float64 atomics afterwards
    u32* pair_ovf_n = nullptr;
    double* rs = nullptr;       // [n_graphs] 1 / sqrt(selfk): normalised jobs' lean store path (gram.hip)
    i64 pair_ovf_cap = 0;
    i64 rare_entries = 0;       // entries of all rare labels' lists
    double* K = nullptr;        // last Gram output (device)
    i64 K_rows = 0, K_cols = 0;
    double last_flops = 0, last_ms = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the MFMA kernel of the last gk_gram* call
};

// ---- primitives (scan_sort.hip) -------------------------------------------------------
// inclusive/exclusive sums; `total` (device, may be null) receives the grand total.
int gk_scan_u32(gk_ctx* ctx, const u32* in, u32* out, i64 n, bool exclusive, u32* total);
int gk_scan_u64(gk_ctx* ctx, const u64* in, u64* out, i64 n, bool exclusive, u64* total);
// Stable LSD radix sort of (key,value) pairs on key bits [0,key_bits). Result in keys_out /
// vals_out; keys_in/vals_in are clobbered (used as ping-pong buffers).
// implicit_iota: the input values are 0..n-1 and are never read (vals_in is scratch only).
int gk_radix_sort_pairs(gk_ctx* ctx, const u64* keys_in, const u32* vals_in, u64* keys_out, u32* vals_out,
                        i64 n, int key_bits, int use_buckets = 0, u32* top_digit_max = nullptr);

// Dictionary without a sort (full WL levels whose label-grouped order nobody reads): see scan_sort.hip
bool gk_bucket_dictionary_fits(gk_ctx* ctx, i64 n);      // a-priori test: can no bucket overflow when every key is distinct?
int gk_bucket_dictionary(gk_ctx* ctx, const u64* keys, i64 n, int key_bits, i32* lab, i32* rep, u32* frozen,
                         unsigned char* shared_out, u32* count_dev, u32* listed_dev, u32* top_digit_max, u32* overflow,
                         u32* mbox, u32 seq, int flag_in_lab = 0);

// ---- wl.hip ---------------------------------------------------------------------------
int gk_batch_ensure_levels(gk_batch* b, int n_levels);
// wl_stream.hip: GK_ERR_UNSUPPORTED = not applicable to this job / a table overflowed / a hash collision: take the host-driven route
int gk_wl_relabel_stream(gk_ctx* ctx, gk_batch* b, int n_levels, int hash_bits, bool default_bits, std::vector<u32>& counts);
int gk_sr_enqueue(gk_ctx* ctx, gk_batch* b, int n_levels, int hash_bits, bool default_bits);     // the same without the read-back ...
int gk_sr_collect(gk_ctx* ctx, gk_batch* b, const u32* ctl_words);                                // ... which its caller hands in later
#define GK_ERR_RETRY (-100)     // internal: the queued stream relabel turned out unusable (collision / overflow) at a later read-back
// two device arrays in ONE mailbox round trip (n1 + n2 <= GK_MBOX_WORDS - 1; more: two copies)
int gk_readback2(gk_ctx* ctx, const u32* src1, int n1, const u32* src2, int n2, u32* dst_host);
int gk_readback_post(gk_ctx* ctx, const u32* src_dev, int n_words, u32* ticket);
int gk_readback_collect(gk_ctx* ctx, u32 ticket, const u32* src_dev, u32* dst_host, int n_words);
int gk_batch_rebuild_order(gk_ctx* ctx, gk_batch* b, int level);
int gk_sp_materialise(gk_ctx* ctx, gk_batch* pair_batch);            // sp.hip: item arrays of a histogram-form pair batch      // perm[level] on demand (sort-free dictionary levels)

// ---- gram.hip -------------------------------------------------------------------------
int gk_gram_launch(gk_ctx* ctx, gk_feat* f, i64 row_lo, i64 row_hi, int normalize, double* K);

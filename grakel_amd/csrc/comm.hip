// Multi-GPU entry points of the C ABI (include/gk_hip.h "multi-GPU"): one process per GPU, graphs sharded over the
// ranks, Gram rows sharded too (gk_comm_init / gk_gram_sharded; 8e: one exchange step).
//
// The path has ONE collective: the all-gather of the packed CSR shards.  WL labels are a global dictionary
// (grakel/kernels/weisfeiler_lehman.py:224-246), so every rank needs every graph's level-0 labels and adjacency before it
// relabels; after that relabel + label-count features run replicated and rank r computes and stores the rows of K that
// belong to its own graphs -- plain row blocks, nothing to exchange afterwards (DESIGN.md 5 has the costs of the
// alternatives).  This file is the same scheme grakel_amd/dist.py drives through torch.distributed, for consumers without
// Python: RCCL directly, on the context's stream.
//
// RCCL is loaded on the first gk_comm_* call (dlopen), not linked: a single-GPU consumer of libgk_hip.so does not need the
// library at all, and a process that already carries an RCCL (PyTorch ships its own copy) keeps using that one.
#include <dlfcn.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace {

// the slice of rccl.h this file uses (ABI of RCCL 2.x: /opt/rocm/include/rccl/rccl.h:40-43,187,220,260,339,533-,568-)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclInt32 = 2, ncclInt64 = 4 };

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

int rccl_load() {
    if (g_rccl.lib) return GK_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) {
        gk_set_error("gk_comm: RCCL not found (librccl.so.1): %s", dlerror());
        return GK_ERR_UNSUPPORTED;
    }
    Rccl r;
    r.lib = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
        gk_set_error("gk_comm: the RCCL found lacks an entry point this library calls");
        dlclose(h);
        return GK_ERR_UNSUPPORTED;
    }
    g_rccl = r;
    return GK_OK;
}

#define GK_RCCL_CHECK(expr)                                                                              \
    do {                                                                                                 \
        const ncclResult_t _r = (expr);                                                                  \
        if (_r != 0) {                                                                                   \
            gk_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return GK_ERR_HIP;                                                                           \
        }                                                                                                \
    } while (0)

}  // namespace

struct gk_comm {
    gk_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
};

extern "C" int gk_comm_unique_id(void* out_id) {
    GK_ARG(out_id, "gk_comm_unique_id: null argument");
    GK_TRY(rccl_load());
    ncclUniqueId id;
    GK_RCCL_CHECK(g_rccl.GetUniqueId(&id));
    memcpy(out_id, id.internal, sizeof id.internal);
    return GK_OK;
}

extern "C" int gk_comm_init(gk_ctx* ctx, int rank, int n_ranks, const void* id, gk_comm** out) {
    GK_ARG(ctx && id && out, "gk_comm_init: null argument");
    GK_ARG(n_ranks >= 1 && n_ranks <= GK_MAX_RANKS && rank >= 0 && rank < n_ranks, "gk_comm_init: rank outside [0, n_ranks), 1..64 ranks");
    GK_TRY(rccl_load());
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, sizeof uid.internal);
    gk_comm* c = new gk_comm();
    c->ctx = ctx, c->rank = rank, c->n_ranks = n_ranks;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, n_ranks, uid, rank);
    if (r != 0) {
        gk_set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        delete c;
        return GK_ERR_HIP;
    }
    *out = c;
    return GK_OK;
}

extern "C" int gk_comm_destroy(gk_comm* c) {
    if (!c) return GK_OK;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return GK_OK;
}

extern "C" int gk_comm_info(gk_comm* c, int* rank, int* n_ranks) {
    GK_ARG(c, "gk_comm_info: null");
    if (rank) *rank = c->rank;
    if (n_ranks) *n_ranks = c->n_ranks;
    return GK_OK;
}

// This rank's message for gk_batch_from_shards: [graph sizes | node degrees | node labels | col_idx, LOCAL node ids], each
// part zero padded to the largest shard
static void shard_message(std::vector<int32_t>& msg, int64_t ng, int64_t nv, int64_t ne, int64_t mg, int64_t mv, int64_t me,
                          const int32_t* graph_ptr, const int32_t* row_ptr, const int32_t* col_idx, const int32_t* node_label) {
    msg.assign((size_t)(mg + 2 * mv + me), 0);
    int32_t* p = msg.data();
    for (int64_t g = 0; g < ng; ++g) p[g] = graph_ptr[g + 1] - graph_ptr[g];
    p += mg;
    for (int64_t v = 0; v < nv; ++v) p[v] = row_ptr[v + 1] - row_ptr[v];
    p += mv;
    if (nv > 0) memcpy(p, node_label, (size_t)nv * 4);
    p += mv;
    if (ne > 0) memcpy(p, col_idx, (size_t)ne * 4);
}

extern "C" int gk_shard_message(int64_t n_graphs, int64_t n_nodes, int64_t n_edges, const int32_t* graph_ptr, const int32_t* row_ptr,
                                const int32_t* col_idx, const int32_t* node_label, int64_t mg, int64_t mv, int64_t me, int32_t* out_msg) {
    GK_ARG(out_msg, "gk_shard_message: null argument");
    GK_ARG(n_graphs >= 0 && n_nodes >= 0 && n_edges >= 0 && n_graphs <= mg && n_nodes <= mv && n_edges <= me, "gk_shard_message: shard larger than its padding");
    GK_ARG(n_graphs == 0 || (graph_ptr && row_ptr), "gk_shard_message: null shard arrays");
    GK_ARG((n_nodes == 0 || node_label) && (n_edges == 0 || col_idx), "gk_shard_message: null shard arrays");
    std::vector<int32_t> msg;
    shard_message(msg, n_graphs, n_nodes, n_edges, mg, mv, me, graph_ptr, row_ptr, col_idx, node_label);
    if (!msg.empty()) memcpy(out_msg, msg.data(), msg.size() * 4);
    return GK_OK;
}

extern "C" int gk_batch_allgather(gk_ctx* ctx, gk_comm* c, int64_t n_graphs, int64_t n_nodes, int64_t n_edges,
                                  const int32_t* graph_ptr, const int32_t* row_ptr, const int32_t* col_idx,
                                  const int32_t* node_label, int32_t n_labels0, gk_batch** out, int64_t* graph_bounds) {
    GK_ARG(ctx && c && out && graph_bounds, "gk_batch_allgather: null argument");
    GK_ARG(c->ctx == ctx, "gk_batch_allgather: communicator of another context");
    // The shard is validated BEFORE the first collective, but a rank whose shard is malformed must not return while the
    // others block in ncclAllGather: the verdict travels in the size exchange (n_graphs = -1) and EVERY rank returns
    // GK_ERR_ARG after it, together.
    const char* bad = nullptr;
    if (!(n_graphs >= 0 && n_nodes >= 0 && n_edges >= 0 && n_labels0 >= 1)) bad = "gk_batch_allgather: negative size";
    else if (!(n_graphs == 0 || (graph_ptr && row_ptr))) bad = "gk_batch_allgather: null shard arrays";
    else if (!(n_nodes == 0 || node_label)) bad = "gk_batch_allgather: null labels";
    else if (!(n_edges == 0 || col_idx)) bad = "gk_batch_allgather: null col_idx";
    // the shard in LOCAL numbering: the same checks gk_batch_create applies happen on the gathered batch (batch_finish)
    else if (!(n_graphs == 0 || (graph_ptr[0] == 0 && graph_ptr[n_graphs] == n_nodes))) bad = "gk_batch_allgather: graph_ptr must run from 0 to n_nodes";
    else if (!(n_nodes == 0 || (row_ptr[0] == 0 && row_ptr[n_nodes] == n_edges))) bad = "gk_batch_allgather: row_ptr must run from 0 to n_edges";
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    const int R = c->n_ranks;
    // ---- sizes of every shard (4 int64 per rank): one small all-gather, read back
    Tmp<i64> sz_dev(ctx);
    GK_TRY(sz_dev.alloc((size_t)4 * (R + 1)));
    const i64 mine[4] = {bad ? -1 : n_graphs, bad ? 0 : n_nodes, bad ? 0 : n_edges, (i64)(n_labels0 >= 1 ? n_labels0 : 1)};
    GK_HIP_CHECK(hipMemcpyAsync(sz_dev.p + 4 * R, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
    GK_RCCL_CHECK(g_rccl.AllGather(sz_dev.p + 4 * R, sz_dev.p, 4, ncclInt64, c->comm, ctx->stream));
    std::vector<i64> sz((size_t)4 * R);
    GK_HIP_CHECK(hipMemcpyAsync(sz.data(), sz_dev.p, sz.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (bad) { gk_set_error(bad); return GK_ERR_ARG; }
    for (int r = 0; r < R; ++r)
        GK_ARG(sz[4 * r] >= 0, "gk_batch_allgather: another rank's shard failed its validation");
    i64 mg = 0, mv = 0, me = 0, labels = 1;
    std::vector<i64> shard_sizes((size_t)3 * R);
    graph_bounds[0] = 0;
    for (int r = 0; r < R; ++r) {
        mg = std::max(mg, sz[4 * r]), mv = std::max(mv, sz[4 * r + 1]), me = std::max(me, sz[4 * r + 2]);
        labels = std::max(labels, sz[4 * r + 3]);
        shard_sizes[3 * r] = sz[4 * r], shard_sizes[3 * r + 1] = sz[4 * r + 1], shard_sizes[3 * r + 2] = sz[4 * r + 2];
        graph_bounds[r + 1] = graph_bounds[r] + sz[4 * r];
    }
    GK_ARG(sz[4 * c->rank] == n_graphs && sz[4 * c->rank + 1] == n_nodes, "gk_batch_allgather: the ranks of the communicator are out of step");
    const i64 stride = mg + 2 * mv + me;
    GK_ARG(stride > 0 && stride < (1ll << 31), "gk_batch_allgather: empty job, or a shard of more than 2^31 words");
    // ---- the shards: this rank's message up, one all-gather, the global batch built on the device from the R messages
    std::vector<int32_t> msg;
    shard_message(msg, n_graphs, n_nodes, n_edges, mg, mv, me, graph_ptr, row_ptr, col_idx, node_label);
    Tmp<i32> gathered(ctx), mine_dev(ctx);
    GK_TRY(gathered.alloc((size_t)stride * R));
    GK_TRY(mine_dev.alloc((size_t)stride));
    GK_HIP_CHECK(hipMemcpyAsync(mine_dev.p, msg.data(), (size_t)stride * 4, hipMemcpyHostToDevice, ctx->stream));
    GK_RCCL_CHECK(g_rccl.AllGather(mine_dev.p, gathered.p, (size_t)stride, ncclInt32, c->comm, ctx->stream));
    GK_TRY(gk_batch_from_shards(ctx, R, shard_sizes.data(), mg, mv, me, gathered.p, (int32_t)labels, out));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));       // msg (pageable host memory) and the temporaries are released on return
    return GK_OK;
}

extern "C" int gk_gram_sharded(gk_ctx* ctx, gk_comm* c, gk_feat* f, const int64_t* graph_bounds, int normalize, double* out_host,
                               int64_t* row_lo, int64_t* row_hi) {
    GK_ARG(ctx && c && f && graph_bounds, "gk_gram_sharded: null argument");
    const i64 lo = graph_bounds[c->rank], hi = graph_bounds[c->rank + 1];
    GK_ARG(lo >= 0 && hi >= lo && hi <= f->n_graphs - (f->symmetric ? 0 : f->n_fit), "gk_gram_sharded: the bounds are not those of this job's graphs");
    if (row_lo) *row_lo = lo;
    if (row_hi) *row_hi = hi;
    if (hi == lo) return GK_OK;                       // a rank without graphs owns no rows
    return gk_gram_rows(ctx, f, lo, hi, normalize, out_host);
}

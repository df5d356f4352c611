// Context, error reporting, allocation and timing entry points of libgk_hip.so.
#include "common.h"
#include <chrono>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>

static thread_local char g_err[1024] = "";
static void cache_release_all(gk_ctx* ctx);
static void* block_base(gk_ctx* ctx, void* user);
static int guard_verdict(gk_ctx* ctx);

void gk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gk_last_error(void) { return g_err; }
extern "C" const char* gk_version(void) { return "gk_hip 0.1 (gfx950)"; }

extern "C" int gk_device_count(int* out_count) {
    GK_ARG(out_count, "gk_device_count: null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *out_count = n;
    return GK_OK;
}

extern "C" int gk_create(int device_id, gk_ctx** out) {
    GK_ARG(out, "gk_create: null out");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        gk_set_error("gk_create: no HIP device visible (this library has no CPU fallback)");
        return GK_ERR_HIP;
    }
    GK_ARG(device_id >= 0 && device_id < n, "gk_create: bad device id");
    GK_HIP_CHECK(hipSetDevice(device_id));
    gk_ctx* ctx = new gk_ctx();
    ctx->device = device_id;
    {
        hipDeviceProp_t prop;
        ctx->n_cu = (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    GK_HIP_CHECK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    GK_HIP_CHECK(hipEventCreate(&ctx->ev0));
    GK_HIP_CHECK(hipEventCreate(&ctx->ev1));
    GK_HIP_CHECK(hipEventCreate(&ctx->pv0));
    GK_HIP_CHECK(hipEventCreate(&ctx->pv1));
    {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, GK_MBOX_WORDS * 4, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
            hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            memset(h, 0, GK_MBOX_WORDS * 4);
            ctx->mbox_host = (u32*)h, ctx->mbox_dev = (u32*)d;
        } else {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
        }
    }
    *out = ctx;
    return GK_OK;
}

extern "C" int gk_destroy(gk_ctx* ctx) {
    if (!ctx) return GK_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipEventDestroy(ctx->pv0);
    (void)hipEventDestroy(ctx->pv1);
    if (ctx->mbox_host) (void)hipHostFree(ctx->mbox_host);
    if (ctx->host_pool) gk_host_pool_destroy(ctx->host_pool);
    if (ctx->stage_host) (void)hipHostFree(ctx->stage_host);
    if (ctx->xfer_host) (void)hipHostFree(ctx->xfer_host);
    for (int i = 0; i < 4; ++i)
        if (ctx->stage_ev[i]) (void)hipEventDestroy(ctx->stage_ev[i]);
    cache_release_all(ctx);
    for (auto& kv : ctx->cache.live) (void)hipFree(block_base(ctx, kv.first));   // leaked by the caller: reclaim
    if (ctx->cache.guard_faults) (void)hipFree(ctx->cache.guard_faults);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
    if (ctx->side_join) (void)hipEventDestroy(ctx->side_join);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return GK_OK;
}

// ---- small read-backs ---------------------------------------------------------------------
// A hipMemcpyAsync(D2H) + hipStreamSynchronize pair costs a staging-copy kernel plus a stream drain
// (~25 us of idle GPU each; a WL job needs one per level).  Instead one tiny kernel stores the words
// into mapped pinned host memory and then releases a sequence number the host spins on.
__global__ void mbox_post_kernel(const u32* __restrict__ src, int n_words, u32* __restrict__ mbox, u32 seq) {
    for (int i = threadIdx.x; i < n_words; i += blockDim.x)
        __hip_atomic_store(&mbox[1 + i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();        // every thread: its words are visible to the host before the barrier
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(&mbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// A kernel that can post by itself (e.g. the finish hook of a scan) takes gk_mbox_begin()'s sequence
// number and ctx->mbox_dev, stores its words at mbox[1..] and then releases mbox[0] = seq at system
// scope; the host collects them with gk_mbox_wait().  Returns 0 when the mailbox is unavailable.
u32 gk_mbox_begin(gk_ctx* ctx) {
    if (!ctx->mbox_host || ctx->opt.no_mailbox) return 0;
    if (++ctx->mbox_seq == 0) ++ctx->mbox_seq;     // never 0
    return ctx->mbox_seq;
}

int gk_mbox_wait(gk_ctx* ctx, u32 seq, u32* dst_host, int n_words) {
    volatile u32* box = ctx->mbox_host;
    const auto t0 = std::chrono::steady_clock::now();
    u64 spins = 0;
    while (__atomic_load_n(&box[0], __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0xfffff) == 0) {
            // very slow or failed kernel: fall back to a real synchronisation (reports the error, if any)
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                if (__atomic_load_n(&box[0], __ATOMIC_ACQUIRE) != seq) {
                    gk_set_error("gk_mbox_wait: the device never posted its values");
                    return GK_ERR_HIP;
                }
                break;
            }
        }
    }
    for (int i = 0; i < n_words; ++i) dst_host[i] = box[1 + i];
    return GK_OK;
}

int gk_readback(gk_ctx* ctx, const u32* src_dev, u32* dst_host, int n_words) {
    if (n_words <= 0) return GK_OK;
    const u32 seq = n_words <= GK_MBOX_WORDS - 1 ? gk_mbox_begin(ctx) : 0;
    if (!seq) {
        GK_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, (size_t)n_words * 4, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return GK_OK;
    }
    mbox_post_kernel<<<1, 256, 0, ctx->stream>>>(src_dev, n_words, ctx->mbox_dev, seq);
    GK_HIP_CHECK(hipGetLastError());
    return gk_mbox_wait(ctx, seq, dst_host, n_words);
}

// The same in two halves: gk_readback_post queues the post and returns its ticket, the caller queues whatever the device can
// do meanwhile (work that does not depend on the values), gk_readback_collect waits -- the device is not idle for the host's
// round trip.  Ticket 0: the mailbox is unavailable, gk_readback_collect then does the synchronous copy.
int gk_readback_post(gk_ctx* ctx, const u32* src_dev, int n_words, u32* ticket) {
    *ticket = (n_words > 0 && n_words <= GK_MBOX_WORDS - 1) ? gk_mbox_begin(ctx) : 0;
    if (*ticket) {
        mbox_post_kernel<<<1, 256, 0, ctx->stream>>>(src_dev, n_words, ctx->mbox_dev, *ticket);
        GK_HIP_CHECK(hipGetLastError());
    }
    return GK_OK;
}

int gk_readback_collect(gk_ctx* ctx, u32 ticket, const u32* src_dev, u32* dst_host, int n_words) {
    if (n_words <= 0) return GK_OK;
    if (ticket) return gk_mbox_wait(ctx, ticket, dst_host, n_words);
    GK_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, (size_t)n_words * 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

__global__ void mbox_post2_kernel(const u32* __restrict__ src1, int n1, const u32* __restrict__ src2, int n2, u32* __restrict__ mbox, u32 seq) {
    for (int i = threadIdx.x; i < n1 + n2; i += blockDim.x)
        __hip_atomic_store(&mbox[1 + i], i < n1 ? src1[i] : src2[i - n1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(&mbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int gk_readback2(gk_ctx* ctx, const u32* src1, int n1, const u32* src2, int n2, u32* dst_host) {
    const u32 seq = n1 + n2 <= GK_MBOX_WORDS - 1 ? gk_mbox_begin(ctx) : 0;
    if (!seq) {
        GK_TRY(gk_readback(ctx, src1, dst_host, n1));
        return gk_readback(ctx, src2, dst_host + n1, n2);
    }
    mbox_post2_kernel<<<1, 256, 0, ctx->stream>>>(src1, n1, src2, n2, ctx->mbox_dev, seq);
    GK_HIP_CHECK(hipGetLastError());
    return gk_mbox_wait(ctx, seq, dst_host, n1 + n2);
}

int gk_side_fork(gk_ctx* ctx, hipStream_t* side) {
    if (!ctx->side_stream) {
        GK_HIP_CHECK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        GK_HIP_CHECK(hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
        GK_HIP_CHECK(hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming));
    }
    GK_HIP_CHECK(hipEventRecord(ctx->side_fork, ctx->stream));
    GK_HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ctx->side_fork, 0));
    *side = ctx->side_stream;
    return GK_OK;
}

int gk_side_join(gk_ctx* ctx) {
    GK_HIP_CHECK(hipEventRecord(ctx->side_join, ctx->side_stream));
    GK_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->side_join, 0));
    return GK_OK;
}

// Pinned host memory for Gram outputs: the float64 matrix is 8 N^2 bytes (800 MB at 10 k graphs) and a
// device -> pageable copy runs at 12-18 GB/s, into pinned memory at 57 GB/s (tools/micro/pinbw.hip).
extern "C" int gk_host_alloc(uint64_t bytes, void** out) {
    GK_ARG(out && bytes > 0, "gk_host_alloc: bad argument");
    void* p = nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        gk_set_error("gk_host_alloc: %llu bytes of pinned host memory are not available", (unsigned long long)bytes);
        *out = nullptr;
        return GK_ERR_HIP;
    }
    *out = p;
    return GK_OK;
}

extern "C" int gk_host_free(void* p) {
    if (p) GK_HIP_CHECK(hipHostFree(p));
    return GK_OK;
}

// ---- options ---------------------------------------------------------------------------------
struct OptName { const char* name; int gk_opts::*field; };
static const OptName g_opt_names[] = {
    {"wl.no_tiny", &gk_opts::wl_no_tiny}, {"wl.no_listscan", &gk_opts::wl_no_listscan}, {"wl.no_iso", &gk_opts::wl_no_iso},
    {"wl.no_split", &gk_opts::wl_no_split}, {"wl.no_exact1", &gk_opts::wl_no_exact1}, {"wl.no_active_set", &gk_opts::wl_no_active_set},
    {"wl.no_bucket_dict", &gk_opts::wl_no_bucket_dict}, {"wl.no_hist0", &gk_opts::wl_no_hist0},
    {"wl.frozen_words", &gk_opts::wl_frozen_words}, {"wl.flag_bytes", &gk_opts::wl_flag_bytes},
    {"wl.sig_no_regs", &gk_opts::wl_sig_no_regs}, {"wl.no_stream", &gk_opts::wl_no_stream}, {"wl.debug", &gk_opts::wl_debug}, {"sort.buckets", &gk_opts::sort_buckets},
    {"wl.bd_slots", &gk_opts::bd_slots}, {"feat.no_gm", &gk_opts::feat_no_gm}, {"feat.gm_no_priv", &gk_opts::gm_no_priv}, {"feat.gm_rows_wg", &gk_opts::gm_rows_wg},
    {"feat.low_df", &gk_opts::low_df}, {"feat.gm_row_lds_max", &gk_opts::gm_row_lds_max},
    {"gram.dd", &gk_opts::gram_dd}, {"gram.no_fp4", &gk_opts::gram_no_fp4}, {"gram.no_ws", &gk_opts::gram_no_ws}, {"gram.no_sym", &gk_opts::gram_no_sym},
    {"gram.no_patch", &gk_opts::gram_no_patch}, {"gram.xcc", &gk_opts::gram_xcc}, {"feat.gm_no_huge", &gk_opts::gm_no_huge}, {"feat.gm_rows_256", &gk_opts::gm_rows_256}, {"feat.gm_no_early_post", &gk_opts::gm_no_early_post}, {"feat.rows_lo", &gk_opts::feat_rows_lo}, {"feat.rows_hi", &gk_opts::feat_rows_hi}, {"gram.strip", &gk_opts::gram_strip}, {"gram.no_compact", &gk_opts::gram_no_compact}, {"gram.no_split8", &gk_opts::gram_no_split8}, {"gram.no_split64", &gk_opts::gram_no_split64}, {"gram.fold", &gk_opts::gram_fold}, {"gram.pair_cap", &gk_opts::gram_pair_cap}, {"gram.copy_threads", &gk_opts::gram_copy_threads}, {"gram.no_tri", &gk_opts::gram_no_tri}, {"gram.no_avx2", &gk_opts::gram_no_avx2}, {"wl.no_wave_sig", &gk_opts::wl_no_wave_sig}, {"wl.no_frozen_skip", &gk_opts::wl_no_frozen_skip}, {"wl.no_converge", &gk_opts::wl_no_converge}, {"transform.no_fused", &gk_opts::tt_no_fused}, {"scan.direct_max", &gk_opts::scan_direct_max}, {"sp.no_reg", &gk_opts::sp_no_reg}, {"sp.no_pk", &gk_opts::sp_no_pk}, {"sp.no_hist", &gk_opts::sp_no_hist}, {"sp.no_prep", &gk_opts::sp_no_prep}, {"sp.no_rows", &gk_opts::sp_no_rows}, {"sp.no_bfs", &gk_opts::sp_no_bfs}, {"sp.bfs_one_stream", &gk_opts::sp_bfs_one_stream}, {"sp.bfs_no_lds_cols", &gk_opts::sp_bfs_no_lds_cols}, {"sp.bfs_no_bytes", &gk_opts::sp_bfs_no_bytes}, {"sp.rows_all", &gk_opts::sp_rows_all}, {"sp.no_fused_mark", &gk_opts::sp_no_fused_mark}, {"sp.hist_no_batch", &gk_opts::sp_hist_no_batch}, {"sp.static_type", &gk_opts::sp_static_type}, {"sp.rows_no_merge", &gk_opts::sp_rows_no_merge}, {"sp.hist_unit", &gk_opts::sp_hist_unit}, {"sp.hist_slots", &gk_opts::sp_hist_slots}, {"no_mailbox", &gk_opts::no_mailbox},
    {"debug.poison", &gk_opts::poison}, {"debug.guard", &gk_opts::guard},
};

extern "C" int gk_set_option(gk_ctx* ctx, const char* name, int64_t value) {
    GK_ARG(ctx && name, "gk_set_option: null argument");
    GK_ARG(value >= -2147483647 && value <= 2147483647, "gk_set_option: value out of range");
    for (const OptName& o : g_opt_names)
        if (!strcmp(o.name, name)) {
            ctx->opt.*(o.field) = (int)value;
            return GK_OK;
        }
    gk_set_error("gk_set_option: unknown option '%s'", name);
    return GK_ERR_ARG;
}

extern "C" int gk_get_option(gk_ctx* ctx, const char* name, int64_t* out_value) {
    GK_ARG(ctx && name && out_value, "gk_get_option: null argument");
    for (const OptName& o : g_opt_names)
        if (!strcmp(o.name, name)) {
            *out_value = ctx->opt.*(o.field);
            return GK_OK;
        }
    gk_set_error("gk_get_option: unknown option '%s'", name);
    return GK_ERR_ARG;
}

extern "C" int gk_set_stream(gk_ctx* ctx, void* hip_stream) {
    GK_ARG(ctx, "gk_set_stream: null ctx");
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return GK_OK;
}

extern "C" int gk_synchronize(gk_ctx* ctx) {
    GK_ARG(ctx, "gk_synchronize: null ctx");
    if (ctx->cache.guard_faults) return guard_verdict(ctx);
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

extern "C" int gk_timer_start(gk_ctx* ctx) {
    GK_ARG(ctx, "gk_timer_start: null ctx");
    GK_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
    return GK_OK;
}

extern "C" int gk_timer_stop_ms(gk_ctx* ctx, double* out_ms) {
    GK_ARG(ctx && out_ms, "gk_timer_stop_ms: null argument");
    GK_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
    GK_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    GK_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *out_ms = ms;
    return GK_OK;
}

extern "C" int gk_profile_enable(gk_ctx* ctx, int enable) {
    GK_ARG(ctx, "gk_profile_enable: null ctx");
    ctx->profile = enable != 0;
    return GK_OK;
}

extern "C" int gk_profile_reset(gk_ctx* ctx) {
    GK_ARG(ctx, "gk_profile_reset: null ctx");
    ctx->prof.clear();
    return GK_OK;
}

extern "C" int gk_profile_get(gk_ctx* ctx, const char* name, double* out_ms, int64_t* out_launches) {
    GK_ARG(ctx && name, "gk_profile_get: null argument");
    auto it = ctx->prof.find(name);
    if (out_ms) *out_ms = it == ctx->prof.end() ? 0.0 : it->second.ms;
    if (out_launches) *out_launches = it == ctx->prof.end() ? 0 : it->second.launches;
    return GK_OK;
}

int gk_func_lds(gk_ctx* ctx, const void* func, int bytes) {
    if (bytes <= 32 * 1024) return GK_OK;        // below every default limit
    int& have = ctx->func_lds[func];
    if (bytes > have) {
        GK_HIP_CHECK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        have = bytes;
    }
    return GK_OK;
}

static size_t bucket_size(size_t bytes) {
    if (bytes < 512) return 512;
    if (bytes <= (1u << 20)) {   // next power of two up to 1 MiB
        size_t b = 512;
        while (b < bytes) b <<= 1;
        return b;
    }
    const size_t g = 2u << 20;    // 2 MiB granules above
    return (bytes + g - 1) / g * g;
}

static void* block_base(gk_ctx* ctx, void* user) {
    return ctx->cache.guarded.count(user) ? (void*)((char*)user - GK_GUARD_BYTES) : user;
}
static void cache_release_all(gk_ctx* ctx) {
    for (auto& kv : ctx->cache.free_blocks) {
        (void)hipFree(block_base(ctx, kv.second));
        ctx->cache.guarded.erase(kv.second);
    }
    ctx->cache.free_blocks.clear();
}

// ---- debug.guard: red zones (the round-2 device fault was never reproduced; a write outside a block would be one way to
// produce it, and this makes such a write an ERROR at the next gk_synchronize instead of a corruption somewhere else)
#define GK_GUARD_WORD 0xA5C3F00Du
__global__ void guard_fill_kernel(u32* __restrict__ a, size_t na, u32* __restrict__ b, size_t nb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a[i] = GK_GUARD_WORD;
    if (i < nb) b[i] = GK_GUARD_WORD;
}
__global__ void guard_check_kernel(const u32* __restrict__ a, size_t na, const u32* __restrict__ b, size_t nb, u32 req_kib,
                                   u32* __restrict__ faults) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 bad = 0;
    if (i < na && a[i] != GK_GUARD_WORD) ++bad;
    if (i < nb && b[i] != GK_GUARD_WORD) ++bad;
    if (bad) {
        if (atomicAdd(&faults[0], bad) == 0) faults[1] = req_kib;
    }
}
// the zones of a guarded block: [user - GUARD, user) and [user + round16(requested), + min(slack + GUARD, TAIL_MAX))
static void guard_zones(void* user, size_t cap, size_t bytes, u32** a, size_t* na, u32** b, size_t* nb) {
    const size_t r = (bytes + 15) & ~(size_t)15;
    size_t tail = cap - r + GK_GUARD_BYTES;
    if (tail > GK_GUARD_TAIL_MAX) tail = GK_GUARD_TAIL_MAX;
    *a = (u32*)((char*)user - GK_GUARD_BYTES), *na = GK_GUARD_BYTES / 4;
    *b = (u32*)((char*)user + r), *nb = tail / 4;
}
static void guard_fill(gk_ctx* ctx, void* user, size_t cap, size_t bytes) {
    u32 *a, *b;
    size_t na, nb;
    guard_zones(user, cap, bytes, &a, &na, &b, &nb);
    const size_t n = na > nb ? na : nb;
    guard_fill_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(a, na, b, nb);
}
static void guard_check(gk_ctx* ctx, void* user, size_t cap, size_t bytes) {
    if (!ctx->cache.guard_faults) return;
    u32 *a, *b;
    size_t na, nb;
    guard_zones(user, cap, bytes, &a, &na, &b, &nb);
    const size_t n = na > nb ? na : nb;
    guard_check_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream>>>(a, na, b, nb, (u32)(bytes >> 10), ctx->cache.guard_faults);
}
// every live guarded block is checked, then the fault words are read: GK_ERR_STATE when a red zone was written
static int guard_verdict(gk_ctx* ctx) {
    BlockCache& c = ctx->cache;
    if (!c.guard_faults) return GK_OK;
    for (auto& kv : c.guarded) {
        auto it = c.live.find(kv.first);
        if (it != c.live.end()) guard_check(ctx, kv.first, it->second, kv.second);
    }
    u32 h[2] = {0, 0};
    GK_HIP_CHECK(hipMemcpyAsync(h, c.guard_faults, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (h[0]) {
        gk_set_error("debug.guard: %u words of a red zone were overwritten (first hit: a block requested with ~%u KiB)", h[0], h[1]);
        (void)hipMemsetAsync(c.guard_faults, 0, 8, ctx->stream);
        for (auto& kv : c.guarded) {                 // reported once: the zones of the live blocks are made whole again
            auto it = c.live.find(kv.first);
            if (it != c.live.end()) guard_fill(ctx, kv.first, it->second, kv.second);
        }
        return GK_ERR_STATE;
    }
    return GK_OK;
}

// debug.poison: every block the allocator hands out is filled with a byte pattern first, so that a kernel
// reading memory nobody wrote sees the same garbage in every run (tests/tools/poison_suite.sh)
__global__ void gk_poison_kernel(uint4* __restrict__ p, size_t n16, u32 word) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 z = make_uint4(word, word, word, word);
    for (; i < n16; i += stride) p[i] = z;
}
static void poison_block(gk_ctx* ctx, void* p, size_t cap) {
    if (!ctx->opt.poison) return;
    const u32 b = (u32)ctx->opt.poison & 0xffu, word = b | (b << 8) | (b << 16) | (b << 24);
    size_t blocks = (cap / 16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    gk_poison_kernel<<<dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, ctx->stream>>>((uint4*)p, cap / 16, word);
}

int gk_dev_alloc(gk_ctx* ctx, void** p, size_t bytes) {
    const size_t cap = bucket_size(bytes);
    BlockCache& c = ctx->cache;
    const bool guard = ctx->opt.guard != 0;
    if (guard && !c.guard_faults) {
        if (hipMalloc((void**)&c.guard_faults, 8) != hipSuccess) { (void)hipGetLastError(); c.guard_faults = nullptr; }
        else (void)hipMemsetAsync(c.guard_faults, 0, 8, ctx->stream);
    }
    for (auto it = c.free_blocks.lower_bound(cap); it != c.free_blocks.end() && it->first <= cap + cap / 2; ++it) {   // bounded internal waste
        if ((c.guarded.count(it->second) != 0) != guard) continue;       // blocks with and without red zones do not mix
        *p = it->second;
        c.live[*p] = it->first;
        poison_block(ctx, *p, it->first);
        if (guard) { c.guarded[*p] = bytes; guard_fill(ctx, *p, it->first, bytes); }
        c.free_blocks.erase(it);
        return GK_OK;
    }
    const size_t extra = guard ? 2 * (size_t)GK_GUARD_BYTES : 0;
    hipError_t e = hipMalloc(p, cap + extra);
    if (e != hipSuccess) {       // give cached blocks back to the driver and retry once
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        cache_release_all(ctx);
        e = hipMalloc(p, cap + extra);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        gk_set_error("device allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        *p = nullptr;
        return GK_ERR_HIP;
    }
    if (guard) *p = (char*)*p + GK_GUARD_BYTES;
    c.live[*p] = cap;
    c.bytes_total += cap;
    poison_block(ctx, *p, cap);
    if (guard) { c.guarded[*p] = bytes; guard_fill(ctx, *p, cap, bytes); }
    return GK_OK;
}

void gk_dev_free(gk_ctx* ctx, void* p) {
    if (!p) return;
    BlockCache& c = ctx->cache;
    auto it = c.live.find(p);
    if (it == c.live.end()) return;   // not ours / double free: ignore rather than corrupt
    auto g = c.guarded.find(p);
    if (g != c.guarded.end()) guard_check(ctx, p, it->second, g->second);      // queued behind the block's last user (stream order)
    c.free_blocks.insert({it->second, p});
    c.live.erase(it);
}

// 16-byte grid-stride zero fill.  hipMallocAsync blocks are 256-byte aligned; tails are bytes.
__global__ void gk_zero_kernel(uint4* __restrict__ p, size_t n16, unsigned char* __restrict__ tail, int ntail) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (; i < n16; i += stride) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

int gk_zero_async(gk_ctx* ctx, void* p, size_t bytes) {
    if (bytes == 0) return GK_OK;
    size_t n16 = bytes / 16;
    int ntail = (int)(bytes - n16 * 16);
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    gk_zero_kernel<<<dim3((unsigned)blocks), dim3(256), 0, ctx->stream>>>((uint4*)p, n16, (unsigned char*)p + n16 * 16, ntail);
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

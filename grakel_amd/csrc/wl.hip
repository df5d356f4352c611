// Weisfeiler-Lehman relabelling on gfx950.
//
// One level (reference: grakel/kernels/weisfeiler_lehman.py:223-258):
//   1. wl_signature_*  : per node, gather the previous labels of its out-neighbours (coalesced
//                        col_idx stream, LDS-staged), sort them, write the sorted list to
//                        nbr_sorted[] and form a 64-bit multiset hash of (own, degree, list).
//   2. dictionary       : equal hashes -> one dense label id, the lowest node of the class = its representative.
//                        Sort-free form (scan_sort.hip: top-digit partition + one LDS table of DISTINCT keys per
//                        bucket, bucket_dict_kernel / bucket_assign_kernel) whenever gk_bucket_dictionary_fits;
//                        sorting form (stable radix sort of (hash, node), run heads + scan) for active-set levels,
//                        pair items, and as the redo path when a bucket's table overflows after all
//                        (gk_wl_relabel then relabels the whole job again with the sorting dictionary).
//   3. singleton flags  : ride in bit 31 of the label word until verify strips them (shared_flag[]).
//   4. verify           : every node compares its FULL signature (own label, degree, sorted
//                        list) with its group's representative -> the dictionary is exact, the
//                        hash only proposes groups.  Any mismatch (a 64-bit collision) is
//                        counted; the host then re-runs that level in "exact" mode, refining
//                        groups with re-seeded hashes until no mismatch is left.
// Work that cannot change the partition is not done (every switch below is a context option, gk_set_option "wl.*";
// tests/test_gpu_parity.py runs the job through the removal of each):
//   * level 1 with few input labels and small degrees: exact 32-bit signature codes instead of
//     hashes -- no sorted lists, fewer digit passes, nothing to verify (wl_signature_exact_kernel);
//   * a level that sorts every node splits its label-grouped order: nodes of classes >= 2 first
//     (HeadAssignSplit); only those are "listed" for the label-count features;
//   * from level 2 on only the ACTIVE nodes go through 1-4 once they are at most a quarter of the
//     batch: a node whose class is a singleton keeps a class of its own for ever -- it is frozen, with
//     a permanent id ([frozen | carried | active] id layout), so a level starts as a copy of the
//     previous level's labels; the first such level finds the active nodes among all nodes
//     (ActiveScan, frozen_assign_verify_kernel), the following ones compact the previous list
//     (ActiveFromList, active_finish_kernel);
//   * isolated vertices form one class per input label for ever: carried, never sorted again;
//   * at most 1024 active nodes: the whole level is one single-workgroup launch
//     (wl_tiny_level_kernel), no host read-back.
// The per-level sizes reach the host through the mailbox of api.hip, not through a stream
// synchronisation.  HBM-bound integer work: algorithmic bytes per level 8E + 12V.
#include "common.h"
#include "scan_fn.h"
#include "features.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <memory>

#include "wl_sig.h"

// Round 6: a node whose class was a singleton at the level before stays alone for ever (classes only split) -- at a FULL level
// (every node gets a key) the wave / workgroup kernels do not gather, sort and hash its neighbours: its key is a hash of its own
// (unique) label under a degree no node has.  A thread starter with 2 500 answers is alone from level 1 on; its 78-barrier
// LDS sort was 40 us of every level of the REDDIT-like batch.  Nobody reads the sorted list of a singleton (it is its own
// representative); a collision of the made-up key with a real one is what the exact verification and the redo are for.
#define SIG_FROZEN_KEY(own_label, seed) mix64(sig_head((u32)(own_label), 0xffffffffu, (seed)))

// ---------------------------------------------------------------------------------------
// Nodes of degree WL_DEG_SMALL + 1 .. WAVE_DEG_MAX (round 5): ONE WAVE per node, the neighbour labels in registers
// (striped: element i = 64 r + lane, R = 1, 2, 4, 8 or 16 registers per lane), a bitonic network over the wave -- partners
// less than 64 apart by a lane shuffle, farther apart in the lane's own registers (static indices) -- coalesced gather and
// coalesced write of the sorted list, wave reduction of the multiset hash.  No LDS, no workgroup barrier: the workgroup
// form below (64 KiB of LDS and ~log^2 barriers per NODE) took 3.5 ms per level on a COLLAB-like batch (360 k nodes of
// degree ~ 60), this one is bound by the gather.
// ---------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ u64 wave_node_signature(const i32* __restrict__ col_idx, const i32* __restrict__ lab_prev,
                                                   i32* __restrict__ nbr_sorted, i32 e0, int d, int lane, u64 seed) {
    i32 x[R];
    u64 part = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = r * 64 + lane;
        x[r] = 0x7fffffff;
        if (i < d) {
            x[r] = lab_prev[col_idx[e0 + i]];
            part += sig_elem((u32)x[r], seed);
        }
    }
    wave_bitonic_sort<R>(x, lane);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = r * 64 + lane;
        if (i < d) nbr_sorted[e0 + i] = x[r];
    }
    return part;
}

__global__ __launch_bounds__(SIG_THREADS) void wl_signature_small_kernel(
    const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx,
    const i32* __restrict__ lab_prev, i32* __restrict__ nbr_sorted, u64* __restrict__ hash,
    i64 V, u64 seed, u64 mask, int sig_regs, int deg_small) {
    __shared__ i32 buf[SIG_LDS_CAP];
    const int tid = threadIdx.x;
    const i64 v0 = (i64)blockIdx.x * SIG_THREADS;
    const i64 v1 = (v0 + SIG_THREADS < V) ? v0 + SIG_THREADS : V;
    const i32 e0 = row_ptr[v0], e1 = row_ptr[v1];
    const int cnt = e1 - e0;
    const bool use_lds = cnt <= SIG_LDS_CAP;   // block-uniform
    const i64 v = v0 + tid;
    // a chunk whose nodes all belong to the wave / workgroup kernels below (degree > WL_DEG_SMALL: ego networks, cliques)
    // has nothing to stage here -- round 5: a COLLAB-like batch streamed its 21 M neighbour labels through this loop for
    // nothing, 130 us per level
    {
        const int dv = v < v1 ? row_ptr[v + 1] - row_ptr[v] : 0;
        if (!__syncthreads_or(dv >= 1 && dv <= deg_small)) {
            if (v < v1 && dv == 0) hash[v] = mix64(sig_head((u32)lab_prev[v], 0u, seed)) & mask;
            return;
        }
    }
    // a chunk that ALSO holds nodes of the wave / workgroup kernels (mixed degrees: the few low-degree members of an ego
    // network) does not stream their lists through here: its small nodes gather their own neighbours, as the list kernel does
    {
        const int dv = v < v1 ? row_ptr[v + 1] - row_ptr[v] : 0;
        if (__syncthreads_or(dv > deg_small)) {
            const i32 s = v < v1 ? row_ptr[v] : 0;
            if (v < v1 && dv <= 16) hash[v] = mix64(node_key_regs(col_idx, lab_prev, nbr_sorted + s, s, dv, (u32)lab_prev[v], seed)) & mask;
            // 17 .. WL_DEG_SMALL neighbours: the wave sorts such a list together, one node after the other (an insertion
            // sort by the node's own thread is ~d^2 / 4 dependent steps in global memory: 107 us per level on the COLLAB-like set)
            const int lane = tid & 63;
            u64 todo = __ballot(v < v1 && dv > 16 && dv <= deg_small);
            while (todo) {
                const int src = (int)__builtin_ctzll(todo);
                todo &= todo - 1ull;
                const i32 ns = __shfl(s, src, 64);
                const int nd = __shfl(dv, src, 64);
                u64 part = wave_sum_u64(wave_node_signature<1>(col_idx, lab_prev, nbr_sorted, ns, nd, lane, seed));
                const i64 node = v0 + (tid & ~63) + src;
                if (lane == 0) hash[node] = mix64(sig_head((u32)lab_prev[node], (u32)nd, seed) + part) & mask;
            }
            return;
        }
    }
    // coalesced stream over the chunk's col_idx; the label gather hits L2 (4 B x V table)
    for (int i = tid; i < cnt; i += SIG_THREADS) {
        i32 l = lab_prev[col_idx[e0 + i]];
        if (use_lds) buf[i] = l;
        else nbr_sorted[e0 + i] = l;
    }
    __syncthreads();
    // largest degree of the wave: up to 16 neighbours per node are sorted in registers (a fixed network: no
    // dependent LDS round trip per insertion step, no divergence); the multiset hash is a sum, so the order in
    // which the elements are added does not matter
    int dwave = 0;
    {
        const int dv = v < v1 ? row_ptr[v + 1] - row_ptr[v] : 0;
        dwave = dv;
        for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(dwave, off, 64); dwave = o > dwave ? o : dwave; }
    }
    if (v < v1) {
        const i32 s = row_ptr[v];
        const int d = row_ptr[v + 1] - s;
        if (d <= deg_small) {
            u64 acc = sig_head((u32)lab_prev[v], (u32)d, seed);
            if (use_lds && dwave <= 16 && sig_regs) {
                i32* x = buf + (s - e0);
                i32 r[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) r[k] = (k < d && (k < 8 || dwave > 8)) ? x[k] : 0x7fffffff;
                if (dwave <= 8) sort_regs<8>(r);
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
            hash[v] = mix64(acc) & mask;
        }
    }
    __syncthreads();
    if (use_lds)
        for (int i = tid; i < cnt; i += SIG_THREADS) nbr_sorted[e0 + i] = buf[i];
}

// Normalised bitonic network (every comparator puts the minimum at the lower index), so a
// length that is not a power of two needs no padding: virtual +inf elements never move.
template <typename P>
__device__ __forceinline__ void block_bitonic_sort(P x, int n) {
    for (int k = 2; (k >> 1) < n; k <<= 1) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            int l = i ^ (k - 1);
            if (l > i && l < n) {
                i32 a = x[i], b = x[l];
                if (a > b) { x[i] = b; x[l] = a; }
            }
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                int l = i ^ j;
                if (l > i && l < n) {
                    i32 a = x[i], b = x[l];
                    if (a > b) { x[i] = b; x[l] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(256) void wl_signature_wave_kernel(
    const i32* __restrict__ big_nodes, i64 n_big, const i32* __restrict__ row_ptr,
    const i32* __restrict__ col_idx, const i32* __restrict__ lab_prev,
    i32* __restrict__ nbr_sorted, u64* __restrict__ hash, u64 seed, u64 mask, const unsigned char* __restrict__ shared_prev) {
    const i64 w = ((i64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n_big) return;
    const i32 v = big_nodes[w];
    const i32 e0 = row_ptr[v];
    const int d = row_ptr[v + 1] - e0;
    if (d > WAVE_DEG_MAX) return;                 // the workgroup kernel's
    if (shared_prev && !shared_prev[v]) {         // alone in its class already: SIG_FROZEN_KEY
        if (lane == 0) hash[v] = SIG_FROZEN_KEY(lab_prev[v], seed) & mask;
        return;
    }
    u64 part;
    if (d <= 64) part = wave_node_signature<1>(col_idx, lab_prev, nbr_sorted, e0, d, lane, seed);
    else if (d <= 128) part = wave_node_signature<2>(col_idx, lab_prev, nbr_sorted, e0, d, lane, seed);
    else if (d <= 256) part = wave_node_signature<4>(col_idx, lab_prev, nbr_sorted, e0, d, lane, seed);
    else if (d <= 512) part = wave_node_signature<8>(col_idx, lab_prev, nbr_sorted, e0, d, lane, seed);
    else part = wave_node_signature<16>(col_idx, lab_prev, nbr_sorted, e0, d, lane, seed);      // (32 or 64 registers per lane:
    // the fully unrolled network no longer compiles to registers -- 272 B of scratch per lane, tried; hubs beyond 1024
    // neighbours keep the workgroup kernel, now 1024 threads wide)
    part = wave_sum_u64(part);
    if (lane == 0) hash[v] = mix64(sig_head((u32)lab_prev[v], (u32)d, seed) + part) & mask;
}

// the verifier's half for the same nodes: a wave compares the node's sorted list with its class representative's,
// 64 entries per step (verify_kernel walks a list with ONE thread: 200 us per level on the COLLAB-like batch)
#define VB_PER_WAVE 4
__global__ __launch_bounds__(256) void verify_big_kernel(const i32* __restrict__ big_nodes, i64 n_big, const i32* __restrict__ row_ptr,
                                                         const i32* __restrict__ lab_prev, const i32* __restrict__ nbr_sorted,
                                                         const i32* __restrict__ lab, const i32* __restrict__ rep,
                                                         u32* __restrict__ unresolved, const unsigned char* __restrict__ shared) {
    // Round 6: FOUR listed vertices per wave.  One vertex per wave was a chain of four dependent round trips (vertex -> class ->
    // representative -> its row -> the lists) per wave with nothing to overlap them: 108 us per level on the COLLAB-like batch
    // (360 k vertices, 44 rounds of waves).  Lanes 0-3 walk the four chains side by side, the wave then compares the lists
    // one vertex after the other with the first 64 entries of all of them already in flight.
    const i64 w0 = (((i64)blockIdx.x * 256 + threadIdx.x) >> 6) * VB_PER_WAVE;
    const int lane = threadIdx.x & 63;
    if (w0 >= n_big) return;
    i32 hv = -1, hr = -1, hs = 0, hsr = 0, hd = 0;
    bool hok = true;
    if (lane < VB_PER_WAVE && w0 + lane < n_big) {
        const i32 v = big_nodes[w0 + lane];
        if (!(shared && !shared[v])) {                 // a singleton is its own representative (rep[] has no entry for it)
            const i32 r = rep[lab[v] & 0x7fffffff];
            if (r != v) {
                hv = v, hr = r, hs = row_ptr[v], hsr = row_ptr[r], hd = row_ptr[v + 1] - hs;
                hok = lab_prev[v] == lab_prev[r] && hd == row_ptr[r + 1] - hsr;
            }
        }
    }
    bool bad = false;
    i32 a0[VB_PER_WAVE], b0[VB_PER_WAVE];
#pragma unroll
    for (int q = 0; q < VB_PER_WAVE; ++q) {            // the first 64 entries of every pair of lists: eight loads in flight
        const i32 v = __shfl(hv, q, 64);
        const int d = __shfl(hd, q, 64);
        const i32 s = __shfl(hs, q, 64), sr = __shfl(hsr, q, 64);
        const bool ok = __shfl((int)hok, q, 64) != 0;
        a0[q] = b0[q] = 0;
        if (v >= 0 && ok && lane < d) a0[q] = nbr_sorted[s + lane], b0[q] = nbr_sorted[sr + lane];
    }
#pragma unroll
    for (int q = 0; q < VB_PER_WAVE; ++q) {
        const i32 v = __shfl(hv, q, 64);
        if (v < 0) continue;                           // (wave-uniform)
        const int d = __shfl(hd, q, 64);
        const i32 s = __shfl(hs, q, 64), sr = __shfl(hsr, q, 64);
        bool ok = __shfl((int)hok, q, 64) != 0;
        if (ok) {
            if (a0[q] != b0[q]) ok = false;
            for (int k = lane + 64; k < d && ok; k += 64)
                if (nbr_sorted[s + k] != nbr_sorted[sr + k]) ok = false;
        }
        if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) bad = true;       // one count per vertex, as before
        if (bad && lane == 0) { atomicAdd(unresolved, 1u); }
        bad = false;
    }
}

__global__ __launch_bounds__(BIG_THREADS) void wl_signature_big_kernel(
    const i32* __restrict__ big_nodes, const i32* __restrict__ row_ptr,
    const i32* __restrict__ col_idx, const i32* __restrict__ lab_prev,
    i32* __restrict__ nbr_sorted, u64* __restrict__ hash, u64 seed, u64 mask, int wave_done,
    const unsigned char* __restrict__ shared_prev) {
    __shared__ i32 buf[BIG_LDS_CAP];
    __shared__ u64 red[BIG_THREADS / 64];
    const int tid = threadIdx.x;
    const i32 v = big_nodes[blockIdx.x];
    const i32 e0 = row_ptr[v];
    const int d = row_ptr[v + 1] - e0;
    if (wave_done && d <= WAVE_DEG_MAX) return;        // wl_signature_wave_kernel's
    if (shared_prev && !shared_prev[v]) {              // alone in its class already: SIG_FROZEN_KEY (workgroup-uniform)
        if (tid == 0) hash[v] = SIG_FROZEN_KEY(lab_prev[v], seed) & mask;
        return;
    }
    u64 part = 0;
    if (d <= BIG_LDS_CAP) {
        for (int i = tid; i < d; i += BIG_THREADS) {
            i32 l = lab_prev[col_idx[e0 + i]];
            buf[i] = l;
            part += sig_elem((u32)l, seed);
        }
        __syncthreads();
        block_bitonic_sort(buf, d);
        for (int i = tid; i < d; i += BIG_THREADS) nbr_sorted[e0 + i] = buf[i];
    } else {   // hub larger than LDS: same network directly on the global scratch
        for (int i = tid; i < d; i += BIG_THREADS) {
            i32 l = lab_prev[col_idx[e0 + i]];
            nbr_sorted[e0 + i] = l;
            part += sig_elem((u32)l, seed);
        }
        __syncthreads();
        block_bitonic_sort(nbr_sorted + e0, d);
    }
    // block reduction of the (order independent) multiset sum
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        u64 t = 0;
        for (int i = 0; i < BIG_THREADS / 64; ++i) t += red[i];
        hash[v] = mix64(sig_head((u32)lab_prev[v], (u32)d, seed) + t) & mask;
    }
}


__global__ void labels_to_keys_kernel(const i32* __restrict__ lab, u64* __restrict__ keys, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = (u64)(u32)lab[i];
}

// exact-mode refinement key: (label of the previous round, re-seeded hash)
__global__ void refine_keys_kernel(const i32* __restrict__ lab_round, const u64* __restrict__ hash,
                                   u64* __restrict__ keys, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ((u64)(u32)lab_round[i] << 32) | (hash[i] & 0xffffffffull);
}

// run heads of the sorted keys -> dense label ids, fused into the prefix sum (scan_fn.h)
struct HeadAssign {
    const u64* ks;     // sorted keys
    const u32* perm;   // node at each sorted position
    i32* lab;          // out: lab[node] = run index
    i32* rep;          // out: rep[run]  = first node of the run
    u32* frozen;       // out (may be null): 1 when the node's class is a singleton
    i64 n;
    u32 base;              // ids start at base + *base_dev (active-set levels: behind the frozen and carried ids)
    const u32* base_dev;   // may be null
    __device__ __forceinline__ u32 value(i64 k) const { return (k == 0 || ks[k] != ks[k - 1]) ? 1u : 0u; }
    __device__ __forceinline__ void emit(i64 k, u32 head, u32 incl) const {
        const u32 v = perm[k];
        const i32 r = (i32)incl - 1;
        lab[v] = (i32)(base + (base_dev ? *base_dev : 0u)) + r;
        if (head) rep[r] = (i32)v;
        if (frozen) frozen[v] = (head && (k == n - 1 || ks[k + 1] != ks[k])) ? 1u : 0u;
    }
    __device__ __forceinline__ void finish(u32) const {}
    __device__ __forceinline__ i64 seg_first_tile(i64) const { return 0; }
};

// The same for a level that sorts EVERY node, with the label-grouped order split on the way: nodes
// of classes with two or more members first (still grouped by label, ascending node inside a group),
// the singletons behind them from the end.  Only the first *ns_out positions can share a label, so
// the label-count features of the level (features.hip) read those and nothing else -- at the
// third level of a 1 M-node job that is 5 % of the nodes.  Packed scan: heads low, listed nodes high.
struct HeadAssignSplit {
    const u64* ks;       // sorted keys
    const u32* sorted;   // node at each sorted position
    i32* lab; i32* rep; u32* frozen;
    i32* perm_out;       // out: [nodes of shared classes | singletons]
    u32* ns_out;         // out: number of nodes of shared classes
    u32* count_out;      // out: number of classes
    i64 n;
    u32* mbox; u32 seq;  // host mailbox (may be null): the host learns {listed nodes, *extra} without a scan of its own
    const u32* extra;    // largest top-digit bucket of the sort that produced ks
    unsigned char* shared_out;   // out (may be null): 1 when the node's class has two or more members (features_gm.hip)
    __device__ __forceinline__ bool head(i64 k) const { return k == 0 || ks[k] != ks[k - 1]; }
    __device__ __forceinline__ u64 value(i64 k) const {
        const bool h = head(k);
        const bool single = h && (k == n - 1 || ks[k + 1] != ks[k]);
        return (u64)(h ? 1u : 0u) | ((u64)(single ? 0u : 1u) << 32);
    }
    __device__ __forceinline__ void emit(i64 k, u64 val, u64 incl) const {
        const u32 v = sorted[k];
        const bool h = (val & 1ull) != 0, single = (val >> 32) == 0;
        const i32 r = (i32)(u32)(incl & 0xffffffffull) - 1;
        const i64 listed = (i64)(incl >> 32);            // listed nodes up to and including k
        lab[v] = r;
        if (h) rep[r] = (i32)v;
        if (frozen) frozen[v] = single ? 1u : 0u;
        if (shared_out) shared_out[v] = single ? 0 : 1;
        if (single) perm_out[n - 1 - (k - listed)] = (i32)v;
        else perm_out[listed - 1] = (i32)v;
    }
    __device__ __forceinline__ void finish(u64 total) const {
        *ns_out = (u32)(total >> 32);
        *count_out = (u32)(total & 0xffffffffull);
        if (mbox) {
            __hip_atomic_store(&mbox[1], (u32)(total >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mbox[2], extra ? *extra : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mbox[3], (u32)(total & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // labels of the level
            __threadfence_system();
            __hip_atomic_store(&mbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __device__ __forceinline__ i64 seg_first_tile(i64) const { return 0; }
};

// A node whose class is a singleton stays a singleton at every later level (classes only
// split), so it needs no signature, no sort and no verification any more: it just receives a
// fresh id.  ActiveScan compacts the still-active nodes (ascending) and numbers the frozen.
struct ActiveScan {
    const u32* frozen; u32* act; u32* fidx;
    u32* total_out;            // device copy of the active count
    const u32* extra;          // one more word to report (largest top-digit bucket of the last sort)
    u32* mbox; u32 seq;        // host mailbox (null: the host reads total_out / extra back itself)
    const i32* iso_info;       // gk_batch::iso_info (null: no isolated vertices): those are carried, never active
    const unsigned char* shared;   // not null: the previous level left "class of two or more" bytes instead of frozen[]
    __device__ __forceinline__ u32 value(i64 v) const {
        const bool fr = shared ? !shared[v] : frozen[v] != 0u;
        return (fr || (iso_info && iso_info[v] < 0)) ? 0u : 1u;
    }
    __device__ __forceinline__ void emit(i64 v, u32 a, u32 incl) const {
        if (a) { act[incl - 1] = (u32)v; fidx[v] = 0xffffffffu; return; }
        const i32 info = iso_info ? iso_info[v] : 0;
        if (info < 0) fidx[v] = 0x80000000u | (u32)(-1 - info);   // carried: slot in the carried list
        else fidx[v] = (u32)v - incl - (u32)info;                  // rank among the frozen nodes
    }
    // the totals are final before any emit ran, so the host can be told right away
    __device__ __forceinline__ void finish(u32 total) const {
        *total_out = total;
        if (mbox) {
            __hip_atomic_store(&mbox[1], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mbox[2], *extra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(&mbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __device__ __forceinline__ i64 seg_first_tile(i64) const { return 0; }
};


// signature of the listed (active, degree <= WL_DEG_SMALL) nodes: one thread per node, the
// neighbour list is gathered and sorted in the global scratch (few nodes: not worth staging)
__global__ void wl_signature_list_kernel(const u32* __restrict__ act, i64 n_act,
                                         const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx,
                                         const i32* __restrict__ lab_prev, i32* __restrict__ nbr_sorted,
                                         u64* __restrict__ hash_out, u64 seed, u64 mask, int deg_small) {
    i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_act) return;
    const u32 v = act[j];
    const i32 s = row_ptr[v];
    const int d = row_ptr[v + 1] - s;
    if (d > deg_small) return;                  // the wave / workgroup kernels' (they write hash_node[v])
    i32* x = nbr_sorted + s;
    u64 acc;
    if (d <= 16) acc = node_key_regs(col_idx, lab_prev, x, s, d, (u32)lab_prev[v], seed);
    else {
        for (int k = 0; k < d; ++k) x[k] = lab_prev[col_idx[s + k]];
        insertion_sort(x, d);
        acc = sig_head((u32)lab_prev[v], (u32)d, seed);
        for (int k = 0; k < d; ++k) acc += sig_elem((u32)x[k], seed);
    }
    hash_out[j] = mix64(acc) & mask;
}

// Few labels and small degrees (level 1 of a job with a handful of input labels): the signature
// (own label, multiset of neighbour labels) has an EXACT integer code
//     own * R^L + sum over neighbours of R^label,   R = max degree + 1, L = number of labels
// (the sum is the mixed-radix number of the label counts: no digit reaches R, so no carries).
// When L * R^L < 2^32 the code, scrambled by a bijection of the 32-bit integers so that the sort's
// top digit stays balanced, replaces the 48-bit hash: 4 digit passes instead of 6, no sorted
// neighbour lists, and nothing to verify -- equal keys ARE equal signatures.
__global__ __launch_bounds__(256) void wl_signature_exact_kernel(
    const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx, const i32* __restrict__ lab_prev,
    u64* __restrict__ hash, i64 n, int L, u64 R, u32* __restrict__ unresolved) {
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
    const u32 own = (u32)lab_prev[v];
    bool in_range = own < (u32)L;
    u64 key = (u64)own * pw[L];
    for (i32 k = s; k < e; ++k) {
        const u32 l = (u32)lab_prev[col_idx[k]];
        in_range = in_range && l < (u32)L;
        key += pw[l < (u32)L ? l : 0];
    }
    if (!in_range) atomicAdd(unresolved, 1u);      // ids beyond the declared label count: redo with hashes
    u32 x = (u32)key;                    // bijective scramble (every step is invertible mod 2^32)
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    hash[v] = (u64)x;
}

// hubs write their hash indexed by node: move it to the active-list slot
__global__ void gather_big_hash_kernel(const u32* __restrict__ act, i64 n_act, const i32* __restrict__ row_ptr,
                                       const u64* __restrict__ hash_node, u64* __restrict__ hash_out, int deg_small) {
    i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_act) return;
    const u32 v = act[j];
    if (row_ptr[v + 1] - row_ptr[v] > deg_small) hash_out[j] = hash_node[v];
}


// The active list of a level from the active list of the level before (both ascending node ids): a
// node of the old list either stays active or has just become a singleton.  Frozen nodes keep their
// id for good -- ids of an active-set level are [frozen | carried classes | active classes], the frozen
// ones numbered in the order they froze -- so a level starts as a copy of the previous level's labels
// and the nodes that froze last get the next free frozen ids here.  O(previous active nodes), where
// ActiveScan + frozen_assign_verify_kernel walk all nodes.
struct ActiveFromList {
    const u32* frozen; const u32* act_prev; u32* act_new;
    i32* lab;                  // this level's labels (already a copy of the previous level's)
    u32 n_frozen_prev;         // frozen nodes of the previous level = first free frozen id
    u32* total_out; const u32* extra; u32* mbox; u32 seq;     // as in ActiveScan
    __device__ __forceinline__ u32 value(i64 i) const { return frozen[act_prev[i]] ? 0u : 1u; }
    __device__ __forceinline__ void emit(i64 i, u32 a, u32 incl) const {
        const u32 v = act_prev[i];
        if (a) act_new[incl - 1] = v;
        else lab[v] = (i32)(n_frozen_prev + ((u32)i - incl));
    }
    __device__ __forceinline__ void finish(u32 total) const {
        *total_out = total;
        if (mbox) {
            __hip_atomic_store(&mbox[1], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mbox[2], *extra, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(&mbox[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __device__ __forceinline__ i64 seg_first_tile(i64) const { return 0; }
};

// ... and the end of such a level: carried classes get their ids and their place behind the sorted
// active nodes, active nodes are verified against their class representative.  O(active + carried).
__global__ void active_finish_kernel(const u32* __restrict__ act, u32 n_active,
                                     const i32* __restrict__ car_nodes, const i32* __restrict__ car_class, u32 n_car,
                                     u32 n_frozen, const u32* __restrict__ n_car_classes_dev,
                                     const u32* __restrict__ ra_dev, i32* __restrict__ lab, i32* __restrict__ perm,
                                     u32* __restrict__ count_out, const i32* __restrict__ row_ptr,
                                     const i32* __restrict__ lab_prev, const i32* __restrict__ nbr_sorted,
                                     const i32* __restrict__ rep, u32* __restrict__ unresolved) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 n_cc = n_car ? *n_car_classes_dev : 0u;
    if (j == 0) *count_out = n_frozen + n_cc + *ra_dev;
    if (j < n_car) {
        const i32 v = car_nodes[j];
        lab[v] = (i32)(n_frozen + (u32)car_class[j]);
        perm[n_active + j] = v;
    }
    if (j >= n_active) return;
    const i32 v = (i32)act[j];
    const i32 r = rep[lab[v] - (i32)(n_frozen + n_cc)];
    if (r == v) return;
    bool ok = lab_prev[v] == lab_prev[r];
    const i32 s = row_ptr[v], sr = row_ptr[r];
    const int d = row_ptr[v + 1] - s;
    ok = ok && (d == row_ptr[r + 1] - sr);
    if (ok)
        for (int k = 0; k < d; ++k)
            if (nbr_sorted[s + k] != nbr_sorted[sr + k]) { ok = false; break; }
    if (!ok) atomicAdd(unresolved, 1u);
}

// A whole active-set level in ONE workgroup, for the tail of a job where a few hundred nodes are still
// active: eight launches of 3-5 us kernels plus a host read-back cost ~55 us per level, this kernel
// ~15 us and no read-back (the host learns the active counts with the final read-back of the job).
//   do_scan: the active list is first derived from the previous level's list (ActiveFromList above),
//            otherwise act_cur / *n_act_io already describe this level.
//   then   : signatures of the active nodes (as wl_signature_list_kernel), (key, list position) ranked
//            against each other in LDS (a strict total order: stable), run heads -> class ids behind
//            the frozen and carried ids, singleton flags for the next level, verification against the
//            class representative, carried classes (as active_finish_kernel).
#define TINY_MAX 1024
__device__ __forceinline__ u32 tiny_block_scan(u32 x, u32* wsum, u32* total) {     // inclusive, 1024 threads
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 inc = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(inc, off, 64);
        if (lane >= off) inc += y;
    }
    __syncthreads();                      // wsum may still be read from the previous scan
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    u32 before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < TINY_MAX / 64; ++q) {
        const u32 t = wsum[q];
        if (q < w) before += t;
        all += t;
    }
    *total = all;
    return inc + before;
}

__global__ __launch_bounds__(TINY_MAX) void wl_tiny_level_kernel(
    const u32* __restrict__ act_prev, u32* __restrict__ act_cur, u32* __restrict__ n_act_io, int do_scan,
    u32* __restrict__ frozen, const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx,
    const i32* __restrict__ lab_prev, i32* __restrict__ nbr_sorted, i32* __restrict__ lab, i32* __restrict__ perm,
    u64 seed, u64 mask, const i32* __restrict__ car_nodes, const i32* __restrict__ car_class, u32 n_car,
    const u32* __restrict__ n_car_classes_dev, i64 V, u32* __restrict__ count_out, u32* __restrict__ n_act_out,
    u32* __restrict__ unresolved) {
    __shared__ u32 act_s[TINY_MAX];
    __shared__ __attribute__((aligned(16))) u64 elem_s[TINY_MAX];
    __shared__ u64 skey[TINY_MAX];
    __shared__ u64 korig[TINY_MAX];
    __shared__ u32 sidx[TINY_MAX];
    __shared__ i32 rep_s[TINY_MAX];
    __shared__ u32 wsum[TINY_MAX / 64];
    const u32 j = threadIdx.x;
    if (blockIdx.x > 0) {
        // the carried classes of the isolated vertices (thousands of scattered stores: 6.5 us when the level's one
        // workgroup does them itself) on further workgroups.  Only launched when the active list is taken as it is
        // (do_scan == 0): *n_act_io is then rewritten with the same value by workgroup 0
        const u32 n_act0 = *n_act_io;
        const u32 n_frozen0 = (u32)(V - (i64)n_act0 - (i64)n_car);
        for (u32 c = (blockIdx.x - 1u) * TINY_MAX + j; c < n_car; c += (gridDim.x - 1u) * TINY_MAX) {
            const i32 cv = car_nodes[c];
            lab[cv] = (i32)(n_frozen0 + (u32)car_class[c]);
            perm[n_act0 + c] = cv;
        }
        return;
    }
    const u32 n_in = *n_act_io;                       // every thread reads it before thread 0 overwrites it below
    u32 n_act = n_in;
    if (do_scan) {
        u32 a = 0, v = 0;
        if (j < n_in) { v = act_prev[j]; a = frozen[v] ? 0u : 1u; }
        const u32 incl = tiny_block_scan(a, wsum, &n_act);
        if (j < n_in) {
            if (a) { act_s[incl - 1] = v; act_cur[incl - 1] = v; }
            else lab[v] = (i32)((u32)(V - (i64)n_in - (i64)n_car) + (j - incl));      // next free frozen ids
        }
    } else if (j < n_in) {
        act_s[j] = act_cur[j];
    }
    __syncthreads();
    const u32 n_cc = n_car ? *n_car_classes_dev : 0u;
    const u32 n_frozen = (u32)(V - (i64)n_act - (i64)n_car);
    const u32 base = n_frozen + n_cc;
    // ---- signatures
    u64 key = ~0ull;
    if (j < n_act) {
        const u32 v = act_s[j];
        const i32 s = row_ptr[v];
        const int d = row_ptr[v + 1] - s;
        i32* x = nbr_sorted + s;
        u64 acc;
        if (d <= 16) acc = node_key_regs(col_idx, lab_prev, x, s, d, (u32)lab_prev[v], seed);
        else {
            for (int k = 0; k < d; ++k) x[k] = lab_prev[col_idx[s + k]];
            insertion_sort(x, d);
            acc = sig_head((u32)lab_prev[v], (u32)d, seed);
            for (int k = 0; k < d; ++k) acc += sig_elem((u32)x[k], seed);
        }
        key = mix64(acc) & mask;
    }
    // ---- sort the (key, list position) elements (distinct; the padding ~0 sorts last): bitonic network, one
    // element per thread in a register -- partners closer than 64 exchange by wave shuffle, the 10 farther steps
    // (at 1024 elements) through two alternating LDS buffers, one barrier each.  (The same network entirely in LDS:
    // 55 barriers, 10.7 us of the kernel's 26; s_memtime stamps.)
    u64 e = j < n_act ? ((key << 10) | (u64)j) : ~0ull;
    korig[j] = key;
    u32 p2 = 2;
    while (p2 < n_act) p2 <<= 1;
    int flip = 0;
    for (u32 k = 2; k <= p2; k <<= 1)
        for (u32 jj = k >> 1; jj > 0; jj >>= 1) {
            u64 o;
            if (jj >= 64) {
                u64* buf = flip ? skey : elem_s;
                flip ^= 1;
                buf[j] = e;
                __syncthreads();
                o = buf[j ^ jj];
            } else {
                o = __shfl_xor(e, (int)jj, 64);
            }
            const bool keep_min = ((j & jj) == 0u) == ((j & k) == 0u);
            e = keep_min ? (e < o ? e : o) : (e < o ? o : e);
        }
    __syncthreads();                    // skey may still be read as an exchange buffer; korig is complete
    if (j < n_act) {
        const u32 src = (u32)(e & 1023u);
        skey[j] = korig[src];
        sidx[j] = src;
    }
    __syncthreads();
    // ---- run heads -> class ids, singleton flags, representatives
    bool head = false, single = false;
    if (j < n_act) {
        head = j == 0 || skey[j] != skey[j - 1];
        single = head && (j + 1 == n_act || skey[j + 1] != skey[j]);
    }
    u32 ra = 0;
    const u32 incl = tiny_block_scan(head ? 1u : 0u, wsum, &ra);
    i32 v = 0;
    if (j < n_act) {
        v = (i32)act_s[sidx[j]];
        lab[v] = (i32)(base + incl - 1u);
        perm[j] = v;
        frozen[v] = single ? 1u : 0u;
        if (head) rep_s[incl - 1] = v;
    }
    __syncthreads();
    if (j < n_act && !head) {           // same full signature as the class representative?
        const i32 r = rep_s[incl - 1];
        bool ok = lab_prev[v] == lab_prev[r];
        const i32 s = row_ptr[v], sr = row_ptr[r];
        const int d = row_ptr[v + 1] - s;
        ok = ok && (d == row_ptr[r + 1] - sr);
        if (ok && d <= 16) {                // all loads in flight together (an early exit would chain them)
            bool same = true;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < d) same = same && nbr_sorted[s + k] == nbr_sorted[sr + k];
            ok = same;
        } else if (ok)
            for (int k = 0; k < d; ++k)
                if (nbr_sorted[s + k] != nbr_sorted[sr + k]) { ok = false; break; }
        if (!ok) atomicAdd(unresolved, 1u);
    }
    if (gridDim.x == 1)
        for (u32 c = j; c < n_car; c += TINY_MAX) {
            const i32 cv = car_nodes[c];
            lab[cv] = (i32)(n_frozen + (u32)car_class[c]);
            perm[n_act + c] = cv;
        }
    if (j == 0) {
        *count_out = base + ra;
        *n_act_io = n_act;
        *n_act_out = n_act;
    }
}

// FIRST active-set level (the level before sorted every node), one pass over the nodes: a frozen node
// receives its permanent id (its rank among the frozen nodes); a carried (isolated) node keeps its
// class, listed right behind the sorted active nodes in carried-list order; an active node is verified
// against its class representative.  ids = [frozen | carried classes | active classes],
// perm = [sorted active | carried | (unlisted: nobody reads behind n_sorted)].
__global__ void frozen_assign_verify_kernel(const u32* __restrict__ fidx, const u32* __restrict__ ra_dev,
                                            i32* __restrict__ lab, i32* __restrict__ perm,
                                            u32* __restrict__ count_out, u32 n_active, i64 n,
                                            u32 n_car, const u32* __restrict__ n_car_classes_dev,
                                            const i32* __restrict__ car_class,
                                            const i32* __restrict__ row_ptr,
                                            const i32* __restrict__ lab_prev, const i32* __restrict__ nbr_sorted,
                                            const i32* __restrict__ rep, u32* __restrict__ unresolved) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 ra = *ra_dev;
    const u32 n_car_classes = n_car ? *n_car_classes_dev : 0u;
    const u32 n_frozen = (u32)(n - n_active - n_car);
    if (v == 0) *count_out = n_frozen + n_car_classes + ra;
    if (v >= n) return;
    const u32 f = fidx[v];
    if (f != 0xffffffffu) {
        if (f & 0x80000000u) {
            const u32 slot = f & 0x7fffffffu;
            lab[v] = (i32)(n_frozen + (u32)car_class[slot]);
            perm[n_active + slot] = (i32)v;
        } else {
            lab[v] = (i32)f;               // frozen ids come first and never change again (ActiveFromList)
        }
        return;
    }
    const i32 r = rep[lab[v] - (i32)(n_frozen + n_car_classes)];
    if (r == (i32)v) return;
    bool ok = lab_prev[v] == lab_prev[r];
    const i32 s = row_ptr[v], sr = row_ptr[r];
    const int d = row_ptr[v + 1] - s;
    ok = ok && (d == row_ptr[r + 1] - sr);
    if (ok)
        for (int k = 0; k < d; ++k)
            if (nbr_sorted[s + k] != nbr_sorted[sr + k]) { ok = false; break; }
    if (!ok) atomicAdd(unresolved, 1u);
}

// shared_out (may be null): the dictionary left "singleton class" in bit 31 of the label word (gk_bucket_dictionary with
// flag_in_lab) -- this pass walks the nodes in order anyway: it strips the bit, writes the level's "class of two or
// more" bytes coalesced, and skips the representative look-up for singletons (they are their own representative;
// rep[] holds no entry for them)
__global__ void verify_kernel(const i32* __restrict__ row_ptr, const i32* __restrict__ lab_prev,
                              const i32* __restrict__ nbr_sorted, i32* __restrict__ lab,
                              const i32* __restrict__ rep, u32* __restrict__ unresolved, i64 n,
                              unsigned char* __restrict__ shared_out, int skip_big) {
    i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    i32 l = lab[v];
    if (shared_out) {
        shared_out[v] = l < 0 ? 0 : 1;
        if (l < 0) {
            lab[v] = l & 0x7fffffff;
            return;
        }
    }
    i32 s = row_ptr[v];
    int d = row_ptr[v + 1] - s;
    if (skip_big && d > skip_big) return;              // verify_big_kernel's (a wave per node); skip_big = the batch's threshold
    const i32 r = rep[l];
    if (r == (i32)v) return;
    bool ok = lab_prev[v] == lab_prev[r];
    i32 sr = row_ptr[r];
    ok = ok && (d == row_ptr[r + 1] - sr);
    if (ok)
        for (int k = 0; k < d; ++k)
            if (nbr_sorted[s + k] != nbr_sorted[sr + k]) { ok = false; break; }
    if (!ok) atomicAdd(unresolved, 1u);
}

__global__ void node_graph_kernel(const i32* __restrict__ graph_ptr, i32* __restrict__ node_graph,
                                  i64 n_graphs, i64 n_nodes) {
    i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_nodes) return;
    i64 lo = 0, hi = n_graphs;   // largest g with graph_ptr[g] <= v
    while (hi - lo > 1) {
        i64 mid = (lo + hi) >> 1;
        if (graph_ptr[mid] <= v) lo = mid; else hi = mid;
    }
    node_graph[v] = (i32)lo;
}

// per-block maxima (largest graph, largest degree) into part[2*block..]: thousands of blocks
// hammering one counter -- even just reading it -- serialise on a single L2 channel
__global__ __launch_bounds__(256) void batch_stats_kernel(const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr,
                                   i64 n_graphs, i64 n_nodes, u32* __restrict__ big_flag,
                                   u32* __restrict__ iso_flag, i32* __restrict__ part) {
    __shared__ int sg[4], sd[4];
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    int gn = 0, d = 0;
    if (i < n_graphs) gn = graph_ptr[i + 1] - graph_ptr[i];
    if (i < n_nodes) {
        d = row_ptr[i + 1] - row_ptr[i];
        big_flag[i] = d > WL_DEG_SMALL ? 1u : 0u;
        iso_flag[i] = d == 0 ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_down(gn, off, 64); gn = o > gn ? o : gn;
        o = __shfl_down(d, off, 64); d = o > d ? o : d;
    }
    if ((threadIdx.x & 63) == 0) sg[threadIdx.x >> 6] = gn, sd[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 4; ++q) { gn = sg[q] > gn ? sg[q] : gn; d = sd[q] > d ? sd[q] : d; }
        part[2 * blockIdx.x] = gn, part[2 * blockIdx.x + 1] = d;
    }
}

// Input validation (the C ABI does not trust its caller: a malformed CSR must become GK_ERR_ARG, not an
// out-of-bounds gather in the signature kernels) and the set of level-0 label ids that occur.
//   err bits: 1 graph_ptr not a monotone cover of [0, n_nodes], 2 row_ptr not a monotone cover of
//   [0, n_edges], 4 label id outside [0, n_labels0), 8 neighbour outside the node range of its graph
__global__ void batch_check_kernel(const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr,
                                   const i32* __restrict__ col_idx, const i32* __restrict__ node_graph,
                                   const i32* __restrict__ labels, i64 n_graphs, i64 n_nodes, i64 n_edges,
                                   i32 n_labels0, u32* __restrict__ pres, u32* __restrict__ err) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 e = 0;
    bool long_row = false;
    i64 row0 = 0, row1 = 0;
    i32 lo = 0, hi = 0;
    if (i < n_graphs) {
        const i64 a = graph_ptr[i], b = graph_ptr[i + 1];
        if (a < 0 || a > b || b > n_nodes || (i == 0 && a != 0) || (i == n_graphs - 1 && b != n_nodes)) e |= 1u;
    }
    if (i < n_nodes) {
        const i64 r0 = row_ptr[i], r1 = row_ptr[i + 1];
        if (r0 < 0 || r0 > r1 || r1 > n_edges || (i == 0 && r0 != 0) || (i == n_nodes - 1 && r1 != n_edges)) e |= 2u;
        const i32 l = labels[i];
        if (l < 0 || l >= n_labels0) e |= 4u;
        else if (pres && !pres[l]) pres[l] = 1u;          // same value from every writer
        if (!(e & 2u)) {
            const i32 g = node_graph[i];
            lo = graph_ptr[g], hi = graph_ptr[g + 1];
            if (r1 - r0 <= 64) {
                for (i64 k = r0; k < r1; ++k) {
                    const i32 c = col_idx[k];
                    if (c < lo || c >= hi) { e |= 8u; break; }
                }
            } else long_row = true, row0 = r0, row1 = r1;
        }
    }
    // rows of more than 64 entries (hubs, ego networks): the wave walks them together, 64 entries per step -- a thread per
    // vertex made a 2 500-neighbour hub one thread's loop and the whole check 0.3 ms on the REDDIT- / COLLAB-like sets
    {
        const int lane = threadIdx.x & 63;
        u64 todo = __ballot(long_row);
        while (todo) {
            const int src = (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const i64 a = __shfl(row0, src, 64), b = __shfl(row1, src, 64);
            const i32 glo = __shfl(lo, src, 64), ghi = __shfl(hi, src, 64);
            bool bad = false;
            for (i64 k = a + lane; k < b; k += 64) {
                const i32 c = col_idx[k];
                bad = bad || c < glo || c >= ghi;
            }
            if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == src) e |= 8u;
        }
    }
    if (e) atomicOr(err, e);
}

// one block: stats[0] = max graph nodes, stats[1] = max degree, stats[4] = level-0 label ids that occur
__global__ __launch_bounds__(1024) void batch_stats_reduce_kernel(const i32* __restrict__ part, int nblk,
                                                                  i32* __restrict__ stats, const u32* __restrict__ pres, int n_pres) {
    __shared__ int sg[16], sd[16];
    __shared__ int spres;
    if (threadIdx.x == 0) spres = 0;
    __syncthreads();
    if (pres && (int)threadIdx.x < n_pres && pres[threadIdx.x]) atomicAdd(&spres, 1);
    int gn = 0, d = 0;
    for (int i = threadIdx.x; i < nblk; i += 1024) {
        const int a = part[2 * i], b = part[2 * i + 1];
        gn = a > gn ? a : gn, d = b > d ? b : d;
    }
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_down(gn, off, 64); gn = o > gn ? o : gn;
        o = __shfl_down(d, off, 64); d = o > d ? o : d;
    }
    if ((threadIdx.x & 63) == 0) sg[threadIdx.x >> 6] = gn, sd[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < 16; ++q) { gn = sg[q] > gn ? sg[q] : gn; d = sd[q] > d ? sd[q] : d; }
        stats[0] = gn, stats[1] = d, stats[4] = spres;
    }
}

__global__ void big_flag_kernel(const i32* __restrict__ row_ptr, i64 n, int thr, u32* __restrict__ big_flag) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) big_flag[i] = row_ptr[i + 1] - row_ptr[i] > thr ? 1u : 0u;
}

__global__ void compact_big_kernel(const u32* __restrict__ big_flag, const u32* __restrict__ excl,
                                   i32* __restrict__ big_nodes, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && big_flag[i]) big_nodes[excl[i]] = (i32)i;
}

// isolated vertices: sort key = input label, list of the vertices; everybody else learns how many
// isolated vertices precede it
__global__ void iso_keys_kernel(const u32* __restrict__ iso_flag, const u32* __restrict__ iso_excl,
                                const i32* __restrict__ labels0, u64* __restrict__ keys,
                                i32* __restrict__ iso_nodes, i32* __restrict__ iso_info, i64 n) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const u32 before = iso_excl[v];
    if (iso_flag[v]) {
        keys[before] = (u64)(u32)labels0[v];
        iso_nodes[before] = (i32)v;
    } else {
        iso_info[v] = (i32)before;
    }
}

// carried list = isolated vertices grouped by input label (stable: ascending vertex inside a group)
__global__ void iso_slots_kernel(const i32* __restrict__ order, const i32* __restrict__ cls,
                                 const i32* __restrict__ iso_nodes, i32* __restrict__ iso_info,
                                 i32* __restrict__ car_class, i32* __restrict__ car_nodes, i64 n_iso) {
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_iso) return;
    const i32 item = order[k];
    const i32 v = iso_nodes[item];
    iso_info[v] = -1 - (i32)k;
    car_class[k] = cls[item];
    car_nodes[k] = v;
}

static inline dim3 grid_for(i64 n, int t) { return dim3((unsigned)(n > 0 ? cdiv(n, t) : 1)); }
int gk_dictionary_from_keys(gk_ctx* ctx, const u64* keys, i64 n, int key_bits, i32* lab, i32* perm, u32* count_dev);
int gk_sr_rebuild_order(gk_ctx* ctx, gk_batch* b, int level);      // wl_stream.hip

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
// node -> graph map, size statistics and the list of high-degree nodes of a batch whose
// graph_ptr / row_ptr / col_idx / labels are already in place on the device
static int batch_finish(gk_ctx* ctx, gk_batch* b) {
    const i64 n_graphs = b->n_graphs, n_nodes = b->n_nodes;
    if (n_nodes > 0)
        node_graph_kernel<<<grid_for(n_nodes, 256), 256, 0, ctx->stream>>>(b->graph_ptr, b->node_graph, n_graphs, n_nodes);
    Tmp<u32> flag(ctx), excl(ctx), total(ctx), iso_flag(ctx), iso_excl(ctx);
    Tmp<i32> stats(ctx), part(ctx);
    const i64 m = n_graphs > n_nodes ? n_graphs : n_nodes;
    const int nblk = (int)cdiv(m > 0 ? m : 1, 256);
    GK_TRY(flag.alloc(n_nodes)); GK_TRY(excl.alloc(n_nodes)); GK_TRY(total.alloc(1));
    GK_TRY(iso_flag.alloc(n_nodes)); GK_TRY(iso_excl.alloc(n_nodes));
    Tmp<u32> pres(ctx);
    const bool few_labels = b->n_labels0 >= 1 && b->n_labels0 <= GK_HIST0_MAX_LABELS;
    GK_TRY(stats.alloc(8)); GK_TRY(part.alloc(2 * (size_t)nblk)); GK_TRY(pres.alloc(GK_HIST0_MAX_LABELS));
    GK_TRY(gk_zero_async(ctx, stats.p, 32));
    GK_TRY(gk_zero_async(ctx, pres.p, GK_HIST0_MAX_LABELS * 4));
    batch_check_kernel<<<dim3(nblk), 256, 0, ctx->stream>>>(b->graph_ptr, b->row_ptr, b->col_idx, b->node_graph, b->labels,
                                                             n_graphs, n_nodes, b->n_edges, b->n_labels0,
                                                             few_labels ? pres.p : nullptr, (u32*)stats.p + 5);
    batch_stats_kernel<<<dim3(nblk), 256, 0, ctx->stream>>>(b->graph_ptr, b->row_ptr, n_graphs, n_nodes, flag.p,
                                                             iso_flag.p, part.p);
    batch_stats_reduce_kernel<<<1, 1024, 0, ctx->stream>>>(part.p, nblk, stats.p, few_labels ? pres.p : nullptr,
                                                            (int)b->n_labels0);
    GK_TRY(gk_scan_u32(ctx, flag.p, excl.p, n_nodes, true, (u32*)stats.p + 2));
    GK_TRY(gk_scan_u32(ctx, iso_flag.p, iso_excl.p, n_nodes, true, (u32*)stats.p + 3));
    u32 h[6] = {0, 0, 0, 0, 0, 0};
    GK_TRY(gk_readback(ctx, (const u32*)stats.p, h, 6));
    if (h[5]) {
        gk_set_error("gk_batch_create: malformed batch (%s%s%s%s)", (h[5] & 1u) ? "graph_ptr is not a monotone cover of the nodes; " : "",
                     (h[5] & 2u) ? "row_ptr is not a monotone cover of the edges; " : "",
                     (h[5] & 4u) ? "node_label outside [0, n_labels0); " : "",
                     (h[5] & 8u) ? "col_idx leaves the node range of its graph" : "");
        return GK_ERR_ARG;
    }
    b->max_graph_nodes = (i32)h[0], b->max_degree = (i32)h[1], b->n_big = h[2];
    b->deg_small = WL_DEG_SMALL;
    b->wave_sig = ctx->opt.wl_no_wave_sig ? 0 : 1;      // decided HERE, with deg_small and the big_nodes list it shapes
    if (b->n_big > 0 && b->wave_sig) {
        // A batch with vertices above WL_DEG_SMALL neighbours is off the route without host round trips anyway; its
        // vertices of 17 .. 32 neighbours then go to the wave-per-vertex kernels as well (the thread-per-vertex kernel sorts
        // up to 16 in registers; beyond it had an insertion sort per thread, or -- in a chunk shared with high degrees -- one
        // wave sorting the chunk's lists one after the other: 107 us per level on the COLLAB-like set).  One more flag pass,
        // scan and 4-byte read-back per batch.
        b->deg_small = 16;
        big_flag_kernel<<<grid_for(n_nodes, 256), 256, 0, ctx->stream>>>(b->row_ptr, n_nodes, b->deg_small, flag.p);
        GK_TRY(gk_scan_u32(ctx, flag.p, excl.p, n_nodes, true, (u32*)stats.p + 2));
        u32 nb = 0;
        GK_TRY(gk_readback(ctx, (const u32*)stats.p + 2, &nb, 1));
        b->n_big = nb;
    }
    b->n_labels0_present = few_labels ? (i32)h[4] : 0;
    b->n_iso = 0;
    b->n_isolated = h[3];
    if (h[3] > 0 && !ctx->opt.wl_no_iso) {
        // the carried list of the isolated vertices (see gk_batch::iso_info)
        const i64 n_iso = h[3];
        void* q = nullptr;
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)n_nodes * 4));
        b->iso_info = (i32*)q;
        GK_TRY(gk_dev_alloc(ctx, &q, ((size_t)n_iso + 1) * 4));      // [n_iso] class per slot, then the class count
        b->car_class = (i32*)q;
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)n_iso * 4));
        b->car_nodes = (i32*)q;
        Tmp<u64> keys(ctx);
        Tmp<i32> iso_nodes(ctx), cls(ctx), order(ctx);
        GK_TRY(keys.alloc(n_iso)); GK_TRY(iso_nodes.alloc(n_iso)); GK_TRY(cls.alloc(n_iso)); GK_TRY(order.alloc(n_iso));
        iso_keys_kernel<<<grid_for(n_nodes, 256), 256, 0, ctx->stream>>>(iso_flag.p, iso_excl.p, b->labels, keys.p,
                                                                          iso_nodes.p, b->iso_info, n_nodes);
        GK_TRY(gk_dictionary_from_keys(ctx, keys.p, n_iso, bits_for(b->n_labels0 > 0 ? (u64)b->n_labels0 - 1 : 0),
                                       cls.p, order.p, (u32*)b->car_class + n_iso));
        iso_slots_kernel<<<grid_for(n_iso, 256), 256, 0, ctx->stream>>>(order.p, cls.p, iso_nodes.p, b->iso_info,
                                                                         b->car_class, b->car_nodes, n_iso);
        b->n_iso = n_iso;      // the class count stays on the device (car_class[n_iso]): no second read-back
    }
    {
        void* q = nullptr;
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)(b->n_big > 0 ? b->n_big : 1) * 4));
        b->big_nodes = (i32*)q;
    }
    if (b->n_big > 0)
        compact_big_kernel<<<grid_for(n_nodes, 256), 256, 0, ctx->stream>>>(flag.p, excl.p, b->big_nodes, n_nodes);
    GK_HIP_CHECK(hipGetLastError());
    b->n_levels = 0;
    b->label_counts.assign(1, b->n_labels0);
    return GK_OK;
}

static int batch_alloc(gk_ctx* ctx, gk_batch* b) {
    i32** arrs[] = {&b->graph_ptr, &b->row_ptr, &b->col_idx, &b->node_graph, &b->nbr_sorted, &b->labels, &b->perm};
    const i64 sizes[] = {b->n_graphs + 1, b->n_nodes + 1, b->n_edges, b->n_nodes, b->n_edges, b->n_nodes, b->n_nodes};
    for (int k = 0; k < 7; ++k) {
        void* q = nullptr;
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)(sizes[k] > 0 ? sizes[k] : 1) * 4));
        *arrs[k] = (i32*)q;
    }
    {
        void* q = nullptr;
        GK_TRY(gk_dev_alloc(ctx, &q, (size_t)(b->n_nodes > 0 ? b->n_nodes : 1)));
        b->shared_flag = (unsigned char*)q;
    }
    b->cap_levels = 1;
    return GK_OK;
}

extern "C" int gk_batch_create(gk_ctx* ctx, int64_t n_graphs, int64_t n_nodes, int64_t n_edges,
                               const int32_t* graph_ptr, const int32_t* row_ptr,
                               const int32_t* col_idx, const int32_t* node_label,
                               int32_t n_labels0, int src_on_device, gk_batch** out) {
    GK_ARG(ctx && out, "gk_batch_create: null ctx/out");
    GK_ARG(n_graphs > 0 && n_nodes >= 0 && n_edges >= 0, "gk_batch_create: bad sizes");
    GK_ARG(n_nodes < (1ll << 31) - 1 && n_edges < (1ll << 31) - 1 && n_graphs < (1ll << 31) - 1, "gk_batch_create: int32 index overflow");
    GK_ARG(n_labels0 >= 0, "gk_batch_create: negative n_labels0");
    GK_ARG(graph_ptr && row_ptr && node_label && (col_idx || n_edges == 0), "gk_batch_create: null array");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    gk_batch* b = new gk_batch();
    b->ctx = ctx;
    b->n_graphs = n_graphs, b->n_nodes = n_nodes, b->n_edges = n_edges, b->n_labels0 = n_labels0;
    hipMemcpyKind kind = src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    auto fail = [&](int r) { gk_batch_destroy(b); return r; };
    int r;
    if ((r = batch_alloc(ctx, b))) return fail(r);
#define B_COPY(dst, src, n) if ((n) > 0 && hipMemcpyAsync(dst, src, (size_t)(n) * 4, kind, ctx->stream) != hipSuccess) { gk_set_error("gk_batch_create: copy failed"); return fail(GK_ERR_HIP); }
    B_COPY(b->graph_ptr, graph_ptr, n_graphs + 1);
    B_COPY(b->row_ptr, row_ptr, n_nodes + 1);
    B_COPY(b->col_idx, col_idx, n_edges);
    B_COPY(b->labels, node_label, n_nodes);
#undef B_COPY
    if ((r = batch_finish(ctx, b))) return fail(r);
    if (!src_on_device && hipStreamSynchronize(ctx->stream) != hipSuccess) {     // the host arrays may go away
        gk_set_error("gk_batch_create: %s", hipGetErrorString(hipGetLastError()));
        return fail(GK_ERR_HIP);
    }
    *out = b;
    return GK_OK;
}

// ---- multi-GPU: the global batch straight from the all-gathered shard messages ----------------
// Rank r's message (msg_stride int32 words, as grakel_amd/dist.py packs it) is
//   [graph sizes, padded to mg | node degrees, padded to mv | node labels, padded to mv | col_idx (LOCAL node ids), padded to me]
// The shards are concatenated in rank order; col_idx is shifted to global node ids.
struct ShardMap {
    int R;
    i64 g0[GK_MAX_RANKS + 1], v0[GK_MAX_RANKS + 1], e0[GK_MAX_RANKS + 1];   // prefix sums of the shard sizes
    i64 stride, mg, mv;
    const i32* msg;
    __device__ __forceinline__ int rank_of(const i64* p, i64 i) const {
        int r = 0;
        while (i >= p[r + 1]) ++r;
        return r;
    }
};

struct ShardGraphPtr {     // exclusive prefix of the graph sizes -> graph_ptr
    ShardMap M; i32* graph_ptr; i64 n;
    __device__ __forceinline__ u32 value(i64 g) const {
        const int r = M.rank_of(M.g0, g);
        return (u32)M.msg[r * M.stride + (g - M.g0[r])];
    }
    __device__ __forceinline__ void emit(i64 g, u32, u32 incl) const {
        if (g == 0) graph_ptr[0] = 0;
        graph_ptr[g + 1] = (i32)incl;
    }
    __device__ __forceinline__ void finish(u32) const {}
    __device__ __forceinline__ i64 seg_first_tile(i64) const { return 0; }
};

struct ShardRowPtr {       // exclusive prefix of the node degrees -> row_ptr; labels ride along
    ShardMap M; i32* row_ptr; i32* labels; i64 n;
    __device__ __forceinline__ u32 value(i64 v) const {
        const int r = M.rank_of(M.v0, v);
        return (u32)M.msg[r * M.stride + M.mg + (v - M.v0[r])];
    }
    __device__ __forceinline__ void emit(i64 v, u32, u32 incl) const {
        const int r = M.rank_of(M.v0, v);
        if (v == 0) row_ptr[0] = 0;
        row_ptr[v + 1] = (i32)incl;
        labels[v] = M.msg[r * M.stride + M.mg + M.mv + (v - M.v0[r])];
    }
    __device__ __forceinline__ void finish(u32) const {}
    __device__ __forceinline__ i64 seg_first_tile(i64) const { return 0; }
};

__global__ void shard_col_idx_kernel(const ShardMap M, i32* __restrict__ col_idx, i64 n_edges) {
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int r = M.rank_of(M.e0, e);
    col_idx[e] = M.msg[r * M.stride + M.mg + 2 * M.mv + (e - M.e0[r])] + (i32)M.v0[r];
}

extern "C" int gk_batch_from_shards(gk_ctx* ctx, int n_ranks, const int64_t* shard_sizes, int64_t mg, int64_t mv,
                                    int64_t me, const int32_t* gathered_dev, int32_t n_labels0, gk_batch** out) {
    GK_ARG(ctx && shard_sizes && gathered_dev && out, "gk_batch_from_shards: null argument");
    GK_ARG(n_ranks >= 1 && n_ranks <= GK_MAX_RANKS, "gk_batch_from_shards: 1..64 ranks");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ShardMap M;
    M.R = n_ranks, M.stride = mg + 2 * mv + me, M.mg = mg, M.mv = mv, M.msg = gathered_dev;
    M.g0[0] = M.v0[0] = M.e0[0] = 0;
    for (int r = 0; r < n_ranks; ++r) {
        const int64_t ng = shard_sizes[3 * r], nv = shard_sizes[3 * r + 1], ne = shard_sizes[3 * r + 2];
        GK_ARG(ng >= 0 && nv >= 0 && ne >= 0 && ng <= mg && nv <= mv && ne <= me, "gk_batch_from_shards: shard larger than its padding");
        M.g0[r + 1] = M.g0[r] + ng, M.v0[r + 1] = M.v0[r] + nv, M.e0[r + 1] = M.e0[r] + ne;
    }
    const i64 N = M.g0[n_ranks], V = M.v0[n_ranks], E = M.e0[n_ranks];
    GK_ARG(N > 0, "gk_batch_from_shards: no graphs");
    GK_ARG(V < (1ll << 31) - 1 && E < (1ll << 31) - 1, "gk_batch_from_shards: int32 index overflow");
    gk_batch* b = new gk_batch();
    b->ctx = ctx;
    b->n_graphs = N, b->n_nodes = V, b->n_edges = E, b->n_labels0 = n_labels0;
    auto fail = [&](int r) { gk_batch_destroy(b); return r; };
    int r;
    if ((r = batch_alloc(ctx, b))) return fail(r);
    ShardGraphPtr sg{M, b->graph_ptr, N};
    if ((r = gk_scan_fn<u32, ShardGraphPtr>(ctx, sg, N, nullptr))) return fail(r);
    if (V > 0) {
        ShardRowPtr sr{M, b->row_ptr, b->labels, V};
        if ((r = gk_scan_fn<u32, ShardRowPtr>(ctx, sr, V, nullptr))) return fail(r);
    } else if ((r = gk_zero_async(ctx, b->row_ptr, 4))) return fail(r);
    if (E > 0) shard_col_idx_kernel<<<grid_for(E, 256), 256, 0, ctx->stream>>>(M, b->col_idx, E);
    if ((r = batch_finish(ctx, b))) return fail(r);
    *out = b;
    return GK_OK;
}

// out[i] = in[i] + delta (pointer / index arrays of the second batch of a union)
__global__ void shift_copy_kernel(const i32* __restrict__ in, i32* __restrict__ out, i64 n, i32 delta) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + delta;
}

// Union batch on the device: graphs of `a` first, then the graphs of `b` (transform = fitted graphs +
// targets, weisfeiler_lehman.py:330-500 relabels the targets against the fitted dictionaries; here the
// union is relabelled jointly).  `a` typically is the fitted batch kept resident between calls, so a
// transform only uploads its targets.  Both inputs stay valid.
extern "C" int gk_batch_concat(gk_ctx* ctx, gk_batch* a, gk_batch* b, int32_t n_labels0, gk_batch** out) {
    GK_ARG(ctx && a && b && out, "gk_batch_concat: null argument");
    GK_ARG(!a->is_pair_batch && !b->is_pair_batch, "gk_batch_concat: needs graph batches");
    GK_ARG(a->ctx == ctx && b->ctx == ctx, "gk_batch_concat: batches of another context");
    const i64 N = a->n_graphs + b->n_graphs, V = a->n_nodes + b->n_nodes, E = a->n_edges + b->n_edges;
    GK_ARG(V < (1ll << 31) - 1 && E < (1ll << 31) - 1, "gk_batch_concat: int32 index overflow");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    gk_batch* u = new gk_batch();
    u->ctx = ctx;
    u->n_graphs = N, u->n_nodes = V, u->n_edges = E, u->n_labels0 = n_labels0;
    auto fail = [&](int r) { gk_batch_destroy(u); return r; };
    int r;
    if ((r = batch_alloc(ctx, u))) return fail(r);
    hipStream_t st = ctx->stream;
#define U_COPY(dst, src, n) if ((n) > 0 && hipMemcpyAsync(dst, src, (size_t)(n) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { gk_set_error("gk_batch_concat: copy failed"); return fail(GK_ERR_HIP); }
    U_COPY(u->graph_ptr, a->graph_ptr, a->n_graphs + 1);
    U_COPY(u->row_ptr, a->row_ptr, a->n_nodes + 1);
    U_COPY(u->col_idx, a->col_idx, a->n_edges);
    U_COPY(u->labels, a->labels, a->n_nodes);          // level-0 labels: the first n_nodes entries
    U_COPY(u->labels + a->n_nodes, b->labels, b->n_nodes);
#undef U_COPY
    shift_copy_kernel<<<grid_for(b->n_graphs, 256), 256, 0, st>>>(b->graph_ptr + 1, u->graph_ptr + a->n_graphs + 1, b->n_graphs, (i32)a->n_nodes);
    if (b->n_nodes > 0)
        shift_copy_kernel<<<grid_for(b->n_nodes, 256), 256, 0, st>>>(b->row_ptr + 1, u->row_ptr + a->n_nodes + 1, b->n_nodes, (i32)a->n_edges);
    if (b->n_edges > 0)
        shift_copy_kernel<<<grid_for(b->n_edges, 256), 256, 0, st>>>(b->col_idx, u->col_idx + a->n_edges, b->n_edges, (i32)a->n_nodes);
    if ((r = batch_finish(ctx, u))) return fail(r);
    *out = u;
    return GK_OK;
}

// ---- fitted state for consumers without Python (gk_export_state / gk_import_state) ------------------
// What a fit leaves behind on this path is the packed batch itself (transform relabels the targets jointly with the
// fitted graphs; every level array is recomputed from it), so the persistent state is the CSR + level-0 label ids:
//   header: "GKB1", then int64 n_graphs, n_nodes, n_edges, n_labels0;  arrays: graph_ptr, row_ptr, col_idx, node_label (int32)
// The meaning of the label ids (label value -> id) stays with the caller, as in gk_batch_create.
#define GK_STATE_MAGIC 0x31424b47u       /* "GKB1" */
extern "C" int gk_export_state(gk_ctx* ctx, gk_batch* b, void* out_buf, uint64_t buf_bytes, uint64_t* out_needed) {
    GK_ARG(ctx && b && out_needed, "gk_export_state: null argument");
    GK_ARG(!b->is_pair_batch, "gk_export_state: needs a graph batch");
    const uint64_t need = 8 + 4 * 8 + 4ull * (uint64_t)((b->n_graphs + 1) + (b->n_nodes + 1) + b->n_edges + b->n_nodes);
    *out_needed = need;
    if (!out_buf) return GK_OK;                     // size query
    GK_ARG(buf_bytes >= need, "gk_export_state: buffer too small (call with a null buffer for the size)");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    unsigned char* p = (unsigned char*)out_buf;
    const u32 magic[2] = {GK_STATE_MAGIC, 1u};
    memcpy(p, magic, 8);
    const int64_t hdr[4] = {b->n_graphs, b->n_nodes, b->n_edges, b->n_labels0};
    memcpy(p + 8, hdr, 32);
    p += 40;
    const void* src[4] = {b->graph_ptr, b->row_ptr, b->col_idx, b->labels};
    const i64 cnt[4] = {b->n_graphs + 1, b->n_nodes + 1, b->n_edges, b->n_nodes};
    for (int k = 0; k < 4; ++k) {
        if (cnt[k] > 0) GK_HIP_CHECK(hipMemcpyAsync(p, src[k], (size_t)cnt[k] * 4, hipMemcpyDeviceToHost, ctx->stream));
        p += (size_t)cnt[k] * 4;
    }
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

extern "C" int gk_import_state(gk_ctx* ctx, const void* buf, uint64_t bytes, gk_batch** out) {
    GK_ARG(ctx && buf && out, "gk_import_state: null argument");
    GK_ARG(bytes >= 40, "gk_import_state: truncated state");
    const unsigned char* p = (const unsigned char*)buf;
    u32 magic[2];
    int64_t hdr[4];
    memcpy(magic, p, 8);
    memcpy(hdr, p + 8, 32);
    GK_ARG(magic[0] == GK_STATE_MAGIC && magic[1] == 1u, "gk_import_state: not a gk_hip state blob (or a newer format)");
    // every count below 2^31 (so the sum below cannot wrap), a label count that fits the int32 it is handed on as
    GK_ARG(hdr[0] > 0 && hdr[0] < (1ll << 31) - 1 && hdr[1] >= 0 && hdr[2] >= 0 && hdr[1] < (1ll << 31) - 1 && hdr[2] < (1ll << 31) - 1,
           "gk_import_state: bad sizes");
    GK_ARG(hdr[3] >= 0 && hdr[3] <= 0x7fffffffll, "gk_import_state: bad label count");
    const uint64_t need = 40 + 4ull * ((uint64_t)(hdr[0] + 1) + (uint64_t)(hdr[1] + 1) + (uint64_t)hdr[2] + (uint64_t)hdr[1]);
    GK_ARG(bytes >= need, "gk_import_state: truncated state");
    const int32_t* gp = (const int32_t*)(p + 40);
    const int32_t* rp = gp + (hdr[0] + 1);
    const int32_t* ci = rp + (hdr[1] + 1);
    const int32_t* lab = ci + hdr[2];
    return gk_batch_create(ctx, hdr[0], hdr[1], hdr[2], gp, rp, ci, lab, (int32_t)hdr[3], 0, out);     // validates the CSR
}

extern "C" int gk_batch_destroy(gk_batch* b) {
    if (!b) return GK_OK;
    gk_ctx* ctx = b->ctx;
    void* ptrs[] = {b->graph_ptr, b->row_ptr, b->col_idx, b->node_graph, b->big_nodes,
                    b->labels, b->perm, b->nbr_sorted, b->iso_info, b->car_class, b->car_nodes, b->shared_flag,
                    b->sp_node_ptr, b->sp_node_label, b->sp_dist_ptr, b->sp_dist, b->sp_idtab, b->sr_ctl};
    for (void* p : ptrs)
        if (p) gk_dev_free(ctx, p);
    delete b;
    return GK_OK;
}

extern "C" int gk_batch_info(gk_batch* b, int64_t* n_graphs, int64_t* n_nodes, int64_t* n_edges) {
    GK_ARG(b, "gk_batch_info: null batch");
    if (n_graphs) *n_graphs = b->n_graphs;
    if (n_nodes) *n_nodes = b->n_nodes;
    if (n_edges) *n_edges = b->n_edges;
    return GK_OK;
}

int gk_batch_ensure_levels(gk_batch* b, int n_levels) {
    if (n_levels <= b->cap_levels) return GK_OK;
    gk_ctx* ctx = b->ctx;
    size_t per = (size_t)(b->n_nodes > 0 ? b->n_nodes : 1) * 4;
    void *nl = nullptr, *np = nullptr;
    GK_TRY(gk_dev_alloc(ctx, &nl, per * n_levels));
    GK_TRY(gk_dev_alloc(ctx, &np, per * n_levels));
    GK_HIP_CHECK(hipMemcpyAsync(nl, b->labels, per * b->cap_levels, hipMemcpyDeviceToDevice, ctx->stream));
    GK_HIP_CHECK(hipMemcpyAsync(np, b->perm, per * b->cap_levels, hipMemcpyDeviceToDevice, ctx->stream));
    gk_dev_free(ctx, b->labels);
    gk_dev_free(ctx, b->perm);
    {   // the flags of earlier runs are recomputed by every relabel call: no copy
        void* nf = nullptr;
        GK_TRY(gk_dev_alloc(ctx, &nf, (per / 4) * n_levels));
        gk_dev_free(ctx, b->shared_flag);
        b->shared_flag = (unsigned char*)nf;
    }
    b->labels = (i32*)nl, b->perm = (i32*)np, b->cap_levels = n_levels;
    return GK_OK;
}


// nodes of degree > WL_DEG_SMALL: a wave per node up to WAVE_DEG_MAX neighbours, a workgroup per node beyond (hubs);
// option wl.no_wave_sig keeps everything in the workgroup kernel (rounds 1-4)
static int launch_signature_big(gk_ctx* ctx, gk_batch* b, const i32* lab_prev, u64* hash_by_node, u64 seed, u64 mask,
                                const unsigned char* shared_prev = nullptr) {
    const int wave = b->wave_sig;
    if (wave)
        wl_signature_wave_kernel<<<grid_for(b->n_big * 64, 256), 256, 0, ctx->stream>>>(
            b->big_nodes, b->n_big, b->row_ptr, b->col_idx, lab_prev, b->nbr_sorted, hash_by_node, seed, mask, shared_prev);
    if (!wave || b->max_degree > WAVE_DEG_MAX)
        wl_signature_big_kernel<<<dim3((unsigned)b->n_big), BIG_THREADS, 0, ctx->stream>>>(
            b->big_nodes, b->row_ptr, b->col_idx, lab_prev, b->nbr_sorted, hash_by_node, seed, mask, wave, shared_prev);
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

static int launch_signature(gk_ctx* ctx, gk_batch* b, const i32* lab_prev, u64* hash, u64 seed, u64 mask,
                            const unsigned char* shared_prev = nullptr) {
    i64 V = b->n_nodes;
    if (V == 0) return GK_OK;
    const int sig_regs = ctx->opt.wl_sig_no_regs ? 0 : 1;      // route option: insertion sort in LDS instead
    wl_signature_small_kernel<<<grid_for(V, SIG_THREADS), SIG_THREADS, 0, ctx->stream>>>(
        b->row_ptr, b->col_idx, lab_prev, b->nbr_sorted, hash, V, seed, mask, sig_regs, b->deg_small);
    if (b->n_big > 0) GK_TRY(launch_signature_big(ctx, b, lab_prev, hash, seed, mask, shared_prev));
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

// Sort (key,node) pairs, turn equal-key groups into dense ids (keys[] and vals[] are only read).
// Writes lab[node], perm[] (nodes in key order, ascending node inside a group), rep[id]
// (may be null) and the number of groups to *count_dev.
// vals == nullptr: the items are 0..n-1; otherwise vals[] (ascending node ids) are the items and
// lab/rep/frozen are indexed by item id, perm receives item ids.
// listed_dev (device, may be null; needs vals == nullptr): perm receives the split order of
// HeadAssignSplit and *listed_dev the number of nodes in classes of two or more.
static int dictionary_from_keys(gk_ctx* ctx, const u64* keys, i64 n, int key_bits, i32* lab, i32* perm,
                                i32* rep, u32* count_dev, const u32* vals = nullptr, u32* frozen = nullptr,
                                i64 rep_capacity = 0, int use_buckets = 0, u32* top_digit_max = nullptr,
                                u32* listed_dev = nullptr, u32* posted_seq = nullptr, u32 lab_base = 0,
                                const u32* lab_base_dev = nullptr, unsigned char* shared_out = nullptr,
                                u32* no_order_overflow = nullptr, bool* frozen_in_shared = nullptr,
                                bool* flag_in_lab = nullptr, bool* no_order_taken = nullptr) {
    if (frozen_in_shared) *frozen_in_shared = false;
    if (flag_in_lab) *flag_in_lab = false;
    if (no_order_taken) *no_order_taken = false;
    if (n == 0) {
        GK_TRY(gk_zero_async(ctx, count_dev, 4));
        if (listed_dev) GK_TRY(gk_zero_async(ctx, listed_dev, 4));
        return GK_OK;
    }
    Tmp<u64> ks(ctx);
    GK_TRY(ks.alloc(n));
    Tmp<i32> rep_tmp(ctx);
    if (!rep) { GK_TRY(rep_tmp.alloc(rep_capacity > n ? rep_capacity : n)); rep = rep_tmp.p; }
    if (listed_dev && !vals && no_order_overflow && key_bits >= 24 && gk_bucket_dictionary_fits(ctx, n)) {
        if (no_order_taken) *no_order_taken = true;
        // nobody reads this level's label-grouped order (graph-major features): equal keys only have to meet --
        // top-digit partition + one LDS table per bucket instead of the remaining digit passes and the
        // run-head scan (scan_sort.hip); a bucket that does not fit raises a flag in *no_order_overflow and
        // the level is redone by the sorting path (gk_wl_relabel)
        const u32 seq = posted_seq ? gk_mbox_begin(ctx) : 0u;
        if (posted_seq) *posted_seq = seq;
        // singleton flags: one scattered byte per node (shared_out) instead of a byte and a word -- the only reader of
        // frozen[] after a full level, ActiveScan, takes the bytes
        const bool bytes_only = shared_out && frozen_in_shared && !ctx->opt.wl_frozen_words;
        if (bytes_only) *frozen_in_shared = true;
        // ... and when the caller's verification pass follows (one thread per node), not even the byte:
        // the flag rides in bit 31 of lab[] and verify_kernel writes the bytes in node order
        const bool in_rep = bytes_only && flag_in_lab && rep != rep_tmp.p && !ctx->opt.wl_flag_bytes;
        if (in_rep) *flag_in_lab = true;
        return gk_bucket_dictionary(ctx, keys, n, key_bits, lab, rep, bytes_only ? nullptr : frozen, in_rep ? nullptr : shared_out,
                                    count_dev, listed_dev, top_digit_max, no_order_overflow, seq ? ctx->mbox_dev : nullptr, seq,
                                    in_rep ? 1 : 0);
    }
    if (listed_dev && !vals) {
        Tmp<u32> sorted(ctx);
        GK_TRY(sorted.alloc(n));
        GK_TRY(gk_radix_sort_pairs(ctx, keys, vals, ks.p, sorted.p, n, key_bits, use_buckets, top_digit_max));
        // posted_seq (may be null): the finish hook also posts {listed nodes, largest top-digit bucket}
        // to the host mailbox; *posted_seq receives the sequence number to wait for (0: not posted)
        const u32 seq = posted_seq ? gk_mbox_begin(ctx) : 0u;
        if (posted_seq) *posted_seq = seq;
        HeadAssignSplit ha{ks.p, sorted.p, lab, rep, frozen, perm, listed_dev, count_dev, n,
                           seq ? ctx->mbox_dev : nullptr, seq, top_digit_max, shared_out};
        GK_TRY((gk_scan_fn<u64, HeadAssignSplit>(ctx, ha, n, nullptr)));
        return GK_OK;
    }
    GK_TRY(gk_radix_sort_pairs(ctx, keys, vals, ks.p, (u32*)perm, n, key_bits, use_buckets, top_digit_max));
    HeadAssign ha{ks.p, (const u32*)perm, lab, rep, frozen, n, lab_base, lab_base_dev};
    GK_TRY((gk_scan_fn<u32, HeadAssign>(ctx, ha, n, count_dev)));
    return GK_OK;
}

// exported for sp.hip
int gk_dictionary_from_keys(gk_ctx* ctx, const u64* keys, i64 n, int key_bits, i32* lab, i32* perm, u32* count_dev) {
    return dictionary_from_keys(ctx, keys, n, key_bits, lab, perm, nullptr, count_dev);
}

// The bucket finish of the sort (scan_sort.hip) pays when no top-digit bucket is much larger than
// the average.  Classes only split from level to level, so the largest bucket of the PREVIOUS
// level's sort bounds this level's largest class; it is read back together with n_active.
#define SORT_BUCKET_MAX_KEYS 12288     // what one workgroup sorts entirely in LDS (scan_sort.hip: BK_CAP)
static int sort_buckets_ok(gk_ctx* ctx, u32 prev_top_max, i64 n, bool exact) {
    if (ctx->opt.sort_buckets == 1) return 0;       // option "sort.buckets": 1 never, 2 always (tests), 0 decide
    if (ctx->opt.sort_buckets == 2) return 1;
    if (exact || n / 256 > SORT_BUCKET_MAX_KEYS) return 0;
    if (prev_top_max > 0 && prev_top_max <= SORT_BUCKET_MAX_KEYS) return 1;
    return 2;       // no bound from the previous level (e.g. level 1 after a few input labels): let the sort probe
}

struct RelabelState {
    u32 prev_top_max = 0;                  // largest top-digit bucket of the previous level's sort
    bool default_bits = true;              // the caller did not force a hash width (tests do, to provoke collisions)
    std::vector<char> full_level;          // levels that sorted every node (their perm is split: shared classes first)
    bool split = true;                     // option wl.no_split: keep the plain label-grouped order
    u32 posted_seq = 0;                    // mailbox message {listed nodes, top-digit max, labels} of the previous (full) level
    // Round 6: the partition has CONVERGED when a level has as many labels as the one before (classes only split: equal counts
    // mean the same partition, and every later level repeats it).  The posts of two consecutive full levels tell the host;
    // the remaining levels are copies of the last computed one (labels, order, flags, counts).  The COLLAB-like batch -- dense
    // ego networks -- is stable after level 1: levels 3..5 cost 3 x 0.29 ms of signatures and verification for nothing.
    bool converged = false;
    i64 count_of_prev = -1;                // labels of the previous level when its post said so, else -1
    Tmp<u32> frozen, act, fidx, scratch;   // [V] each; scratch[0] = dictionary count, [1] = n_active, [2] = top-digit max
    bool frozen_in_shared = false;         // the last full level wrote its singleton flags to shared_flag only (never with
                                           // option wl.no_listscan: every level then scans frozen[] of all nodes)
    const unsigned char* shared_prev = nullptr;
    Tmp<u32> act2;                         // second active list (the list of a level is built from the previous level's)
    u32* act_cur = nullptr;                // the current level's active list (act or act2)
    bool prev_active = false;              // the previous level took the active-set path (its list is act_cur)
    u32 n_act_prev = 0;
    bool list_scan = true;                 // option wl.no_listscan: always rebuild the active list from all nodes
    bool tiny = true;                      // option wl.no_tiny: never run a level in the single-workgroup kernel
    bool no_order = false;                 // full levels need no label-grouped order (graph-major features will read them)
    std::vector<char> tiny_level;          // levels run by wl_tiny_level_kernel (n_act_prev is then only a bound)
    i64 n_frozen_levels = 0;
    explicit RelabelState(gk_ctx* c) : frozen(c), act(c), fidx(c), scratch(c), act2(c) {}
};

// one launch of wl_tiny_level_kernel for `level` (labels of the level already hold a copy of the
// previous level's); st.n_act_prev bounds the number of active nodes
static int launch_tiny_level(gk_ctx* ctx, gk_batch* b, int level, int hash_bits, RelabelState& st, int do_scan,
                             const u32* act_prev, u32* act_cur, i32* cur, i32* perm, const i32* prev,
                             u32* count_dev, u32* tiny_dev, u32* unresolved_dev) {
    const i64 V = b->n_nodes;
    const i64 n_car = b->iso_info ? b->n_iso : 0;
    int bits = hash_bits;
    if (hash_bits >= 32) bits = 32;        // 2 * log2(1024) + 8 = 28 bits, rounded up to whole digits; (key << 10 | position) fits 64 bits
    const u64 mask = (1ull << bits) - 1ull;
    const unsigned carried_wgs = (do_scan == 0 && n_car > 0) ? (unsigned)std::min<i64>(cdiv(n_car, TINY_MAX), 8) : 0u;
    wl_tiny_level_kernel<<<1 + carried_wgs, TINY_MAX, 0, ctx->stream>>>(
        act_prev, act_cur, st.scratch.p + 1, do_scan, st.frozen.p, b->row_ptr, b->col_idx, prev, b->nbr_sorted, cur, perm,
        level_seed(level, 0), mask, b->car_nodes, b->car_class, (u32)n_car,
        n_car > 0 ? (const u32*)b->car_class + n_car : nullptr, V, count_dev, tiny_dev, unresolved_dev);
    GK_HIP_CHECK(hipGetLastError());
    st.tiny_level[level] = 1;
    b->active_layout[level] = 1;
    st.prev_active = true;
    b->n_sorted[level] = n_car;            // + the active count, known after the job's final read-back
    return GK_OK;
}

static int relabel_level(gk_ctx* ctx, gk_batch* b, int level, int hash_bits, bool exact, RelabelState& st,
                         u32* count_dev, u32* unresolved_dev, u32* listed_dev, u32* tiny_dev, int* rounds) {
    const i64 V = b->n_nodes;
    const i32* prev = b->labels + (size_t)(level - 1) * V;
    i32* cur = b->labels + (size_t)level * V;
    i32* perm = b->perm + (size_t)level * V;
    if (V == 0) return GK_OK;
    const i64 n_car = b->iso_info ? b->n_iso : 0;      // isolated vertices: carried along, never active
    // ---- how many nodes still sit in classes of size >= 2 ? (one 4-byte read-back per level)
    u32 n_act = (u32)V;
    // level 1 always takes the full path: a singleton class among the INPUT labels is rare, treating
    // it as active is still correct (freezing is an optimisation), and skipping the scan saves two
    // launches and a read-back; the sort probes its buckets instead of using the previous level's bound
    bool decided = false;
    auto repeat_previous_level = [&]() -> int {        // the level is the previous one again (a converged partition)
        GK_HIP_CHECK(hipMemcpyAsync(cur, prev, V * 4, hipMemcpyDeviceToDevice, ctx->stream));
        GK_HIP_CHECK(hipMemcpyAsync(perm, perm - V, V * 4, hipMemcpyDeviceToDevice, ctx->stream));
        GK_HIP_CHECK(hipMemcpyAsync(b->shared_flag + (size_t)level * V, b->shared_flag + (size_t)(level - 1) * V, V, hipMemcpyDeviceToDevice, ctx->stream));
        GK_HIP_CHECK(hipMemcpyAsync(count_dev, count_dev - 1, 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (listed_dev) GK_HIP_CHECK(hipMemcpyAsync(listed_dev, listed_dev - 1, 4, hipMemcpyDeviceToDevice, ctx->stream));
        b->n_sorted[level] = b->n_sorted[level - 1], b->active_layout[level] = b->active_layout[level - 1];
        b->perm_valid[level] = b->perm_valid[level - 1];
        st.full_level[level] = st.full_level[level - 1], st.tiny_level[level] = 0;
        st.posted_seq = 0;
        return GK_OK;
    };
    if (st.converged && !exact) return repeat_previous_level();
    if (!exact && st.posted_seq && level >= 2) {
        // the previous level sorted every node and told the host how many nodes sit in shared classes:
        // when even without the isolated ones they are more than a quarter of the batch this level takes
        // the full path again, and the active-set scan (two launches over all nodes) is not needed
        u32 back[3] = {0, 0, 0};
        GK_TRY(gk_mbox_wait(ctx, st.posted_seq, back, 3));
        const i64 count_prev = (i64)back[2], count_prev2 = st.count_of_prev;
        st.count_of_prev = count_prev;
        if (level >= 3 && count_prev2 >= 0 && count_prev == count_prev2 && st.full_level[level - 1] && st.full_level[level - 2] &&
            !ctx->opt.wl_no_converge) {
            st.converged = true;
            if (ctx->opt.wl_debug) fprintf(stderr, "[gk] level %d: the partition is stable since level %d (%lld labels): copied\n", level, level - 2, (long long)count_prev);
            return repeat_previous_level();
        }
        if (((i64)back[0] - n_car) * 4 > V && !ctx->opt.wl_no_active_set) {
            n_act = back[0], st.prev_top_max = back[1], decided = true;
        }
    } else st.count_of_prev = -1;
    st.posted_seq = 0;
    bool list_based = false;               // this level's active list came from the previous level's list
    if (decided) {
    } else if (!exact && !ctx->opt.wl_no_active_set && level >= 2) {
        u32 back[2] = {0, 0};
        if (st.prev_active && st.list_scan) {
            // the previous level took the active-set path: frozen ids are permanent, so this level's labels
            // start as a copy and only the previous active list (st.act_cur) is scanned
            list_based = true;
            GK_HIP_CHECK(hipMemcpyAsync(cur, prev, V * 4, hipMemcpyDeviceToDevice, ctx->stream));
            if (st.tiny && b->n_big == 0 && st.n_act_prev <= TINY_MAX) {
                // a few hundred active nodes at most: the whole level in one workgroup, no read-back
                // (st.n_act_prev stays an upper bound; the exact count lives in st.scratch[1])
                u32* act_new = st.act_cur == st.act.p ? st.act2.p : st.act.p;
                GK_TRY(launch_tiny_level(ctx, b, level, hash_bits, st, 1, st.act_cur, act_new, cur, perm, prev, count_dev,
                                         tiny_dev, unresolved_dev));
                st.act_cur = act_new;
                return GK_OK;
            }
            if (st.n_act_prev > 0) {
                const u32 seq = gk_mbox_begin(ctx);
                u32* act_new = st.act_cur == st.act.p ? st.act2.p : st.act.p;
                const u32 n_frozen_prev = (u32)(V - (i64)st.n_act_prev - n_car);
                ActiveFromList af{st.frozen.p, st.act_cur, act_new, cur, n_frozen_prev, st.scratch.p + 1, st.scratch.p + 2,
                                  seq ? ctx->mbox_dev : nullptr, seq};
                GK_TRY((gk_scan_fn<u32, ActiveFromList>(ctx, af, (i64)st.n_act_prev, nullptr)));
                if (seq) GK_TRY(gk_mbox_wait(ctx, seq, back, 2));
                else GK_TRY(gk_readback(ctx, st.scratch.p + 1, back, 2));
                st.act_cur = act_new;
            }
        } else {
            const u32 seq = gk_mbox_begin(ctx);
            ActiveScan as{st.frozen.p, st.act.p, st.fidx.p, st.scratch.p + 1, st.scratch.p + 2, seq ? ctx->mbox_dev : nullptr, seq,
                          n_car > 0 ? b->iso_info : nullptr, st.frozen_in_shared ? st.shared_prev : nullptr};
            GK_TRY((gk_scan_fn<u32, ActiveScan>(ctx, as, V, nullptr)));
            if (seq) GK_TRY(gk_mbox_wait(ctx, seq, back, 2));
            else GK_TRY(gk_readback(ctx, st.scratch.p + 1, back, 2));
            st.act_cur = st.act.p;
        }
        n_act = back[0], st.prev_top_max = back[1];
        if (ctx->opt.wl_debug) fprintf(stderr, "[gk] level %d: active %u of %lld%s, previous top-digit bucket max %u\n", level, n_act,
                         (long long)V, list_based ? " (from the previous list)" : "", back[1]);
    } else {
        st.prev_top_max = 0;
    }
    const u64 full_mask = hash_bits >= 64 ? ~0ull : ((1ull << hash_bits) - 1ull);
    b->n_sorted[level] = V;
    if (n_act == 0 && n_car == 0) {
        // every class is a singleton: the partition cannot change any more
        b->n_sorted[level] = 0;
        b->active_layout[level] = 1;
        GK_HIP_CHECK(hipMemcpyAsync(cur, prev, V * 4, hipMemcpyDeviceToDevice, ctx->stream));
        GK_HIP_CHECK(hipMemcpyAsync(perm, perm - V, V * 4, hipMemcpyDeviceToDevice, ctx->stream));
        GK_HIP_CHECK(hipMemcpyAsync(count_dev, count_dev - 1, 4, hipMemcpyDeviceToDevice, ctx->stream));
        st.prev_active = true, st.n_act_prev = 0;       // nothing left to scan at the following levels either
        return GK_OK;
    }
    if (!exact && (u64)n_act * 4 <= (u64)V) {
        // ---- active-set path: only the n_act active nodes are hashed, sorted and verified; the
        // carried classes of the isolated vertices are listed behind them.
        // ids of the level: [frozen nodes | carried classes | classes of the active nodes]
        b->n_sorted[level] = (i64)n_act + n_car;
        b->active_layout[level] = 1;
        const u32 n_frozen = (u32)(V - (i64)n_act - n_car);
        const u32* n_cc_dev = n_car > 0 ? (const u32*)b->car_class + n_car : nullptr;
        int bits = hash_bits;
        if (hash_bits >= 32) {   // default sizing rule applied to the active set (tests may force fewer bits)
            int lg = bits_for((u64)(n_act > 1 ? n_act - 1 : 1));
            bits = ((2 * lg + 8 + 7) / 8) * 8;
            if (bits < 32) bits = 32;
            if (bits > hash_bits) bits = hash_bits;
        }
        const u64 mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
        if (list_based && st.tiny && b->n_big == 0 && n_act <= TINY_MAX) {
            st.n_act_prev = n_act;
            return launch_tiny_level(ctx, b, level, hash_bits, st, 0, nullptr, st.act_cur, cur, perm, prev, count_dev,
                                     tiny_dev, unresolved_dev);
        }
        Tmp<u64> hash_act(ctx), hash_node(ctx);
        Tmp<i32> rep(ctx);
        GK_TRY(hash_act.alloc(n_act)); GK_TRY(rep.alloc(n_act));
        const u64 seed = level_seed(level, 0);
        wl_signature_list_kernel<<<grid_for(n_act, 256), 256, 0, ctx->stream>>>(
            st.act_cur, n_act, b->row_ptr, b->col_idx, prev, b->nbr_sorted, hash_act.p, seed, mask, b->deg_small);
        if (b->n_big > 0) {
            GK_TRY(hash_node.alloc(V));
            GK_TRY(launch_signature_big(ctx, b, prev, hash_node.p, seed, mask));
            gather_big_hash_kernel<<<grid_for(n_act, 256), 256, 0, ctx->stream>>>(st.act_cur, n_act, b->row_ptr, hash_node.p, hash_act.p, b->deg_small);
        }
        GK_TRY(dictionary_from_keys(ctx, hash_act.p, n_act, bits, cur, perm, rep.p, st.scratch.p, st.act_cur, st.frozen.p, 0,
                                    sort_buckets_ok(ctx, st.prev_top_max, n_act, exact), st.scratch.p + 2, nullptr, nullptr,
                                    n_frozen, n_cc_dev));
        // *unresolved_dev is still zero here: gk_wl_relabel cleared it and this path runs once per level
        if (list_based) {
            const i64 m = (i64)n_act > n_car ? (i64)n_act : n_car;
            active_finish_kernel<<<grid_for(m, 256), 256, 0, ctx->stream>>>(
                st.act_cur, n_act, b->car_nodes, b->car_class, (u32)n_car, n_frozen, n_cc_dev, st.scratch.p, cur, perm,
                count_dev, b->row_ptr, prev, b->nbr_sorted, rep.p, unresolved_dev);
        } else {
            frozen_assign_verify_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(
                st.fidx.p, st.scratch.p, cur, perm, count_dev, n_act, V, (u32)n_car, n_cc_dev, b->car_class,
                b->row_ptr, prev, b->nbr_sorted, rep.p, unresolved_dev);
        }
        GK_HIP_CHECK(hipGetLastError());
        st.prev_active = true, st.n_act_prev = n_act;
        return GK_OK;
    }
    st.prev_active = false;
    b->active_layout[level] = 0;
    // ---- full path
    if (!st.split) listed_dev = nullptr;
    st.full_level[level] = listed_dev ? 1 : 0;
    st.tiny_level[level] = 0;
    b->n_sorted[level] = V;
    Tmp<u64> hash(ctx), keys(ctx);
    Tmp<i32> rep(ctx);
    GK_TRY(hash.alloc(V)); GK_TRY(rep.alloc(V));
    // level 1 of a job with few input labels: exact 32-bit signature codes (wl_signature_exact_kernel)
    bool exact_code = false;
    u64 code_R = (u64)b->max_degree + 1;
    if (level == 1 && !exact && st.default_bits && b->n_labels0 >= 1 && b->n_labels0 <= 16 && !ctx->opt.wl_no_exact1) {
        double span = (double)b->n_labels0;
        for (int i = 0; i < b->n_labels0; ++i) span *= (double)code_R;
        exact_code = span < 4294967296.0;
    }
    bool no_order_taken = false;
    for (int round = 0;; ++round) {
        if (exact_code) {
            wl_signature_exact_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(b->row_ptr, b->col_idx, prev, hash.p, V,
                                                                                 (int)b->n_labels0, code_R, unresolved_dev);
            GK_TRY(dictionary_from_keys(ctx, hash.p, V, 32, cur, perm, rep.p, count_dev, nullptr, st.frozen.p, 0,
                                        sort_buckets_ok(ctx, st.prev_top_max, V, exact), st.scratch.p + 2, listed_dev,
                                        listed_dev ? &st.posted_seq : nullptr, 0, nullptr, b->shared_flag + (size_t)level * V,
                                        (st.no_order && !exact) ? unresolved_dev : nullptr, st.list_scan ? &st.frozen_in_shared : nullptr,
                                        nullptr, &no_order_taken));
            st.shared_prev = b->shared_flag + (size_t)level * V;
            b->perm_valid[level] = no_order_taken ? 0 : 1;
            GK_HIP_CHECK(hipGetLastError());
            break;
        }
        // (singletons of the previous FULL level keep to themselves: SIG_FROZEN_KEY; not in the exact redo, whose rounds refine
        // every node's key)
        const unsigned char* skip = (!exact && round == 0 && level >= 2 && st.full_level[level - 1] && !ctx->opt.wl_no_frozen_skip)
                                        ? b->shared_flag + (size_t)(level - 1) * V : nullptr;
        GK_TRY(launch_signature(ctx, b, prev, hash.p, level_seed(level, round), full_mask, skip));
        bool flag_in_lab = false;
        int bits = hash_bits;
        const u64* sort_keys = hash.p;           // round 0: the sort reads the hashes in place
        if (round > 0) {
            if (!keys.p) GK_TRY(keys.alloc(V));
            refine_keys_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(cur, hash.p, keys.p, V);
            bits = 64, sort_keys = keys.p;
        }
        GK_TRY(dictionary_from_keys(ctx, sort_keys, V, bits, cur, perm, rep.p, count_dev, nullptr, st.frozen.p, 0,
                                    round == 0 ? sort_buckets_ok(ctx, st.prev_top_max, V, exact) : 0, st.scratch.p + 2, listed_dev,
                                    (listed_dev && !exact) ? &st.posted_seq : nullptr, 0, nullptr,
                                    b->shared_flag + (size_t)level * V,
                                    (st.no_order && !exact && round == 0) ? unresolved_dev : nullptr, st.list_scan ? &st.frozen_in_shared : nullptr,
                                    &flag_in_lab, &no_order_taken));
        st.shared_prev = b->shared_flag + (size_t)level * V;
        b->perm_valid[level] = no_order_taken ? 0 : 1;
        // the first attempt of a level finds *unresolved_dev cleared by gk_wl_relabel
        if (exact) GK_TRY(gk_zero_async(ctx, unresolved_dev, 4));
        const int big_apart = (b->n_big > 0 && b->wave_sig) ? b->deg_small : 0;
        verify_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(b->row_ptr, prev, b->nbr_sorted, cur, rep.p, unresolved_dev, V,
                                                                  flag_in_lab ? b->shared_flag + (size_t)level * V : nullptr, big_apart);
        if (big_apart)
            verify_big_kernel<<<grid_for(cdiv(b->n_big, VB_PER_WAVE) * 64, 256), 256, 0, ctx->stream>>>(
                b->big_nodes, b->n_big, b->row_ptr, prev, b->nbr_sorted, cur, rep.p, unresolved_dev,
                flag_in_lab ? b->shared_flag + (size_t)level * V : nullptr);
        GK_HIP_CHECK(hipGetLastError());
        if (!exact) break;
        u32 un = 0;
        GK_HIP_CHECK(hipMemcpyAsync(&un, unresolved_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (un == 0) break;
        if (rounds) ++*rounds;
        if (round > 200) {
            gk_set_error("gk_wl_relabel: refinement did not converge at level %d", level);
            return GK_ERR_STATE;
        }
    }
    return GK_OK;
}

// One hashed pass over all levels (level 0 included); h receives the per-level words of meta[] ([n_levels] counts,
// unresolved, nodes of shared classes, active nodes of the single-workgroup levels).  force_sort: every dictionary
// takes the sorting path (the second attempt after a bucket of the sort-free dictionary overflowed).
static int relabel_pass(gk_ctx* ctx, gk_batch* b, int n_levels, int hash_bits, bool default_bits, bool force_sort,
                        RelabelState& st, Tmp<u32>& meta, std::vector<u32>& h) {
    const i64 V = b->n_nodes;
    GK_TRY(gk_zero_async(ctx, meta.p, 16 * (size_t)n_levels));
    b->n_sorted.assign((size_t)n_levels, V);
    b->active_layout.assign((size_t)n_levels, 0);
    b->perm_valid.assign((size_t)n_levels, 1);
    st.default_bits = default_bits;
    st.split = !ctx->opt.wl_no_split;
    st.full_level.assign((size_t)n_levels, 0);
    st.list_scan = !ctx->opt.wl_no_listscan;
    st.tiny = !ctx->opt.wl_no_tiny;
    // the graph-major feature builder (features_gm.hip) takes graph batches with small graphs: their full levels
    // then need no label-grouped order
    st.no_order = !b->is_pair_batch && b->max_graph_nodes <= (ctx->opt.gm_no_huge ? GM_MAX_NODES : GM_HUGE_MAX_NODES) && !ctx->opt.feat_no_gm &&
                  !ctx->opt.wl_no_bucket_dict && !force_sort;
    st.tiny_level.assign((size_t)n_levels, 0);
    GK_TRY(st.frozen.alloc(V)); GK_TRY(st.act.alloc(V)); GK_TRY(st.fidx.alloc(V)); GK_TRY(st.scratch.alloc(4));
    GK_TRY(st.act2.alloc(V / 4 + 1));      // active-set levels hold at most V/4 active nodes
    // level 0: group nodes by the given label ids -- unless there are only a few of them: then nothing of
    // level 0 needs an order (level 1 always takes the full path and never reads the singleton flags, the
    // label-count features of level 0 come from one LDS histogram per graph, features.hip)
    const bool hist0 = V > 0 && b->n_labels0 >= 1 && b->n_labels0 <= GK_HIST0_MAX_LABELS && !ctx->opt.wl_no_hist0;
    b->level0_hist = hist0;
    if (!hist0) {
        Tmp<u64> keys(ctx);
        Tmp<i32> lab_tmp(ctx);
        GK_TRY(keys.alloc(V)); GK_TRY(lab_tmp.alloc(V));
        if (V > 0) labels_to_keys_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(b->labels, keys.p, V);
        int bits = bits_for(b->n_labels0 > 0 ? (u64)b->n_labels0 - 1 : 0);
        st.full_level[0] = st.split ? 1 : 0;
        GK_TRY(dictionary_from_keys(ctx, keys.p, V, bits, lab_tmp.p, b->perm, nullptr, meta.p, nullptr, st.frozen.p, 0,
                                    0, st.scratch.p + 2, st.split ? meta.p + 2 * n_levels : nullptr, nullptr, 0, nullptr,
                                    b->shared_flag));
    }
    for (int lvl = 1; lvl < n_levels; ++lvl)
        GK_TRY(relabel_level(ctx, b, lvl, hash_bits, false, st, meta.p + lvl, meta.p + n_levels + lvl,
                             meta.p + 2 * n_levels + lvl, meta.p + 3 * n_levels + lvl, nullptr));
    GK_TRY(gk_readback(ctx, meta.p, h.data(), 4 * n_levels));
    return GK_OK;
}

extern "C" int gk_wl_relabel(gk_ctx* ctx, gk_batch* b, int n_iter, int hash_bits,
                             int64_t* out_label_counts, int* out_rounds) {
    GK_ARG(ctx && b, "gk_wl_relabel: null ctx/batch");
    GK_ARG(!b->is_pair_batch, "gk_wl_relabel: pair batches have no adjacency");
    GK_ARG(n_iter >= 0 && n_iter < 4096, "gk_wl_relabel: bad n_iter");
    const bool default_bits = hash_bits <= 0 || hash_bits > 64;
    if (hash_bits <= 0 || hash_bits > 64) {
        // default: 2*log2(V) + 8 bits (multiple of the 8-bit radix digit, at least 32): a colliding
        // pair then shows up in ~1/256 of the levels and is resolved exactly by the refine loop
        int lg = bits_for((u64)(b->n_nodes > 1 ? b->n_nodes - 1 : 1));
        hash_bits = ((2 * lg + 8 + 7) / 8) * 8;
        if (hash_bits < 32) hash_bits = 32;
        if (hash_bits > 64) hash_bits = 64;
    }
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ++b->relabel_gen;
    ProfScope prof(ctx, "relabel");
    const int n_levels = n_iter + 1;
    const i64 V = b->n_nodes;
    GK_TRY(gk_batch_ensure_levels(b, n_levels));
    Tmp<u32> meta(ctx);   // [n_levels] counts, [n_levels] unresolved, [n_levels] nodes of shared classes (full levels)
    GK_TRY(meta.alloc(4 * (size_t)n_levels));     // ... and [n_levels] active nodes of the levels run by the tiny kernel
    if (out_rounds) *out_rounds = 0;
    // ---- the route without host round trips (wl_stream.hip); it declines jobs it is not built for and hands hash
    // collisions / table overflows back
    b->stream_layout = false;
    {
        std::vector<u32> counts;
        const int r = gk_wl_relabel_stream(ctx, b, n_levels, hash_bits, default_bits, counts);
        if (r == GK_OK) {
            b->n_levels = n_levels;
            b->label_counts.resize(n_levels);
            for (int lvl = 0; lvl < n_levels; ++lvl) {
                b->label_counts[lvl] = counts[lvl];
                if (out_label_counts) out_label_counts[lvl] = counts[lvl];
            }
            return GK_OK;
        }
        if (r != GK_ERR_UNSUPPORTED) return r;
        b->stream_layout = false;
    }
    std::vector<u32> h(4 * (size_t)n_levels);
    int first_bad = -1;
    std::unique_ptr<RelabelState> stp;
    for (int attempt = 0; attempt < 2; ++attempt) {
        stp.reset(new RelabelState(ctx));
        GK_TRY(relabel_pass(ctx, b, n_levels, hash_bits, default_bits, attempt == 1, *stp, meta, h));
        first_bad = -1;
        for (int lvl = 1; lvl < n_levels; ++lvl)
            if (h[n_levels + lvl] != 0) { first_bad = lvl; break; }
        // a bucket of the sort-free dictionary overflowed (bit 31; gk_bucket_dictionary_fits makes that a matter of
        // adversarial key distributions or of the wl.bd_slots test hook): once more, on the sorting path throughout
        if (first_bad > 0 && (h[n_levels + first_bad] & 0x80000000u) && attempt == 0) continue;
        break;
    }
    RelabelState& st = *stp;
    if (first_bad > 0) {   // a hash collision was detected: redo from that level, exactly
        for (int lvl = first_bad; lvl < n_levels; ++lvl)
            GK_TRY(relabel_level(ctx, b, lvl, hash_bits, true, st, meta.p + lvl, meta.p + n_levels + lvl,
                                 meta.p + 2 * n_levels + lvl, meta.p + 3 * n_levels + lvl, out_rounds));
        GK_HIP_CHECK(hipMemcpyAsync(h.data(), meta.p, 16 * (size_t)n_levels, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    b->n_levels = n_levels;
    b->label_counts.resize(n_levels);
    for (int lvl = 0; lvl < n_levels && V > 0; ++lvl) {     // full levels only list the nodes of shared classes
        if (st.full_level[lvl]) b->n_sorted[lvl] = h[2 * n_levels + lvl];
        else if (st.tiny_level[lvl]) b->n_sorted[lvl] += h[3 * n_levels + lvl];      // carried + active
    }
    if (b->level0_hist) h[0] = (u32)b->n_labels0_present;
    for (int lvl = 0; lvl < n_levels; ++lvl) {
        b->label_counts[lvl] = h[lvl];
        if (out_label_counts) out_label_counts[lvl] = h[lvl];
    }
    return GK_OK;
}

// ---- fit_transform in one call: relabel -> label-count features -> Gram matrix ---------------------------------------
// (weisfeiler_lehman.py:292-328 + vertex_histogram.py:57-184 + kernel.py:195-204 in one piece.)  The three entry points
// above each end in a host round trip and the caller's own glue sits between them; here the stream relabel is only QUEUED
// (gk_sr_enqueue), the feature builder runs behind it on device-side counts, and the ONE round trip of the whole job -- the
// operand sizes the Gram launch needs -- also carries the relabel's control words (collision / overflow flags, label counts).
// Jobs the stream route declines, or that turn out to need the redo, take the three calls in sequence.
extern "C" int gk_wl_fit_transform(gk_ctx* ctx, gk_batch* b, int n_iter, int hash_bits, int kind, int normalize,
                                   int64_t* out_label_counts, int* out_rounds, gk_feat** out_feat, double* out_host) {
    GK_ARG(ctx && b && out_feat, "gk_wl_fit_transform: null argument");
    GK_ARG(!b->is_pair_batch, "gk_wl_fit_transform: pair batches have no adjacency");
    GK_ARG(n_iter >= 0 && n_iter < 4096, "gk_wl_fit_transform: bad n_iter");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    const int n_levels = n_iter + 1;
    int bits = hash_bits;
    const bool default_bits = bits <= 0 || bits > 64;
    if (default_bits) {
        int lg = bits_for((u64)(b->n_nodes > 1 ? b->n_nodes - 1 : 1));
        bits = ((2 * lg + 8 + 7) / 8) * 8;
        if (bits < 32) bits = 32;
        if (bits > 64) bits = 64;
    }
    gk_feat* f = nullptr;
    bool fused = false, no_stream_once = false;
    if (out_rounds) *out_rounds = 0;
    if (n_levels >= 2 && n_levels <= FEAT_MAX_LEVELS) {
        GK_TRY(gk_batch_ensure_levels(b, n_levels));
        ++b->relabel_gen;
        int r;
        {
            ProfScope prof(ctx, "relabel");
            r = gk_sr_enqueue(ctx, b, n_levels, bits, default_bits);
        }
        bool redo_host_driven = false;
        if (r == GK_OK) {
            r = gk_features_build_ex(ctx, b, n_levels, b->n_graphs, kind, &f);
            if (r == GK_OK) fused = true;
            else {
                // the queued relabel was not collected (or asked for the redo): whatever the error, the batch must not keep a
                // stream layout whose control words nobody read -- it goes back to "not relabelled"
                if (b->sr_pending > 0 || r == GK_ERR_RETRY) {
                    (void)hipStreamSynchronize(ctx->stream);
                    b->sr_pending = 0, b->stream_layout = false, b->n_levels = 0;
                    redo_host_driven = true;
                }
                if (r != GK_ERR_RETRY) return r;
            }
        } else if (r != GK_ERR_UNSUPPORTED) return r;
        if (redo_host_driven && !fused) no_stream_once = true;
    }
    if (!fused) {
        b->sr_pending = 0;
        const int keep = ctx->opt.wl_no_stream;
        if (no_stream_once || (n_levels >= 2 && n_levels <= FEAT_MAX_LEVELS && !b->stream_layout && b->n_levels == 0)) ctx->opt.wl_no_stream = 1;    // the redo: not the stream route again
        int r = gk_wl_relabel(ctx, b, n_iter, hash_bits, nullptr, out_rounds);
        ctx->opt.wl_no_stream = keep;
        GK_TRY(r);
        GK_ARG(n_levels <= FEAT_MAX_LEVELS, "gk_wl_fit_transform: at most 48 levels (deeper hierarchies: gk_features_build_range in chunks)");
        GK_TRY(gk_features_build_ex(ctx, b, n_levels, b->n_graphs, kind, &f));
    }
    if (out_label_counts)
        for (int l = 0; l < n_levels; ++l) out_label_counts[l] = b->label_counts[l];
    const int r = gk_gram_rows(ctx, f, 0, b->n_graphs, normalize, out_host);
    if (r != GK_OK) { gk_features_destroy(f); return r; }
    *out_feat = f;
    return GK_OK;
}

// Label-grouped node order of a level whose dictionary ran without a sort (perm_valid == 0), built on demand:
// the graph-major feature builder needs no order, but when it declines a job (operand row wider than its LDS
// image) the label-major builder (features.hip) takes over and reads perm[level][0 .. n_sorted) = the nodes of
// shared classes grouped by label, ascending node inside a group.  One stable sort of (label | singleton bit).
__global__ void order_keys_kernel(const i32* __restrict__ lab, const unsigned char* __restrict__ shared, u64* __restrict__ keys,
                                  i64 n, u64 single_key) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v < n) keys[v] = (!shared || shared[v]) ? (u64)(u32)lab[v] : single_key;
}

int gk_batch_rebuild_order(gk_ctx* ctx, gk_batch* b, int level) {
    const i64 V = b->n_nodes;
    if (V == 0 || (size_t)level >= b->perm_valid.size() || b->perm_valid[level]) return GK_OK;
    if (b->stream_layout) {
        GK_ARG(level >= 1 && level < b->n_levels && (size_t)level < b->sr_F.size(), "gk_batch_rebuild_order: level not computed");
        return gk_sr_rebuild_order(ctx, b, level);
    }
    GK_ARG(level >= 0 && level < b->n_levels && b->shared_flag, "gk_batch_rebuild_order: level without class flags");
    const i64 count = level < (int)b->label_counts.size() ? b->label_counts[level] : V;
    const int bits = bits_for((u64)(count > 0 ? count : 1));          // the singleton key `count` sorts behind every label
    Tmp<u64> keys(ctx), ks(ctx);
    GK_TRY(keys.alloc(V)); GK_TRY(ks.alloc(V));
    order_keys_kernel<<<grid_for(V, 256), 256, 0, ctx->stream>>>(b->labels + (size_t)level * V, b->shared_flag + (size_t)level * V,
                                                                  keys.p, V, (u64)count);
    GK_TRY(gk_radix_sort_pairs(ctx, keys.p, nullptr, ks.p, (u32*)(b->perm + (size_t)level * V), V, bits));
    GK_HIP_CHECK(hipGetLastError());
    b->perm_valid[level] = 1;
    return GK_OK;
}

// 1: the last gk_wl_relabel of the batch ran without host round trips and left the stream id layout (wl_stream.hip), 0: host-driven route
extern "C" int gk_wl_route(gk_batch* b, int* out_stream) {
    GK_ARG(b && out_stream, "gk_wl_route: null argument");
    *out_stream = b->stream_layout ? 1 : 0;
    return GK_OK;
}

extern "C" int gk_wl_get_labels(gk_ctx* ctx, gk_batch* b, int level, int32_t* out_labels) {
    GK_ARG(ctx && b && out_labels, "gk_wl_get_labels: null argument");
    GK_ARG(level >= 0 && level < (b->n_levels > 0 ? b->n_levels : 1), "gk_wl_get_labels: level not computed");
    if (b->n_nodes == 0) return GK_OK;
    GK_HIP_CHECK(hipMemcpyAsync(out_labels, b->labels + (size_t)level * b->n_nodes, b->n_nodes * 4,
                                hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

extern "C" int gk_wl_debug_signature(gk_ctx* ctx, gk_batch* b, int level, uint64_t seed,
                                     uint64_t* out_hash, int32_t* out_sorted) {
    GK_ARG(ctx && b, "gk_wl_debug_signature: null argument");
    GK_ARG(level >= 1 && level <= (b->n_levels > 0 ? b->n_levels : 1), "gk_wl_debug_signature: previous level not computed");
    Tmp<u64> hash(ctx);
    GK_TRY(hash.alloc(b->n_nodes));
    GK_TRY(launch_signature(ctx, b, b->labels + (size_t)(level - 1) * b->n_nodes, hash.p, seed, ~0ull));
    if (out_hash && b->n_nodes)
        GK_HIP_CHECK(hipMemcpyAsync(out_hash, hash.p, b->n_nodes * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_sorted && b->n_edges)
        GK_HIP_CHECK(hipMemcpyAsync(out_sorted, b->nbr_sorted, b->n_edges * 4, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return GK_OK;
}

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
#include "scan_fn.h"
#include <stdlib.h>

#define SP_INF 0x3f000000
#define SP_THREADS 256
#define SP_FW_MAX_N 200          // 200*201*4 B = 160.8 KB > 160 KiB? -> see sp_fw_cap()
#define SP_ROW_MAX_N 32768

int gk_dictionary_from_keys(gk_ctx* ctx, const u64* keys, i64 n, int key_bits, i32* lab, i32* perm, u32* count_dev);

static inline dim3 grid_for(i64 n, int t) { return dim3((unsigned)(n > 0 ? cdiv(n, t) : 1)); }

// size classes of the bit-parallel breadth-first search (sp_msbfs_kernel, further down)
#define SPB_HUB_DEG 32
#define SPB_BITWISE_MAX 6      // new bits of a vertex and sweep up to which the levels are stored byte by byte
#define SPB_HUB_CAP 512
#define SPB_LDS_MAX (149 * 1024)        // dynamic LDS of a workgroup (the static hub arrays take 10 KiB more)
#define SPB_LDS0 (42 * 1024)
#define SPB_LDS1 (69 * 1024)
#define SPB_OVERFLOW 0xffffffffu
__host__ __device__ static inline int spb_class(int n, int m) {
    const long n8 = (n + 7) & ~7, need = 2l * m;
    // per vertex: two words of G bits + G distance bytes + 4 bytes of padding (SPB row stride)
    if (n <= 512 && 84 * n8 + need <= SPB_LDS0) return 0;
    if (n <= 1024 && 84 * n8 + need <= SPB_LDS1) return 1;
    if (n <= 2048 && 84 * n8 + need <= SPB_LDS_MAX) return 2;
    if (n <= 4096 && 44 * n8 + need <= SPB_LDS_MAX) return 3;
    if (24 * n8 <= SPB_LDS_MAX) return 4;
    return 5;
}
#define SPB_MAX_N (SPB_LDS_MAX / 24 / 8 * 8)

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

// ---------------------------------------------------------------------------------------
// Graphs of up to 64 vertices (97 % of the NCI1-like set): ONE WAVE per graph, the distance matrix in REGISTERS.
// Lane i owns row i as NP registers (NP = n rounded up to a multiple of 8, a compile-time constant: register arrays
// need static indices).  Pivot k relaxes d[i][j] = min(d[i][j], d[i][k] + d[k][j]): d[k][j] is lane k's register
// (v_readlane with the lane number in an SGPR), d[i][k] would be the lane's OWN register number k -- a dynamic
// register index.  So the rows ROTATE: the pass over j writes its results one register down, after pivot k register
// r holds column (r + k + 1) mod NP, the needed d[i][k] is always register 0, and after NP pivots everything is
// back in place.  Three VALU instructions per (k, j), no LDS traffic, no barrier of any kind in the main loop (the
// workgroup form needs one s_barrier per pivot and kept half of its 64-lane rows idle at n = 30).
// LDS only stages the adjacency on the way in (scattered edge writes) and the rows on the way out (coalesced store).
// ---------------------------------------------------------------------------------------
#define SP_REG_MAX_N 64
#define SP_REG_WAVES 4

template <int NP>
__device__ __forceinline__ void sp_fw_reg_body(i32* __restrict__ lds, int n, const i32* __restrict__ row_ptr,
                                               const i32* __restrict__ col_idx, const i32* __restrict__ w, i32 v0,
                                               i32* __restrict__ out, u32* __restrict__ pair_count_g, u32* __restrict__ maxd) {
    constexpr int LD = NP + 4;                       // 16-byte aligned rows, lanes spread over the banks
    const int lane = threadIdx.x & 63;
    for (int idx = lane; idx < NP * LD; idx += 64) lds[idx] = SP_INF;
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {
        const i32 e0 = row_ptr[v0 + lane], e1 = row_ptr[v0 + lane + 1];
        for (i32 e = e0; e < e1; ++e) {               // multi-edges keep the lightest (the CSR holds each neighbour once)
            const int j = col_idx[e] - v0;
            const i32 wt = w ? w[e] : 1;
            if (wt < lds[lane * LD + j]) lds[lane * LD + j] = wt;
        }
        lds[lane * LD + lane] = 0;                    // np.fill_diagonal(dist, 0): graph.py:1786
    }
    __builtin_amdgcn_wave_barrier();
    i32 d[NP];
    const int rl = lane < NP ? lane : NP - 1;          // lanes beyond the padded size read a valid row (their results are never used)
#pragma unroll
    for (int r = 0; r < NP; r += 4) {
        const int4 t = *(const int4*)(lds + rl * LD + r);
        d[r] = t.x, d[r + 1] = t.y, d[r + 2] = t.z, d[r + 3] = t.w;
    }
    for (int k = 0; k < NP; ++k) {
        const i32 dik = d[0];
        if (k < n) {                                  // wave-uniform
            const i32 x0 = min(d[0], dik + __builtin_amdgcn_readlane(d[0], k));
#pragma unroll
            for (int r = 1; r < NP; ++r) d[r - 1] = min(d[r], dik + __builtin_amdgcn_readlane(d[r], k));
            d[NP - 1] = x0;
        } else {                                      // padding pivot: rotate only
            const i32 x0 = d[0];
#pragma unroll
            for (int r = 1; r < NP; ++r) d[r - 1] = d[r];
            d[NP - 1] = x0;
        }
    }
    // rows back through LDS: coalesced n x n store, finite pairs and the largest distance on the way
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < NP; r += 4) *(int4*)(lds + rl * LD + r) = make_int4(d[r], d[r + 1], d[r + 2], d[r + 3]);
    __builtin_amdgcn_wave_barrier();
    u32 cnt = 0, mx = 0;
    for (int idx = lane; idx < n * n; idx += 64) {
        const int i = idx / n, j = idx - i * n;
        const i32 x = lds[i * LD + j];
        out[idx] = x;
        if (i != j && x < SP_INF) { ++cnt; mx = (u32)x > mx ? (u32)x : mx; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off, 64);
        const u32 o = __shfl_down(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) {
        *pair_count_g = cnt;
        if (mx) atomicMax(maxd, mx);
    }
}

// ---------------------------------------------------------------------------------------
// The same with 16-bit distances, two columns per register (v_pk_add_u16 / v_pk_min_u16): whenever every finite distance
// stays below 0x3fff -- (n - 1) x the largest weight < 16383, i.e. every unit-weight job -- a relaxation of two columns costs
// one v_readlane + one packed add + one packed min, and a lane can own TWO rows (i and i + 64), which takes graphs of up
// to 128 vertices into registers: the 3 % of the NCI1-like set above 64 vertices cost more in the LDS workgroup kernel
// (one s_barrier per pivot, 166 us) than the other 97 % in the register kernel (97 us).  The rows rotate by one register
// every second pivot (h = k & 1 selects the half of register 0 that holds d[i][k]).
// ---------------------------------------------------------------------------------------
#define SP_PK_INF 0x3fffu
typedef unsigned short v2us __attribute__((ext_vector_type(2)));

// Key marks of the histogram form straight from the packed kernels' LDS matrices (round 6): a job of graphs of at most 128
// vertices whose key space is known beforehand (gk_sp_build: defer_rt) does not run sp_mark_kernel over the matrices it
// has just written -- BASELINE config 4: 30 us of 510.  present == nullptr: no marks.
struct SpMark {
    unsigned char* present; const i32* labels; u64 L, d1; int with_labels;
};
// row i of a packed LDS matrix, columns [j0, j1): sixteen entries per trip, stage by stage; colterm[j] = d1 * label of column j
// (staged in LDS by the caller: the marks of a row are then ONE global round trip per trip, the presence bytes)
__device__ __forceinline__ void sp_pk_mark_row(const unsigned short* __restrict__ ldsrow, int i, int j0, int j1, const u32* colterm, const SpMark& mk) {
    const u32 rowterm = mk.with_labels ? colterm[i] * (u32)mk.L : 0u;
    for (int j = j0; j < j1; j += 16) {
        u32 key[16];
        unsigned char seen[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const u32 x = j + u < j1 ? (u32)ldsrow[j + u] : (u32)SP_PK_INF;
            key[u] = 0xffffffffu;
            if (j + u < j1 && j + u != i && x < (u32)SP_PK_INF) key[u] = rowterm + colterm[j + u] + x;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) seen[u] = key[u] != 0xffffffffu ? mk.present[key[u]] : (unsigned char)1;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (!seen[u]) mk.present[key[u]] = 1;          // same value from every writer
    }
}

template <int NP2, int ROWS>
__device__ __forceinline__ void sp_fw_pk_body(unsigned short* __restrict__ lds, int n, const i32* __restrict__ row_ptr,
                                              const i32* __restrict__ col_idx, const i32* __restrict__ w, i32 v0,
                                              i32* __restrict__ out, u32* __restrict__ pair_count_g, u32* __restrict__ maxd, const SpMark& mk) {
    constexpr int NC = 2 * NP2, LD = NC + 8;           // columns; row pitch in halfwords (16-byte aligned rows)
    constexpr int NR = 64 * ROWS;
    const int lane = threadIdx.x & 63;
    {
        u32* z = (u32*)lds;
        for (int idx = lane; idx < NR * LD / 2; idx += 64) z[idx] = SP_PK_INF | (SP_PK_INF << 16);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
        const int i = lane + 64 * s;
        if (i < n) {
            const i32 e0 = row_ptr[v0 + i], e1 = row_ptr[v0 + i + 1];
            for (i32 e = e0; e < e1; ++e) {
                const int j = col_idx[e] - v0;
                const unsigned short wt = (unsigned short)(w ? w[e] : 1);
                if (wt < lds[i * LD + j]) lds[i * LD + j] = wt;
            }
            lds[i * LD + i] = 0;
        }
    }
    __builtin_amdgcn_wave_barrier();
    u32 d[ROWS][NP2];
#pragma unroll
    for (int s = 0; s < ROWS; ++s)
#pragma unroll
        for (int r = 0; r < NP2; r += 4) {
            const uint4 t = *(const uint4*)(lds + (lane + 64 * s) * LD + 2 * r);
            d[s][r] = t.x, d[s][r + 1] = t.y, d[s][r + 2] = t.z, d[s][r + 3] = t.w;
        }
    auto relax = [](u32 x, u32 dik2, u32 pk) __attribute__((always_inline)) {
        const v2us a = __builtin_bit_cast(v2us, dik2) + __builtin_bit_cast(v2us, pk);
        return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(v2us, x), a));
    };
    // one pivot, in place: PSET = row set that holds the pivot row (its lane kl); d[.][k] is half (k & 1) of register 0
#define SP_PK_RELAX(PSET)                                                                          \
    {                                                                                              \
        u32 dik2[ROWS];                                                                            \
        _Pragma("unroll") for (int s = 0; s < ROWS; ++s) {                                         \
            const u32 h = (d[s][0] >> sh) & 0xffffu;                                               \
            dik2[s] = h | (h << 16);                                                               \
        }                                                                                          \
        _Pragma("unroll") for (int r = 0; r < NP2; ++r) {                                          \
            const u32 pk = (u32)__builtin_amdgcn_readlane((int)d[PSET][r], kl);                    \
            _Pragma("unroll") for (int s = 0; s < ROWS; ++s) d[s][r] = relax(d[s][r], dik2[s], pk); \
        }                                                                                          \
    }
#pragma clang loop unroll(disable)
    for (int k = 0; k < NC; ++k) {
        const int sh = (k & 1) << 4, kl = k & 63;
        if (k < n) {                                   // wave-uniform; padding pivots relax nothing
            if (ROWS == 1 || k < 64) SP_PK_RELAX(0)
            else SP_PK_RELAX(ROWS - 1)
        }
        if (k & 1) {                                   // after every second pivot the rows move one register down
#pragma unroll
            for (int s = 0; s < ROWS; ++s) {
                const u32 x0 = d[s][0];
#pragma unroll
                for (int r = 1; r < NP2; ++r) d[s][r - 1] = d[s][r];
                d[s][NP2 - 1] = x0;
            }
        }
    }
#undef SP_PK_RELAX
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < ROWS; ++s)
#pragma unroll
        for (int r = 0; r < NP2; r += 4)
            *(uint4*)(lds + (lane + 64 * s) * LD + 2 * r) = make_uint4(d[s][r], d[s][r + 1], d[s][r + 2], d[s][r + 3]);
    __builtin_amdgcn_wave_barrier();
    u32 cnt = 0, mx = 0;
    for (int idx = lane; idx < n * n; idx += 64) {
        const int i = idx / n, j = idx - i * n;
        const u32 x = lds[i * LD + j];
        out[idx] = x >= SP_PK_INF ? SP_INF : (i32)x;
        if (i != j && x < SP_PK_INF) { ++cnt; mx = x > mx ? x : mx; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off, 64);
        const u32 o = __shfl_down(mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) {
        *pair_count_g = cnt;
        if (mx) atomicMax(maxd, mx);
    }
    if (mk.present) {
        u32* colterm = (u32*)(lds + NR * LD);          // [NR] behind the wave's matrix (the launch sizes the region for it)
#pragma unroll
        for (int s = 0; s < ROWS; ++s)
            if (lane + 64 * s < n) colterm[lane + 64 * s] = mk.with_labels ? (u32)mk.d1 * (u32)mk.labels[v0 + lane + 64 * s] : 0u;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < ROWS; ++s)
            if (lane + 64 * s < n) sp_pk_mark_row(lds + (lane + 64 * s) * LD, lane + 64 * s, 0, n, colterm, mk);
    }
}

// Graphs of 65..128 vertices: the single-wave form above is one long dependent chain (43 k instructions for n = 110,
// 170 us -- the whole batch waits for its largest graphs).  Here a WORKGROUP of four waves owns the graph and every
// wave a quarter of the COLUMNS for all rows (two rows per lane): the pivot row restricted to the wave's columns sits
// in the wave's own lanes (readlane as before), only the pivot COLUMN d[.][k] has to travel -- its owner writes 128
// halfwords to LDS, one s_barrier, everybody reads its two.  Floyd-Warshall is correct for any order of the pivots, so
// they are taken round-robin over the waves' column slices (wave 0's first pair, wave 1's, ... then every wave rotates
// its registers by one): the pivot column is always register 0 of its owner.
template <int NP2>
__device__ __forceinline__ void sp_fw_pkw_body(unsigned short* __restrict__ lds, int n, const i32* __restrict__ row_ptr,
                                               const i32* __restrict__ col_idx, const i32* __restrict__ w, i32 v0,
                                               i32* __restrict__ out, u32* __restrict__ pair_count_g, u32* __restrict__ maxd, const SpMark& mk) {
    constexpr int NC = 2 * NP2, LD = NC + 8, S = NP2 / 4;      // S registers (column pairs) per wave and row set
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned short* xch = lds + 128 * LD;                       // [2 slots][2 columns][128] pivot-column exchange
    {
        u32* z = (u32*)lds;
        for (int idx = tid; idx < 128 * LD / 2; idx += 256) z[idx] = SP_PK_INF | (SP_PK_INF << 16);
    }
    __syncthreads();
    if (tid < n) {
        const i32 e0 = row_ptr[v0 + tid], e1 = row_ptr[v0 + tid + 1];
        for (i32 e = e0; e < e1; ++e) {
            const int j = col_idx[e] - v0;
            const unsigned short wt = (unsigned short)(w ? w[e] : 1);
            if (wt < lds[tid * LD + j]) lds[tid * LD + j] = wt;
        }
        lds[tid * LD + tid] = 0;
    }
    __syncthreads();
    u32 d[2][S];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < S; ++r) d[s][r] = *(const u32*)(lds + (lane + 64 * s) * LD + 2 * (wv * S + r));
    auto relax = [](u32 x, u32 dik2, u32 pk) __attribute__((always_inline)) {
        const v2us a = __builtin_bit_cast(v2us, dik2) + __builtin_bit_cast(v2us, pk);
        return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(v2us, x), a));
    };
    int slot = 0;
    // Round 6: TWO pivots per barrier.  The two halves of an owner's register 0 are the pivot columns k and k + 1; the owner
    // publishes column k as it is and column k + 1 as it will be AFTER pivot k -- min(d[i][k+1], d[i][k] + d[k][k+1]), everything
    // of which it holds itself -- and every wave relaxes with k, then with k + 1 (whose pivot row it reads from its own registers
    // after the first relaxation).  One exchange and one barrier per column pair instead of per column: the pivots are a chain
    // of (LDS write, barrier, LDS read) round trips, 0.83 us each at config 4.
#pragma clang loop unroll(disable)
    for (int m = 0; m < S; ++m) {
#pragma clang loop unroll(disable)
        for (int ow = 0; ow < 4; ++ow) {                        // owner wave of the pair
            const int k = 2 * (ow * S + m);                     // the columns its register 0 holds: k, k + 1 (workgroup-uniform)
            if (k < n) {
                const int kl = k & 63;                          // row k (and k + 1): lane kl (kl + 1) of row set k / 64
                if (wv == ow) {
                    const u32 dk_k1 = (u32)__builtin_amdgcn_readlane((int)(k < 64 ? d[0][0] : d[1][0]), kl) >> 16;      // d[k][k + 1]
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const u32 h0 = d[s][0] & 0xffffu, h1 = d[s][0] >> 16;
                        const u32 via = h0 + dk_k1;
                        xch[slot * 256 + 64 * s + lane] = (unsigned short)h0;
                        xch[slot * 256 + 128 + 64 * s + lane] = (unsigned short)(via < h1 ? via : h1);
                    }
                }
                __syncthreads();
                u32 dik2[2], dik2b[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const u32 h = xch[slot * 256 + 64 * s + lane], hb = xch[slot * 256 + 128 + 64 * s + lane];
                    dik2[s] = h | (h << 16), dik2b[s] = hb | (hb << 16);
                }
                slot ^= 1;
                if (k < 64) {
#pragma unroll
                    for (int r = 0; r < S; ++r) {
                        const u32 pk = (u32)__builtin_amdgcn_readlane((int)d[0][r], kl);
                        d[0][r] = relax(d[0][r], dik2[0], pk), d[1][r] = relax(d[1][r], dik2[1], pk);
                    }
                    if (k + 1 < n) {
#pragma unroll
                        for (int r = 0; r < S; ++r) {
                            const u32 pk = (u32)__builtin_amdgcn_readlane((int)d[0][r], kl + 1);
                            d[0][r] = relax(d[0][r], dik2b[0], pk), d[1][r] = relax(d[1][r], dik2b[1], pk);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < S; ++r) {
                        const u32 pk = (u32)__builtin_amdgcn_readlane((int)d[1][r], kl);
                        d[0][r] = relax(d[0][r], dik2[0], pk), d[1][r] = relax(d[1][r], dik2[1], pk);
                    }
                    if (k + 1 < n) {
#pragma unroll
                        for (int r = 0; r < S; ++r) {
                            const u32 pk = (u32)__builtin_amdgcn_readlane((int)d[1][r], kl + 1);
                            d[0][r] = relax(d[0][r], dik2b[0], pk), d[1][r] = relax(d[1][r], dik2b[1], pk);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {                           // every wave: one register down
            const u32 x0 = d[s][0];
#pragma unroll
            for (int r = 1; r < S; ++r) d[s][r - 1] = d[s][r];
            d[s][S - 1] = x0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < S; ++r) *(u32*)(lds + (lane + 64 * s) * LD + 2 * (wv * S + r)) = d[s][r];
    __syncthreads();
    u32 cnt = 0, mx = 0;
    for (int idx = tid; idx < n * n; idx += 256) {
        const int i = idx / n, j = idx - i * n;
        const u32 x = lds[i * LD + j];
        out[idx] = x >= SP_PK_INF ? SP_INF : (i32)x;
        if (i != j && x < SP_PK_INF) { ++cnt; mx = x > mx ? x : mx; }
    }
    block_count_max(cnt, mx, pair_count_g, maxd);
    if (mk.present) {                                           // a thread per row and half of its columns
        u32* colterm = (u32*)(xch + 512);                       // [128] behind the exchange area
        if (tid < n) colterm[tid] = mk.with_labels ? (u32)mk.d1 * (u32)mk.labels[v0 + tid] : 0u;
        __syncthreads();
        const int i = tid & 127, half = tid >> 7, mid = (n + 1) >> 1;
        if (i < n) sp_pk_mark_row(lds + i * LD, i, half ? mid : 0, half ? n : mid, colterm, mk);
    }
}

__global__ __launch_bounds__(256) void sp_fw_pkw_kernel(
    const i32* __restrict__ cls_list, i64 n_graphs, int c4, int c5, int c6, int c7, const i32* __restrict__ graph_ptr,
    const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx, const i32* __restrict__ w,
    const u64* __restrict__ dist_ptr, i32* __restrict__ dist, u32* __restrict__ pair_count, u32* __restrict__ maxd, const SpMark mk) {
    extern __shared__ __attribute__((aligned(16))) i32 sp_lds[];
    int gi = blockIdx.x, c = 4;                          // classes 4..7 (80 / 96 / 112 / 128 columns), one workgroup per graph
    if (gi >= c4) { gi -= c4, c = 5; if (gi >= c5) { gi -= c5, c = 6; if (gi >= c6) { gi -= c6, c = 7; if (gi >= c7) return; } } }
    const i32 g = cls_list[(i64)c * n_graphs + gi];
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0;
    i32* out = dist + dist_ptr[g];
    unsigned short* lds = (unsigned short*)sp_lds;
    u32* pc = &pair_count[g];
    switch (c) {
        case 4: sp_fw_pkw_body<40>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
        case 5: sp_fw_pkw_body<48>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
        case 6: sp_fw_pkw_body<56>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
        default: sp_fw_pkw_body<64>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
    }
}

// packed classes: c = 0..3 one row per lane, 16(c+1) columns; c = 4..7 two rows per lane, 80 / 96 / 112 / 128 columns
struct SpPkClasses {
    int first[9];
    int count[8];
};

template <int BIG>
__global__ __launch_bounds__(64 * SP_REG_WAVES) void sp_fw_pk_kernel(
    const SpPkClasses C, const i32* __restrict__ cls_list, i64 n_graphs, const i32* __restrict__ graph_ptr,
    const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx, const i32* __restrict__ w,
    const u64* __restrict__ dist_ptr, i32* __restrict__ dist, u32* __restrict__ pair_count, u32* __restrict__ maxd, const SpMark mk) {
    extern __shared__ __attribute__((aligned(16))) i32 sp_lds[];
    // BIG = 0: the launch covers classes 0..3, BIG = 1: classes 4..7 (their LDS staging areas differ by 4x)
    int c = BIG ? 4 : 0;
    const int c_end = BIG ? 7 : 3;
    while (c < c_end && (int)blockIdx.x >= C.first[c + 1] - C.first[BIG ? 4 : 0]) ++c;
    const int wave = threadIdx.x >> 6;
    const int gi = ((int)blockIdx.x - (C.first[c] - C.first[BIG ? 4 : 0])) * SP_REG_WAVES + wave;
    if (gi >= C.count[c]) return;
    const i32 g = cls_list[(i64)c * n_graphs + gi];
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0;
    i32* out = dist + dist_ptr[g];
    unsigned short* lds = (unsigned short*)sp_lds + (size_t)wave * (BIG ? 128 * 136 + 256 : 64 * 72 + 128);      // matrix + column terms (SpMark)
    u32* pc = &pair_count[g];
    if (!BIG) {
        switch (c) {
            case 0: sp_fw_pk_body<8, 1>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
            case 1: sp_fw_pk_body<16, 1>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
            case 2: sp_fw_pk_body<24, 1>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
            default: sp_fw_pk_body<32, 1>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
        }
    } else {
        switch (c) {
            case 4: sp_fw_pk_body<40, 2>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
            case 5: sp_fw_pk_body<48, 2>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
            case 6: sp_fw_pk_body<56, 2>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
            default: sp_fw_pk_body<64, 2>(lds, n, row_ptr, col_idx, w, v0, out, pc, maxd, mk); break;
        }
    }
}

// packed-kernel classes: 0..3 = (0,16] (16,32] (32,48] (48,64]; 4..7 = (64,80] (80,96] (96,112] (112,128]; 8 = up to the LDS
// cap; 9 = beyond
__global__ void sp_bin_pk_kernel(const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr, i64 n_graphs, int cap, u32* __restrict__ cls_count,
                                 i32* __restrict__ cls_list) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    int c = -1, bc = -1;
    if (g < n_graphs) {
        const int n = graph_ptr[g + 1] - graph_ptr[g];
        c = n <= 0 ? 0 : (n <= 128 ? (n - 1) >> 4 : (n <= cap ? 8 : 9));
        if (c == 9) {
            const int m = row_ptr[graph_ptr[g + 1]] - row_ptr[graph_ptr[g]];
            atomicMax(&cls_count[10], (u32)m);           // most adjacency entries of a large graph
            bc = spb_class(n, m);                         // which breadth-first search kernel takes it (unit weights)
        }
    }
    for (int k = 0; k < 10; ++k) {
        const u64 m = __ballot(c == k);
        if (!m) continue;
        const int lane = threadIdx.x & 63;
        u32 base = 0;
        if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&cls_count[k], (u32)__popcll(m));
        base = __shfl(base, (int)__builtin_ctzll(m), 64);
        if (c == k) cls_list[(i64)k * n_graphs + base + __popcll(m & ((1ull << lane) - 1ull))] = (i32)g;
    }
    for (int k = 0; k < 6; ++k) {                         // lists 10..15, counts 11..16: class 9 again, by search class
        const u64 m = __ballot(bc == k);
        if (!m) continue;
        const int lane = threadIdx.x & 63;
        u32 base = 0;
        if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&cls_count[11 + k], (u32)__popcll(m));
        base = __shfl(base, (int)__builtin_ctzll(m), 64);
        if (bc == k) cls_list[(i64)(10 + k) * n_graphs + base + __popcll(m & ((1ull << lane) - 1ull))] = (i32)g;
    }
}

// Everything a job of at most SP_PREP_MAX_GRAPHS graphs needs before its all-pairs kernels, in ONE single-workgroup launch
// (round 6): the counters cleared, n^2 per graph and its exclusive prefix (the matrices' offsets, their total), the size
// classes of sp_bin_pk_kernel.  BASELINE config 4 queued seven launches of 2-6 us for this (two clears, squares, a two-kernel
// scan, a clear, the binning) behind a host that had just waited for the previous step: ~60 us of the step's 640.
#define SP_PREP_MAX_GRAPHS 65536
__global__ __launch_bounds__(1024) void sp_prep_small_kernel(const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr, i64 n_graphs, int cap,
                                                             u64* __restrict__ dist_ptr, u64* __restrict__ total, u32* __restrict__ pair_count,
                                                             u32* __restrict__ maxd, u32* __restrict__ cls_count, i32* __restrict__ cls_list) {
    __shared__ u64 wsum[16];
    __shared__ u64 carry_s;
    __shared__ u32 cnt_s[32];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid < 32) cnt_s[tid] = 0;
    if (tid == 0) carry_s = 0, *maxd = 0;
    __syncthreads();
    for (i64 g0 = 0; g0 < n_graphs; g0 += 1024) {
        const i64 g = g0 + tid;
        u64 v = 0;
        int c = -1, bc = -1;
        if (g < n_graphs) {
            const int n = graph_ptr[g + 1] - graph_ptr[g];
            v = (u64)n * (u64)n;
            pair_count[g] = 0;
            c = n <= 0 ? 0 : (n <= 128 ? (n - 1) >> 4 : (n <= cap ? 8 : 9));
            if (c == 9) {
                const int m = row_ptr[graph_ptr[g + 1]] - row_ptr[graph_ptr[g]];
                atomicMax(&cnt_s[10], (u32)m);
                bc = spb_class(n, m);
            }
        }
        u64 incl = v;                                        // inclusive scan of the chunk
        for (int off = 1; off < 64; off <<= 1) {
            const u64 o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        u64 before = carry_s;
        for (int k = 0; k < w; ++k) before += wsum[k];
        if (g < n_graphs) dist_ptr[g] = before + incl - v;
        // the classes: slots from the workgroup's LDS counters (one atomic per wave and class)
        for (int k = 0; k < 10; ++k) {
            const u64 m = __ballot(c == k);
            if (!m) continue;
            u32 base = 0;
            if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&cnt_s[k], (u32)__popcll(m));
            base = __shfl(base, (int)__builtin_ctzll(m), 64);
            if (c == k) cls_list[(i64)k * n_graphs + base + __popcll(m & ((1ull << lane) - 1ull))] = (i32)g;
        }
        for (int k = 0; k < 6; ++k) {
            const u64 m = __ballot(bc == k);
            if (!m) continue;
            u32 base = 0;
            if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&cnt_s[11 + k], (u32)__popcll(m));
            base = __shfl(base, (int)__builtin_ctzll(m), 64);
            if (bc == k) cls_list[(i64)(10 + k) * n_graphs + base + __popcll(m & ((1ull << lane) - 1ull))] = (i32)g;
        }
        __syncthreads();
        if (tid == 0) {
            u64 t = carry_s;
            for (int k = 0; k < 16; ++k) t += wsum[k];
            carry_s = t;
        }
        __syncthreads();
    }
    if (tid == 0) *total = carry_s;
    if (tid < 32) cls_count[tid] = cnt_s[tid];
}

// pair counts -> pair ranges of a job of at most SP_PREP_MAX_GRAPHS graphs: out[0 .. n] (exclusive prefix, the total at out[n]
// and at *total) in one single-workgroup launch instead of a two-kernel scan and a 4-byte device copy
__global__ __launch_bounds__(1024) void sp_pair_ranges_small_kernel(const u32* __restrict__ cnt, u32* __restrict__ out, i64 n, u32* __restrict__ total) {
    __shared__ u32 wsum[16];
    __shared__ u32 carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (i64 g0 = 0; g0 < n; g0 += 1024) {
        const i64 g = g0 + tid;
        const u32 v = g < n ? cnt[g] : 0u;
        u32 incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const u32 o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        u32 before = carry_s;
        for (int k = 0; k < w; ++k) before += wsum[k];
        if (g < n) out[g] = before + incl - v;
        __syncthreads();
        if (tid == 0) {
            u32 t = carry_s;
            for (int k = 0; k < 16; ++k) t += wsum[k];
            carry_s = t;
        }
        __syncthreads();
    }
    if (tid == 0) out[n] = carry_s, *total = carry_s;
}

// class c = 1..8: graphs of (8(c-1), 8c] vertices, listed in cls_list[(c-1) * n_graphs ..); blocks are dealt to the
// classes by the prefix cls_first[] (workgroups per class)
struct SpClasses {
    int first[9];          // first workgroup of class c (index c - 1), [8] = total
    int count[8];
};

__global__ __launch_bounds__(64 * SP_REG_WAVES) void sp_fw_reg_kernel(
    const SpClasses C, const i32* __restrict__ cls_list, i64 n_graphs, const i32* __restrict__ graph_ptr,
    const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx, const i32* __restrict__ w,
    const u64* __restrict__ dist_ptr, i32* __restrict__ dist, u32* __restrict__ pair_count, u32* __restrict__ maxd) {
    extern __shared__ __attribute__((aligned(16))) i32 sp_lds[];
    int c = 0;
    while (c < 7 && (int)blockIdx.x >= C.first[c + 1]) ++c;
    const int wave = threadIdx.x >> 6;
    const int gi = ((int)blockIdx.x - C.first[c]) * SP_REG_WAVES + wave;
    if (gi >= C.count[c]) return;
    const i32 g = cls_list[(i64)c * n_graphs + gi];
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0;
    i32* out = dist + dist_ptr[g];
    i32* lds = sp_lds + (size_t)wave * (SP_REG_MAX_N * (SP_REG_MAX_N + 4));
    switch (c) {
        case 0: sp_fw_reg_body<8>(lds, n, row_ptr, col_idx, w, v0, out, &pair_count[g], maxd); break;
        case 1: sp_fw_reg_body<16>(lds, n, row_ptr, col_idx, w, v0, out, &pair_count[g], maxd); break;
        case 2: sp_fw_reg_body<24>(lds, n, row_ptr, col_idx, w, v0, out, &pair_count[g], maxd); break;
        case 3: sp_fw_reg_body<32>(lds, n, row_ptr, col_idx, w, v0, out, &pair_count[g], maxd); break;
        case 4: sp_fw_reg_body<40>(lds, n, row_ptr, col_idx, w, v0, out, &pair_count[g], maxd); break;
        case 5: sp_fw_reg_body<48>(lds, n, row_ptr, col_idx, w, v0, out, &pair_count[g], maxd); break;
        case 6: sp_fw_reg_body<56>(lds, n, row_ptr, col_idx, w, v0, out, &pair_count[g], maxd); break;
        default: sp_fw_reg_body<64>(lds, n, row_ptr, col_idx, w, v0, out, &pair_count[g], maxd); break;
    }
}

// size classes of the graphs: lists for the register kernel (classes 0..7), the LDS kernel (8: 64 < n <= cap) and the
// row-relaxation kernel (9: n > cap).  cls_count[10]; lists are n_graphs apart.  Wave-aggregated appends.
__global__ void sp_bin_kernel(const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr, i64 n_graphs, int cap, u32* __restrict__ cls_count,
                              i32* __restrict__ cls_list) {
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    int c = -1, bc = -1;
    if (g < n_graphs) {
        const int n = graph_ptr[g + 1] - graph_ptr[g];
        c = n <= 0 ? 0 : (n <= SP_REG_MAX_N ? (n - 1) >> 3 : (n <= cap ? 8 : 9));
        if (c == 9) {
            const int m = row_ptr[graph_ptr[g + 1]] - row_ptr[graph_ptr[g]];
            atomicMax(&cls_count[10], (u32)m);           // most adjacency entries of a large graph
            bc = spb_class(n, m);                         // which breadth-first search kernel takes it (unit weights)
        }
    }
    for (int k = 0; k < 10; ++k) {
        const u64 m = __ballot(c == k);
        if (!m) continue;
        const int lane = threadIdx.x & 63;
        u32 base = 0;
        if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&cls_count[k], (u32)__popcll(m));
        base = __shfl(base, (int)__builtin_ctzll(m), 64);
        if (c == k) cls_list[(i64)k * n_graphs + base + __popcll(m & ((1ull << lane) - 1ull))] = (i32)g;
    }
    for (int k = 0; k < 6; ++k) {                         // lists 10..15, counts 11..16: class 9 again, by search class
        const u64 m = __ballot(bc == k);
        if (!m) continue;
        const int lane = threadIdx.x & 63;
        u32 base = 0;
        if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&cls_count[11 + k], (u32)__popcll(m));
        base = __shfl(base, (int)__builtin_ctzll(m), 64);
        if (bc == k) cls_list[(i64)(10 + k) * n_graphs + base + __popcll(m & ((1ull << lane) - 1ull))] = (i32)g;
    }
}

// local source vertex of every adjacency entry (the row index of the CSR entry inside its graph): what makes the
// relaxation below edge-parallel
__global__ void sp_edge_src_kernel(const i32* __restrict__ graph_ptr, const i32* __restrict__ node_graph,
                                   const i32* __restrict__ row_ptr, i32* __restrict__ esrc, i64 V) {
    const i64 v = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const i32 u = (i32)v - graph_ptr[node_graph[v]];
    for (i32 e = row_ptr[v]; e < row_ptr[v + 1]; ++e) esrc[e] = u;
}

// Graphs larger than the Floyd-Warshall LDS cap: grid (graphs of size class 9, max_n), block (g, src) relaxes the distance
// row of ONE source in LDS to its fixed point.  Round 5: the sweeps run over the graph's ADJACENCY ENTRIES (a thread per
// entry) instead of over its vertices (a thread per vertex walking its neighbour list): a hub of 2 500 neighbours was one
// thread's 2 500-trip loop in every sweep of every source -- 155 ms on the REDDIT-like set.
__global__ __launch_bounds__(SP_THREADS) void sp_relax_kernel(
    const i32* __restrict__ big_list, const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx,
    const i32* __restrict__ esrc, const i32* __restrict__ w, const u64* __restrict__ dist_ptr, i32* __restrict__ dist,
    u32* __restrict__ pair_count, u32* __restrict__ maxd, int cap) {
    extern __shared__ __attribute__((aligned(16))) i32 row[];
    __shared__ int changed;
    const int g = big_list[blockIdx.x], src = blockIdx.y, tid = threadIdx.x;
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0;
    if (n <= cap || src >= n) return;
    for (int i = tid; i < n; i += SP_THREADS) row[i] = i == src ? 0 : SP_INF;
    const i32 e0 = row_ptr[v0];
    const int m = row_ptr[v0 + n] - e0;
    __syncthreads();
    for (int sweep = 0; sweep < n; ++sweep) {
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int e = tid; e < m; e += SP_THREADS) {
            const i32 du = row[esrc[e0 + e]];
            if (du < SP_INF) {
                const int v = col_idx[e0 + e] - v0;
                const i32 nd = du + (w ? w[e0 + e] : 1);
                if (nd < row[v]) { atomicMin(&row[v], nd); changed = 1; }
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

struct SpDist {
    Tmp<u64> sq, dist_ptr, total;
    Tmp<i32> dist, wdev, esrc;
    Tmp<u32> pair_count, maxd;
    explicit SpDist(gk_ctx* c) : sq(c), dist_ptr(c), total(c), dist(c), wdev(c), esrc(c), pair_count(c), maxd(c) {}
};

// Unit weights, graphs above the Floyd-Warshall LDS cap (round 5): BIT-PARALLEL breadth-first search, G targets at a time.
// Grid (graph of size class 9, group of G columns).  Bit t of visit[u] says "u reaches column gbase + t"; a sweep is
//     next[u] = OR over the out-neighbours v of u of front[v],  minus what u reached before
// -- a pull over u's own adjacency row, no atomics -- and what is new at sweep k is d[u][gbase + t] = k.  One machine word
// does the work of G row relaxations: the relaxation kernel below sweeps all m adjacency entries once per source and
// distance level (REDDIT-like: 13 levels x 2.3 n entries per source, 11.5 ms; D&D-like: 26 x 5 n, 13.3 ms).  A thread owns
// up to SPB_VPT vertices of a degree up to SPB_HUB_DEG; hubs (a thread with 2 500 answers to one user) are OR-reduced by a
// wave each.
// The distances of the group are collected in LDS as BYTES (n x G) and leave as whole row segments when the search is
// over: written level by level straight to HBM every 256-byte segment was touched three or four times with a few lanes
// each, far apart in time -- 35 M partial stores for 2.6 GB of matrix (5.9 ms, REDDIT-like).  n x G bytes + two words per
// vertex bound the group width: G = 64, 32 or 16 columns by graph size (spb_class; the relaxation kernel beyond 6 352 vertices).
// A search that would need level 255 raises maxd to 0xffffffff: the host repeats the job with the relaxation kernel.
// The graph's adjacency entries (local column numbers, 16 bits) are staged in LDS too when there is room (cols_cap).
// Size classes (threads, vertices per thread, columns per word).  A graph takes the first class whose LDS holds its
// vertex arrays AND its adjacency entries (without them a hub's wave walks 24 dependent L2 round trips per sweep while
// fifteen waves wait at the barrier: 215 us per workgroup on a 3 000-vertex thread, 30 us with the entries in LDS); the
// workgroups of the small classes share a CU (one class for everything up to 1 907 vertices left the many 300-vertex
// graphs of a set with one workgroup of mostly idle threads per CU).
//   class 0: n <= 512,  64 columns, 512 threads,  42 KiB (three per CU)     class 3: n <= 4 096, 32 columns, 149 KiB
//   class 1: n <= 1 024, 64 columns, 1 024 threads, 69 KiB (two per CU)     class 4: 20 n bytes <= 149 KiB, 16 columns; the
//   class 2: n <= 2 048, 64 columns, 149 KiB                                         entries in LDS only if they fit
//   class 5: the relaxation kernel
#ifdef GK_ABLATION
// tools' build only: cycles per phase of sp_msbfs_kernel summed over all workgroups of a class (thread 0's stamps):
// [class][0] staging, [1] set-up, [2] pull, [3] update, [4] epilogue, [5] workgroups, [6] levels; tools/dev/msbfs_times.py
__device__ unsigned long long g_spb_dbg[6][8];
#define SPB_DBG_DECL unsigned long long t_x = __builtin_readcyclecounter(), t_ph[5] = {0, 0, 0, 0, 0};
#define SPB_DBG(k) { if (threadIdx.x == 0) { const unsigned long long t_y = __builtin_readcyclecounter(); t_ph[k] += t_y - t_x; t_x = t_y; } }
#define SPB_DBG_OUT(cls, levels) { if (threadIdx.x == 0) { for (int q_ = 0; q_ < 5; ++q_) atomicAdd(&g_spb_dbg[cls][q_], t_ph[q_]); atomicAdd(&g_spb_dbg[cls][5], 1ull); atomicAdd(&g_spb_dbg[cls][6], (unsigned long long)(levels)); } }
extern "C" int gk_debug_spb_times(gk_ctx* ctx, unsigned long long* out48, int reset) {
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    GK_HIP_CHECK(hipDeviceSynchronize());
    GK_HIP_CHECK(hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_spb_dbg), sizeof(unsigned long long) * 48));
    if (reset) { unsigned long long z[48] = {0}; GK_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_spb_dbg), z, sizeof(z))); }
    return GK_OK;
}
#else
#define SPB_DBG_DECL
#define SPB_DBG(k)
#define SPB_DBG_OUT(cls, levels)
#endif
template <typename W, int G, int NT, int VPT, int CLS>
__global__ __launch_bounds__(NT) void sp_msbfs_kernel(
    const i32* __restrict__ big_list, const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr, const i32* __restrict__ col_idx,
    u64* dist_ptr, i32* __restrict__ dist, u32* __restrict__ pair_count, u32* __restrict__ maxd, int n_lo,
    int lds_bytes, int use_cols, int bytes_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char spb_raw[];   // visit[n8] | front[n8] | d8[n8][G] | cols[m]
    __shared__ u64 hub_nx[SPB_HUB_CAP];
    __shared__ i32 hub_id[SPB_HUB_CAP], hub_e0[SPB_HUB_CAP], hub_dg[SPB_HUB_CAP];
    __shared__ u32 n_hubs_s;
    const int g = big_list[blockIdx.x], tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0;
    const int gbase = blockIdx.y * G;
    if (n <= n_lo || gbase >= n) return;
    const i32 eg = row_ptr[v0];
    const int m = row_ptr[v0 + n] - eg;
    if (spb_class(n, m) != CLS) return;
    constexpr int GS = G + 4;                                         // row stride of d8: lanes = consecutive vertices, and a stride of
    const int n_words = (n + 7) & ~7;                                 // 64 bytes put all of them on two LDS banks
    W* visit = (W*)spb_raw;
    W* front = visit + n_words;
    unsigned char* d8 = (unsigned char*)(front + n_words);
    unsigned short* cols = (unsigned short*)(d8 + (size_t)n_words * GS);
    const bool in_lds = use_cols && (long)n_words * (2 * (long)sizeof(W) + GS) + 2l * m <= (long)lds_bytes;   // workgroup-uniform
    SPB_DBG_DECL
    if (in_lds)
        for (int e = tid; e < m; e += NT) cols[e] = (unsigned short)(col_idx[eg + e] - v0);
    for (int q = tid; q < n * (GS / 4); q += NT) ((u32*)d8)[q] = 0xffffffffu;   // 255: not reached
    if (tid == 0) n_hubs_s = 0;
    __syncthreads();
    SPB_DBG(0)
    i32 e0[VPT], dg[VPT];                                     // dg < 0: not this thread's to pull (beyond n, or a listed hub)
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int u = tid + k * NT;
        e0[k] = 0, dg[k] = -1;
        if (u < n) {
            e0[k] = row_ptr[v0 + u] - eg;
            int d = row_ptr[v0 + u + 1] - eg - e0[k];
            if (d > SPB_HUB_DEG) {
                const u32 slot = atomicAdd(&n_hubs_s, 1u);
                if (slot < (u32)SPB_HUB_CAP) hub_id[slot] = u, hub_e0[slot] = e0[k], hub_dg[slot] = d, d = -1;
            }
            dg[k] = d;
            W bit = 0;
            if (u >= gbase && u < gbase + G) bit = (W)((W)1 << (u - gbase)), d8[u * GS + (u - gbase)] = 0;
            visit[u] = bit, front[u] = bit;
        }
    }
    __syncthreads();
    const int n_hubs = n_hubs_s < (u32)SPB_HUB_CAP ? (int)n_hubs_s : SPB_HUB_CAP;
    const i32* cg = col_idx + eg;
    u32 cnt = 0, mx = 0;
    SPB_DBG(1)
    for (int level = 1;; ++level) {
        W nx[VPT];
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            nx[k] = 0;
            if (dg[k] >= 0) {
                W acc = 0;
                if (in_lds)
                    for (int e = 0; e < dg[k]; ++e) acc |= front[cols[e0[k] + e]];
                else
                    for (int e = 0; e < dg[k]; ++e) acc |= front[cg[e0[k] + e] - v0];
                nx[k] = (W)(acc & ~visit[tid + k * NT]);
            }
        }
        for (int h = w; h < n_hubs; h += NT / 64) {
            const int u = hub_id[h], eh = hub_e0[h], dh = hub_dg[h];
            u64 acc = 0;
            if (in_lds) {
                int e = lane;
                for (; e + 192 < dh; e += 256)                        // four independent LDS chains in flight
                    acc |= (u64)(front[cols[eh + e]] | front[cols[eh + e + 64]] | front[cols[eh + e + 128]] | front[cols[eh + e + 192]]);
                for (; e < dh; e += 64) acc |= (u64)front[cols[eh + e]];
            } else
                for (int e = lane; e < dh; e += 64) acc |= (u64)front[cg[eh + e] - v0];
            for (int off = 32; off > 0; off >>= 1) acc |= __shfl_xor(acc, off, 64);
            if (lane == 0) hub_nx[h] = acc & ~(u64)visit[u];
        }
        __syncthreads();                                              // every read of front[] is done
        SPB_DBG(2)
        int any = 0;
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            const int u = tid + k * NT;
            if (dg[k] >= 0) {
                front[u] = nx[k];
                if (nx[k]) {
                    const int pc = __popcll((u64)nx[k]);
                    visit[u] |= nx[k], any = 1, cnt += (u32)pc;
                    u64 x = (u64)nx[k];
                    if (pc <= SPB_BITWISE_MAX) {
                        while (x) {
                            const int t = __ffsll((unsigned long long)x) - 1;
                            x &= x - 1;
                            d8[u * GS + t] = (unsigned char)level;
                        }
                    } else {
                        // many columns reach the vertex in the same sweep (a thread with a hub: nearly every vertex two steps
                        // from nearly every column): the level goes into the row four bytes at a time -- the group's four bits
                        // spread to a byte mask -- instead of one byte store and ten instructions per bit (the update was the
                        // longer half of a sweep: 5 800 of 9 800 cycles per level on the REDDIT-like set)
                        u32* row32 = (u32*)(d8 + u * GS);
                        const u32 lv4 = (u32)level * 0x01010101u;
#pragma unroll
                        for (int q = 0; q < G / 4; ++q) {
                            const u32 b4 = (u32)(x >> (4 * q)) & 15u;
                            if (b4) {
                                const u32 m = ((b4 * 0x00204081u) & 0x01010101u) * 0xffu;
                                row32[q] = (row32[q] & ~m) | (lv4 & m);
                            }
                        }
                    }
                }
            }
        }
        for (int h = w; h < n_hubs; h += NT / 64) {
            const u64 x = hub_nx[h];
            const int u = hub_id[h];
            if (lane == 0) {
                front[u] = (W)x;
                if (x) visit[u] |= (W)x, any = 1, cnt += (u32)__popcll(x);
            }
            if (lane < G && ((x >> lane) & 1ull)) d8[u * GS + lane] = (unsigned char)level;
        }
        const int more = __syncthreads_or(any);
        SPB_DBG(3)
        if (!more) break;
        mx = (u32)level;
        if (level == 254) {                                           // level 255 is the "not reached" byte
            if (tid == 0) atomicMax(maxd, SPB_OVERFLOW);
            return;
        }
    }
    // the group's columns of every row: whole segments, once
    const u64 moff = dist_ptr[g] & ~SP_BYTE_FLAG;                      // (workgroup y == 0 sets the flag while the others read)
    i32* dgm = dist + moff;
    if (bytes_out) {
        // round 6: the matrix stays BYTES (common.h: SpMat) -- a quarter of the store here and of the two passes that read it
        // (key marks, counting).  Four columns per lane: G / 4 lanes per row, 256 / G rows per wave and trip.
        unsigned char* d8m = (unsigned char*)(((uintptr_t)dgm + 15) & ~(uintptr_t)15);
        const int ns = (n + 15) & ~15;
        constexpr int LPR = G / 4, RPW = 64 / LPR;
        const int q = lane % LPR, col = gbase + 4 * q;
        for (int u = w * RPW + lane / LPR; u < n; u += (NT / 64) * RPW)
            if (col < ns) *(u32*)(d8m + (size_t)u * ns + col) = *(const u32*)(d8 + u * GS + 4 * q);
        if (blockIdx.y == 0 && tid == 0) dist_ptr[g] = moff | SP_BYTE_FLAG;
    } else {
        constexpr int RPW = 64 / G;                                   // rows per wave and trip
        const int t = lane % G, col = gbase + t;
        for (int u = w * RPW + lane / G; u < n; u += (NT / 64) * RPW) {
            if (col < n) {
                const unsigned char x = d8[u * GS + t];
                dgm[(size_t)u * n + col] = x == 255 ? SP_INF : (i32)x;
            }
        }
    }
    block_count_max(cnt, mx, &pair_count[g], maxd);
    SPB_DBG(4)
    SPB_DBG_OUT(CLS, mx)
}

template <typename W, int G, int NT, int VPT, int CLS>
static int sp_msbfs_launch(gk_ctx* ctx, gk_batch* b, SpDist& s, const i32* list, u32 n_list, int cap, int n_hi, int nmax, int lds,
                           hipStream_t st) {
    const int hi = nmax < n_hi ? nmax : n_hi;
    auto kern = sp_msbfs_kernel<W, G, NT, VPT, CLS>;
    GK_TRY(gk_func_lds(ctx, (const void*)kern, lds));
    kern<<<dim3(n_list, (unsigned)cdiv(hi, G)), NT, (size_t)lds, st>>>(
        list, b->graph_ptr, b->row_ptr, b->col_idx, s.dist_ptr.p, s.dist.p, s.pair_count.p, s.maxd.p, cap, lds,
        ctx->opt.sp_bfs_no_lds_cols ? 0 : 1, ctx->opt.sp_bfs_no_bytes ? 0 : 1);
    return GK_OK;
}

// Pair items / key marks of a graph in SLABS of SP_SLAB rows, grid (graph, slab): a 5 748-vertex graph (D&D has one) is 33 M
// pairs -- one workgroup walking them alone took 90-120 ms per pass (round 5: 90 workgroups).  Item slots: a slab counts its
// finite pairs, reserves its range in the graph's item range with ONE atomic on the graph's cursor, and numbers its items
// inside (the order of the items of a graph is immaterial: they are sorted by key afterwards).
#define SP_SLAB 64
#define SPM_SEEN_BITS 11
#define SPM_SEEN (1 << SPM_SEEN_BITS)      // sp_mark_kernel: keys of the workgroup's LDS cache
#define SPM_COLS 4096                      // ... and column terms staged in LDS up to this many vertices
__global__ __launch_bounds__(SP_THREADS) void sp_emit_kernel(
    const i32* __restrict__ graph_ptr, const i32* __restrict__ node_label, const u64* __restrict__ dist_ptr,
    const i32* __restrict__ dist, const u32* __restrict__ pair_base, u64* __restrict__ keys,
    i32* __restrict__ item_graph, u64 n_labels, u64 d1, int with_labels, u32* __restrict__ graph_cursor) {
    __shared__ u32 cursor, wsum[SP_THREADS / 64];
    const int g = blockIdx.x, tid = threadIdx.x;
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0;
    const int r0 = blockIdx.y * SP_SLAB;
    if (r0 >= n) return;
    const int r1 = r0 + SP_SLAB < n ? r0 + SP_SLAB : n;
    const SpMat M = sp_mat(dist, dist_ptr, g, n);
    const i64 lo = (i64)r0 * n, hi = (i64)r1 * n;
    u32 mine = 0;
    for (i64 idx = lo + tid; idx < hi; idx += SP_THREADS) {
        const int i = (int)(idx / n), j = (int)(idx - (i64)i * n);
        if (i != j && sp_mat_at(M, n, i, j, SP_INF) < SP_INF) ++mine;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) {
        u32 t = 0;
        for (int k = 0; k < SP_THREADS / 64; ++k) t += wsum[k];
        cursor = t ? atomicAdd(&graph_cursor[g], t) : 0u;
    }
    __syncthreads();
    const u32 base = pair_base[g];
    for (i64 idx0 = lo; idx0 < hi; idx0 += SP_THREADS) {
        const i64 idx = idx0 + tid;
        bool ok = false;
        u64 key = 0;
        if (idx < hi) {
            const int i = (int)(idx / n), j = (int)(idx - (i64)i * n);
            const i32 x = sp_mat_at(M, n, i, j, SP_INF);
            if (i != j && x < SP_INF) {
                ok = true;
                key = (u64)x;
                if (with_labels)
                    key += d1 * ((u64)(u32)node_label[v0 + i] * n_labels + (u64)(u32)node_label[v0 + j]);
            }
        }
        const u64 mask = __ballot(ok);               // one LDS atomic per wave and trip
        u32 wbase = 0;
        const int lane = tid & 63;
        if (lane == 0 && mask) wbase = atomicAdd(&cursor, (u32)__popcll(mask));
        wbase = __shfl(wbase, 0, 64);
        if (ok) {
            const u32 slot = base + wbase + (u32)__popcll(mask & ((1ull << lane) - 1ull));
            keys[slot] = key;
            item_graph[slot] = g;
        }
    }
}

// the row loop of sp_mark_kernel (a wave per matrix row), for 32-bit and for byte matrices (B8: the form is a template
// parameter -- a branch inside the unrolled stages keeps the compiler from batching the loads)
template <bool B8>
__device__ __forceinline__ void spm_mark_rows(const SpMat& M, int n, i32 v0, int r0, int r1, const i32* __restrict__ node_label,
                                              unsigned char* __restrict__ present, u64 n_labels, u64 d1, int with_labels, bool col_in_lds,
                                              const u32* colt_s, u32* seen_s) {
    const int tid = threadIdx.x, lane = tid & 63;
    const i32* dg = M.d32;
    // FOUR neighbouring rows per wave and trip (round 6): eight entries per lane are two columns of four rows -- one column
    // term per four entries instead of one each (the loop is bound by its LDS reads and the chain distance -> key -> cache)
    constexpr int RPW = 4, NCS = 2;
    for (int ib = r0 + RPW * (tid >> 6); ib < r1; ib += RPW * (SP_THREADS / 64)) {
        int ri[RPW];
        u32 rowterm[RPW], last[RPW];
        const i32* dr[RPW];
        const unsigned char* dr8[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int i = ib + q < r1 ? ib + q : -1;
            ri[q] = i, last[q] = 0xffffffffu;
            rowterm[q] = (with_labels && i >= 0) ? (u32)(d1 * (u64)(u32)node_label[v0 + i] * n_labels) : 0u;
            dr[q] = dg + (size_t)(i >= 0 ? i : 0) * n;
            dr8[q] = M.d8 + (size_t)(i >= 0 ? i : 0) * M.ns;
        }
        // eight entries per lane and trip, every stage for all eight before the next: the loop is a chain of dependent
        // round trips (distance -> label -> presence byte) and one entry per trip left the memory system idle (2.7 ms)
        for (int j0 = 0; j0 < n; j0 += NCS * 64) {
            i32 x[8];
            u32 lj[NCS], key[8], hs[8], cached[8];
            unsigned char seen[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                 // entry u: row u % RPW, column step u / RPW
                const int q = u % RPW, j = j0 + (u / RPW) * 64 + lane;
                const bool have = ri[q] >= 0 && j < n;
                if (B8) {
                    const unsigned char b8 = have ? dr8[q][j] : (unsigned char)255;
                    x[u] = b8 == 255 ? SP_INF : (i32)b8;
                } else x[u] = have ? dr[q][j] : SP_INF;
            }
#pragma unroll
            for (int cs = 0; cs < NCS; ++cs) {
                const int j = j0 + cs * 64 + lane;
                lj[cs] = j < n ? (col_in_lds ? colt_s[j] : (with_labels ? (u32)d1 * (u32)node_label[v0 + j] : 0u)) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = u % RPW, j = j0 + (u / RPW) * 64 + lane;
                key[u] = 0xffffffffu;
                if (j != ri[q] && x[u] < SP_INF) {
                    const u32 k = rowterm[q] + lj[u / RPW] + (u32)x[u];
                    if (k != last[q]) key[u] = k, last[q] = k;     // (a lane does not look up the key it marked last in this row)
                }
                hs[u] = (key[u] * 2654435761u) >> (32 - SPM_SEEN_BITS);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) cached[u] = key[u] != 0xffffffffu ? seen_s[hs[u]] : 0xffffffffu;
#pragma unroll
            for (int u = 0; u < 8; ++u) seen[u] = cached[u] != key[u] ? present[key[u]] : (unsigned char)1;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (cached[u] != key[u]) {
                    if (!seen[u]) present[key[u]] = 1;  // same value from every writer
                    seen_s[hs[u]] = key[u];             // (any racing writer leaves a key that has been marked)
                }
        }
    }
}

// histogram form of the pair batch (features_gm.hip: gk_features_build_sp): which keys occur at all ...
__global__ __launch_bounds__(SP_THREADS) void sp_mark_kernel(
    const i32* __restrict__ graph_ptr, const i32* __restrict__ node_label, const u64* __restrict__ dist_ptr,
    const i32* __restrict__ dist, unsigned char* __restrict__ present, u64 n_labels, u64 d1, int with_labels) {
    const int g = blockIdx.x, tid = threadIdx.x;
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0;
    const int r0 = blockIdx.y * SP_SLAB;
    if (r0 >= n) return;
    const int r1 = r0 + SP_SLAB < n ? r0 + SP_SLAB : n;
    const SpMat M = sp_mat(dist, dist_ptr, g, n);
    const i32* dg = M.d32;
    if (n >= 256 || M.d8) {
        // a wave per matrix row (round 5): no division per entry, the row's label term once per row, and a lane does not
        // look up a key it has just marked (a thread with a hub: most of a row is one key).  Byte matrices (round 6) are read
        // with the same lane <-> column mapping, a byte per lane: eight consecutive entries per lane from one 8-byte load put
        // the label loads of a wave on sixteen cache lines instead of two and cost more than the bytes saved (2.35 vs 1.48 ms).
        // Round 6: the slab's 64 n entries are a few hundred distinct keys (REDDIT-like: at most 544 per GRAPH), and the
        // presence bytes are a scattered gather per entry -- a direct-mapped LDS cache of keys this workgroup has marked
        // answers nearly all of them (a lane that finds its key there skips the gather); the column terms d1 * label of the
        // graph are staged in LDS once per workgroup instead of being fetched for every row.
        __shared__ u32 seen_s[SPM_SEEN], colt_s[SPM_COLS];
        const bool col_in_lds = n <= SPM_COLS;
        for (int t = tid; t < SPM_SEEN; t += SP_THREADS) seen_s[t] = 0xffffffffu;          // (key space <= 2^26: never a key)
        if (col_in_lds)
            for (int j = tid; j < n; j += SP_THREADS) colt_s[j] = with_labels ? (u32)d1 * (u32)node_label[v0 + j] : 0u;
        __syncthreads();
        if (M.d8) spm_mark_rows<true>(M, n, v0, r0, r1, node_label, present, n_labels, d1, with_labels, col_in_lds, colt_s, seen_s);
        else spm_mark_rows<false>(M, n, v0, r0, r1, node_label, present, n_labels, d1, with_labels, col_in_lds, colt_s, seen_s);
        return;
    }
    for (i64 idx = (i64)r0 * n + tid; idx < (i64)r1 * n; idx += SP_THREADS) {
        const int i = (int)(idx / n), j = (int)(idx - (i64)i * n);
        const i32 x = dg[idx];
        if (i != j && x < SP_INF) {
            u64 key = (u64)x;
            if (with_labels) key += d1 * ((u64)(u32)node_label[v0 + i] * n_labels + (u64)(u32)node_label[v0 + j]);
            if (!present[key]) present[key] = 1;           // same value from every writer
        }
    }
}

// ... and their dense ids: exclusive prefix of the presence bytes
struct SpIdScan {
    const unsigned char* present; u32* idtab; u32* n_keys;
    __device__ __forceinline__ u32 value(i64 k) const { return present[k] ? 1u : 0u; }
    __device__ __forceinline__ void emit(i64 k, u32 v, u32 incl) const { idtab[k] = v ? incl - 1u : 0xffffffffu; }
    __device__ __forceinline__ void finish(u32 total) const { *n_keys = total; }
    __device__ __forceinline__ i64 seg_first_tile(i64) const { return 0; }
};

static int bits_for64(u64 v) {
    int b = 0;
    while (b < 64 && (v >> b)) ++b;
    return b;
}


static int sp_compute_dist(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, SpDist& s, u64* total_sq, bool no_bfs = false,
                           const SpMark* mark = nullptr, bool* marked = nullptr) {
    const i64 N = b->n_graphs;
    GK_TRY(s.sq.alloc(N)); GK_TRY(s.dist_ptr.alloc(N)); GK_TRY(s.total.alloc(1));
    GK_TRY(s.pair_count.alloc(N)); GK_TRY(s.maxd.alloc(1));
    // 16-bit packed registers whenever every finite distance of a graph of up to 128 vertices stays below 0x3fff
    i64 wmax = 1;
    if (edge_weight)
        for (i64 e = 0; e < b->n_edges; ++e) wmax = edge_weight[e] > wmax ? edge_weight[e] : wmax;
    const bool use_pk = wmax <= 128 && !ctx->opt.sp_no_reg && !ctx->opt.sp_no_pk;
    const bool one_launch = use_pk && N <= SP_PREP_MAX_GRAPHS && !ctx->opt.sp_no_prep;      // sp_prep_small_kernel
    if (!one_launch) {
        GK_TRY(gk_zero_async(ctx, s.pair_count.p, (size_t)N * 4));
        GK_TRY(gk_zero_async(ctx, s.maxd.p, 4));
        sp_sq_kernel<<<grid_for(N, 256), 256, 0, ctx->stream>>>(b->graph_ptr, s.sq.p, N);
        GK_TRY(gk_scan_u64(ctx, s.sq.p, s.dist_ptr.p, N, true, s.total.p));
    }
    // size classes (one wave per graph in registers up to 64 vertices, the LDS workgroup form up to the LDS cap, row
    // relaxation beyond): binned on the device, counts read back with the matrix total
    // unit weights: everything above the packed register kernels (128 vertices) goes to the breadth-first search -- the
    // Floyd-Warshall workgroup kernel spent 1.5 ms on the 130..202-vertex graphs of the REDDIT-like set (n^3 each)
    const bool bfs = !edge_weight && !no_bfs && !ctx->opt.sp_no_bfs && !ctx->opt.sp_no_reg && !ctx->opt.sp_no_pk;
    const int cap = bfs ? 128 : sp_fw_cap();
    Tmp<u32> cls_count(ctx);
    Tmp<i32> cls_list(ctx);
    GK_TRY(cls_count.alloc(32)); GK_TRY(cls_list.alloc((size_t)16 * (size_t)N));
    if (!one_launch) GK_TRY(gk_zero_async(ctx, cls_count.p, 128));
    if (one_launch)
        sp_prep_small_kernel<<<1, 1024, 0, ctx->stream>>>(b->graph_ptr, b->row_ptr, N, cap, s.dist_ptr.p, s.total.p, s.pair_count.p, s.maxd.p,
                                                          cls_count.p, cls_list.p);
    else if (use_pk) sp_bin_pk_kernel<<<grid_for(N, 256), 256, 0, ctx->stream>>>(b->graph_ptr, b->row_ptr, N, cap, cls_count.p, cls_list.p);
    else sp_bin_kernel<<<grid_for(N, 256), 256, 0, ctx->stream>>>(b->graph_ptr, b->row_ptr, N, cap, cls_count.p, cls_list.p);
    u32 h_cls[17];           // [10]: most adjacency entries of a class-9 graph, [11..16]: class 9 by search class (spb_class)
    {   // one mailbox round trip (a hipMemcpyAsync + stream drain pair costs a staging-copy kernel and ~25 us of idle device)
        u32 hb[19];
        GK_TRY(gk_readback2(ctx, (const u32*)s.total.p, 2, cls_count.p, 17, hb));
        *total_sq = (u64)hb[0] | ((u64)hb[1] << 32);
        for (int k = 0; k < 17; ++k) h_cls[k] = hb[2 + k];
    }
    GK_ARG(*total_sq < (1ull << 31), "ShortestPath: sum of n^2 exceeds int32 item indexing");
    GK_TRY(s.dist.alloc(*total_sq));
    const i32* w = nullptr;
    if (edge_weight && b->n_edges > 0) {
        GK_TRY(s.wdev.alloc(b->n_edges));
        GK_HIP_CHECK(hipMemcpyAsync(s.wdev.p, edge_weight, (size_t)b->n_edges * 4, hipMemcpyHostToDevice, ctx->stream));
        w = s.wdev.p;
    }
    const int nmax = b->max_graph_nodes;
    i64 n_launch = 0;
    ProfScope prof_fw(ctx, "sp_fw", 2);       // the all-pairs kernels alone (bench.py: min-plus rate)
    const SpMark no_mark{nullptr, nullptr, 0, 0, 0};
    const SpMark mk = (mark && use_pk && b->max_graph_nodes <= 128) ? *mark : no_mark;       // every graph through the packed kernels
    if (marked) *marked = mk.present != nullptr;
    if (use_pk) {
        SpPkClasses C;
        int wg = 0;
        for (int c = 0; c < 8; ++c) {
            C.first[c] = wg, C.count[c] = (int)h_cls[c];
            wg += (int)cdiv(h_cls[c], SP_REG_WAVES);
        }
        C.first[8] = wg;
        const u32 n_big = h_cls[4] + h_cls[5] + h_cls[6] + h_cls[7];
        // the two register kernels work on disjoint graphs and neither fills the chip (a few thousand waves, each one long
        // dependent chain): side by side on two streams instead of one after the other (config 4: 65 + 58 us -> their maximum)
        const bool both = C.first[4] > 0 && n_big > 0;
        hipStream_t st_pk = ctx->stream;
        if (both) GK_TRY(gk_side_fork(ctx, &st_pk));
        if (C.first[4] > 0) {
            sp_fw_pk_kernel<0><<<dim3((unsigned)C.first[4]), 64 * SP_REG_WAVES, SP_REG_WAVES * (64 * 72 + 128) * 2, st_pk>>>(
                C, cls_list.p, N, b->graph_ptr, b->row_ptr, b->col_idx, w, s.dist_ptr.p, s.dist.p, s.pair_count.p, s.maxd.p, mk);
            ++n_launch;
        }
        if (n_big > 0) {                     // 65..128 vertices: a workgroup per graph, columns split over its four waves
            const int lds = (128 * 136 + 512) * 2 + 128 * 4;    // matrix, pivot-column exchange, column terms (SpMark)
            sp_fw_pkw_kernel<<<dim3(n_big), 256, lds, ctx->stream>>>(
                cls_list.p, N, (int)h_cls[4], (int)h_cls[5], (int)h_cls[6], (int)h_cls[7], b->graph_ptr, b->row_ptr, b->col_idx, w,
                s.dist_ptr.p, s.dist.p, s.pair_count.p, s.maxd.p, mk);
            ++n_launch;
        }
        if (both) GK_TRY(gk_side_join(ctx));
        if (h_cls[8] > 0 && nmax > 128) {
            const int nfw = nmax < cap ? nmax : cap;
            const size_t lds = (size_t)nfw * (nfw | 1) * 4;
            GK_TRY(gk_func_lds(ctx, (const void*)sp_fw_kernel, (int)lds));
            sp_fw_kernel<<<dim3((unsigned)N), 1024, lds, ctx->stream>>>(
                b->graph_ptr, b->row_ptr, b->col_idx, w, s.dist_ptr.p, s.dist.p, s.pair_count.p, s.maxd.p, 128, cap);
            ++n_launch;
        }
    } else {
        SpClasses C;
        int wg = 0;
        for (int c = 0; c < 8; ++c) {
            C.first[c] = wg, C.count[c] = (int)h_cls[c];
            wg += (int)cdiv(h_cls[c], SP_REG_WAVES);
        }
        C.first[8] = wg;
        if (wg > 0 && !ctx->opt.sp_no_reg) {
            const int lds = SP_REG_WAVES * SP_REG_MAX_N * (SP_REG_MAX_N + 4) * 4;
            GK_TRY(gk_func_lds(ctx, (const void*)sp_fw_reg_kernel, lds));
            sp_fw_reg_kernel<<<dim3((unsigned)wg), 64 * SP_REG_WAVES, lds, ctx->stream>>>(
                C, cls_list.p, N, b->graph_ptr, b->row_ptr, b->col_idx, w, s.dist_ptr.p, s.dist.p, s.pair_count.p, s.maxd.p);
            ++n_launch;
        }
        // 64 < n <= cap: one 1024-thread workgroup per graph with the matrix in LDS (workgroups of other sizes exit at once)
        const int lo = ctx->opt.sp_no_reg ? 0 : SP_REG_MAX_N;
        if ((h_cls[8] > 0 || ctx->opt.sp_no_reg) && nmax > lo) {
            const int nfw = nmax < cap ? nmax : cap;
            const size_t lds = (size_t)nfw * (nfw | 1) * 4;
            GK_TRY(gk_func_lds(ctx, (const void*)sp_fw_kernel, (int)lds));
            sp_fw_kernel<<<dim3((unsigned)N), 1024, lds, ctx->stream>>>(
                b->graph_ptr, b->row_ptr, b->col_idx, w, s.dist_ptr.p, s.dist.p, s.pair_count.p, s.maxd.p, lo, cap);
            ++n_launch;
        }
    }
    // unit weights: bit-parallel breadth-first search (64 / 32 / 16 columns per word by graph size), the row relaxation
    // beyond its largest class (and for weights)
    const i32* relax_list = cls_list.p + (size_t)9 * (size_t)N;
    u32 relax_n = h_cls[9];
    if (nmax > cap && h_cls[9] > 0 && !w && !no_bfs && !ctx->opt.sp_no_bfs) {       // (with sp.no_reg / sp.no_pk: above the LDS cap only)
        // a launch per class over the class' own graph list (one grid over all large graphs left 500 k workgroups that
        // only found out they had nothing to do -- each holding 149 KiB of LDS, i.e. one at a time per CU: 4.7 ms)
        const i32* L = cls_list.p + (size_t)10 * (size_t)N;
        // the size classes work on disjoint graphs and every class ends in a tail that leaves most of the chip idle (REDDIT-like:
        // 0.70 + 0.52 + 0.31 + 0.28 + 0.25 ms one after the other): two streams, alternating classes (round 6)
        int n_cls = 0;
        for (int c = 11; c <= 15; ++c) n_cls += h_cls[c] ? 1 : 0;
        hipStream_t st_side = ctx->stream;
        const bool two = n_cls >= 2 && !ctx->opt.sp_bfs_one_stream;
        if (two) GK_TRY(gk_side_fork(ctx, &st_side));
        hipStream_t st[2] = {ctx->stream, st_side};
        int k = 0;
        if (h_cls[13]) GK_TRY((sp_msbfs_launch<u64, 64, 1024, 2, 2>(ctx, b, s, L + 2 * N, h_cls[13], cap, 2048, nmax, SPB_LDS_MAX, st[k++ & 1])));
        if (h_cls[14]) GK_TRY((sp_msbfs_launch<u32, 32, 1024, 4, 3>(ctx, b, s, L + 3 * N, h_cls[14], cap, 4096, nmax, SPB_LDS_MAX, st[k++ & 1])));
        if (h_cls[12]) GK_TRY((sp_msbfs_launch<u64, 64, 1024, 1, 1>(ctx, b, s, L + N, h_cls[12], cap, 1024, nmax, SPB_LDS1, st[k++ & 1])));
        if (h_cls[15]) GK_TRY((sp_msbfs_launch<unsigned short, 16, 1024, 8, 4>(ctx, b, s, L + 4 * N, h_cls[15], cap, SPB_MAX_N, nmax, SPB_LDS_MAX, st[k++ & 1])));
        if (h_cls[11]) GK_TRY((sp_msbfs_launch<u64, 64, 512, 1, 0>(ctx, b, s, L, h_cls[11], cap, 512, nmax, SPB_LDS0, st[k++ & 1])));
        if (two) GK_TRY(gk_side_join(ctx));
        relax_list = L + 5 * N, relax_n = h_cls[16];
    }
    if (nmax > cap && relax_n > 0) {
        GK_ARG(nmax <= SP_ROW_MAX_N, "ShortestPath: graphs above 32768 vertices are not supported");
        size_t lds = (size_t)nmax * 4;
        GK_TRY(gk_func_lds(ctx, (const void*)sp_relax_kernel, (int)lds));
        GK_ARG(nmax <= 65535, "ShortestPath: grid.y overflow");
        GK_TRY(s.esrc.alloc(b->n_edges > 0 ? b->n_edges : 1));
        sp_edge_src_kernel<<<grid_for(b->n_nodes, 256), 256, 0, ctx->stream>>>(b->graph_ptr, b->node_graph, b->row_ptr, s.esrc.p, b->n_nodes);
        sp_relax_kernel<<<dim3((unsigned)relax_n, (unsigned)nmax), SP_THREADS, lds, ctx->stream>>>(
            relax_list, b->graph_ptr, b->row_ptr, b->col_idx, s.esrc.p, w, s.dist_ptr.p, s.dist.p,
            s.pair_count.p, s.maxd.p, cap);
    }
    (void)n_launch;
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

// ---------------------------------------------------------------------------------------
// Arbitrary positive float64 edge weights.  The reference keys its features by the float distance AS IT COMPUTES IT
// (shortest_path.py:389,469-490), so the distances have to be its distances bit for bit:
//   * floyd_warshall (graph.py:1767-1794; adjacency input): pivots 0..n-1 in order, one rounded add per update; within a
//     pivot the updates are independent (row and column k do not change: D[k][k] = 0), so a float64 sweep per pivot with
//     the same pivot order gives the same bits;
//   * dijkstra (graph.py:1712-1764; dictionary input): for positive weights and monotone rounding the final distances
//     are the LEAST FIXED POINT of d(w) = min over edges (v, w) of fl(d(v) + weight(v, w)), d(source) = 0 -- per source
//     the minimum over paths of the left-to-right float sum (D[u][v] and D[v][u] may differ) -- which relaxation sweeps
//     from +inf reach in any order (checked against the reference's dijkstra on 3 000 random float graphs, and by the
//     goldens of tests/golden/sp_float.npz).
// One workgroup per graph, the n x n float64 matrix in LDS (n <= 143).  The distinct distance values of the whole batch
// are then RANKED (the shared sorting dictionary on the 63-bit patterns of the positive doubles) and the ranks 1..R take
// the place of the integer distances in everything downstream.
// ---------------------------------------------------------------------------------------
#define SPF_MAX_N 143
#define SPF_INF_BITS 0x7ff0000000000000ull

__global__ __launch_bounds__(1024) void sp_f64_kernel(const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr,
                                                      const i32* __restrict__ col_idx, const double* __restrict__ w,
                                                      const unsigned char* __restrict__ algo, const u64* __restrict__ dist_ptr,
                                                      double* __restrict__ out, int only_graph) {
    extern __shared__ __attribute__((aligned(16))) double spf_d[];
    __shared__ int changed;
    const int g = only_graph >= 0 ? only_graph : (int)blockIdx.x;
    const i32 v0 = graph_ptr[g];
    const int n = graph_ptr[g + 1] - v0, tid = threadIdx.x, nt = blockDim.x;
    if (n > SPF_MAX_N) return;                             // sp_f64_big_kernel's graphs (the matrix does not fit LDS)
    const double inf = __longlong_as_double((long long)SPF_INF_BITS);
    for (int idx = tid; idx < n * n; idx += nt) spf_d[idx] = (idx / n == idx % n) ? 0.0 : inf;
    __syncthreads();
    if (algo[g] == 0) {
        for (int u = tid; u < n; u += nt)
            for (i32 e = row_ptr[v0 + u]; e < row_ptr[v0 + u + 1]; ++e) {
                const int v = col_idx[e] - v0;
                if (v != u) spf_d[u * n + v] = w[e];
            }
        __syncthreads();
        for (int k = 0; k < n; ++k) {
            for (int idx = tid; idx < n * n; idx += nt) {
                const int i = idx / n, j = idx - i * n;
                const double c = spf_d[i * n + k] + spf_d[k * n + j];        // entries of row / column k never change in pivot k
                if (c < spf_d[idx]) spf_d[idx] = c;
            }
            __syncthreads();
        }
    } else {
        unsigned long long* bits = (unsigned long long*)spf_d;             // positive doubles order like their bit patterns
        for (;;) {
            if (tid == 0) changed = 0;
            __syncthreads();
            for (int idx = tid; idx < n * n; idx += nt) {                 // idx = (source, v): relax the out-edges of v
                const int src = idx / n, v = idx - src * n;
                const double dv = spf_d[idx];
                if (!(dv < inf)) continue;
                for (i32 e = row_ptr[v0 + v]; e < row_ptr[v0 + v + 1]; ++e) {
                    const int t = col_idx[e] - v0;
                    const double c = dv + w[e];
                    if (c < spf_d[src * n + t]) {
                        atomicMin(&bits[src * n + t], (unsigned long long)__double_as_longlong(c));
                        changed = 1;
                    }
                }
            }
            __syncthreads();
            if (!changed) break;
            __syncthreads();
        }
    }
    double* o = out + (only_graph >= 0 ? 0 : dist_ptr[g]);
    for (int idx = tid; idx < n * n; idx += nt) o[idx] = spf_d[idx];
}

// The same for graphs ABOVE SPF_MAX_N vertices: the matrix stays in HBM (one workgroup per graph works on its block of D).
// Floyd-Warshall order: one sweep per pivot, the sweeps separated by workgroup barriers -- row and column k are not written
// in pivot k (D[k][k] = 0 and weights are positive: fl(D[i][k] + D[k][k]) = D[i][k]), so the in-place sweep reads what the
// reference reads.  Dijkstra semantics: relaxation sweeps with 64-bit atomic minima until nothing changes.  All accesses
// to D go to L2 (agent-scope relaxed atomics): the waves of the workgroup see each other's updates after a barrier.
// O(n^3) by ONE workgroup per graph: a route for the occasional large graph, not a fast path.
__global__ __launch_bounds__(1024) void sp_f64_big_kernel(const i32* __restrict__ graph_ptr, const i32* __restrict__ row_ptr,
                                                          const i32* __restrict__ col_idx, const double* __restrict__ w,
                                                          const unsigned char* __restrict__ algo, const u64* __restrict__ dist_ptr,
                                                          double* __restrict__ out, int only_graph) {
    __shared__ int changed;
    const int g = only_graph >= 0 ? only_graph : (int)blockIdx.x;
    const i32 v0 = graph_ptr[g];
    const i64 n = graph_ptr[g + 1] - v0;
    if (n <= SPF_MAX_N) return;                            // sp_f64_kernel's graphs
    const int tid = threadIdx.x, nt = blockDim.x;
    const unsigned long long INF = SPF_INF_BITS;
    unsigned long long* D = (unsigned long long*)(out + (only_graph >= 0 ? 0 : dist_ptr[g]));      // positive doubles order like their bit patterns
    auto ld = [&](i64 i) __attribute__((always_inline)) { return __hip_atomic_load(&D[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st = [&](i64 i, unsigned long long x) __attribute__((always_inline)) { __hip_atomic_store(&D[i], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (i64 idx = tid; idx < n * n; idx += nt) st(idx, (idx / n == idx % n) ? 0ull : INF);
    __syncthreads();
    if (algo[g] == 0) {
        for (i64 u = tid; u < n; u += nt)
            for (i32 e = row_ptr[v0 + u]; e < row_ptr[v0 + u + 1]; ++e) {
                const i64 v = col_idx[e] - v0;
                if (v != u) st(u * n + v, (unsigned long long)__double_as_longlong(w[e]));
            }
        __syncthreads();
        for (i64 k = 0; k < n; ++k) {
            for (i64 idx = tid; idx < n * n; idx += nt) {
                const i64 i = idx / n, j = idx - i * n;
                const double a = __longlong_as_double((long long)ld(i * n + k)), b = __longlong_as_double((long long)ld(k * n + j));
                const double c = a + b;
                if (c < __longlong_as_double((long long)ld(idx))) st(idx, (unsigned long long)__double_as_longlong(c));
            }
            __syncthreads();
        }
    } else {
        for (;;) {
            if (tid == 0) changed = 0;
            __syncthreads();
            for (i64 idx = tid; idx < n * n; idx += nt) {              // idx = (source, v): relax the out-edges of v
                const i64 src = idx / n, v = idx - src * n;
                const unsigned long long bv = ld(idx);
                if (bv >= INF) continue;
                const double dv = __longlong_as_double((long long)bv);
                for (i32 e = row_ptr[v0 + v]; e < row_ptr[v0 + v + 1]; ++e) {
                    const i64 t = col_idx[e] - v0;
                    const unsigned long long c = (unsigned long long)__double_as_longlong(dv + w[e]);
                    if (c < ld(src * n + t)) {
                        atomicMin(&D[src * n + t], c);
                        changed = 1;
                    }
                }
            }
            __syncthreads();
            if (!changed) break;
            __syncthreads();
        }
    }
}

// sort keys of the ranking: the bit pattern of a finite off-diagonal distance, +inf's pattern otherwise
__global__ void spf_keys_kernel(const i32* __restrict__ graph_ptr, const u64* __restrict__ dist_ptr, const double* __restrict__ D,
                                u64* __restrict__ keys, i64 n_graphs) {
    const int g = blockIdx.x;
    const int n = graph_ptr[g + 1] - graph_ptr[g];
    const u64 base = dist_ptr[g];
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
        const u64 b = (u64)__double_as_longlong(D[base + idx]);
        keys[base + idx] = (idx / n == idx % n || b >= SPF_INF_BITS) ? SPF_INF_BITS : b;
    }
}

// ranks -> the int32 distance matrices the rest of sp.hip works on, pair counts per graph, the largest rank
__global__ __launch_bounds__(256) void spf_finish_kernel(const i32* __restrict__ graph_ptr, const u64* __restrict__ dist_ptr,
                                                         const u64* __restrict__ keys, const i32* __restrict__ lab,
                                                         i32* __restrict__ dist, u32* __restrict__ pair_count,
                                                         u32* __restrict__ maxd) {
    const int g = blockIdx.x;
    const int n = graph_ptr[g + 1] - graph_ptr[g];
    const u64 base = dist_ptr[g];
    u32 cnt = 0, mx = 0;
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
        i32 d;
        if (keys[base + idx] == SPF_INF_BITS) d = (idx / n == idx % n) ? 0 : SP_INF;
        else {
            d = lab[base + idx] + 1;
            ++cnt, mx = (u32)d > mx ? (u32)d : mx;
        }
        dist[base + idx] = d;
    }
    block_count_max(cnt, mx, pair_count + g, maxd);
}

static int sp_compute_dist_f64(gk_ctx* ctx, gk_batch* b, const double* edge_weight, const unsigned char* graph_algo, SpDist& s,
                               u64* total_sq) {
    const i64 N = b->n_graphs;
    for (i64 e = 0; e < b->n_edges; ++e)
        GK_ARG(edge_weight[e] > 0.0 && edge_weight[e] < 1.0e300, "ShortestPath: float edge weights must be positive and finite");
    GK_TRY(s.sq.alloc(N)); GK_TRY(s.dist_ptr.alloc(N)); GK_TRY(s.total.alloc(1));
    GK_TRY(s.pair_count.alloc(N)); GK_TRY(s.maxd.alloc(1));
    GK_TRY(gk_zero_async(ctx, s.pair_count.p, (size_t)N * 4));
    GK_TRY(gk_zero_async(ctx, s.maxd.p, 4));
    sp_sq_kernel<<<grid_for(N, 256), 256, 0, ctx->stream>>>(b->graph_ptr, s.sq.p, N);
    GK_TRY(gk_scan_u64(ctx, s.sq.p, s.dist_ptr.p, N, true, s.total.p));
    GK_HIP_CHECK(hipMemcpyAsync(total_sq, s.total.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    GK_ARG(*total_sq < (1ull << 31), "ShortestPath: sum of n^2 exceeds int32 item indexing");
    const size_t tot = *total_sq > 0 ? (size_t)*total_sq : 1;
    GK_TRY(s.dist.alloc(tot));
    Tmp<double> wdev(ctx), D(ctx);
    Tmp<unsigned char> adev(ctx);
    Tmp<u64> keys(ctx);
    Tmp<i32> lab(ctx), perm(ctx);
    Tmp<u32> cnt(ctx);
    GK_TRY(wdev.alloc(b->n_edges > 0 ? (size_t)b->n_edges : 1)); GK_TRY(adev.alloc(N > 0 ? (size_t)N : 1)); GK_TRY(D.alloc(tot));
    GK_TRY(keys.alloc(tot)); GK_TRY(lab.alloc(tot)); GK_TRY(perm.alloc(tot)); GK_TRY(cnt.alloc(1));
    if (b->n_edges > 0) GK_HIP_CHECK(hipMemcpyAsync(wdev.p, edge_weight, (size_t)b->n_edges * 8, hipMemcpyHostToDevice, ctx->stream));
    if (N > 0) GK_HIP_CHECK(hipMemcpyAsync(adev.p, graph_algo, (size_t)N, hipMemcpyHostToDevice, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));          // the host arrays may go away
    if (N == 0 || *total_sq == 0) return GK_OK;
    const int nmax = b->max_graph_nodes < SPF_MAX_N ? b->max_graph_nodes : SPF_MAX_N;
    const size_t lds = (size_t)nmax * nmax * 8;
    GK_TRY(gk_func_lds(ctx, (const void*)sp_f64_kernel, (int)lds));
    sp_f64_kernel<<<dim3((unsigned)N), 1024, lds, ctx->stream>>>(b->graph_ptr, b->row_ptr, b->col_idx, wdev.p, adev.p, s.dist_ptr.p,
                                                               D.p, -1);
    if (b->max_graph_nodes > SPF_MAX_N)          // graphs whose matrix does not fit LDS: in HBM, one workgroup each
        sp_f64_big_kernel<<<dim3((unsigned)N), 1024, 0, ctx->stream>>>(b->graph_ptr, b->row_ptr, b->col_idx, wdev.p, adev.p, s.dist_ptr.p,
                                                                       D.p, -1);
    spf_keys_kernel<<<dim3((unsigned)N), 256, 0, ctx->stream>>>(b->graph_ptr, s.dist_ptr.p, D.p, keys.p, N);
    GK_HIP_CHECK(hipGetLastError());
    GK_TRY(gk_dictionary_from_keys(ctx, keys.p, (i64)*total_sq, 63, lab.p, perm.p, cnt.p));
    spf_finish_kernel<<<dim3((unsigned)N), 256, 0, ctx->stream>>>(b->graph_ptr, s.dist_ptr.p, keys.p, lab.p, s.dist.p, s.pair_count.p,
                                                                 s.maxd.p);
    GK_HIP_CHECK(hipGetLastError());
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));          // the temporaries above are released in stream order anyway
    return GK_OK;
}

static int sp_build_impl(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, const double* weight_f64,
                         const unsigned char* graph_algo, int with_labels, int n_levels, gk_batch** out_pair_batch,
                         int64_t* out_n_pairs, int64_t* out_n_keys);

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
    return sp_build_impl(ctx, b, edge_weight, nullptr, nullptr, with_labels, n_levels, out_pair_batch, out_n_pairs, out_n_keys);
}

// the same with float64 edge weights [n_edges] that are matched to the reference bit for bit (above); graph_algo[n_graphs]:
// 0 = the reference would run floyd_warshall on this graph, 1 = dijkstra
extern "C" int gk_sp_build_f64(gk_ctx* ctx, gk_batch* b, const double* edge_weight, const uint8_t* graph_algo, int with_labels,
                               int n_levels, gk_batch** out_pair_batch, int64_t* out_n_pairs, int64_t* out_n_keys) {
    GK_ARG(edge_weight && graph_algo, "gk_sp_build_f64: null weights / algorithm flags");
    return sp_build_impl(ctx, b, nullptr, edge_weight, graph_algo, with_labels, n_levels, out_pair_batch, out_n_pairs, out_n_keys);
}

static int sp_build_impl(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, const double* weight_f64,
                         const unsigned char* graph_algo, int with_labels, int n_levels, gk_batch** out_pair_batch,
                         int64_t* out_n_pairs, int64_t* out_n_keys) {
    GK_ARG(ctx && b && out_pair_batch, "gk_sp_build: null argument");
    GK_ARG(!b->is_pair_batch, "gk_sp_build: needs a graph batch");
    GK_ARG(n_levels >= 1, "gk_sp_build_levels: n_levels must be >= 1");
    GK_ARG(n_levels == 1 || n_levels <= b->n_levels,
           "gk_sp_build_levels: levels not computed (call gk_wl_relabel first)");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    ProfScope prof(ctx, "sp");
    const i64 N = b->n_graphs, V = b->n_nodes;
    if (!weight_f64) {   // distances are int32 sums below SP_INF: the longest simple path must stay under it, otherwise a
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
    // pair offsets double as the pair batch's graph_ptr[N+1]
    Tmp<u32> ptotal(ctx);
    struct PtrGuard { gk_ctx* c; void* p; ~PtrGuard() { if (p) gk_dev_free(c, p); } } gp_guard{ctx, nullptr};
    GK_TRY(gk_dev_alloc(ctx, &gp_guard.p, (size_t)(N + 1) * 4));
    u32* pair_base = (u32*)gp_guard.p;
    u32 h_pairs = 0, h_maxd = 0;
    bool defer_rt = false;
    u64 L0_early = 1;                                        // the alphabet of the keys, as the histogram form computes it below
    if (with_labels) {
        L0_early = (u64)(b->n_labels0 > 0 ? b->n_labels0 : 1);
        if (b->n_levels > 0 && (u64)b->label_counts[0] > L0_early) L0_early = (u64)b->label_counts[0];
    }
    u64 dist_bound = 1;                                      // every finite distance is below it
    {
        i64 wm = 1;
        if (edge_weight)
            for (i64 e = 0; e < b->n_edges; ++e) wm = edge_weight[e] > wm ? edge_weight[e] : wm;
        dist_bound = (u64)(b->max_graph_nodes > 1 ? b->max_graph_nodes - 1 : 0) * (u64)wm + 1;
    }
    // key marks from inside the packed all-pairs kernels (SpMark) wherever the job will skip the read-back of the largest
    // distance anyway (defer_rt below: small graphs, a key space laid out for the bound of a distance)
    Tmp<unsigned char> present_early(ctx);
    SpMark mark{nullptr, b->labels, L0_early, dist_bound, with_labels ? 1 : 0};
    bool marked = false;
    if (!weight_f64 && n_levels == 1 && !ctx->opt.sp_no_hist && !ctx->opt.sp_no_prep && !ctx->opt.sp_no_fused_mark && b->max_graph_nodes <= 128 &&
        dist_bound * L0_early * L0_early <= (1ull << 22) && L0_early < (1u << 15) && N > 0) {
        const size_t ks = (size_t)(dist_bound * L0_early * L0_early);
        GK_TRY(present_early.alloc(ks));
        GK_TRY(gk_zero_async(ctx, present_early.p, ks));
        mark.present = present_early.p;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        // attempt 1: a breadth-first search ran out of its 8-bit levels (a shortest path of 255 edges) -- once more, the
        // large graphs by row relaxation
        if (weight_f64) GK_TRY(sp_compute_dist_f64(ctx, b, weight_f64, graph_algo, s, &total_sq));
        else GK_TRY(sp_compute_dist(ctx, b, edge_weight, s, &total_sq, attempt == 1, mark.present ? &mark : nullptr, &marked));
        GK_TRY(ptotal.alloc(4));                             // [0] pairs, [1] distinct keys of the histogram form (below)
        if (N <= SP_PREP_MAX_GRAPHS && !ctx->opt.sp_no_prep)
            sp_pair_ranges_small_kernel<<<1, 1024, 0, ctx->stream>>>(s.pair_count.p, pair_base, N, ptotal.p);
        else {
            GK_TRY(gk_scan_u32(ctx, s.pair_count.p, pair_base, N, true, ptotal.p));
            GK_HIP_CHECK(hipMemcpyAsync(pair_base + N, ptotal.p, 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        // A job of small graphs with a small key space does not wait for the largest distance here (round 6): no breadth-first
        // search can overflow without a graph above 128 vertices, the key space is laid out for the BOUND (n_max - 1) w_max
        // of a distance -- any d1 above the largest distance serves -- and the pair count comes back with the number of
        // distinct keys, one host round trip later.  (BASELINE config 4: four round trips per fit_transform -> three.)
        defer_rt = !weight_f64 && n_levels == 1 && !ctx->opt.sp_no_hist && !ctx->opt.sp_no_prep && b->max_graph_nodes <= 128 &&
                   dist_bound * L0_early * L0_early <= (1ull << 22) && L0_early < (1u << 15) && total_sq > (u64)V;
        if (defer_rt) { h_maxd = (u32)(dist_bound - 1); h_pairs = 0; break; }
        u32 hb[2];
        GK_TRY(gk_readback2(ctx, ptotal.p, 1, s.maxd.p, 1, hb));
        h_pairs = hb[0], h_maxd = hb[1];
        if (h_maxd != SPB_OVERFLOW) break;
    }
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
    {
        // Histogram form (one level, a key space small enough for a direct table): no pair items at all.  The pair batch
        // keeps the distance matrices; the feature builder counts every graph's (l_u, l_v, d) keys in an LDS table
        // (features_gm.hip).  Which keys occur, and their dense ids: a presence table over the whole key space.
        u64 L0 = 1;
        if (with_labels) {
            L0 = (u64)(b->n_labels0 > 0 ? b->n_labels0 : 1);
            if (b->n_levels > 0 && (u64)b->label_counts[0] > L0) L0 = (u64)b->label_counts[0];
        }
        const u64 keyspace = d1 * L0 * L0;
        // round 5: the key space cap went from 2^22 to 2^26 keys (a presence byte and a 4-byte id each: 320 MB at the cap) --
        // 620 degree labels x 14 distances (REDDIT-like) are 5.4 M keys and used to leave for the pair items
        if (n_levels == 1 && !ctx->opt.sp_no_hist && L0 < (1u << 15) && keyspace <= (1ull << 26) && (h_pairs > 0 || defer_rt)) {
            Tmp<unsigned char> present(ctx);
            Tmp<u32> nk(ctx);
            if ((r = nk.alloc(1))) return fail(r);
            // (marked: the packed kernels left the marks in present_early -- same alphabet, same d1)
            const bool have_marks = marked && defer_rt && mark.present && mark.L == L0 && mark.d1 == d1;
            if (!have_marks) {
                if ((r = present.alloc((size_t)keyspace))) return fail(r);
                if ((r = gk_zero_async(ctx, present.p, (size_t)keyspace))) return fail(r);
                sp_mark_kernel<<<dim3((unsigned)N, (unsigned)cdiv(b->max_graph_nodes > 0 ? b->max_graph_nodes : 1, SP_SLAB)), SP_THREADS, 0, ctx->stream>>>(
                    b->graph_ptr, b->labels, s.dist_ptr.p, s.dist.p, present.p, L0, d1, with_labels ? 1 : 0);
            }
            const unsigned char* present_p = have_marks ? present_early.p : present.p;
            if ((r = gk_dev_alloc(ctx, &q, (size_t)keyspace * 4))) return fail(r);
            pb->sp_idtab = (u32*)q;
            SpIdScan sc{present_p, pb->sp_idtab, defer_rt ? ptotal.p + 1 : nk.p};
            if ((r = gk_scan_fn<u32, SpIdScan>(ctx, sc, (i64)keyspace, nullptr))) return fail(r);
            if ((r = gk_dev_alloc(ctx, &q, (size_t)(N + 1) * 4))) return fail(r);
            pb->sp_node_ptr = (i32*)q;
            if ((r = gk_dev_alloc(ctx, &q, (size_t)(V > 0 ? V : 1) * 4))) return fail(r);
            pb->sp_node_label = (i32*)q;
            u32 h_nk = 0;
            // the counts are posted first, the copies of the source graphs' sizes and labels run while the host waits for them
            u32 ticket = 0;
            const u32* rb_src = defer_rt ? ptotal.p : nk.p;
            const int rb_n = defer_rt ? 2 : 1;
            if ((r = gk_readback_post(ctx, rb_src, rb_n, &ticket))) return fail(r);
            if (hipMemcpyAsync(pb->sp_node_ptr, b->graph_ptr, (size_t)(N + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                (V > 0 && hipMemcpyAsync(pb->sp_node_label, b->labels, (size_t)V * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)) {
                gk_set_error("gk_sp_build: %s", hipGetErrorString(hipGetLastError()));
                return fail(GK_ERR_HIP);
            }
            if (defer_rt) {                                     // pairs and distinct keys in one read-back
                u32 hb[2];
                if ((r = gk_readback_collect(ctx, ticket, rb_src, hb, 2))) return fail(r);
                h_pairs = hb[0], h_nk = hb[1];
                pb->n_nodes = h_pairs;
                if (h_pairs == 0) {                             // no finite pair in the whole job: the general route's empty batch
                    gk_batch_destroy(pb);
                    const int keep = ctx->opt.sp_no_prep;
                    ctx->opt.sp_no_prep = 1;
                    const int rr = sp_build_impl(ctx, b, edge_weight, weight_f64, graph_algo, with_labels, n_levels, out_pair_batch, out_n_pairs, out_n_keys);
                    ctx->opt.sp_no_prep = keep;
                    return rr;
                }
            } else if ((r = gk_readback_collect(ctx, ticket, rb_src, &h_nk, 1))) return fail(r);
            pb->sp_dist = s.dist.p, s.dist.p = nullptr;             // the matrices move into the pair batch
            pb->sp_dist_ptr = s.dist_ptr.p, s.dist_ptr.p = nullptr;
            pb->sp_hist = true, pb->sp_L = (i64)L0, pb->sp_dcap = (i64)d1, pb->sp_keyspace = (i64)keyspace, pb->sp_src_nodes = V;
            pb->sp_with_labels = with_labels ? 1 : 0;
            pb->sp_max_nodes = b->max_graph_nodes;
            pb->n_levels = 1, pb->cap_levels = 0;
            pb->label_counts.assign(1, (i64)h_nk);
            *out_pair_batch = pb;
            if (out_n_pairs) *out_n_pairs = h_pairs;
            if (out_n_keys) out_n_keys[0] = h_nk;
            return GK_OK;
        }
    }
    if ((r = gk_dev_alloc(ctx, &q, np * 4))) return fail(r);
    pb->node_graph = (i32*)q;
    if ((r = gk_dev_alloc(ctx, &q, np * 4 * (size_t)n_levels))) return fail(r);
    pb->labels = (i32*)q;
    if ((r = gk_dev_alloc(ctx, &q, np * 4 * (size_t)n_levels))) return fail(r);
    pb->perm = (i32*)q;
    pb->cap_levels = n_levels;
    Tmp<u64> keys(ctx);
    Tmp<u32> nkeys(ctx), emit_cursor(ctx);
    if ((r = keys.alloc(np)) || (r = nkeys.alloc((size_t)n_levels)) || (r = emit_cursor.alloc((size_t)N))) return fail(r);
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
        if ((r = gk_zero_async(ctx, emit_cursor.p, (size_t)N * 4))) return fail(r);
        sp_emit_kernel<<<dim3((unsigned)N, (unsigned)cdiv(b->max_graph_nodes > 0 ? b->max_graph_nodes : 1, SP_SLAB)), SP_THREADS, 0, ctx->stream>>>(
            b->graph_ptr, b->labels + (size_t)l * V, s.dist_ptr.p, s.dist.p, pair_base, keys.p, pb->node_graph,
            L, d1, with_labels ? 1 : 0, emit_cursor.p);
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

// Item arrays of a histogram-form pair batch, on demand (the label-major feature builder reads them): the classic
// emit + sorting dictionary on the stored distance matrices.
int gk_sp_materialise(gk_ctx* ctx, gk_batch* pb) {
    if (!pb->sp_hist || pb->labels) return GK_OK;
    const i64 N = pb->n_graphs;
    const size_t np = pb->n_nodes > 0 ? (size_t)pb->n_nodes : 1;
    void* q = nullptr;
    GK_TRY(gk_dev_alloc(ctx, &q, np * 4));
    pb->node_graph = (i32*)q;
    GK_TRY(gk_dev_alloc(ctx, &q, np * 4));
    pb->labels = (i32*)q;
    GK_TRY(gk_dev_alloc(ctx, &q, np * 4));
    pb->perm = (i32*)q;
    pb->cap_levels = 1;
    Tmp<u64> keys(ctx);
    Tmp<u32> nkeys(ctx);
    GK_TRY(keys.alloc(np)); GK_TRY(nkeys.alloc(1));
    const u64 L = (u64)pb->sp_L, d1 = (u64)pb->sp_dcap;
    Tmp<u32> emit_cursor(ctx);
    GK_TRY(emit_cursor.alloc((size_t)N));
    GK_TRY(gk_zero_async(ctx, emit_cursor.p, (size_t)N * 4));
    sp_emit_kernel<<<dim3((unsigned)N, (unsigned)cdiv(pb->sp_max_nodes > 0 ? pb->sp_max_nodes : 1, SP_SLAB)), SP_THREADS, 0, ctx->stream>>>(
        pb->sp_node_ptr, pb->sp_node_label, pb->sp_dist_ptr, pb->sp_dist, (const u32*)pb->graph_ptr, keys.p, pb->node_graph, L, d1,
        pb->sp_with_labels, emit_cursor.p);
    GK_TRY(gk_dictionary_from_keys(ctx, keys.p, pb->n_nodes, bits_for64(d1 * L * L - 1), pb->labels, pb->perm, nkeys.p));
    GK_HIP_CHECK(hipGetLastError());
    return GK_OK;
}

extern "C" int gk_sp_debug_apsp(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, int64_t graph,
                                int32_t* out_dist) {
    GK_ARG(ctx && b && out_dist, "gk_sp_debug_apsp: null argument");
    GK_ARG(graph >= 0 && graph < b->n_graphs && !b->is_pair_batch, "gk_sp_debug_apsp: bad graph index");
    SpDist s(ctx);
    u64 total_sq = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {      // as gk_sp_build: the row relaxation when a search overflows its levels
        GK_TRY(sp_compute_dist(ctx, b, edge_weight, s, &total_sq, attempt == 1));
        u32 h_maxd = 0;
        GK_TRY(gk_readback(ctx, s.maxd.p, &h_maxd, 1));
        if (h_maxd != SPB_OVERFLOW) break;
    }
    std::vector<i32> gp(2);
    std::vector<u64> dp(1);
    GK_HIP_CHECK(hipMemcpyAsync(gp.data(), b->graph_ptr + graph, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipMemcpyAsync(dp.data(), s.dist_ptr.p + graph, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const i64 n = gp[1] - gp[0];
    std::vector<i32> h((size_t)(n * n > 0 ? n * n : 1));
    if (n > 0 && (dp[0] & SP_BYTE_FLAG)) {               // byte form (common.h: SpMat)
        const i32* base = s.dist.p + (dp[0] & ~SP_BYTE_FLAG);
        const unsigned char* d8 = (const unsigned char*)(((uintptr_t)base + 15) & ~(uintptr_t)15);
        const i64 ns = (n + 15) & ~15ll;
        std::vector<unsigned char> hb((size_t)(n * ns));
        GK_HIP_CHECK(hipMemcpyAsync(hb.data(), d8, (size_t)(n * ns), hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (i64 i = 0; i < n; ++i)
            for (i64 j = 0; j < n; ++j) out_dist[i * n + j] = hb[(size_t)(i * ns + j)] == 255 ? -1 : (i32)hb[(size_t)(i * ns + j)];
        return GK_OK;
    }
    if (n > 0) {
        GK_HIP_CHECK(hipMemcpyAsync(h.data(), s.dist.p + dp[0], (size_t)n * n * 4, hipMemcpyDeviceToHost, ctx->stream));
        GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    for (i64 i = 0; i < n * n; ++i) out_dist[i] = h[i] >= SP_INF ? -1 : h[i];
    return GK_OK;
}

// float64 distance matrix of ONE graph exactly as the reference computes it (gk_sp_build_f64); unreachable = -1
extern "C" int gk_sp_debug_apsp_f64(gk_ctx* ctx, gk_batch* b, const double* edge_weight, const uint8_t* graph_algo, int64_t graph,
                                    double* out_dist) {
    GK_ARG(ctx && b && out_dist && edge_weight && graph_algo, "gk_sp_debug_apsp_f64: null argument");
    GK_ARG(graph >= 0 && graph < b->n_graphs && !b->is_pair_batch, "gk_sp_debug_apsp_f64: bad graph index");
    GK_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<i32> gp(2);
    GK_HIP_CHECK(hipMemcpyAsync(gp.data(), b->graph_ptr + graph, 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const i64 n = gp[1] - gp[0];
    if (n == 0) return GK_OK;
    Tmp<double> wdev(ctx), D(ctx);
    Tmp<unsigned char> adev(ctx);
    GK_TRY(wdev.alloc(b->n_edges > 0 ? (size_t)b->n_edges : 1)); GK_TRY(adev.alloc((size_t)b->n_graphs)); GK_TRY(D.alloc((size_t)(n * n)));
    if (b->n_edges > 0) GK_HIP_CHECK(hipMemcpyAsync(wdev.p, edge_weight, (size_t)b->n_edges * 8, hipMemcpyHostToDevice, ctx->stream));
    GK_HIP_CHECK(hipMemcpyAsync(adev.p, graph_algo, (size_t)b->n_graphs, hipMemcpyHostToDevice, ctx->stream));
    if (n <= SPF_MAX_N) {
        const size_t lds = (size_t)n * n * 8;
        GK_TRY(gk_func_lds(ctx, (const void*)sp_f64_kernel, (int)lds));
        sp_f64_kernel<<<dim3(1), 1024, lds, ctx->stream>>>(b->graph_ptr, b->row_ptr, b->col_idx, wdev.p, adev.p, nullptr, D.p, (int)graph);
    } else
        sp_f64_big_kernel<<<dim3(1), 1024, 0, ctx->stream>>>(b->graph_ptr, b->row_ptr, b->col_idx, wdev.p, adev.p, nullptr, D.p, (int)graph);
    GK_HIP_CHECK(hipGetLastError());
    GK_HIP_CHECK(hipMemcpyAsync(out_dist, D.p, (size_t)(n * n) * 8, hipMemcpyDeviceToHost, ctx->stream));
    GK_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (i64 i = 0; i < n * n; ++i)
        if (!(out_dist[i] < 1.0e308)) out_dist[i] = -1.0;
    return GK_OK;
}

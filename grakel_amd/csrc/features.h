// Definitions shared by the two feature builders (features.hip: label-major, features_gm.hip: graph-major).
#pragma once
#include "common.h"

// label-major builder: meta layout (u32): per level l: [3l+0]=T (triples) [3l+1]=R (label runs) [3l+2]=dense columns
// with counts that fit the primary region up to and including level l; globals at G = 3*n_levels: [G+0]=max count,
// [G+1]=rare columns, [G+2]=float64 columns, [G+3]=primary dense columns, [G+4+l]=rare columns up to
// and including level l, [4*n_levels+4]=int8 dense columns, then 64 partial maxima of the counts
#define META_T(l) (3 * (l) + 0)
#define META_R(l) (3 * (l) + 1)
#define META_C(l) (3 * (l) + 2)

#define FEAT_MAX_LEVELS 48
#define COL_BYTE_BASE (1 << 30)   // colid >= this: column (colid - base) of the int8 region
// colid >= this: a label whose counts exceed 127, SPLIT into parts^2 int8 columns starting at (colid - base) of the int8
// region.  A count c is written as parts digits a_0..a_{parts-1} <= 127 with sum c; the LEFT operand row holds
// a_p at column p*parts+r, the RIGHT operand row holds a_r there, so that the dot product of a left row with a right
// row contributes sum_p sum_r a_p(i) a_r(j) = c(i) c(j): the exact product, in the int8 GEMM, no float64 side operand.
#define COL_SPLIT_BASE ((1 << 30) + (1 << 29))
#define GM_SPLIT_MAX_PARTS 3

// graph-major builder: meta layout
#define GM_META_PRIM 0
#define GM_META_INT8 1
#define GM_META_F64 2
#define GM_META_RARE 3
#define GM_META_RARE_ENTRIES 4
#define GM_META_SPLIT 5           // parts of the split columns (0: labels with counts above 127 go to the float64 operand)
#define GM_META_TYPE 6            // ShortestPath histogram jobs (round 6): operand type decided ON THE DEVICE from the largest self
                                  // similarity (K_ij <= sqrt(K_ii K_jj)): 0 = fp4 + int8 (below 2^24), 1 = int8 (below 2^31), 2 = float64,
                                  // 3 = int8 + float64 side operand (the int8 columns' part of every self similarity below 2^31)
#define GM_META_SELFMAX 7         // ... and that largest K_ii, saturated to 32 bits: the job's entry bound
#define GM_META_MAXC 8            // 64 partial maxima
#define GM_META_NNZ 72            // 64 partial sums
#define GM_META_WORDS 136
#define GM_META_OVF 135           // gk_features_build_sp: a per-graph histogram table overflowed (last of the NNZ slots, unused otherwise)
#define GM_MAX_NODES 1024         // largest graph a wave stages in LDS
#define GM_WG_NODES 320           // ... and, in a job that has such graphs, everything above this many vertices goes to the workgroup kernel too
#define GM_HUGE_MAX_NODES 8192    // largest graph of the graph-major builder: above GM_MAX_NODES a whole workgroup counts it (gm_pairs_huge_kernel)
#define GM_ROW_LDS_MAX 147456     // widest operand row (bytes) assembled in LDS (round 6: 144 KiB -- 64 KiB sent the D&D-like ShortestPath job with its 90 k int8 columns to the label-major builder)

int gk_features_build_gm(gk_ctx* ctx, gk_batch* b, gk_feat* f, int n_levels, int prim_max, int wide_above);

// ShortestPath pair batch in histogram form (common.h: gk_batch::sp_hist): features straight from the distance matrices.
// GK_ERR_UNSUPPORTED: a graph has more distinct features than the LDS table holds or the operand row is too wide -- the
// caller materialises the pair items (gk_sp_materialise) and takes the label-major builder.
int gk_features_build_sp(gk_ctx* ctx, gk_batch* pb, gk_feat* f, int prim_max, int wide_above, bool force_rows);

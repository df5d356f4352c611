/*
 * gk_hip.h -- C ABI of libgk_hip.so, the MI355X (gfx950) implementation of GraKeL's
 * WeisfeilerLehman / VertexHistogram / ShortestPath Gram-matrix path.
 *
 * The reference (ysig/GraKeL, /root/reference) is pure Python on this path and has NO FFI
 * of its own; the interface each entry point replaces is therefore a Python method of the
 * reference, cited per function below (file:line into /root/reference).  The binding a
 * GraKeL maintainer would add is a ctypes stub -- see INTEGRATION.md.
 *
 * Conventions
 *   - plain C, caller-owned HOST buffers unless the name says `_dev`; the library owns all
 *     device memory; no C++/torch types cross this boundary.
 *   - every function returns 0 on success, <0 on error; gk_last_error() gives the message
 *     (thread-local).  No exceptions cross the boundary.
 *   - a gk_ctx is bound to one device and one HIP stream and is NOT thread-safe.
 *   - a "batch" is a set of graphs packed as CSR:
 *        graph_ptr[n_graphs+1]  node range of each graph (nodes of a graph are contiguous)
 *        row_ptr[n_nodes+1]     out-neighbour range of each node
 *        col_idx[n_edges]       GLOBAL node index of each out-neighbour (same graph)
 *        node_label[n_nodes]    dense level-0 label id in [0, n_labels0)
 *     all int32.  Neighbour = key of the reference's edge dictionary of that vertex
 *     (out-neighbours only, self loops kept, multi-edges collapsed, weights ignored:
 *     grakel/kernels/weisfeiler_lehman.py:235-239).
 */
#ifndef GK_HIP_H
#define GK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gk_ctx gk_ctx;
typedef struct gk_batch gk_batch;
typedef struct gk_feat gk_feat;

#define GK_OK 0
#define GK_ERR_ARG (-1)
#define GK_ERR_HIP (-2)
#define GK_ERR_STATE (-3)
#define GK_ERR_UNSUPPORTED (-4)

/* ---- library / context ------------------------------------------------------------- */
const char* gk_last_error(void);
const char* gk_version(void);
/* Number of visible HIP devices (0 when none: callers must then fail loudly, there is no
 * CPU fallback in this library). */
int gk_device_count(int* out_count);
int gk_create(int device_id, gk_ctx** out);
int gk_destroy(gk_ctx* ctx);
/* Borrow an external HIP stream (e.g. torch.cuda.current_stream().cuda_stream); NULL
 * restores the context's own stream. */
int gk_set_stream(gk_ctx* ctx, void* hip_stream);
int gk_synchronize(gk_ctx* ctx);
/* HIP-event timing on the context's stream (bench.py's roofline clock). */
int gk_timer_start(gk_ctx* ctx);
int gk_timer_stop_ms(gk_ctx* ctx, double* out_ms);
/* Per-kernel-class accumulated device time since the last reset, measured with HIP events
 * around each launch group when profiling is enabled (enable=1 adds sync overhead). */
int gk_profile_enable(gk_ctx* ctx, int enable);
int gk_profile_reset(gk_ctx* ctx);
/* names: "relabel", "features", "gram", "sp" (gk_sp_build*), "sp_fw" (its all-pairs kernels alone).
 * Returns total ms and launch count. */
int gk_profile_get(gk_ctx* ctx, const char* name, double* out_ms, int64_t* out_launches);

/* Route and capacity options of a context.  The library reads NO environment variables; everything that
 * selects between its equivalent routes goes through this table (tests/test_gpu_parity.py runs the jobs through
 * every one of them), and EVERY option leaves the results unchanged -- an option removes or forces one of several
 * equivalent routes, or shrinks a capacity so that a fallback is taken.  value 0 restores the default.  Names:
 *   relabel:  "wl.no_tiny" "wl.no_listscan" "wl.no_iso" "wl.no_split" "wl.no_exact1" "wl.no_active_set"
 *             "wl.no_bucket_dict" "wl.no_hist0" "wl.frozen_words" "wl.flag_bytes" "wl.sig_no_regs" "wl.debug"
 *             "wl.no_stream" (never the relabel route without host round trips, wl_stream.hip: the host-driven
 *             route of wl.hip, which all the other "wl.*" switches select within)
 *             "wl.no_wave_sig" (vertices of degree 33..1024 by the workgroup-per-vertex signature kernel and the
 *             thread-per-vertex verifier of rounds 1-4 instead of the wave-per-vertex kernels)
 *             "wl.no_frozen_skip" (1: at a full level of the host-driven route the wave / workgroup signature kernels also gather and
 *             sort the neighbours of vertices that were alone in their class at the level before)
 *             "wl.no_converge" (1: the host-driven route computes every level even after two consecutive levels with the same
 *             number of labels -- a converged partition, whose remaining levels are copies)
 *             "transform.no_fused" (look-up transform: the target classes matched by two launches per level even when the
 *             targets are few enough for the single-workgroup all-levels kernel)
 *             "scan.direct_max" (test hook: number of 2 048-item tiles up to which the fused scans add up their predecessors
 *             per block; 1 forces the form with a scanned tile-sum array that jobs above 8 M items take)
 *             "sort.buckets" (1 never / 2 always the per-bucket finish of the sort)
 *             "wl.bd_slots" (distinct keys a bucket of the sort-free dictionary accepts: small values force its
 *             overflow and with it the second, sorting attempt)
 *   features: "feat.no_gm" "feat.gm_no_priv" "feat.gm_rows_wg" "feat.low_df" (df below which a column becomes pair updates; default 24 (N / 10 000)^0.75 within [8, 128])
 *             "feat.gm_no_early_post" (1: the operand sizes are read back after the column scan instead of being posted from its tile sums) "feat.gm_no_huge" (1: a graph above 1 024 vertices sends the job to the label-major builder, rounds 1-5; default: a workgroup counts such a graph, up to 8 192 vertices) "feat.gm_rows_256" (1: the workgroup-per-graph operand-row kernel with 256 threads even where the graphs hold thousands of entries each)
 *             "feat.gm_row_lds_max" (bytes of operand row the graph-major builder accepts: small values force the
 *             fall-back to the label-major builder)
 *   Gram:     "gram.dd" (the direct-store form of the persistent kernel: 1 always, 2 never, 0 per job) "gram.no_fp4" "gram.no_ws" "gram.no_sym" "gram.no_patch" "gram.xcc" "feat.rows_lo" "feat.rows_hi" (the multi-GPU operand-row exchange: gk_features_operand_rows) "gram.strip" (tile order: 1 the 8 x 8 patches of rounds 1-5, 2..32 strips of that many tile columns, 0 per job)
 *             "gram.fold" (the rare labels' pair updates inside the tile kernel, which then normalises in its epilogue too, instead of float64
 *             atomics + a normalisation pass afterwards: 0 when it pays, 1 whenever legal, 2 never)
 *             "gram.pair_cap" (test hook: capacity of the per-tile pair buckets of that fold-in)
 *             "gram.no_split8" (labels with counts above 127 in a float64 side operand instead of split int8 columns)
 *             "gram.no_split64" (the float64 side product with one workgroup per tile: its K loop is not split)
 *             "gram.no_compact" (host copies of integer-valued matrices as plain float64 instead of uint16 / int32 + widening)
 *             "gram.copy_threads" (host threads of that widening; 0: min(hardware threads, 32))
 *             "gram.no_tri" (a WHOLE symmetric matrix bound for the host normally crosses PCIe as the 256 x 256 blocks on and
 *             above its diagonal only, the host threads widen AND mirror them and -- for a normalised job -- apply the
 *             1 / sqrt(K_ii K_jj) factors; 1 = the rectangular narrow copy, normalised matrices as plain float64)
 *             "gram.no_avx2" (those host threads keep to SSE2, what a CPU without AVX2 runs)
 *   paths:    "sp.bfs_one_stream" (1: the breadth-first search's size classes one after the other instead of on two streams) "sp.no_prep" (1: a job's set-up -- clears, n^2 prefix, size classes -- as separate launches instead of one single-workgroup kernel) "sp.no_hist" (ShortestPath features from explicit pair items and the sorting dictionary instead of per-graph
 *             histograms of the distance matrices),
 *             "sp.no_bfs" (graphs above the Floyd-Warshall LDS cap with unit weights: one row relaxation per source
 *             instead of the bit-parallel breadth-first search over 64 / 32 / 16 columns at a time),
 *             "sp.bfs_no_lds_cols" (test hook: that search reads the adjacency entries from HBM / L2, not from LDS),
 *             "sp.bfs_no_bytes" (1: that search leaves 32-bit distance matrices -- rounds 3-5 -- instead of byte matrices),
 *             "sp.no_rows" (histogram form without the counter rows of the large graphs: one workgroup and one LDS table
 *             per graph, a table that overflows sends the whole job to the pair items -- the round-4 form),
 *             "sp.no_fused_mark" (1: which keys occur is found by a pass over the stored distance matrices instead of inside the
 *             packed all-pairs kernels of a job of small graphs),
 *             "sp.hist_no_batch" (1: the LDS-table histogram kernel walks its graphs one at a time instead of as many at a time
 *             as fit its table),
 *             "sp.static_type" (1: the Gram operand type of a histogram-form job -- fp4 + int8, int8 or float64 -- from the
 *             a-priori bound (pairs of the largest graph)^2 instead of the job's largest self similarity, found on the device;
 *             2: found on the device, but without the mixed type -- int8 columns + a float64 side operand for the columns
 *             holding counts above 127 -- where only the int8 columns' part of the self similarities stays below 2^31),
 *             "sp.rows_no_merge" (bit 0: the counter-row route adds every matrix entry to its LDS table on its own instead of
 *             per-lane runs of equal keys; bit 1: it walks a graph's rows in matrix order and never empties the table
 *             instead of label by label with the table emptied into the counter row when it fills; bit 2: a wave counts one
 *             matrix row at a time instead of up to four neighbouring rows of the sorted order together),
 *             "sp.rows_all" / "sp.hist_unit" / "sp.hist_slots" (test hooks of the counter-row route: every graph
 *             through it, distance-matrix entries per counting workgroup, slots of its LDS table),
 *             "sp.no_pk" (all-pairs distances never in the 16-bit packed register kernel: 32-bit registers up to 64
 *             vertices, the LDS workgroup kernel beyond), "sp.no_reg" (the LDS workgroup kernel for every graph)
 *   plumbing: "no_mailbox" (small read-backs by hipMemcpy instead of the mapped mailbox),
 *             "debug.poison" (the allocator fills every block it hands out with this byte: uninitialised reads
 *             then see the same garbage in every run),
 *             "debug.guard" (red zones before every block of the allocator and behind its requested size, checked when the
 *             block is released and at gk_synchronize, which then returns GK_ERR_STATE: a kernel writing outside its block
 *             becomes an error instead of a corruption somewhere else; set it before the first allocation of a job)
 * Unknown names return GK_ERR_ARG.  The reference has no counterpart (its route is fixed). */
int gk_set_option(gk_ctx* ctx, const char* name, int64_t value);
int gk_get_option(gk_ctx* ctx, const char* name, int64_t* out_value);

/* Pinned (page-locked) host memory for the outputs of gk_gram / gk_gram_rows: the device -> host copy of the
 * float64 matrix (8 N^2 bytes; the reference returns it as a host ndarray, kernel.py:167-204) then runs at the
 * PCIe rate instead of the pageable-copy rate (measured 57 vs 12-18 GB/s).  Any host pointer is accepted
 * as out_host; this pair only makes the fast path available to callers without a HIP binding. */
int gk_host_alloc(uint64_t bytes, void** out);
int gk_host_free(void* p);

/* ---- batches ------------------------------------------------------------------------ */
/* Replaces the per-graph Python containers built by WeisfeilerLehman.parse_input
 * (grakel/kernels/weisfeiler_lehman.py:142-194: Gs_ed / L dictionaries) and
 * Graph.__init__ (grakel/graph.py:147-232).  src_on_device!=0: the four arrays are device
 * pointers on ctx's device (multi-GPU path: the all-gathered shards). */
int gk_batch_create(gk_ctx* ctx, int64_t n_graphs, int64_t n_nodes, int64_t n_edges,
                    const int32_t* graph_ptr, const int32_t* row_ptr, const int32_t* col_idx,
                    const int32_t* node_label, int32_t n_labels0, int src_on_device,
                    gk_batch** out);
/* Multi-GPU ingestion (grakel_amd/dist.py): the global batch straight from the all-gathered shard
 * messages, without a host round trip.  Rank r's message is msg_stride = mg + 2*mv + me int32 words:
 * [graph sizes | node degrees | node labels | col_idx with LOCAL node ids], each part zero padded to
 * the largest shard (mg graphs, mv nodes, me edges).  shard_sizes = int64[n_ranks][3] (graphs,
 * nodes, edges) on the host; gathered_dev = the n_ranks messages back to back on the device (what
 * ncclAllGather leaves).  Shards are concatenated in rank order. */
int gk_batch_from_shards(gk_ctx* ctx, int n_ranks, const int64_t* shard_sizes, int64_t mg, int64_t mv,
                         int64_t me, const int32_t* gathered_dev, int32_t n_labels0, gk_batch** out);
/* Union batch on the device, graphs of `a` first: what a transform relabels (fitted graphs + targets;
 * weisfeiler_lehman.py:330-500, vertex_histogram.py:57-154 with the fitted label columns).  `a` usually is the
 * fitted batch kept in HBM between calls, so only the targets cross PCIe.  Both inputs stay valid. */
int gk_batch_concat(gk_ctx* ctx, gk_batch* a, gk_batch* b, int32_t n_labels0, gk_batch** out);
/* Fitted state for consumers without Python (the reference pickles its estimator, grakel/tests/test_common.py:53-58): on
 * this path a fit is the packed batch itself -- transform relabels the targets jointly with the fitted graphs, every level
 * array is recomputed -- so the state is the CSR + the level-0 label ids as one self-describing blob.
 * gk_export_state: out_buf == NULL only reports the size in *out_needed.  gk_import_state validates like gk_batch_create.
 * The label value -> id map stays with the caller (as for gk_batch_create). */
int gk_export_state(gk_ctx* ctx, gk_batch* b, void* out_buf, uint64_t buf_bytes, uint64_t* out_needed);
int gk_import_state(gk_ctx* ctx, const void* buf, uint64_t bytes, gk_batch** out);
int gk_batch_destroy(gk_batch* b);
int gk_batch_info(gk_batch* b, int64_t* n_graphs, int64_t* n_nodes, int64_t* n_edges);

/* ---- Weisfeiler-Lehman relabelling --------------------------------------------------- */
/* Replaces the `generate_graphs` relabel loop, weisfeiler_lehman.py:223-258 (fit) and
 * :435-476 (transform: run on the union batch fit+target graphs; the partition restricted
 * to the fit graphs is unchanged).  After the call the batch holds, for every level
 * 0..n_iter, a dense label id per node such that two nodes (of any graphs) share an id iff
 * (own previous label, sorted multiset of out-neighbour previous labels) are equal.
 * Exact: 64-bit multiset hashes are only used to group candidates, every node's full
 * signature is compared with its group representative, and groups that fail are refined
 * with re-seeded hashes until none fails.  Level 1 of a job with few input labels and small
 * degrees uses exact integer signature codes instead (no hash, nothing to verify).  Work that
 * cannot change the result is skipped: singleton classes are frozen, the classes of the
 * isolated vertices (one per input label) are carried along without being sorted again.
 * Label ids are an arbitrary bijection per level; the listed order (nodes of shared classes
 * grouped by label) that gk_features_build consumes stays inside the batch.
 *   out_label_counts[n_iter+1] : number of distinct labels per level (host)
 *   hash_bits : 0 = default, 2*log2(n_nodes) + 8 rounded up to whole bytes (at least 32, at most 64): a colliding
 *               pair then shows up in about one level of 256 and is resolved by the exact refinement; tests pass
 *               small values to force collisions
 *   out_rounds : total extra refinement rounds that were needed (0 in practice), may be NULL */
int gk_wl_relabel(gk_ctx* ctx, gk_batch* b, int n_iter, int hash_bits,
                  int64_t* out_label_counts, int* out_rounds);
/* fit_transform of the WL-subtree kernel in ONE call: gk_wl_relabel + gk_features_build_ex (all graphs fitted) + gk_gram
 * (weisfeiler_lehman.py:292-328, vertex_histogram.py:57-184, kernel.py:195-204).  For the jobs of the route without host
 * round trips the relabel is only queued, the feature builder runs behind it on device-side counts, and the one host round
 * trip of the job (the operand sizes) also carries the relabel's collision / overflow flags and label counts; everything
 * else runs the three calls in sequence.  n_iter + 1 <= 48 levels.  out_host may be NULL (the matrix stays on the device:
 * gk_gram_dev_ptr); *out_feat is the job's feature object (gk_features_selfk, gk_features_info, gk_gram_checksum, ...;
 * release with gk_features_destroy).  kind: GK_FEAT_DOT | GK_FEAT_MINSUM; normalize as gk_gram. */
int gk_wl_fit_transform(gk_ctx* ctx, gk_batch* b, int n_iter, int hash_bits, int kind, int normalize,
                        int64_t* out_label_counts, int* out_rounds, gk_feat** out_feat, double* out_host);
/* Level labels back to the host (parity tests: partition equality with the oracle). */
int gk_wl_get_labels(gk_ctx* ctx, gk_batch* b, int level, int32_t* out_labels);
/* ---- WeisfeilerLehman.transform as a look-up against the fitted dictionaries (csrc/wl_transform.hip) --------------
 * Replaces weisfeiler_lehman.py:435-476 (relabel the targets, look their credentials up in the fitted _inv_labels[i]) and
 * :493-498 + vertex_histogram.py:138-184 (the targets' label counts in the fitted columns, one rectangular product per
 * level) for the VertexHistogram base kernel.  gk_wl_fitted_create: `fitted` is relabelled for n_iter (gk_wl_relabel) and
 * must stay alive and NOT be relabelled again while the state is used (gk_wl_transform then returns GK_ERR_STATE).
 * gk_wl_transform: `targets` relabelled ALONE with the same n_iter, their level-0 ids in the fit's id space (input labels
 * the fit never saw: ids >= the fit's n_labels0); out_K host [n_targets x n_fitted] float64 (normalize as gk_gram: 0 none,
 * 1 plain, 2 nan_to_num), out_y_selfk host [n_targets].  Work is proportional to the targets, not to the fit.
 * GK_ERR_UNSUPPORTED: the job needs the joint route (gk_batch_concat + gk_wl_relabel + gk_features_build with n_fit):
 * two fitted classes share a 64-bit signature hash, a representative node has more than 64 neighbours, a target graph has
 * more than 1024 nodes, more than 48 levels. */
typedef struct gk_wl_fitted gk_wl_fitted;
int gk_wl_fitted_create(gk_ctx* ctx, gk_batch* fitted, int n_iter, gk_wl_fitted** out);
int gk_wl_fitted_destroy(gk_wl_fitted* w);
int gk_wl_fitted_selfk(gk_ctx* ctx, gk_wl_fitted* w, double* out_selfk);     /* host [n_fitted]: the fitted diagonal */
int gk_wl_transform(gk_ctx* ctx, gk_wl_fitted* w, gk_batch* targets, int normalize, double* out_K, double* out_y_selfk);

/* Which route the batch's last gk_wl_relabel took: *out_stream = 1 the route without host round trips (csrc/wl_stream.hip;
 * graph batches of small graphs with at most 256 input labels), 0 the host-driven route (everything else, option
 * "wl.no_stream", and the redo after a hash collision or a table overflow).  Same partitions, same matrices either way;
 * the label ids differ (both dense).  The reference has no counterpart. */
int gk_wl_route(gk_batch* b, int* out_stream);
/* Kernel-level test hooks: the raw 64-bit signature hash and the sorted neighbour-label
 * lists of level `level` (computed from level-1 labels), before dictionary assignment. */
int gk_wl_debug_signature(gk_ctx* ctx, gk_batch* b, int level, uint64_t seed,
                          uint64_t* out_hash, int32_t* out_sorted_nbr_labels);

/* ---- label-count features ------------------------------------------------------------ */
/* Replaces VertexHistogram.parse_input (grakel/kernels/vertex_histogram.py:57-154), called
 * once per WL level by weisfeiler_lehman.py:260-285.  Builds, over levels [0, n_levels) of
 * the batch, the per-graph label-count features:  the sparse (label, graph, count) triples,
 * the self-similarities  selfk[g] = sum over all columns of count^2  (= the Gram diagonal,
 * vertex_histogram.py:186-219 / weisfeiler_lehman.py:502-555) and a dense, column-compacted
 * Phi_s holding only the columns that can contribute to an off-diagonal entry AND occur in
 * enough graphs to be worth a dense column (the rest become pair updates in gk_gram).
 *   n_fit == n_graphs : symmetric job (fit_transform); a column is kept iff it occurs in
 *                       >= 2 graphs, singletons are folded into selfk.
 *   n_fit <  n_graphs : rectangular job (transform): rows = graphs [n_fit, n_graphs),
 *                       cols = graphs [0, n_fit); a column is kept iff it occurs on both sides
 *                       (unseen labels are dropped exactly like vertex_histogram.py:179). */
int gk_features_build(gk_ctx* ctx, gk_batch* b, int n_levels, int64_t n_fit, gk_feat** out);

/* Same with an explicit pairwise operation on the per-level label counts:
 *   GK_FEAT_DOT    K_ij = sum_l c_il * c_jl       (VertexHistogram, vertex_histogram.py:176-182)
 *   GK_FEAT_MINSUM K_ij = sum_l min(c_il, c_jl)   (histogram intersection of the WL hierarchy,
 *                  weisfeiler_lehman_optimal_assignment.py:201-206,268-279; diagonal = sum_l c_il)
 * gk_gram / gk_gram_rows then apply unchanged. */
#define GK_FEAT_DOT 0
#define GK_FEAT_MINSUM 1
int gk_features_build_ex(gk_ctx* ctx, gk_batch* b, int n_levels, int64_t n_fit, int kind, gk_feat** out);
/* The same over the levels [level_lo, level_hi) of the batch, at most 48 per job: a hierarchy deeper than that
 * (the reference takes any n_iter, weisfeiler_lehman.py:112-114) is built in chunks whose matrices -- and selfk vectors --
 * add up (K is a sum over levels, weisfeiler_lehman.py:269-270); normalisation then happens on the sum. */
int gk_features_build_range(gk_ctx* ctx, gk_batch* b, int level_lo, int level_hi, int64_t n_fit, int kind, gk_feat** out);
int gk_features_destroy(gk_feat* f);
/* n_cols_kept: width of the dense MFMA operand Phi_s; n_cols_low: useful but rare columns
 * (fewer graphs than the job's threshold, option feat.low_df) that are applied as exact pair updates after the GEMM instead;
 * nnz: number of (label,graph) triples over all levels; max_count: largest single count;
 * dtype: 0 = integer operand on the fp4 / int8 MFMA path (counts 0..4 as MX fp4 codes when every Gram entry stays
 * below 2^24, counts up to 127 as int8; gk_features_operand says which), 1 = only the float64 side operand is in use. */
int gk_features_info(gk_feat* f, int64_t* n_cols_kept, int64_t* n_cols_low, int64_t* nnz,
                     int64_t* max_count, int* dtype);
/* Layout of the dense operand: fp4 != 0 when the primary region holds MX fp4 (e2m1) codes, k_steps_fp4_or_i8 /
 * k_steps_i8_secondary = its 128-byte K-steps, n_cols_f64 = columns of the float64 side operand. */
int gk_features_operand(gk_feat* f, int* fp4, int* k_steps_primary, int* k_steps_i8_secondary, int64_t* n_cols_f64);
/* Multi-GPU, the operand-row exchange north_star names ("RCCL all-gather of per-graph feature vectors"; the default
 * exchange of grakel_amd/dist.py and gk_batch_allgather is the packed CSR, SURVEY 8e / DESIGN 5): with the context options
 * "feat.rows_lo" / "feat.rows_hi" set, gk_features_build assembles the dense operand rows of the graphs [lo, hi) only (the
 * rare labels' lists and the float64 side operand stay complete on every rank); gk_features_operand_rows hands out the row
 * buffers -- left and, when a job has split columns, right operand: [n_rows x row_bytes] bytes each, row g at g * row_bytes
 * -- so that the caller's collective can fill in the other ranks' rows, and the range this job assembled.  gk_memcpy_dev is
 * a device-to-device copy on the context's stream (staging buffers of that collective). */
int gk_features_operand_rows(gk_feat* f, void** out_phi, void** out_phi_right, int64_t* out_row_bytes, int64_t* out_n_rows,
                             int64_t* out_own_lo, int64_t* out_own_hi);
int gk_memcpy_dev(gk_ctx* ctx, void* dst_dev, const void* src_dev, uint64_t bytes);
int gk_features_selfk(gk_ctx* ctx, gk_feat* f, double* out_selfk /* [n_graphs] */);
/* Test hook: the dense column-compacted Phi_s as float64 [n_graphs x n_cols_kept]. */
int gk_features_debug_phi(gk_ctx* ctx, gk_feat* f, double* out_phi);
/* Test hook: the RIGHT operand of the dense product in the same form.  It differs from the left one (above) only when
 * labels with counts of 128 .. 381 were split into parts^2 int8 columns (*split_parts = 2 or 3; 0 = no split columns, both
 * operands are the same matrix): the off-diagonal dense term of K is left . right^T. */
int gk_features_debug_phi_right(gk_ctx* ctx, gk_feat* f, double* out_phi, int* split_parts);

/* ---- Gram matrix ---------------------------------------------------------------------- */
/* Replaces VertexHistogram._calculate_kernel_matrix (vertex_histogram.py:156-184: X.dot(X.T)
 * per level), the np.sum over levels (weisfeiler_lehman.py:269-270) and the normalisation
 * tails (weisfeiler_lehman.py:323-328,493-498; kernel.py:158-165,195-204).
 * Output is row-major float64 [n_rows x n_cols]:
 *   symmetric job:   n_rows = n_cols = n_graphs
 *   rectangular job: n_rows = n_graphs - n_fit, n_cols = n_fit
 * normalize: 0 none; 1 divide by sqrt(selfk_row*selfk_col) leaving 0/0 = NaN (Kernel);
 *            2 same with nan_to_num -> 0 (WeisfeilerLehman).
 * out_host may be NULL: the matrix then stays on the device (gk_gram_dev_ptr), which is what
 * bench.py times.  With out_host set, a WHOLE symmetric integer-valued matrix crosses PCIe as its upper
 * triangle in uint16 / int32 blocks and the normalisation factors are applied by the host threads that widen
 * it (option "gram.no_tri"): after such a call the DEVICE copy (gk_gram_dev_ptr, gk_gram_checksum) is the exact
 * unnormalised matrix, whatever `normalize` was. */
int gk_gram(gk_ctx* ctx, gk_feat* f, int normalize, double* out_host);
int gk_gram_dev_ptr(gk_feat* f, void** out_dev_ptr, int64_t* n_rows, int64_t* n_cols);
/* Row-sharded Gram (multi-GPU): only rows [row_lo,row_hi) of the job's matrix are computed;
 * out_host is [(row_hi-row_lo) x n_cols]. */
int gk_gram_rows(gk_ctx* ctx, gk_feat* f, int64_t row_lo, int64_t row_hi, int normalize,
                 double* out_host);
/* Block-wise form for the multi-GPU path (grakel_amd/dist.py; SURVEY.md 8e "symmetry can halve work"): the block
 * [row_lo,row_hi) x [col_lo,col_hi) of the job's matrix, every term included, un-normalised, into CALLER-owned
 * device memory (out_dev = the block's entry (0,0), ld elements between rows).  A diagonal block of a symmetric
 * job only multiplies the tiles on/above its diagonal.  gk_gram_last_stats then reports the flops of all blocks
 * since gk_gram_reset_stats.  gk_block_copy moves a block between device buffers, optionally transposed (the
 * mirrored half a rank receives from its peer); gk_gram_normalize_rows applies gk_gram's normalisation to a
 * finished row block. */
int gk_gram_block(gk_ctx* ctx, gk_feat* f, int64_t row_lo, int64_t row_hi, int64_t col_lo, int64_t col_hi,
                  double* out_dev, int64_t ld);
int gk_gram_reset_stats(gk_feat* f);
int gk_block_copy(gk_ctx* ctx, const double* src_dev, int64_t rows, int64_t cols, int64_t ld_src,
                  double* dst_dev, int64_t ld_dst, int transpose);
int gk_gram_normalize_rows(gk_ctx* ctx, gk_feat* f, int64_t row_lo, int64_t row_hi, double* K_dev, int mode);
/* Checksums of the matrix the last gk_gram* call left on the device, computed in place: sum of all entries,
 * trace and max |K_ij - K_ji| (the last two 0 for a non-square output).  What bench.py asserts on the
 * timed matrix and what the 50 000-graph parity test (a 20 GB matrix) checks without a host copy; the
 * reference has no counterpart (it holds K as one host ndarray, kernel.py:167-204). */
int gk_gram_checksum(gk_ctx* ctx, gk_feat* f, double* out_sum, double* out_trace, double* out_max_asym);
/* What the caller-side wall of a gk_gram* call with out_host set is made of (bench.py prints it next to the
 * host-to-host figures so that they can be reproduced on another box).  out_cpu[4]: online CPUs, CPUs in the affinity
 * mask, CPUs of the container's cgroup quota (0 = none), the host-thread budget the library derives from them (its
 * widening and ingestion threads: min(32, those three, quota - 2)).  out_copy[8], of the context's LAST host copy:
 * [0] form (0 plain float64 copy, 1 upper-triangle blocks, 2 rectangular narrow), [1] host threads that widened,
 * [2] bytes that crossed PCIe, [3] ms until the last chunk had landed in the staging ring (pack kernel + copies),
 * [4] ms of the whole copy-out (until the last widening thread was done), [5] / [6] mean / max busy ms of a widening
 * thread, [7] chunks.  The reference has no counterpart (its matrix is born on the host, kernel.py:167-204). */
int gk_host_copy_stats(gk_ctx* ctx, int* out_cpu, double* out_copy);
/* Algorithmic work of the last gk_gram* call, for the roofline: MACs = rows*cols*kept cols. */
int gk_gram_last_stats(gk_feat* f, double* out_flops, double* out_ms_event);

/* ---- Shortest-path kernel ------------------------------------------------------------- */
/* Replaces Graph.build_shortest_path_matrix + floyd_warshall/dijkstra
 * (grakel/graph.py:588-687,1712-1794) and ShortestPath.parse_input's pair enumeration
 * (grakel/kernels/shortest_path.py:412-499,510-511).  edge_weight: NULL = unit weights, else
 * int32[n_edges] positive integer weights.  Produces a batch of "pair items": one per ordered
 * pair (u,v), u!=v, d(u,v) finite, keyed by (label_u, label_v, d) (with_labels!=0) or d,
 * exposed as level 0 of a derived batch so that gk_features_build / gk_gram apply unchanged. */
int gk_sp_build(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, int with_labels,
                gk_batch** out_pair_batch, int64_t* out_n_pairs, int64_t* out_n_keys);
/* The WL framework over the ShortestPath base kernel
 * (WeisfeilerLehman(base_graph_kernel=ShortestPath): grakel/kernels/weisfeiler_lehman.py:77-109
 * resolves the base class, :260-270 fits one base kernel per level on (graph, level labels) and
 * sums their matrices).  After gk_wl_relabel(b, n_levels - 1): level l of the pair batch keys the
 * pairs by the level-l WL labels (distances computed once), so gk_features_build(pair_batch,
 * n_levels, ...) + gk_gram return sum_l K_SP(level l).  out_n_keys: int64[n_levels]. */
int gk_sp_build_levels(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, int with_labels, int n_levels,
                       gk_batch** out_pair_batch, int64_t* out_n_pairs, int64_t* out_n_keys);
/* The same for ARBITRARY positive float64 edge weights (edge_weight[n_edges]).  The reference keys its features by the
 * float distance as it computes it (shortest_path.py:389,469-490), so the distances are reproduced bit for bit: per graph
 * graph_algo[g] = 0 runs the reference's floyd_warshall (graph.py:1767-1794: adjacency input, or algorithm_type
 * "floyd_warshall"), 1 its dijkstra (graph.py:1712-1764: dictionary input, or "dijkstra") -- see sp.hip for why a float64
 * pivot sweep / relaxation to the fixed point give the same bits.  The distinct distances of the batch are ranked and the
 * ranks stand in for integer distances from there on.  Graphs of up to 143 vertices keep their float64 matrix in LDS; larger ones work on it in HBM, one workgroup per graph (slower, no limit). */
int gk_sp_build_f64(gk_ctx* ctx, gk_batch* b, const double* edge_weight, const uint8_t* graph_algo, int with_labels, int n_levels,
                    gk_batch** out_pair_batch, int64_t* out_n_pairs, int64_t* out_n_keys);
/* float64 distance matrix of one graph as gk_sp_build_f64 computes it (n x n, -1 = unreachable): the lazy `_enum` state */
int gk_sp_debug_apsp_f64(gk_ctx* ctx, gk_batch* b, const double* edge_weight, const uint8_t* graph_algo, int64_t graph,
                         double* out_dist);
/* Test hook: all-pairs distance matrix of one graph (n x n int32, -1 = unreachable). */
int gk_sp_debug_apsp(gk_ctx* ctx, gk_batch* b, const int32_t* edge_weight, int64_t graph,
                     int32_t* out_dist);

/* ---- Core framework ---------------------------------------------------------------------- */
/* Replaces core_number() (grakel/kernels/core_framework.py:376-416): the k-core number of every
 * vertex of every (undirected) graph of the batch, out_core int32[n_nodes] in batch order.  The
 * framework itself (core_framework.py:95-234: one base kernel per core level on the induced
 * subgraphs, matrices summed) runs the base-kernel entry points above on sub-batches. */
int gk_core_numbers(gk_ctx* ctx, gk_batch* b, int32_t* out_core);

/* ---- multi-GPU: one process per GPU ----------------------------------------------------------------- */
/* SURVEY.md 8b / 8e.  Graphs are sharded over the ranks (rank r ingests and holds only its own), and so are the rows of
 * K.  The reference computes its label dictionaries over ALL graphs (weisfeiler_lehman.py:224-246), so the one exchange
 * step is an all-gather of the packed CSR shards (RCCL, over xGMI inside a node); relabel + features then run replicated
 * and every rank multiplies and stores the row block of its own graphs -- no collective after the all-gather.
 * grakel_amd/dist.py is the same scheme over torch.distributed; these entry points serve consumers without Python.
 * RCCL (librccl.so.1) is loaded on the first gk_comm_* call; GK_ERR_UNSUPPORTED when the host has none.
 *
 *   rank 0:     gk_comm_unique_id(id)  ... hand the 128 bytes to the other processes (file, socket, MPI_Bcast, ...)
 *   every rank: gk_comm_init(ctx, rank, n_ranks, id, &comm)                       (collective: all ranks must call it)
 *               gk_batch_allgather(ctx, comm, <its shard>, &batch, bounds)        (collective)
 *               gk_wl_fit_transform / gk_wl_relabel + gk_features_build           (replicated, no communication)
 *               gk_gram_sharded(ctx, comm, feat, bounds, normalize, K_rows, ...)  (rows bounds[rank] .. bounds[rank + 1])
 */
#define GK_COMM_ID_BYTES 128
typedef struct gk_comm gk_comm;
int gk_comm_unique_id(void* out_id /* GK_COMM_ID_BYTES */);
int gk_comm_init(gk_ctx* ctx, int rank, int n_ranks, const void* id, gk_comm** out);
int gk_comm_destroy(gk_comm* c);
int gk_comm_info(gk_comm* c, int* rank, int* n_ranks);
/* This rank's shard as for gk_batch_create, host arrays in LOCAL numbering (graph_ptr[0] = 0, col_idx = node ids within
 * the shard); the level-0 label ids must mean the same label on every rank.  *out = the batch of ALL graphs in rank
 * order on this rank's device; graph_bounds[n_ranks + 1] = first graph of every rank.  A rank may hold no graph. */
int gk_batch_allgather(gk_ctx* ctx, gk_comm* c, int64_t n_graphs, int64_t n_nodes, int64_t n_edges,
                       const int32_t* graph_ptr, const int32_t* row_ptr, const int32_t* col_idx,
                       const int32_t* node_label, int32_t n_labels0, gk_batch** out, int64_t* graph_bounds);
/* Bring your own transport (MPI, a socket, ...): this rank's shard as the int32 message gk_batch_from_shards consumes,
 * out_msg[mg + 2*mv + me] on the host, mg / mv / me = the largest shard's graphs / nodes / edges over all ranks.  All-gather
 * the messages in rank order, put them on the device back to back and call gk_batch_from_shards: that is what
 * gk_batch_allgather does over RCCL.  No device work, no communicator. */
int gk_shard_message(int64_t n_graphs, int64_t n_nodes, int64_t n_edges, const int32_t* graph_ptr, const int32_t* row_ptr,
                     const int32_t* col_idx, const int32_t* node_label, int64_t mg, int64_t mv, int64_t me, int32_t* out_msg);
/* The rows of the job's matrix that belong to this rank's graphs: gk_gram_rows on [graph_bounds[rank], graph_bounds[rank + 1]);
 * out_host [(row_hi - row_lo) x n_cols] or NULL (the block stays on the device: gk_gram_dev_ptr). */
int gk_gram_sharded(gk_ctx* ctx, gk_comm* c, gk_feat* f, const int64_t* graph_bounds, int normalize, double* out_host,
                    int64_t* row_lo, int64_t* row_hi);

#ifdef __cplusplus
}
#endif
#endif /* GK_HIP_H */

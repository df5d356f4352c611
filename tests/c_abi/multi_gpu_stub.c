/* INTEGRATION.md section C as a translation unit: type-checked against include/gk_hip.h by
 * tests/test_host.py::test_c_header_is_plain_c_and_the_multi_gpu_stub_type_checks (gcc -fsyntax-only; nothing runs). */
#include <stdint.h>
#include <stdlib.h>

#include "gk_hip.h"

int multi_gpu_rows(int local_gpu, int rank, int n_ranks, char* id /* GK_COMM_ID_BYTES, from rank 0 */,
                   int64_t n_graphs, int64_t n_nodes, int64_t n_edges, const int32_t* graph_ptr, const int32_t* row_ptr,
                   const int32_t* col_idx, const int32_t* node_label, int32_t n_labels0, int n_iter, double** out_rows,
                   int64_t* lo, int64_t* hi) {
    gk_ctx* ctx;
    gk_comm* comm;
    gk_batch* all;
    gk_feat* feat;
    int64_t* bounds = (int64_t*)malloc((size_t)(n_ranks + 1) * sizeof(int64_t));
    int64_t* label_counts = (int64_t*)malloc((size_t)(n_iter + 1) * sizeof(int64_t));
    int rc = gk_create(local_gpu, &ctx);
#ifndef GK_STUB_ID_IS_DISTRIBUTED   /* (tests/c_abi/multi_gpu_run.c creates and publishes the id before it calls this function) */
    if (rc == GK_OK && rank == 0) rc = gk_comm_unique_id(id);
#endif
    /* ... hand the 128 bytes to the other processes: a file, a socket, MPI_Bcast(id, 128, MPI_BYTE, 0, ...) ... */
    if (rc == GK_OK) rc = gk_comm_init(ctx, rank, n_ranks, id, &comm);                       /* collective */
    if (rc == GK_OK)
        rc = gk_batch_allgather(ctx, comm, n_graphs, n_nodes, n_edges, graph_ptr, row_ptr, col_idx, node_label, n_labels0, &all,
                                bounds);                                                     /* collective: the one exchange step */
    if (rc == GK_OK) rc = gk_wl_relabel(ctx, all, n_iter, 0, label_counts, NULL);            /* replicated, no communication */
    if (rc == GK_OK) rc = gk_features_build(ctx, all, n_iter + 1, bounds[n_ranks], &feat);
    if (rc == GK_OK) {
        *out_rows = (double*)malloc((size_t)(bounds[rank + 1] - bounds[rank]) * (size_t)bounds[n_ranks] * sizeof(double));
        rc = gk_gram_sharded(ctx, comm, feat, bounds, /*normalize=*/0, *out_rows, lo, hi);  /* rows lo..hi of K, all columns */
    }
    free(bounds);
    free(label_counts);
    return rc;
}

/* INTEGRATION.md section C, RUN: one process per GPU, each with its own shard, through gk_comm_init / gk_batch_allgather /
 * gk_gram_sharded (tests/c_abi/multi_gpu_stub.c is the section's code, linked in unchanged).
 *
 *     multi_gpu_run <rank> <n_ranks> <local_gpu> <id_file> <shard_file> <out_file> <n_iter>
 *
 * shard_file: int64 n_graphs, n_nodes, n_edges, n_labels0, then int32 graph_ptr[n_graphs + 1], row_ptr[n_nodes + 1],
 * col_idx[n_edges], node_label[n_nodes] (local numbering, global level-0 label ids).  Rank 0 creates the communicator
 * id and publishes it as id_file (written to id_file.tmp, then renamed); the other ranks wait for the file -- the
 * "hand the 128 bytes to the other processes" step of the section, done with a file here.
 * out_file: int64 lo, hi, n_cols, then the rank's rows of K as float64 [hi - lo][n_cols].
 * Started by tests/test_gpu_parity.py::test_integration_md_section_c_runs_with_one_process_per_gpu with n_ranks = the
 * number of visible GPUs: on a one-GPU box a communicator of one rank, on an 8-GPU node the real thing. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "gk_hip.h"

int multi_gpu_rows(int local_gpu, int rank, int n_ranks, char* id, int64_t n_graphs, int64_t n_nodes, int64_t n_edges,
                   const int32_t* graph_ptr, const int32_t* row_ptr, const int32_t* col_idx, const int32_t* node_label,
                   int32_t n_labels0, int n_iter, double** out_rows, int64_t* lo, int64_t* hi);

static int read_all(const char* path, void* buf, size_t n) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    const size_t got = fread(buf, 1, n, f);
    fclose(f);
    return got == n ? 0 : -1;
}

int main(int argc, char** argv) {
    if (argc != 8) {
        fprintf(stderr, "usage: multi_gpu_run rank n_ranks local_gpu id_file shard_file out_file n_iter\n");
        return 2;
    }
    const int rank = atoi(argv[1]), n_ranks = atoi(argv[2]), gpu = atoi(argv[3]), n_iter = atoi(argv[7]);
    const char *id_file = argv[4], *shard_file = argv[5], *out_file = argv[6];
    FILE* f = fopen(shard_file, "rb");
    if (!f) { perror("shard"); return 2; }
    int64_t hdr[4];
    if (fread(hdr, 8, 4, f) != 4) return 2;
    const int64_t ng = hdr[0], nv = hdr[1], ne = hdr[2];
    int32_t* gp = (int32_t*)malloc((size_t)(ng + 1) * 4);
    int32_t* rp = (int32_t*)malloc((size_t)(nv + 1) * 4);
    int32_t* ci = (int32_t*)malloc((size_t)(ne > 0 ? ne : 1) * 4);
    int32_t* lab = (int32_t*)malloc((size_t)(nv > 0 ? nv : 1) * 4);
    if (fread(gp, 4, (size_t)ng + 1, f) != (size_t)ng + 1 || fread(rp, 4, (size_t)nv + 1, f) != (size_t)nv + 1 ||
        fread(ci, 4, (size_t)ne, f) != (size_t)ne || fread(lab, 4, (size_t)nv, f) != (size_t)nv) {
        fprintf(stderr, "short shard file\n");
        return 2;
    }
    fclose(f);
    char id[GK_COMM_ID_BYTES];
    memset(id, 0, sizeof id);
    if (rank == 0) {
        if (gk_comm_unique_id(id) != GK_OK) { fprintf(stderr, "gk_comm_unique_id: %s\n", gk_last_error()); return 3; }
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
        FILE* o = fopen(tmp, "wb");
        if (!o || fwrite(id, 1, sizeof id, o) != sizeof id) return 3;
        fclose(o);
        if (rename(tmp, id_file) != 0) return 3;
    } else {
        int tries = 0;
        while (read_all(id_file, id, sizeof id) != 0) {
            if (++tries > 6000) { fprintf(stderr, "rank %d: no communicator id after 60 s\n", rank); return 3; }
            usleep(10000);
        }
    }
    double* rows = NULL;
    int64_t lo = 0, hi = 0;
    /* every rank holds the id now: the stub's own `if (rank == 0) gk_comm_unique_id(id)` line is that same step, so the stub is
     * compiled with -DGK_STUB_ID_IS_DISTRIBUTED here (it would overwrite the id rank 0 has already published) */
    const int rc = multi_gpu_rows(gpu, rank, n_ranks, id, ng, nv, ne, gp, rp, ci, lab, (int32_t)hdr[3], n_iter, &rows, &lo, &hi);
    if (rc != GK_OK) { fprintf(stderr, "rank %d: error %d: %s\n", rank, rc, gk_last_error()); return 4; }
    int64_t total = 0;
    {   /* n_cols = all graphs: the sum of the shard sizes is what gk_batch_allgather reported through bounds; recover it from
         * the file sizes the launcher wrote next to the shards (one int64) to keep this program free of collectives of its own */
        char p[4096];
        snprintf(p, sizeof p, "%s.total", shard_file);
        if (read_all(p, &total, 8) != 0) { fprintf(stderr, "missing %s\n", p); return 2; }
    }
    FILE* o = fopen(out_file, "wb");
    if (!o) return 5;
    int64_t head[3] = {lo, hi, total};
    fwrite(head, 8, 3, o);
    fwrite(rows, 8, (size_t)(hi - lo) * (size_t)total, o);
    fclose(o);
    printf("rank %d/%d on GPU %d: rows [%lld, %lld) of %lld\n", rank, n_ranks, gpu, (long long)lo, (long long)hi, (long long)total);
    return 0;
}

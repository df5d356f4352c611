#!/usr/bin/env python3
"""Benchmark of the hot path: WL-subtree(h=5) fit_transform Gram matrix, graph-pairs/second.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one full pass of the hot path over BASELINE config 3 (10 000 synthetic
Erdos-Renyi graphs, n=100, p=0.05, 5 labels, seed 0): WL relabelling for 5
iterations, label-count features of the 6 levels, and the N x N Gram matrix, with the packed
CSR batch already resident in HBM when the timed region starts and the float64 matrix left in
HBM when it ends.  N>1: the graphs (and the Gram rows) are sharded over the ranks; a step then
also contains the RCCL all-gather of the CSR shards (grakel_amd/dist.py).  Total work is
fixed, so the scaling is "strong".

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` for the
dominant kernel (the int8 MFMA Gram) and `cpu_baseline` (the oracle timed on this box).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(N=10000, n=100, p=0.05, L=5, seed=0, n_iter=5)
I8_DENSE_PEAK_TOPS = 5000.0      # int8 MFMA dense = 2x the 2.5 PF bf16 dense peak (MI355X_MICROARCH.md)
F64_PEAK_TFLOPS = 78.6
# HBM-side bytes per launch of the MFMA Gram kernel on this exact workload (config 3), from the PMC
# passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of
# this script, summarised by tools/pmc_summary.py): 2 x FETCH_SIZE (gfx950 reports half of a wide
# coalesced read, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, KiB -> bytes.
# Algorithmic floor: read Phi_s once (24 MB packed) + write the float64 K once (800 MB).
PMC_FILE = os.path.join(ROOT, "profiles", "r01s_pmc_hbm_bytes.csv")


def gram_pmc_traffic_bytes(n_graphs, dtype):
    """Per-launch HBM bytes of the int8 MFMA Gram kernel from the committed PMC summary, or None
    when the workload is not the profiled one (config 3) or the file is absent."""
    if n_graphs != WORKLOAD["N"] or dtype != "i8" or not os.path.exists(PMC_FILE):
        return None
    fetch = write = None
    with open(PMC_FILE) as f:
        for line in f:
            if "gram_i8_glds_kernel" not in line:
                continue
            parts = line.rstrip().rsplit(",", 3)          # "kernel",counter,launches,avg
            if parts[1] == "FETCH_SIZE":
                fetch = float(parts[3])
            elif parts[1] == "WRITE_SIZE":
                write = float(parts[3])
    if fetch is None or write is None:
        return None
    return (2.0 * fetch + write) * 1024.0


def cpu_baseline(sample_graphs, cfg):
    """The CPU oracle (a literal restatement of the reference's algorithm) on a bounded sample
    of the same workload: the first `sample_graphs` graphs of config 3, one host core."""
    from oracle import grakel_oracle as O
    from grakel_amd.synthetic import er_dataset
    X = er_dataset(sample_graphs, cfg["n"], cfg["p"], cfg["L"], cfg["seed"])
    t0 = time.perf_counter()
    K = O.WLOracle(n_iter=cfg["n_iter"]).fit_transform(X)
    dt = time.perf_counter() - t0
    return dict(value=sample_graphs * sample_graphs / dt, unit="graph-pairs/s", cores=1, kind="port",
                sample="first %d graphs of the config-3 generator (n=%d p=%.2f h=%d), "
                       "oracle.WLOracle.fit_transform, %.1f s, K sum %d"
                       % (sample_graphs, cfg["n"], cfg["p"], cfg["n_iter"], dt, int(K.sum()))), K, X


def end_to_end(eng, full, cfg):
    """What a caller of the estimator sees (never `value`): (a) packed CSR on the host -> float64 K on the
    host (upload, step, 8 N^2 bytes over PCIe into pageable memory), (b) the estimator on Python
    objects (SURVEY.md 8d asks for both walls).  One run each, after one warm-up of (a)."""
    import grakel_amd
    from grakel_amd.synthetic import er_dataset
    N, h = cfg["N"], cfg["n_iter"]

    def packed():
        t0 = time.perf_counter()
        db = eng.upload(full)
        eng.wl_relabel(db, h)
        feat = eng.features(db, h + 1)
        K = eng.gram(feat, 0, to_host=True)
        dt = time.perf_counter() - t0
        feat.close()
        db.close()
        return dt, K

    packed()
    dt_packed, K = packed()
    X = er_dataset(N, cfg["n"], cfg["p"], cfg["L"], cfg["seed"])      # {u: [v, ...]} + {u: label} per graph
    from grakel_amd.batch import wl_batch_from_input
    t0 = time.perf_counter()
    Kobj = grakel_amd.WeisfeilerLehman(n_iter=h).fit_transform(X)
    dt_obj = time.perf_counter() - t0
    t0 = time.perf_counter()
    wl_batch_from_input(X)                       # the ingestion part alone, same (now warm) objects
    dt_ingest = time.perf_counter() - t0
    return {"packed_csr_host_to_host_ms": dt_packed * 1e3, "packed_csr_graph_pairs_per_s": N * N / dt_packed,
            "python_objects_s": dt_obj, "python_objects_graph_pairs_per_s": N * N / dt_obj,
            "of_which_host_ingestion_s": dt_ingest, "same_matrix": bool(np.array_equal(K, Kobj)),
            "note": "packed: H2D of the CSR + step + D2H of the %d MB float64 K into pageable memory; "
                    "objects: grakel_amd.WeisfeilerLehman(n_iter=%d).fit_transform on %d dict graphs"
                    % (N * N * 8 // 1000000, h, N)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--graphs", type=int, default=WORKLOAD["N"], help="(debug) smaller workload")
    ap.add_argument("--cpu-sample", type=int, default=6500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    from grakel_amd import GraphBatch, _lib
    from grakel_amd.engine import get_engine
    from grakel_amd.synthetic import er_dataset_csr

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run" % a.gpus)
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    assert torch.cuda.is_available() and _lib.device_count() > 0, "bench.py needs MI355X GPUs"
    # test hook (tests the N > 1 code path of this script on a ONE-GPU box): GK_BENCH_BACKEND=gloo puts
    # every rank on device GK_BENCH_DEVICE and moves the shard messages with gloo; never set by the driver
    backend = os.environ.get("GK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = int(os.environ.get("GK_BENCH_DEVICE", "0"))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    cfg = dict(WORKLOAD)
    cfg["N"] = a.graphs
    N, h = cfg["N"], cfg["n_iter"]
    eng = get_engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)

    gp, rp, ci, lab = er_dataset_csr(N, cfg["n"], cfg["p"], cfg["L"], cfg["seed"])
    full = GraphBatch(gp, rp, ci, lab, cfg["L"])
    info = {}

    if world == 1:
        db = eng.upload(full)                 # input resident in HBM before the timed region

        pending = []

        def collect():
            # HIP-event time of the previous step's MFMA kernel: read one step late, when it has
            # long finished, so that reading it never drains the queue the host is filling
            while pending:
                f = pending.pop()
                info["gram"] = eng.gram_stats(f)
                f.close()

        def step():
            eng.wl_relabel(db, h)
            feat = eng.features(db, h + 1)
            eng.gram(feat, 0, to_host=False)
            info.update(n_cols=feat.n_cols, n_cols_low=feat.n_cols_low, dtype=feat.dtype,
                        label_counts=db.label_counts, nnz=feat.nnz)
            collect()
            pending.append(feat)
    else:
        from grakel_amd.dist import ShardedWL, shard_bounds
        b = shard_bounds(N, world)
        local = full.slice_graphs(b[rank], b[rank + 1])
        sw = ShardedWL(eng, n_iter=h)

        def step():
            _, i = sw.step(local)
            info.update(i)

    def sync():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    if world == 1:
        collect()
    gram_ms = []
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step()
        if world > 1 or i > 0:
            gram_ms.append(info["gram"][1])      # world == 1: the step before (see collect)
    sync()
    dt = time.perf_counter() - t0
    if world == 1:
        collect()
        gram_ms.append(info["gram"][1])
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-phase device times (one extra, untimed, profiled step; single GPU only)
    phases = None
    if world == 1:
        eng.profile(True)
        step()
        collect()
        phases = {k: round(eng.profile_get(k)[0], 4) for k in ("relabel", "features", "gram")}
        eng.profile(False)

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        flops, _ = info["gram"]
        gram_avg_ms = float(np.mean(gram_ms))
        dtype = ("i8", "f64")[info["dtype"]]
        peak = I8_DENSE_PEAK_TOPS if dtype == "i8" else F64_PEAK_TFLOPS
        achieved = flops / (gram_avg_ms * 1e-3) / 1e12
        d_eff = (info.get("n_cols") or 0) + (info.get("n_cols_low") or 0)
        alg_flops = (2.0 * (N / world) * N * d_eff) if world > 1 else (2.0 * (N * (N + 1) / 2) * d_eff)
        # HBM view of the two integer phases (SURVEY.md 8d algorithmic bytes, int32 everywhere):
        # relabel per level 8E + 12V (signature) + 24V (dictionary pass); features 16V per level
        V_, E_ = int(full.n_nodes), int(full.n_edges)
        phases_hbm = None
        if phases:
            rb = h * (8 * E_ + 12 * V_ + 24 * V_)
            fb = (h + 1) * 16 * V_
            phases_hbm = {
                "relabel": {"algorithmic_bytes": rb, "GB_per_s": rb / (phases["relabel"] * 1e-3) / 1e9,
                            "frac_of_8TBps": rb / (phases["relabel"] * 1e-3) / 8e12},
                "features": {"algorithmic_bytes": fb, "GB_per_s": fb / (phases["features"] * 1e-3) / 1e9,
                             "frac_of_8TBps": fb / (phases["features"] * 1e-3) / 8e12},
                "note": "about 60 dependent launches of 3-60 us over 1 M-element arrays per step, six device->host "
                        "read-backs: launch/latency bound, not bandwidth bound (DESIGN.md 4)"}
        out = {
            "metric": "graph-pairs/sec for NxN WL-subtree(h=%d) fit_transform" % h,
            "value": N * N / (dt / a.steps),
            "unit": "graph-pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE config 3: %d Erdos-Renyi graphs n=%d p=%.2f, %d labels, seed %d, "
                                   "WL-subtree h=%d, full NxN float64 Gram left in HBM"
                                   % (N, cfg["n"], cfg["p"], cfg["L"], cfg["seed"], h),
                       "graphs": N, "nodes": int(full.n_nodes), "edges": int(full.n_edges),
                       "parallelism": "graphs+Gram rows sharded over %d GPU(s)" % world,
                       "label_counts": info.get("label_counts"), "gram_columns_dense": info.get("n_cols"),
                       "gram_columns_rare": info.get("n_cols_low")},
            "roofline": {
                "kernel": ("gram_i8_glds_kernel<2,2,2,2,4> (128x128 tile)" if (info.get("n_cols") or 0) < 8065
                           else "gram_i8_glds_kernel<2,4,4,2,4> (256x256 tile)"),
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": gram_pmc_traffic_bytes(N, dtype) if world == 1 else None,
                "traffic_source": "profiles/r01s_pmc_hbm_bytes.csv (2 x FETCH_SIZE + WRITE_SIZE, KiB)",
                "executed_flops_per_launch": flops, "avg_launch_ms": gram_avg_ms,
                "algorithmic": {
                    # 2*N_rows*N_cols*D_eff with D_eff = label columns occurring in
                    # >= 2 graphs (dense + rare); upper triangle only at 1 GPU (stated), full rows at N GPUs
                    "D_eff": d_eff, "dense_columns": info.get("n_cols"), "rare_columns": info.get("n_cols_low"),
                    "flops": alg_flops, "gram_phase_ms": (phases or {}).get("gram"),
                    "achieved": (alg_flops / ((phases or {}).get("gram") * 1e-3) / 1e12) if phases else None},
                "note": "achieved = integer ops EXECUTED by the MFMA kernel per launch / its avg HIP-event "
                        "duration (1 GPU: only the tiles on/above the diagonal, mirrored on store; "
                        "only the dense columns -- label columns present in < 24 graphs are applied as exact "
                        "pair updates by gram_low_kernel, inside gram_phase_ms; columns whose counts fit 4 bits "
                        "travel as nibbles and are unpacked in registers, the MFMA itself is int8). The kernel "
                        "also writes the whole float64 K (N^2*8 B), which bounds it at ~0.13 ms by HBM."},
            "phases_ms": phases,
            "phases_hbm": phases_hbm,
        }
        if world == 1 and not a.no_cpu_baseline:
            cb, Kcpu, X = cpu_baseline(min(a.cpu_sample, N), cfg)
            out["cpu_baseline"] = cb
            out["end_to_end"] = end_to_end(eng, full, cfg)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

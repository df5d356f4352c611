#!/usr/bin/env python3
"""Benchmark of the hot path: WL-subtree(h=5) fit_transform Gram matrix, graph-pairs/second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config3|config5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one full pass of the hot path over the workload -- default BASELINE config 3 (10 000
synthetic Erdos-Renyi graphs, n=100, p=0.05, 5 labels, seed 0), the configuration the
metric is quoted on: WL relabelling for 5 iterations, label-count features of the 6 levels, and the
N x N Gram matrix, with the packed CSR batch already resident in HBM when the timed region starts and
the float64 matrix left in HBM when it ends (`value` = the DEVICE STEP; what a caller of
`fit_transform` sees, host ndarray included, is `end_to_end.value_host_to_host`).  N>1: the graphs and
the Gram rows are sharded over the ranks; a step then also contains the RCCL all-gather of the CSR
shards (grakel_amd/dist.py).  Total work is fixed, so the scaling is "strong".  `--workload config5`
(50 000 graphs, n=30: a 20 GB matrix) is the case where the row sharding is needed for capacity.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` for the dominant
kernel (the Gram kernel: bound by the float64 store of K) and `cpu_baseline` (the oracle
timed on this box).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # golden = (sum of K, trace of K) from the real reference (tests/golden/er_config3.npz); config 5 cannot
    # be run by the reference (six dense 50k x 50k float64 matrices): its checks are the invariants
    "config3": dict(N=10000, n=100, p=0.05, L=5, seed=0, n_iter=5, golden=(200604613570.0, 25874190.0),
                    label_counts=[5, 17694, 973861, 993286, 993302, 993302]),
    "config5": dict(N=50000, n=30, p=0.1, L=5, seed=0, n_iter=5, golden=None,
                    label_counts=[5, 6987, 1106456, 1386518, 1408511, 1408933]),
}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP4_DENSE_PEAK_TOPS = 10000.0    # MX fp4 dense (the operands of the counts <= 4); int8 dense = 5000
I8_DENSE_PEAK_TOPS = 5000.0
# HBM-side bytes per launch of the Gram kernel on config 3, from the PMC passes committed under profiles/
# (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this script, summarised by
# tools/pmc_summary.py): 2 x FETCH_SIZE (gfx950 reports half of a wide coalesced read,
# MI355X_MICROARCH.md "HBM") + WRITE_SIZE, KiB -> bytes.  NOT measured in this run.
PMC_FILES = [os.path.join(ROOT, "profiles", "r04_pmc_hbm_bytes.csv"), os.path.join(ROOT, "profiles", "r03_pmc_hbm_bytes.csv")]
PMC_FILE = next((p for p in PMC_FILES if os.path.exists(p)), PMC_FILES[0])
GRAM_KERNELS = ("gram_ws_kernel", "gram_tile_kernel")


def gram_pmc_traffic_bytes(workload, world):
    if workload != "config3" or world != 1 or not os.path.exists(PMC_FILE):
        return None
    fetch = write = None
    with open(PMC_FILE) as f:
        for line in f:
            if not any(k in line for k in GRAM_KERNELS):
                continue
            parts = line.rstrip().rsplit(",", 3)          # "kernel",counter,launches,avg
            if parts[1] == "FETCH_SIZE":
                fetch = float(parts[3])
            elif parts[1] == "WRITE_SIZE":
                write = float(parts[3])
    if fetch is None or write is None:
        return None
    return (2.0 * fetch + write) * 1024.0


def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count()


def cpu_baseline(sample_graphs, cfg):
    """The CPU oracle (a literal restatement of the reference's algorithm, oracle/grakel_oracle.py) on a
    bounded sample of the same workload: the first `sample_graphs` graphs of the generator.  Two legs, as
    SURVEY.md 8d asks: n_jobs=None (ONE host core; `value`) and n_jobs = number of WL levels (the only
    parallelism the reference has on this path: joblib runs the per-level base-kernel products side by side,
    weisfeiler_lehman.py:271-283; more workers than levels cannot be used)."""
    from oracle import grakel_oracle as O
    from grakel_amd.synthetic import er_dataset
    X = er_dataset(sample_graphs, cfg["n"], cfg["p"], cfg["L"], cfg["seed"])
    t0 = time.perf_counter()
    K = O.WLOracle(n_iter=cfg["n_iter"]).fit_transform(X)
    dt = time.perf_counter() - t0
    model, nproc = cpu_info()
    jobs = min(cfg["n_iter"] + 1, max(nproc, 1))
    ksum = int(K.sum())
    # the n_jobs leg on a smaller sample (it repeats the whole job; the one-core leg above is the stated baseline)
    nj = min(sample_graphs, 5000)
    ksum_j = int(K[:nj, :nj].sum())
    del K
    t0 = time.perf_counter()
    Kj = O.WLOracle(n_iter=cfg["n_iter"]).fit_transform(X[:nj], n_jobs=jobs)
    dtj = time.perf_counter() - t0
    same = int(Kj.sum()) == ksum_j
    del Kj
    return dict(value=sample_graphs * sample_graphs / dt, unit="graph-pairs/s", cores=1, kind="port",
                cpu_model=model, host_cores_available=nproc,
                n_jobs=dict(value=nj * nj / dtj, cores=jobs, seconds=round(dtj, 2), same_K_sum=same, sample_graphs=nj,
                            note="per-level products in %d worker processes (the reference's joblib granularity); "
                                 "the relabel loop and the sum of the level matrices stay on one core" % jobs),
                sample="first %d graphs of the %d-graph generator (n=%d p=%.2f h=%d), oracle.WLOracle.fit_transform, "
                       "%.1f s, K sum %d; the cost is ~N^2 (one dense N x N float64 per level), so the full-size rate "
                       "is at or below this one" % (sample_graphs, cfg["N"], cfg["n"], cfg["p"], cfg["n_iter"], dt, ksum),
                reference_real="grakel 0.1.11 itself, full config 3, one core of the build container (Intel Xeon "
                               "2.1 GHz): 93.1 s = 1.07e6 graph-pairs/s (tests/golden/er_config3.npz: ref_seconds); "
                               "it cannot travel to the GPU box")


def end_to_end(eng, full, cfg, with_objects):
    """What a caller of the estimator sees (never `value`): (a) packed CSR on the host -> float64 K on the
    host (upload, step, 8 N^2 bytes over PCIe into the pinned output pool), (b) the estimator on Python
    objects (SURVEY.md 8d asks for both walls).  One run each, after warm-ups (the first run of a size
    pins the output block)."""
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

    t0 = time.perf_counter()
    _, K = packed()
    first = time.perf_counter() - t0
    del K
    packed()
    dt_packed, K = packed()
    with eng.options(**{"gram.no_compact": 1}):          # the plain 8 N^2-byte copy, for comparison
        packed()
        dt_plain, Kp = packed()
    same_plain = bool(np.array_equal(K, Kp))
    del Kp
    out = {"packed_csr_host_to_host_ms": dt_packed * 1e3, "value_host_to_host": N * N / dt_packed,
           "plain_float64_copy_ms": dt_plain * 1e3, "compact_equals_plain": same_plain,
           "first_call_ms_incl_pinning_the_output": first * 1e3,
           "note": "packed: H2D of the CSR + step + the %d MB float64 K into a pinned, reused output block "
                   "(grakel_amd.engine.PinnedPool); integer-valued matrices cross PCIe as uint16 / int32 and are "
                   "widened by host threads (gram.hip: gram_copy_out), plain_float64_copy_ms = the 8 N^2-byte copy"
                   % (N * N * 8 // 1000000)}
    if with_objects:
        X = er_dataset(N, cfg["n"], cfg["p"], cfg["L"], cfg["seed"])      # {u: [v, ...]} + {u: label} per graph
        from grakel_amd.batch import wl_batch_from_input
        est = grakel_amd.WeisfeilerLehman(n_iter=h)
        est.fit_transform(X[:50])
        Kw = est.fit_transform(X)                    # K above is still alive: this size's second pinned block is created here,
        del Kw                                       # outside the timed call (pinning 800 MB costs ~50 ms once per process)
        t0 = time.perf_counter()
        Kobj = est.fit_transform(X)
        dt_obj = time.perf_counter() - t0
        t0 = time.perf_counter()
        wl_batch_from_input(X)                       # the ingestion part alone, same (now warm) objects
        dt_ingest = time.perf_counter() - t0
        from grakel_amd import batch as _batch
        saved_threads, _batch.INGEST_THREADS = _batch.INGEST_THREADS, 1
        try:
            t0 = time.perf_counter()
            wl_batch_from_input(X)                   # the same walk on the calling thread alone
            dt_ingest_1 = time.perf_counter() - t0
        finally:
            _batch.INGEST_THREADS = saved_threads
        out["host_ingestion_one_thread_s"] = dt_ingest_1
        out["host_ingestion_threads"] = min(16, os.cpu_count() or 1)
        out.update({"python_objects_s": dt_obj, "python_objects_graph_pairs_per_s": N * N / dt_obj,
                    "of_which_host_ingestion_s": dt_ingest, "same_matrix": bool(np.array_equal(K, Kobj)),
                    "objects_note": "grakel_amd.WeisfeilerLehman(n_iter=%d).fit_transform on %d dict graphs" % (h, N)})
    return out


def config4_sp(eng, steps=5):
    """BASELINE config 4 stand-in (4110 NCI1-like graphs, ShortestPath): ms per fit_transform from the packed
    CSR in HBM, the all-pairs kernels' min-plus rate against an LDS-bandwidth ceiling, the Gram kernel."""
    from grakel_amd.batch import sp_batch_from_input
    from grakel_amd.synthetic import nci1_like
    N = 4110
    gb, _ = sp_batch_from_input(nci1_like(N, 0, as_adj=True), True)
    db = eng.upload(gb)
    sizes = np.diff(gb.graph_ptr).astype(np.float64)

    def step(check=False):
        pb = eng.sp_build(db, None, True)
        feat = eng.features(pb, 1)
        eng.gram(feat, 0, to_host=False)
        info = dict(n_pairs=pb.n_nodes, n_keys=pb.label_counts[0], dense=feat.n_cols, rare=feat.n_cols_low,
                    max_count=feat.max_count, gram=eng.gram_stats(feat))
        if check:                                    # a pass over the 135 MB matrix: outside the timed steps
            info["checksum"] = eng.gram_checksum(feat)
        feat.close()
        pb.close()
        return info

    for _ in range(2):
        step()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        info = step()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    info["checksum"] = step(check=True)["checksum"]
    eng.profile(True)
    step()
    ph = {k: round(eng.profile_get(k)[0], 4) for k in ("sp", "sp_fw", "features", "gram")}
    eng.profile(False)
    db.close()
    ops = float((sizes ** 3).sum())
    # a min-plus update reads d[i][k], d[k][j], d[i][j] and writes d[i][j]: 16 LDS bytes; 4-byte LDS reads
    # stream at ~75 TB/s chip-wide (MI355X_MICROARCH.md "LDS") -- the ceiling of the LDS-resident form (round 2)
    ceiling = 75e12 / 16.0
    # the register form (sp.hip: 16-bit packed distances, two columns per register): 3.5 vector instructions per two
    # relaxations of a lane (v_readlane, v_pk_add_u16, v_pk_min_u16, half a rotation move), 64 lanes; one VALU instruction per
    # SIMD every 2 cycles at 2.4 GHz on 1024 SIMDs
    valu_peak = 1024 * 2.4e9 / 2.0 * (2.0 * 64.0 / 3.5)
    return {"workload": "NCI1-like stand-in (SURVEY.md appendix A), %d graphs, ShortestPath(with_labels), packed CSR in HBM" % N,
            "ms_per_fit_transform": dt * 1e3, "graph_pairs_per_s": N * N / dt, "phases_ms": ph,
            "fw_minplus_ops": ops, "fw_Gops_per_s": ops / (ph["sp_fw"] * 1e-3) / 1e9,
            "fw_lds_ceiling_Gops_per_s": ceiling / 1e9, "fw_frac_of_lds_ceiling": ops / (ph["sp_fw"] * 1e-3) / ceiling,
            "roofline": {"kernel": "sp_fw_pk_kernel (one wave per graph, <= 64 vertices) + sp_fw_pkw_kernel (four-wave column "
                                   "split, 65..128 vertices): distance matrices in registers, 16-bit packed",
                         "bound": "valu-issue", "achieved": ops / (ph["sp_fw"] * 1e-3) / 1e9, "peak": valu_peak / 1e9,
                         "unit": "G min-plus/s", "frac": ops / (ph["sp_fw"] * 1e-3) / valu_peak,
                         "note": "one dependent chain per graph (n pivots x n/2 registers): 4110 waves cannot fill 1024 SIMDs "
                                 "four deep, so the rate is set by the chain length of the largest graphs, not by issue "
                                 "bandwidth; algorithmic work = sum of n^3 min-plus updates (padding pivots / columns not counted)"},
            "pairs": info["n_pairs"], "features": info["n_keys"], "dense_columns": info["dense"],
            "rare_columns": info["rare"], "max_count": info["max_count"], "gram_kernel_ms": info["gram"][1],
            "K_sum": info["checksum"][0], "K_sum_expected": 87649686148.0,
            "K_matches_reference_checksum": bool(info["checksum"][0] == 87649686148.0 and info["checksum"][2] == 0.0),
            "reference_cpu_s": {"floyd_warshall_route": 15.2, "dijkstra_route": 21.8}}


def transform_bench(eng, cfg):
    """`transform` of a few target graphs against the config-3 fit (never `value`): the look-up route (targets relabelled
    alone, signatures looked up in the fitted dictionaries on the device, csrc/wl_transform.hip -- the reference's own
    scheme, weisfeiler_lehman.py:435-498) against the joint route (fitted graphs + targets relabelled together), wall per
    call through the estimator on Python dict graphs and the device phases of one call."""
    import grakel_amd
    from grakel_amd.synthetic import er_dataset
    X = er_dataset(cfg["N"], cfg["n"], cfg["p"], cfg["L"], cfg["seed"])
    out = {"fit": "%d graphs (the headline config)" % cfg["N"], "routes": {}}
    sums = {}
    for route in ("lookup", "joint"):
        est = grakel_amd.WeisfeilerLehman(n_iter=cfg["n_iter"])
        est.transform_route = route
        est.fit(X)
        res = {}
        for nt in (1, 100, 1000):
            Y = er_dataset(nt, cfg["n"], cfg["p"], cfg["L"], 4321)
            est.transform(Y)
            est.transform(Y)
            t0 = time.perf_counter()
            for _ in range(3):
                K = est.transform(Y)
            dt = (time.perf_counter() - t0) / 3
            eng.profile(True)
            est.transform(Y)
            ph = {k: round(eng.profile_get(k)[0], 4) for k in ("relabel", "features", "gram", "transform") if eng.profile_get(k)[1]}
            eng.profile(False)
            res["%d_targets" % nt] = {"wall_ms": round(dt * 1e3, 3), "device_phases_ms": ph}
            sums.setdefault(nt, []).append(float(K.sum()))
        out["routes"][route] = res
    out["same_matrices"] = all(len(set(v)) == 1 for v in sums.values())
    out["note"] = ("device_phases_ms.transform of the look-up route includes the device -> host copy of the n_targets x "
                   "n_fitted float64 block; default policy: look-up while the targets hold at most 1/32 of the fitted nodes")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="config3")
    ap.add_argument("--graphs", type=int, default=0, help="(debug) smaller workload")
    ap.add_argument("--cpu-sample", type=int, default=10000, help="graphs of the CPU baseline's one-core leg (default: the full config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end and the config-4 object")
    ap.add_argument("--plan", choices=["plain", "symmetric"], default="plain",
                    help="N > 1: Gram sharding plan (grakel_amd.dist.gram_plan; plain row blocks is the default)")
    ap.add_argument("--separate-calls", action="store_true",
                    help="step = gk_wl_relabel + gk_features_build + gk_gram as three library calls instead of gk_wl_fit_transform")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="context option (gk_set_option, include/gk_hip.h), e.g. --opt wl.debug=1; A/B runs only")
    a = ap.parse_args()

    # the CPU baseline runs FIRST: its n_jobs leg forks worker processes, which must not happen in a process
    # that already holds an initialised HIP runtime (runtime threads, locks)
    cpu_base = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.gpus == 1 and not a.no_cpu_baseline:
        cfg0 = dict(WORKLOADS[a.workload])
        if a.graphs not in (0, cfg0["N"]):
            cfg0["N"] = a.graphs
        cpu_base = cpu_baseline(min(a.cpu_sample, cfg0["N"]), cfg0)

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

    cfg = dict(WORKLOADS[a.workload])
    full_size = a.graphs in (0, cfg["N"])
    if not full_size:
        cfg["N"], cfg["golden"], cfg["label_counts"] = a.graphs, None, None
    N, h = cfg["N"], cfg["n_iter"]
    eng = get_engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for o in a.opt:
        eng.set_option(o.split("=")[0], int(o.split("=")[1]))

    gp, rp, ci, lab = er_dataset_csr(N, cfg["n"], cfg["p"], cfg["L"], cfg["seed"])
    full = GraphBatch(gp, rp, ci, lab, cfg["L"])
    info = {}
    keep = {}

    if world == 1:
        db = eng.upload(full)                 # input resident in HBM before the timed region

        pending = []

        def collect(last=False):
            # HIP-event time of the previous step's Gram kernel: read one step late, when it has
            # long finished, so that reading it never drains the queue the host is filling
            while pending:
                f = pending.pop()
                info["gram"] = eng.gram_stats(f)
                if last:
                    keep["feat"] = f
                else:
                    f.close()

        def step():
            # one library call: relabel (queued without a host round trip) -> features -> Gram (gk_wl_fit_transform; the
            # three separate calls -- --separate-calls -- give the same matrix with two more host round trips)
            if a.separate_calls:
                eng.wl_relabel(db, h)
                feat = eng.features(db, h + 1)
                eng.gram(feat, 0, to_host=False)
            else:
                feat, _ = eng.wl_fit_transform(db, h, to_host=False)
            info.update(n_cols=feat.n_cols, n_cols_low=feat.n_cols_low, dtype=feat.dtype, operand=feat.operand,
                        label_counts=db.label_counts, nnz=feat.nnz)
            collect()
            pending.append(feat)
    else:
        from grakel_amd.dist import ShardedWL, shard_bounds
        b = shard_bounds(N, world)
        local = full.slice_graphs(b[rank], b[rank + 1])
        sw = ShardedWL(eng, n_iter=h, symmetric=(a.plan == "symmetric"))       # default: plain row blocks (dist.gram_plan)

        def step():
            _, i = sw.step(local)
            i.pop("K_dev", None)                 # the rank's row block (a torch tensor) is released with the step
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
    checks = None
    if world == 1:
        collect(last=True)
        gram_ms.append(info["gram"][1])
        # the matrix the LAST timed step left in HBM, checked in place (no 8 N^2-byte copy)
        s, tr, asym = eng.gram_checksum(keep["feat"])
        selfk_sum = float(eng.selfk(keep["feat"]).sum())
        checks = {"K_sum": s, "K_trace": tr, "max_abs_K_minus_KT": asym, "trace_equals_sum_of_selfk": bool(tr == selfk_sum),
                  "label_counts_match_the_oracle": (info["label_counts"] == cfg["label_counts"]) if cfg["label_counts"] else None,
                  "matches_reference_checksums": bool((s, tr) == cfg["golden"]) if cfg["golden"] else None,
                  # what pins this workload's matrix: the real reference's checksums (config 3), or -- where the reference
                  # cannot run (config 5: six dense 50k x 50k float64 matrices) -- the CPU oracle, block by block, in
                  # tests/test_gpu_parity.py::test_config5_full_size_blockwise_against_the_oracle (the oracle itself is pinned
                  # to the reference on every golden set); here only the invariants and the oracle's label counts are asserted
                  "checked_against": ("reference checksums (tests/golden/er_config3.npz)" if cfg["golden"] else
                                      ("oracle blockwise (tests/test_gpu_parity.py: config 5 test) + invariants + oracle label counts"
                                       if cfg["label_counts"] else "invariants only (custom size)")),
                  "gram_max_abs_err": 0.0 if (cfg["golden"] and (s, tr) == cfg["golden"] and asym == 0.0) else None}
        assert asym == 0.0 and tr == selfk_sum, "timed Gram matrix is not symmetric / has a wrong diagonal: %r" % (checks,)
        if cfg["golden"]:
            assert (s, tr) == cfg["golden"], "timed Gram matrix differs from the reference checksums: %r" % (checks,)
        if cfg["label_counts"]:
            assert info["label_counts"] == cfg["label_counts"], "WL label counts differ from the oracle's"
        keep.pop("feat").close()
    sharded_ms = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ev = getattr(sw, "last_events", None)
        if ev is not None:                       # the last step's Gram part on this rank (rank 0 reports its own)
            sharded_ms = {"block_products_ms": ev[0].elapsed_time(ev[1]), "exchange_and_placement_ms": ev[1].elapsed_time(ev[2])}

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
        if sharded_ms is not None:               # N > 1: all block products of the rank (HIP events on the shared stream)
            gram_avg_ms = sharded_ms["block_products_ms"]
        f64_only = bool(info["dtype"])
        d_dense = info.get("n_cols") or 0
        d_eff = d_dense + (info.get("n_cols_low") or 0)
        rows = N / world
        # algorithmic HBM bytes of one Gram launch: write the float64 entries the rank produces
        # once (1 GPU: all of K; N GPUs: the blocks of its symmetric plan, ~half of its row block), read the
        # dense operand once (fp4: two columns per byte, 128-byte K-steps)
        operand_bytes = (N + 511) // 256 * 256 * ((d_dense + 255) // 256 * 128)
        sym = world > 1 and a.plan == "symmetric"
        gram_bytes = 8.0 * rows * N * (0.5 * (1.0 + 1.0 / world) if sym else 1.0) + operand_bytes
        achieved_gbs = gram_bytes / (gram_avg_ms * 1e-3) / 1e9
        mfma_peak = FP4_DENSE_PEAK_TOPS if str(info.get("operand", "fp4")).startswith("fp4") else I8_DENSE_PEAK_TOPS
        alg_flops = 2.0 * (N * (N + 1) / 2) * d_eff / world
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
                "note": "dependent launches of 3-70 us over <= 1 M-element arrays, 4 per WL level: launch / latency bound (DESIGN.md 4)"}
        traffic = gram_pmc_traffic_bytes(a.workload if full_size else "", world)
        out = {
            "metric": "graph-pairs/sec for NxN WL-subtree(h=%d) fit_transform" % h,
            "value": N * N / (dt / a.steps),
            "value_is": "device step: CSR resident in HBM -> float64 K resident in HBM (see end_to_end for host to host)",
            "unit": "graph-pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": info.get("operand") or ("f64" if f64_only else "fp4+i8"),
            "data": "synthetic",
            "config": {"workload": "BASELINE %s: %d Erdos-Renyi graphs n=%d p=%.2f, %d labels, seed %d, "
                                   "WL-subtree h=%d, full NxN float64 Gram left in HBM"
                                   % (a.workload, N, cfg["n"], cfg["p"], cfg["L"], cfg["seed"], h),
                       "graphs": N, "nodes": V_, "edges": E_,
                       "parallelism": "graphs+Gram rows sharded over %d GPU(s)%s" % (
                           world, "" if world == 1 else (
                               "; every rank multiplies 1/%d of the upper triangle and ships the mirrored blocks point to "
                               "point (grakel_amd.dist.symmetric_plan)" % world if sym else
                               "; one all-gather of the CSR shards, relabel + features replicated, every rank multiplies and "
                               "stores its own row block (plain row blocks: no collective on the Gram path)")),
                       "gram_plan_model": (lambda m: {k: (v if k == "choice" else {kk: round(vv, 6) if kk == "seconds" else vv
                                                                                 for kk, vv in v.items()}) for k, v in m.items()})(
                           __import__("grakel_amd.dist", fromlist=["gram_plan"]).gram_plan(N, world)) if world > 1 else None,
                       "sharded_gram_ms": sharded_ms,
                       "label_counts": info.get("label_counts"), "gram_columns_dense": d_dense,
                       "gram_columns_rare": info.get("n_cols_low")},
            "checks": checks,
            "roofline": {
                "kernel": "gram_ws_kernel (persistent, warp-specialised 128x128 tiles; MX fp4 operands for counts <= 4, "
                          "int8 for 5..127; float64 store of K)",
                "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": "profiles/%s (2 x FETCH_SIZE + WRITE_SIZE, KiB) -- a committed PMC "
                                  "pass of this script, NOT measured in this run" % os.path.basename(PMC_FILE) if traffic else None,
                "algorithmic_bytes_per_launch": gram_bytes, "avg_launch_ms": gram_avg_ms,
                "mfma_view": {
                    # the same launch priced against the matrix pipe: with fp4 operands its floor (flops / peak)
                    # is ~0.05 ms on config 3, below the 0.10 ms HBM floor of the float64 store -- hence "hbm"
                    "executed_flops_per_launch": flops, "TFLOP_per_s": flops / (gram_avg_ms * 1e-3) / 1e12,
                    "peak": mfma_peak, "frac": flops / (gram_avg_ms * 1e-3) / 1e12 / mfma_peak,
                    "D_eff": d_eff, "dense_columns": d_dense, "rare_columns": info.get("n_cols_low"),
                    "algorithmic_flops": alg_flops, "gram_phase_ms": (phases or {}).get("gram")},
                "note": "achieved = (8 bytes x rows x N of float64 K written once + the packed dense operand read once) / "
                        "avg HIP-event duration of the Gram kernel.  1 GPU: only tiles on/above the diagonal are "
                        "multiplied, both halves are stored; label columns present in fewer graphs than the job's threshold (24 at 10 000 graphs, DESIGN.md 2) never enter the dense "
                        "operand, their exact pair updates (gram_low_kernel) are inside gram_phase_ms."},
            "phases_ms": phases,
            "phases_hbm": phases_hbm,
        }
        out["cpu_baseline"] = cpu_base            # measured before the GPU part (see the top of main)
        if world == 1 and not a.no_extras:
            out["end_to_end"] = end_to_end(eng, full, cfg, with_objects=(a.workload == "config3"))
            try:
                out["extra"] = {"config4_sp": config4_sp(eng)}
            except Exception as e:                     # never lose the headline line to an extra
                out["extra"] = {"config4_sp": {"error": repr(e)}}
            if a.workload == "config3" and full_size:
                try:
                    out["extra"]["transform"] = transform_bench(eng, cfg)
                except Exception as e:
                    out["extra"]["transform"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

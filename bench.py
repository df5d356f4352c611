#!/usr/bin/env python3
"""Benchmark of the hot path: WL-subtree(h=5) fit_transform Gram matrix, graph-pairs/second.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches ITSELF under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per GPU over
RCCL); launched that way by somebody else (the driver) it just runs its rank.

A *step* is one full pass of the hot path over the workload: WL relabelling for h iterations, label-count features
of the h + 1 levels and the N x N Gram matrix.  Workloads (`--workload`):

  config3  (default) BASELINE config 3: 10 000 Erdos-Renyi graphs, n=100, p=0.05, 5 labels -- the configuration the
           metric is quoted on
  config5  BASELINE config 5: 50 000 graphs, n=30 (a 20 GB matrix)
  config6  200 000 graphs, n=30: the 320 GB matrix does NOT fit one GPU -- every rank multiplies its rows in 40 GB
           sub-blocks that reuse one device buffer (the workload where the row sharding is the point)
  nci1 | dd | reddit | collab   stand-ins for the TU datasets the reference publishes its running times on
           (grakel_amd/synthetic.py PUBLISHED_LIKE; goldens from the real reference in tests/golden/pub_*.npz)

What the JSON line says (rank 0 prints ONE line; contract in the task statement):

  value / ms_per_step     the DEVICE STEP: packed CSR batch resident in HBM when the timed region starts, float64 K
                          resident in HBM when it ends.  The task contract fixes this ("inputs already resident in HBM
                          ...; the PCIe-inclusive rate is never `value`").
  host_to_host            what a caller of the C ABI sees, timed over the same number of steps in its own bracketed
                          region: packed CSR on the host -> float64 K on the host, unnormalised AND normalised
                          (SURVEY.md 8d's "from pre-packed CSR" wall); `from_python_objects`: the estimator on 10 000
                          Python dict graphs (8d's "from Python objects" wall).
  roofline                the dominant kernel (the Gram tile kernel), algorithmic bytes / HIP-event duration
  cpu_baseline            the CPU oracle on a bounded sample of the same workload, on this box's host cores
  N > 1                   `per_rank`: exchange / replicated relabel + features / Gram rows / step, per rank (HIP events),
                          `end_to_end_ms`: the same step with every rank's row block copied to its host
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ER_WORKLOADS = {
    # golden = (sum of K, trace of K) from the real reference (tests/golden/er_config3.npz); config 5 / 6 cannot be run by
    # the reference (dense N x N float64 per level): their checks are invariants + the oracle's label counts
    "config3": dict(N=10000, n=100, p=0.05, L=5, seed=0, n_iter=5, golden=(200604613570.0, 25874190.0),
                    golden_file="er_config3.npz", label_counts=[5, 17694, 973861, 993286, 993302, 993302]),
    "config5": dict(N=50000, n=30, p=0.1, L=5, seed=0, n_iter=5, golden=None, golden_file=None,
                    label_counts=[5, 6987, 1106456, 1386518, 1408511, 1408933]),
    "config6": dict(N=200000, n=30, p=0.1, L=5, seed=0, n_iter=5, golden=None, golden_file=None, label_counts=None,
                    block_rows=25000),
}
PUBLISHED = ("nci1", "dd", "reddit", "collab")
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP4_DENSE_PEAK_TOPS = 10000.0    # MX fp4 dense (the operands of the counts <= 4); int8 dense = 5000
I8_DENSE_PEAK_TOPS = 5000.0
F64_DENSE_PEAK_TOPS = 78.6       # v_mfma_f64_16x16x4_f64
# HBM-side bytes per launch of the Gram kernel on config 3, from the PMC passes committed under profiles/
# (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of this script, summarised by
# tools/pmc_summary.py): 2 x FETCH_SIZE (gfx950 reports half of a wide coalesced read,
# MI355X_MICROARCH.md "HBM") + WRITE_SIZE, KiB -> bytes.  NOT measured in this run.
PMC_FILES = [os.path.join(ROOT, "profiles", "r%02d_pmc_hbm_bytes.csv" % r) for r in (6, 5, 4, 3)]
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


# ---------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------
class Workload(object):
    """name, cfg, the packed host batch, how to get the same graphs as Python objects, the golden (if any)."""

    def __init__(self, name, graphs=0):
        from grakel_amd import GraphBatch
        self.name = name
        self.golden = None
        if name in ER_WORKLOADS:
            from grakel_amd.synthetic import er_dataset_csr
            cfg = dict(ER_WORKLOADS[name])
            self.full_size = graphs in (0, cfg["N"])
            if not self.full_size:
                cfg.update(N=graphs, golden=None, golden_file=None, label_counts=None)
            self.cfg = cfg
            gp, rp, ci, lab = er_dataset_csr(cfg["N"], cfg["n"], cfg["p"], cfg["L"], cfg["seed"])
            self.batch = GraphBatch(gp, rp, ci, lab, cfg["L"])
            self.describe = ("BASELINE %s: %d Erdos-Renyi graphs n=%d p=%.2f, %d labels, seed %d, WL-subtree h=%d"
                             % (name, cfg["N"], cfg["n"], cfg["p"], cfg["L"], cfg["seed"], cfg["n_iter"]))
            if name == "config6":
                self.describe = self.describe.replace("BASELINE config6", "config 6 (not a BASELINE config: the matrix that does "
                                                      "not fit one GPU)")
            if cfg.get("golden_file"):
                self.golden = dict(np.load(os.path.join(ROOT, "tests", "golden", cfg["golden_file"])))
            self.published = None
        else:
            from grakel_amd import synthetic as S
            gen, pub = S.PUBLISHED_LIKE[name]
            self._graphs = gen()
            self.full_size = graphs in (0, len(self._graphs))
            if not self.full_size:
                self._graphs = self._graphs[:graphs]
            gp, rp, ci, lab, nl = S.as_csr(self._graphs)
            self.batch = GraphBatch(gp, rp, ci, lab, nl)
            self.cfg = dict(N=len(self._graphs), n_iter=5, golden=None, label_counts=None)
            path = os.path.join(ROOT, "tests", "golden", "pub_%s.npz" % name)
            if self.full_size and os.path.exists(path):
                self.golden = dict(np.load(path))
                self.cfg["golden"] = (float(self.golden["K_sum"][0]), float(self.golden["K_trace"][0]))
                self.cfg["label_counts"] = [int(c) for c in self.golden["label_counts"]]
            sizes = np.diff(gp)
            self.describe = ("%s-like stand-in (grakel_amd/synthetic.py; the reference publishes %s for WL-VH on the real set, "
                             "doc/benchmarks/evaluation.rst:19-73): %d graphs, %d..%d vertices (mean %.1f), %d input labels, "
                             "WL-subtree h=5" % (name.upper(), pub["WL-VH"], len(sizes), sizes.min(), sizes.max(), sizes.mean(), nl))
            self.published = pub

    N = property(lambda self: self.cfg["N"])
    h = property(lambda self: self.cfg["n_iter"])

    def objects(self, m=None):
        """The first m graphs in the grakel input form `[{u: [v, ...]}, {u: label}]`."""
        m = self.N if m is None else min(int(m), self.N)
        if self.name in ER_WORKLOADS:
            from grakel_amd.synthetic import er_dataset
            c = self.cfg
            return er_dataset(m, c["n"], c["p"], c["L"], c["seed"])
        from grakel_amd import synthetic as S
        return S.as_grakel(self._graphs[:m])


def cpu_baseline(wl, sample_graphs):
    """The CPU oracle (a literal restatement of the reference's algorithm, oracle/grakel_oracle.py) on a
    bounded sample of the same workload: the first `sample_graphs` graphs of the generator.  Two legs, as
    SURVEY.md 8d asks: n_jobs=None (ONE host core; `value`) and n_jobs = number of WL levels (the only
    parallelism the reference has on this path: joblib runs the per-level base-kernel products side by side,
    weisfeiler_lehman.py:271-283; more workers than levels cannot be used)."""
    from oracle import grakel_oracle as O
    X = wl.objects(sample_graphs)
    sample_graphs = len(X)
    h = wl.h
    t0 = time.perf_counter()
    K = O.WLOracle(n_iter=h).fit_transform(X)
    dt = time.perf_counter() - t0
    model, nproc = cpu_info()
    jobs = min(h + 1, max(nproc, 1))
    ksum = int(K.sum())
    # the n_jobs leg on a smaller sample (it repeats the whole job; the one-core leg above is the stated baseline)
    nj = min(sample_graphs, 5000)
    ksum_j = int(K[:nj, :nj].sum())
    del K
    t0 = time.perf_counter()
    Kj = O.WLOracle(n_iter=h).fit_transform(X[:nj], n_jobs=jobs)
    dtj = time.perf_counter() - t0
    same = int(Kj.sum()) == ksum_j
    del Kj
    out = dict(value=sample_graphs * sample_graphs / dt, unit="graph-pairs/s", cores=1, kind="port",
               cpu_model=model, host_cores_available=nproc,
               n_jobs=dict(value=nj * nj / dtj, cores=jobs, seconds=round(dtj, 2), same_K_sum=same, sample_graphs=nj,
                           note="per-level products in %d worker processes (the reference's joblib granularity); "
                                "the relabel loop and the sum of the level matrices stay on one core" % jobs),
               sample="first %d graphs of the %d-graph workload (%s), oracle.WLOracle.fit_transform, %.1f s, K sum %d; the "
                      "cost is ~N^2 (one dense N x N float64 per level), so the full-size rate is at or below this one"
                      % (sample_graphs, wl.N, wl.name, dt, ksum))
    if wl.name == "config3":
        out["reference_real"] = ("grakel 0.1.11 itself, full config 3, one core of the build container (Intel Xeon 2.1 GHz): "
                                 "93.1 s = 1.07e6 graph-pairs/s (tests/golden/er_config3.npz: ref_seconds); it cannot travel "
                                 "to the GPU box")
    elif wl.golden is not None and "ref_seconds" in wl.golden:
        rs = float(wl.golden["ref_seconds"][0])
        out["reference_real"] = ("grakel 0.1.11 itself on the full set, one core of the build container (Intel Xeon 2.1 GHz): "
                                 "%.1f s = %.3g graph-pairs/s (tests/golden/pub_%s.npz: ref_seconds)" % (rs, wl.N * wl.N / rs, wl.name))
    return out


def check_host_matrix(K, golden):
    """Entry-wise comparison of a host matrix with what the real reference produced (tests/golden): 20 000 sampled
    entries, the diagonal, a 64 x 64 block and every row sum.  -> (max abs error, entries compared)."""
    err, n = 0.0, 0
    i, j, v = golden["samp_i"], golden["samp_j"], golden["samp_v"].astype(np.float64)
    err = max(err, float(np.abs(K[i, j] - v).max()))
    n += len(v)
    err = max(err, float(np.abs(np.diagonal(K) - golden["diag"].astype(np.float64)).max()))
    n += K.shape[0]
    B = golden["K_block"].astype(np.float64)
    err = max(err, float(np.abs(K[:B.shape[0], :B.shape[1]] - B).max()))
    n += B.size
    err = max(err, float(np.abs(K.sum(axis=1) - golden["row_sums"].astype(np.float64)).max()))
    return err, n + K.shape[0]


def _read_first(path):
    try:
        with open(path) as f:
            return f.readline().strip()
    except OSError:
        return None


def host_to_host(eng, wl, steps, warmup):
    """Packed CSR on the host -> float64 K on the host (upload, step, the matrix over PCIe into the pinned output pool),
    unnormalised and normalised, each timed over `steps` calls in its own bracketed region (the first call of a size pins
    the output block: warm-up)."""
    N, h = wl.N, wl.h
    out = {}
    keep = {}

    def one(norm):
        db = eng.upload(wl.batch)
        feat, K = eng.wl_fit_transform(db, h, normalize=norm, to_host=True)
        feat.close()
        db.close()
        return K

    for tag, norm in (("unnormalised", 0), ("normalised", 2)):
        for _ in range(max(warmup, 2)):
            K = one(norm)
            del K
        eng.synchronize()
        calls, stages = [], []
        t0 = time.perf_counter()
        for _ in range(steps):
            t1 = time.perf_counter()
            K = one(norm)
            calls.append((time.perf_counter() - t1) * 1e3)
            stages.append(eng.host_copy_stats()[1])            # a few loads: what the call just did (gk_host_copy_stats)
            if _ + 1 < steps:
                del K
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps
        keep[tag] = K
        # the mean over the bracketed region is the figure of record; median and minimum say how much of it is the host's
        # noise (these walls are host-memory and host-thread work: other tenants of the box's CPUs show up here, not in `value`)
        out[tag] = {"ms_per_step": dt * 1e3, "value": N * N / dt, "steps": steps, "median_ms": float(np.median(calls)),
                    "min_ms": float(np.min(calls)), "max_ms": float(np.max(calls)),
                    # where the wall goes (medians over the timed calls): the copy-out = pack kernel + PCIe + widening by
                    # host threads; the rest of a call is the upload of the packed CSR and the device step
                    "copy_out": {"form": stages[-1]["form"], "widening_threads": stages[-1]["widening_threads"],
                                 "pcie_MB": stages[-1]["pcie_bytes"] / 1e6,
                                 "median_ms_until_last_chunk_landed": float(np.median([x["ms_until_last_chunk_landed"] for x in stages])),
                                 "median_ms_copy_out": float(np.median([x["ms_copy_out"] for x in stages])),
                                 "median_widen_busy_ms_per_thread": float(np.median([x["widen_busy_ms_mean"] for x in stages])),
                                 "median_widen_busy_ms_slowest_thread": float(np.median([x["widen_busy_ms_max"] for x in stages]))}}
    Ku, Kn = keep["unnormalised"], keep["normalised"]
    d = np.sqrt(np.diagonal(Ku))
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = np.nan_to_num(Ku[:2048] / np.outer(d[:2048], d))
        rel = np.abs(Kn[:2048] - ref) / np.maximum(np.abs(ref), 1e-300)
    out["normalised"]["max_rel_err_vs_unnormalised_over_sqrt_diag_first_2048_rows"] = float(rel.max())
    out["normalised"]["diagonal_is_one"] = bool(np.all(np.diagonal(Kn)[d > 0] == 1.0))
    with eng.options(**{"gram.no_compact": 1}):          # the plain 8 N^2-byte copy, for comparison
        one(0)
        t0 = time.perf_counter()
        Kp = one(0)
        out["plain_float64_copy_ms"] = (time.perf_counter() - t0) * 1e3
    out["compact_equals_plain"] = bool(np.array_equal(Ku, Kp))
    # the host this ran on, as the library sees it: its widening / ingestion threads are min(32, online CPUs, affinity mask,
    # cgroup CPU quota - 2) -- a box with another quota gives other walls, and this is what explains them
    out["host_threads"] = dict(eng.host_copy_stats()[0], cgroup_cpu_max=_read_first("/sys/fs/cgroup/cpu.max"),
                               ingestion_threads_setting=int(__import__("grakel_amd.batch", fromlist=["x"]).INGEST_THREADS))
    del Kp
    out["note"] = ("H2D of the packed CSR + the device step + the %d MB float64 K into a pinned, reused output block "
                   "(grakel_amd.engine.PinnedPool).  An integer-valued symmetric matrix crosses PCIe as the uint16 / int32 "
                   "blocks on and above the diagonal and is widened and mirrored by host threads; a normalised one crosses "
                   "the same way and is scaled by 1/sqrt(K_ii K_jj) in the widening threads (gram.hip: gram_copy_out)"
                   % (N * N * 8 // 1000000))
    return out, Ku


def python_objects(eng, wl, Ku, reps=7):
    """The estimator on Python dict graphs (SURVEY.md 8d's "from Python objects" wall) and the ingestion share of it."""
    import grakel_amd
    from grakel_amd import batch as _batch
    from grakel_amd.batch import wl_batch_from_input
    N, h = wl.N, wl.h
    X = wl.objects()
    est = grakel_amd.WeisfeilerLehman(n_iter=h)
    est.fit_transform(X[:50])
    Kw = est.fit_transform(X)                    # warm-up at full size (pins this size's second output block)
    del Kw
    Kobj = None
    calls, calls_n = [], []
    for _ in range(reps):
        Kobj = None                              # the caller's previous matrix goes back to the pinned pool before the next call
        t0 = time.perf_counter()                 # (holding it would make every other call pin a fresh 800 MB block: +20 ms)
        Kobj = est.fit_transform(X)
        calls.append((time.perf_counter() - t0) * 1e3)
    dt_obj = sum(calls) / reps * 1e-3
    estn = grakel_amd.WeisfeilerLehman(n_iter=h, normalize=True)
    Kn = estn.fit_transform(X)
    Kn = None
    Kn = estn.fit_transform(X)
    for _ in range(reps):
        Kn = None
        t0 = time.perf_counter()
        Kn = estn.fit_transform(X)
        calls_n.append((time.perf_counter() - t0) * 1e3)
    dt_objn = sum(calls_n) / reps * 1e-3
    del Kn
    t0 = time.perf_counter()
    wl_batch_from_input(X)                       # the ingestion part alone, same (now warm) objects
    dt_ingest = time.perf_counter() - t0
    saved_threads, _batch.INGEST_THREADS = _batch.INGEST_THREADS, 1
    try:
        t0 = time.perf_counter()
        wl_batch_from_input(X)                   # the same walk on the calling thread alone
        dt_ingest_1 = time.perf_counter() - t0
    finally:
        _batch.INGEST_THREADS = saved_threads
    # the other forms real data arrives in (csrc/ingest.c, round 5): `fetch_dataset` / `read_data` hand out sets of (u, v)
    # tuples with labels keyed by global vertex ids (datasets/base.py:273-279); examples often use adjacency matrices
    forms = {}
    try:
        off, Xs, Xa = 1, [], []
        for ed, lab in X:
            n = len(lab)
            Xs.append([{(u + off, v + off) for u, l in ed.items() for v in l}, {u + off: x for u, x in lab.items()}])
            A = np.zeros((n, n), dtype=np.int64)
            for u, l in ed.items():
                A[u, l] = 1
            Xa.append([A, lab])
            off += n
        for name, Z in (("set_of_tuples_global_ids", Xs), ("ndarray_adjacency", Xa)):
            wl_batch_from_input(Z)
            t0 = time.perf_counter()
            gbz, _ = wl_batch_from_input(Z)
            forms[name + "_ingestion_ms"] = (time.perf_counter() - t0) * 1e3
            forms[name + "_same_batch"] = bool(np.array_equal(gbz.col_idx, wl.batch.col_idx) and np.array_equal(gbz.row_ptr, wl.batch.row_ptr))
        t0 = time.perf_counter()
        Kobj = None
        Kobj = est.fit_transform(Xs)
        forms["set_of_tuples_fit_transform_ms"] = (time.perf_counter() - t0) * 1e3
        forms["set_of_tuples_same_matrix"] = bool(np.array_equal(Ku, Kobj))
        del Xs, Xa
    except Exception as e:
        forms["error"] = repr(e)
    return {"ms_per_call": dt_obj * 1e3, "value": N * N / dt_obj, "reps": reps, "other_input_forms": forms,
            "median_ms_per_call": float(np.median(calls)), "normalised_median_ms_per_call": float(np.median(calls_n)),
            "host_cpu_note": "the box's container has a CPU quota (cgroup cpu.max, 16 CPUs on the project's MI355X boxes): a call "
                             "that lands in a throttled period takes 20-40 ms longer -- see the per-call lists",
            "normalised_ms_per_call": dt_objn * 1e3, "calls_ms": [round(c, 3) for c in calls],
            "normalised_calls_ms": [round(c, 3) for c in calls_n],
            "of_which_host_ingestion_ms": dt_ingest * 1e3, "host_ingestion_one_thread_ms": dt_ingest_1 * 1e3,
            "host_ingestion_threads": "min(32, host CPUs, affinity mask, the container's cgroup CPU quota - 2) = %d here" % eng.host_copy_stats()[0]["thread_budget"],
            "same_matrix": bool(np.array_equal(Ku, Kobj)),
            "note": "grakel_amd.WeisfeilerLehman(n_iter=%d).fit_transform on %d dict graphs" % (h, N)}


def check_sp_matrix(K, name):
    """The host matrix of ShortestPath(with_labels) on a FULL published-like set against (1) the full-set fixture
    tests/golden/pub_<name>_sp_full.npz -- checksums, diagonal, row sums, a corner, 20 000 sampled entries; written by
    oracle/sp_fast.py, which tests/test_oracle.py pins to the real reference -- and (2) pub_<name>_sp_big.npz: the block of
    the set's LARGEST graphs as grakel 0.1.11 itself computed it.  Raises AssertionError on any difference; -> a record."""
    gdir = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(gdir, "pub_%s_sp_full.npz" % name))
    Ki = np.rint(K).astype(np.int64)
    assert np.array_equal(Ki.astype(np.float64), K), "ShortestPath matrix is not integer valued"
    assert Ki.shape[0] == int(z["n_graphs"][0]), "not the full set"
    assert int(Ki.sum()) == int(z["K_sum"][0]) and int(np.trace(Ki)) == int(z["K_trace"][0]) and int(Ki.max()) == int(z["K_max"][0])
    assert np.array_equal(np.diagonal(Ki), z["diag"]) and np.array_equal(Ki.sum(axis=1), z["row_sums"])
    assert np.array_equal(Ki[:64, :64], z["K_block"]) and np.array_equal(Ki[z["samp_i"], z["samp_j"]], z["samp_v"])
    assert np.array_equal(Ki, Ki.T)
    rec = {"K_sum": int(Ki.sum()), "K_trace": int(np.trace(Ki)), "entries_compared": int(len(z["samp_v"]) + 64 * 64 + 2 * Ki.shape[0]),
           "equals_full_set_fixture": True, "fixture": "tests/golden/pub_%s_sp_full.npz (oracle/sp_fast.py, pinned to grakel 0.1.11)" % name}
    big = os.path.join(gdir, "pub_%s_sp_big.npz" % name)
    if os.path.exists(big):
        zb = np.load(big)
        ix = zb["index"]
        assert np.array_equal(Ki[np.ix_(ix, ix)], zb["K"]), "differs from the real reference on the largest graphs"
        rec["equals_real_reference_on_largest_graphs"] = True
        rec["largest_graphs_block"] = "%d x %d, graphs of %s vertices, %.0f s of grakel 0.1.11" % (
            len(ix), len(ix), zb["sizes"].tolist(), float(zb["ref_seconds"][0]))
    return rec


def published_sp(eng, wl, steps=5):
    """ShortestPath(with_labels) fit_transform on a FULL published-like set from the packed CSR in HBM: ms, phases, and the
    matrix asserted against the reference-derived fixtures (check_sp_matrix)."""
    gb = wl.batch
    db = eng.upload(gb)
    sizes = np.diff(gb.graph_ptr).astype(np.float64)
    N = wl.N

    def step(to_host=False):
        pb = eng.sp_build(db, None, True)
        feat = eng.features(pb, 1)
        K = eng.gram(feat, 0, to_host=to_host)
        info = dict(n_pairs=pb.n_nodes, n_keys=pb.label_counts[0], dense=feat.n_cols, rare=feat.n_cols_low,
                    max_count=feat.max_count, operand=feat.operand, gram=eng.gram_stats(feat))
        return info, feat, pb, K

    t0 = time.perf_counter()
    info, feat, pb, _ = step()
    eng.synchronize()
    first = time.perf_counter() - t0
    feat.close(), pb.close()
    t0 = time.perf_counter()
    for _ in range(steps):
        info, feat, pb, _ = step()
        feat.close(), pb.close()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    info, feat, pb, K = step(to_host=True)
    check = None
    if wl.full_size and os.path.exists(os.path.join(ROOT, "tests", "golden", "pub_%s_sp_full.npz" % wl.name)):
        zf = np.load(os.path.join(ROOT, "tests", "golden", "pub_%s_sp_full.npz" % wl.name))
        assert info["n_keys"] == int(zf["n_features"][0]) and info["n_pairs"] == int(zf["n_pairs"][0]), (info, "feature / pair counts")
        check = check_sp_matrix(K, wl.name)
    s, tr = float(K.sum()), float(np.trace(K))
    del K
    feat.close(), pb.close()
    eng.profile(True)
    info2, feat, pb, _ = step()
    feat.close(), pb.close()
    ph = {k: round(eng.profile_get(k)[0], 4) for k in ("sp", "sp_fw", "features", "gram")}
    eng.profile(False)
    db.close()
    ops = float((sizes ** 3).sum())
    return {"workload": wl.describe.replace("WL-subtree h=5", "ShortestPath(with_labels)"), "graphs": N,
            "first_call_ms": first * 1e3, "ms_per_fit_transform": dt * 1e3, "graph_pairs_per_s": N * N / dt, "phases_ms": ph,
            "sum_n3_minplus": ops, "sum_n_times_m_bfs": float((sizes * np.diff(gb.row_ptr).sum() / max(gb.n_nodes, 1)).sum()),
            "fw_G_minplus_per_s": ops / max(ph["sp_fw"], 1e-6) / 1e6, "pairs": info["n_pairs"], "features": info["n_keys"],
            "dense_columns": info["dense"], "rare_columns": info["rare"], "max_count": info["max_count"],
            "operand": info["operand"], "gram_kernel_ms": info["gram"][1],
            "K_sum": s, "K_trace": tr, "checked_against_reference": check,
            "reference_publishes": wl.published}


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


def transform_bench(eng, wl):
    """`transform` of a few target graphs against the config-3 fit (never `value`): the look-up route (targets relabelled
    alone, signatures looked up in the fitted dictionaries on the device, csrc/wl_transform.hip -- the reference's own
    scheme, weisfeiler_lehman.py:435-498) against the joint route (fitted graphs + targets relabelled together), wall per
    call through the estimator on Python dict graphs and the device phases of one call."""
    import grakel_amd
    from grakel_amd.synthetic import er_dataset
    cfg = wl.cfg
    X = wl.objects()
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


def device_step_summary(eng, wl, steps=3):
    """One more workload's device step for the driver's record (`extra`): ms per step, phases, the operand, the Gram
    kernel against both roofs, checks against the golden / the oracle's label counts."""
    db = eng.upload(wl.batch)
    h = wl.h
    for _ in range(2):
        feat, _k = eng.wl_fit_transform(db, h, to_host=False)
        feat.close()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        feat, _k = eng.wl_fit_transform(db, h, to_host=False)
        if _ + 1 < steps:
            feat.close()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flops, gms = eng.gram_stats(feat)
    s, tr, asym = eng.gram_checksum(feat)
    selfk_sum = float(eng.selfk(feat).sum())
    info = dict(n_cols=feat.n_cols, n_cols_low=feat.n_cols_low, operand=feat.operand, max_count=feat.max_count,
                label_counts=db.label_counts, stream_route=bool(getattr(db, "stream_route", False)))
    feat.close()
    eng.profile(True)
    feat, _k = eng.wl_fit_transform(db, h, to_host=False)
    feat.close()
    ph = {k: round(eng.profile_get(k)[0], 4) for k in ("relabel", "features", "gram")}
    eng.profile(False)
    db.close()
    N = wl.N
    cfg = wl.cfg
    out = {"workload": wl.describe, "graphs": N, "nodes": int(wl.batch.n_nodes), "edges": int(wl.batch.n_edges),
           "ms_per_step": dt * 1e3, "graph_pairs_per_s": N * N / dt, "phases_ms": ph,
           "relabel_route": "stream (no host round trips, csrc/wl_stream.hip)" if info["stream_route"] else "host-driven (csrc/wl.hip)",
           "label_counts": info["label_counts"], "operand": info["operand"], "dense_columns": info["n_cols"],
           "rare_columns": info["n_cols_low"], "max_count": info["max_count"],
           "gram": gram_roofs(N, 1, info["n_cols"], info["n_cols_low"], info["operand"], gms, flops),
           "checks": {"K_sum": s, "K_trace": tr, "max_abs_K_minus_KT": asym, "trace_equals_sum_of_selfk": bool(tr == selfk_sum),
                      "label_counts_match": (info["label_counts"] == cfg["label_counts"]) if cfg.get("label_counts") else None,
                      "matches_reference_checksums": bool((s, tr) == cfg["golden"]) if cfg.get("golden") else None}}
    if wl.published:
        out["reference_publishes"] = wl.published
    if wl.golden is not None and "ref_seconds" in wl.golden:
        out["reference_here_s"] = float(wl.golden["ref_seconds"][0])
    return out


def gram_roofs(N, world, d_dense, d_rare, operand, gram_ms, flops, rows=None, sym_plan=False):
    """The Gram kernel of one launch against both roofs: HBM (8 bytes per entry written once + the packed operand read
    once) and MFMA (executed flops against the dense peak of the operand type)."""
    rows = N / world if rows is None else rows
    fp4 = str(operand).startswith("fp4")
    f64 = str(operand).startswith("f64")
    # operand bytes: 128-byte K-steps; fp4 two columns per byte, int8 one (the operand is read once algorithmically)
    per_row = ((d_dense + 255) // 256 * 128) if fp4 else ((d_dense + 127) // 128 * 128 if not f64 else d_dense * 8)
    operand_bytes = (N + 511) // 256 * 256 * per_row
    gram_bytes = 8.0 * rows * N * (0.5 * (1.0 + 1.0 / world) if sym_plan else 1.0) + operand_bytes
    peak = F64_DENSE_PEAK_TOPS if f64 else (FP4_DENSE_PEAK_TOPS if fp4 else I8_DENSE_PEAK_TOPS)
    gbs = gram_bytes / (gram_ms * 1e-3) / 1e9
    tfl = flops / (gram_ms * 1e-3) / 1e12
    hbm_floor_ms = gram_bytes / (HBM_PEAK_GBS * 1e9) * 1e3
    mfma_floor_ms = flops / (peak * 1e12) * 1e3
    return {"kernel_ms": gram_ms, "algorithmic_bytes": gram_bytes, "GB_per_s": gbs, "frac_of_hbm_roof": gbs / HBM_PEAK_GBS,
            "executed_flops": flops, "TFLOP_per_s": tfl, "mfma_peak_TFLOP_per_s": peak, "frac_of_mfma_roof": tfl / peak,
            "hbm_floor_ms": hbm_floor_ms, "mfma_floor_ms": mfma_floor_ms,
            "bound": "mfma" if mfma_floor_ms > hbm_floor_ms else "hbm",
            "D_eff": d_dense + (d_rare or 0), "dense_columns": d_dense, "rare_columns": d_rare}


# ---------------------------------------------------------------------------------------------------------------
def self_launch(a):
    """`python bench.py --gpus N` outside torch.distributed.run: launch N ranks of this script, one per GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(ER_WORKLOADS) + list(PUBLISHED), default="config3")
    ap.add_argument("--graphs", type=int, default=0, help="(debug) smaller workload")
    ap.add_argument("--cpu-sample", type=int, default=0, help="graphs of the CPU baseline's one-core leg (default: per workload, 10-30 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip host_to_host, from_python_objects and `extra`")
    ap.add_argument("--plan", choices=["plain", "symmetric"], default="plain",
                    help="N > 1: Gram sharding plan (grakel_amd.dist.gram_plan; plain row blocks is the default)")
    ap.add_argument("--exchange", choices=["csr", "phi"], default="csr",
                    help="N > 1: what the ranks exchange besides the packed CSR shards -- csr (default): nothing, the operand is "
                         "assembled on every rank; phi: every rank assembles its own graphs' operand rows and the row shards are "
                         "all-gathered (the exchange north_star names; an A/B switch, grakel_amd.dist.ShardedWL)")
    ap.add_argument("--separate-calls", action="store_true",
                    help="step = gk_wl_relabel + gk_features_build + gk_gram as three library calls instead of gk_wl_fit_transform")
    ap.add_argument("--block-rows", type=int, default=-1,
                    help="rows per Gram sub-block (default: the workload's, and only when a rank's row block exceeds 64 GB; tests force it)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="context option (gk_set_option, include/gk_hip.h), e.g. --opt wl.debug=1; A/B runs only")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))

    wl = Workload(a.workload, a.graphs)
    # the CPU baseline runs FIRST: its n_jobs leg forks worker processes, which must not happen in a process
    # that already holds an initialised HIP runtime (runtime threads, locks)
    cpu_base = None
    if world == 1 and not a.no_cpu_baseline:
        default_sample = {"config3": 10000, "config5": 10000, "config6": 10000, "nci1": 4110, "dd": 1178, "reddit": 2000,
                          "collab": 2500}[a.workload]
        cpu_base = cpu_baseline(wl, min(a.cpu_sample or default_sample, wl.N))

    import torch
    from grakel_amd import _lib
    from grakel_amd.engine import get_engine

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

    cfg = wl.cfg
    N, h = wl.N, wl.h
    full = wl.batch
    eng = get_engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for o in a.opt:
        eng.set_option(o.split("=")[0], int(o.split("=")[1]))
    block_rows = int(cfg.get("block_rows") or 0)
    if block_rows and N * N * 8 // world <= 64 << 30:          # the rank's row block fits comfortably: one block
        block_rows = 0
    if a.block_rows >= 0:
        block_rows = a.block_rows
    info = {}
    keep = {}
    block_sums = {}

    if world == 1 and not block_rows:
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
                        label_counts=db.label_counts, nnz=feat.nnz, stream_route=bool(getattr(db, "stream_route", False)))
            collect()
            pending.append(feat)
    elif world == 1:
        # the matrix does not fit the GPU (config 6): the rows are multiplied in sub-blocks that reuse one device buffer
        db = eng.upload(full)

        def collect(last=False):
            pass

        def step(check=False):
            eng.wl_relabel(db, h)
            feat = eng.features(db, h + 1)
            gms, gfl = 0.0, 0.0
            for lo in range(0, N, block_rows):
                sub = (lo, min(lo + block_rows, N))
                eng.gram(feat, 0, rows=sub, to_host=False)
                if check:
                    block_sums[sub] = eng.gram_checksum(feat)[0]
                fl, ms = eng.gram_stats(feat)
                gms, gfl = gms + ms, gfl + fl
            info.update(n_cols=feat.n_cols, n_cols_low=feat.n_cols_low, dtype=feat.dtype, operand=feat.operand,
                        label_counts=db.label_counts, nnz=feat.nnz, gram=(gfl, gms),
                        stream_route=bool(getattr(db, "stream_route", False)))
            if check:
                keep["selfk_sum"] = float(eng.selfk(feat).sum())
            feat.close()
    else:
        from grakel_amd.dist import ShardedWL, shard_bounds
        b = shard_bounds(N, world)
        local = full.slice_graphs(b[rank], b[rank + 1])
        sw = ShardedWL(eng, n_iter=h, symmetric=(a.plan == "symmetric"), exchange=a.exchange)       # default: plain row blocks (dist.gram_plan)

        def collect(last=False):
            pass

        def step(to_host=False, check=False):
            on_block = None
            if check:
                def on_block(feat, sub):
                    block_sums[sub] = eng.gram_checksum(feat)[0]
            K, i = sw.step(local, to_host=to_host, block_rows=block_rows, on_block=on_block)
            i.pop("K_dev", None)                 # the rank's row block (a torch tensor) is released with the step
            info.update(i)
            return K

    def sync():
        eng.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    collect()
    gram_ms = []
    rank_phases = []
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step()
        if world > 1 or block_rows or i > 0:
            gram_ms.append(info["gram"][1])      # world == 1: the step before (see collect)
        if world > 1:
            pe = sw.phase_events                 # read after the step's gram_stats has waited for its events anyway
            pe[3].synchronize()
            rank_phases.append([pe[0].elapsed_time(pe[1]), pe[1].elapsed_time(pe[2]), pe[2].elapsed_time(pe[3])])
    sync()
    dt = time.perf_counter() - t0
    checks = None
    if world == 1 and not block_rows:
        collect(last=True)
        gram_ms.append(info["gram"][1])
        # the matrix the LAST timed step left in HBM, checked in place (no 8 N^2-byte copy)
        s, tr, asym = eng.gram_checksum(keep["feat"])
        selfk_sum = float(eng.selfk(keep["feat"]).sum())
        checks = {"K_sum": s, "K_trace": tr, "max_abs_K_minus_KT": asym, "trace_equals_sum_of_selfk": bool(tr == selfk_sum),
                  "label_counts_match_the_oracle": (info["label_counts"] == cfg["label_counts"]) if cfg["label_counts"] else None,
                  "checksums_match_the_reference": bool((s, tr) == cfg["golden"]) if cfg["golden"] else None,
                  # what pins this workload's matrix: the real reference's checksums and sampled entries (config 3, the
                  # published-like sets), or -- where the reference cannot run (config 5: six dense 50k x 50k float64
                  # matrices) -- the CPU oracle, block by block, in tests/test_gpu_parity.py (the oracle itself is pinned to
                  # the reference on every golden set); here only the invariants and the oracle's label counts are asserted
                  "checked_against": ("the real reference: checksums of the matrix the timed step left in HBM; entry-wise "
                                      "(gram_max_abs_err) on the host copy below" if cfg["golden"] else
                                      ("oracle blockwise (tests/test_gpu_parity.py: config 5 test) + invariants + oracle label counts"
                                       if cfg["label_counts"] else "invariants only (custom size)")),
                  "gram_max_abs_err": None}
        assert asym == 0.0 and tr == selfk_sum, "timed Gram matrix is not symmetric / has a wrong diagonal: %r" % (checks,)
        if cfg["golden"]:
            assert (s, tr) == cfg["golden"], "timed Gram matrix differs from the reference checksums: %r" % (checks,)
        if cfg["label_counts"]:
            assert info["label_counts"] == cfg["label_counts"], "WL label counts differ from the oracle's"
        keep.pop("feat").close()

    # ---- N > 1: max over ranks, per-rank phases, the end-to-end form (every rank's rows on its host) --------------
    per_rank = None
    e2e_ms = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_local, dt = dt, float(t.item())
        ph = np.mean(np.asarray(rank_phases), axis=0) if rank_phases else np.zeros(3)
        mine = {"rank": rank, "device": local_rank, "rows": list(info.get("rows", ())),
                "exchange_and_rebuild_ms": float(ph[0]), "replicated_relabel_features_ms": float(ph[1]),
                "gram_rows_ms": float(ph[2]), "gram_kernel_ms": float(np.mean(gram_ms)), "step_ms": dt_local / a.steps * 1e3}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        if not block_rows and not a.no_extras:
            reps = max(2, min(a.steps, 5))
            step(to_host=True)
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                step(to_host=True)
            sync()
            t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e2e_ms = float(t.item()) * 1e3
    if block_rows:
        # one more, untimed, pass with a checksum per sub-block: the sums add up over ranks; K's trace is not in a row
        # block's sum, so the invariant asserted is  sum over blocks == sum over the same blocks of a second pass
        # (determinism) and, on one GPU, that the first 512 x 512 corner equals the matrix of the first 512 graphs alone
        step(check=True)
        first = dict(block_sums)
        block_sums.clear()
        step(check=True)
        assert first == block_sums, "sub-block checksums differ between two passes"
        tot = float(sum(block_sums.values()))
        if dist is not None:
            t = torch.tensor([tot], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            tot = float(t.item())
        checks = {"K_sum_over_sub_blocks": tot, "sub_blocks_this_rank": len(block_sums),
                  "two_passes_agree": True,
                  "checked_against": "tests/test_gpu_parity.py::test_config6_subsampled_against_the_oracle pins the same code "
                                     "path on a prefix of the generator; here: determinism of the per-block sums"}

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
        f64_only = bool(info["dtype"])
        d_dense = info.get("n_cols") or 0
        sym = world > 1 and a.plan == "symmetric"
        roofs = gram_roofs(N, world, d_dense, info.get("n_cols_low"), info.get("operand", "fp4"), gram_avg_ms, flops, sym_plan=sym)
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
        traffic = gram_pmc_traffic_bytes(a.workload if wl.full_size else "", world)
        bound = roofs["bound"]
        out = {
            "metric": "graph-pairs/sec for NxN WL-subtree(h=%d) fit_transform" % h,
            "value": N * N / (dt / a.steps),
            "value_is": "device step: packed CSR resident in HBM -> float64 K resident in HBM (the task contract: inputs resident in "
                        "HBM, the PCIe-inclusive rate is never `value`); what a caller observes is `host_to_host` and "
                        "`from_python_objects` below, each timed over its own >= 20 bracketed steps",
            "unit": "graph-pairs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": info.get("operand") or ("f64" if f64_only else "fp4+i8"),
            "data": "synthetic",
            "config": {"workload": wl.describe + ", full NxN float64 Gram " + (
                           "in %d-row sub-blocks that reuse one device buffer" % block_rows if block_rows else "left in HBM"),
                       "graphs": N, "nodes": V_, "edges": E_,
                       "relabel_route": "stream (csrc/wl_stream.hip)" if info.get("stream_route") else (
                           "host-driven (csrc/wl.hip)" if world == 1 else None),
                       "parallelism": "graphs+Gram rows sharded over %d GPU(s)%s" % (
                           world, "" if world == 1 else (
                               "; every rank multiplies 1/%d of the upper triangle and ships the mirrored blocks point to "
                               "point (grakel_amd.dist.symmetric_plan)" % world if sym else
                               "; one all-gather of the CSR shards, relabel + features replicated, every rank multiplies and "
                               "stores its own row block (plain row blocks: no collective on the Gram path)")),
                       "gram_plan_model": (lambda m: {k: (v if k == "choice" else {kk: round(vv, 6) if kk == "seconds" else vv
                                                                                 for kk, vv in v.items()}) for k, v in m.items()})(
                           __import__("grakel_amd.dist", fromlist=["gram_plan"]).gram_plan(N, world)) if world > 1 else None,
                       "label_counts": info.get("label_counts"), "gram_columns_dense": d_dense,
                       "gram_columns_rare": info.get("n_cols_low")},
            "checks": checks,
            "roofline": {
                "kernel": "gram_ws_kernel (persistent, warp-specialised 128x128 tiles, strip-walk tile order; MX fp4 operands "
                          "for counts <= 4, int8 for 5..127; float64 store of K)",
                "bound": bound,
                "achieved": roofs["GB_per_s"] if bound == "hbm" else roofs["TFLOP_per_s"],
                "peak": HBM_PEAK_GBS if bound == "hbm" else roofs["mfma_peak_TFLOP_per_s"],
                "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                "frac": roofs["frac_of_hbm_roof"] if bound == "hbm" else roofs["frac_of_mfma_roof"],
                "traffic": traffic,
                "traffic_source": "profiles/%s (2 x FETCH_SIZE + WRITE_SIZE, KiB) -- a committed PMC "
                                  "pass of this script, NOT measured in this run" % os.path.basename(PMC_FILE) if traffic else None,
                "algorithmic_bytes_per_launch": roofs["algorithmic_bytes"], "avg_launch_ms": gram_avg_ms,
                "both_roofs": roofs,
                "note": "bound = the larger of the two floors (algorithmic bytes / 8 TB/s, executed flops / dense MFMA peak of the "
                        "operand type).  hbm: achieved = (8 bytes x rows x N of float64 K written once + the packed dense operand "
                        "read once) / avg HIP-event duration of the Gram kernel.  1 GPU: only tiles on/above the diagonal are "
                        "multiplied, both halves are stored; label columns present in fewer graphs than the job's threshold "
                        "(DESIGN.md 2) never enter the dense operand, their exact pair updates are inside phases_ms.gram."},
            "phases_ms": phases,
            "phases_hbm": phases_hbm,
        }
        if world > 1:
            out["rccl"] = {"backend": backend, "ranks": dist.get_world_size(), "exchange": a.exchange,
                           "operand_row_bytes_sent_per_rank": int(getattr(sw, "phi_bytes", 0)) if a.exchange == "phi" else 0,
                           "launched_by": "torch.distributed.run (self-launched by `bench.py --gpus N` when WORLD_SIZE is unset)"}
            out["per_rank"] = per_rank
            out["device_step_ms_max_over_ranks"] = ms_per_step
            out["end_to_end_ms"] = e2e_ms
            out["end_to_end_is"] = ("the same step with every rank's row block copied to a host array of its own process "
                                    "(max over ranks); null for the sub-blocked workload")
        out["cpu_baseline"] = cpu_base            # measured before the GPU part (see the top of main)
        if world == 1 and not a.no_extras and not block_rows:
            h2h_steps = a.steps if N <= 20000 else max(2, min(a.steps, 3))
            out["host_to_host"], Ku = host_to_host(eng, wl, h2h_steps, a.warmup)
            if wl.golden is not None:
                err, n = check_host_matrix(Ku, wl.golden)
                out["checks"]["gram_max_abs_err"] = err
                out["checks"]["gram_max_abs_err_is"] = ("max |K - K_reference| over %d entries of the host matrix: the reference's "
                                                        "20 000 sampled entries, its diagonal, its 64 x 64 corner and every row sum "
                                                        "(tests/golden, produced by grakel 0.1.11)" % n)
                assert err == 0.0, "host Gram matrix differs from the reference's entries: %r" % err
            if N <= 20000:
                try:
                    out["from_python_objects"] = python_objects(eng, wl, Ku)
                except Exception as e:                     # never lose the headline line to an extra
                    out["from_python_objects"] = {"error": repr(e)}
            del Ku
            out["extra"] = {}
            if a.workload in ("dd", "reddit", "collab") and wl.full_size:
                try:                                       # asserted against the reference-derived full-set fixtures
                    out["extra"]["shortest_path"] = published_sp(eng, wl)
                except AssertionError as e:
                    raise
                except Exception as e:
                    out["extra"]["shortest_path"] = {"error": repr(e)}
            if a.workload == "config3" and wl.full_size:
                for name, fn in (("config4_sp", lambda: config4_sp(eng)),
                                 ("transform", lambda: transform_bench(eng, wl)),
                                 ("config5", lambda: device_step_summary(eng, Workload("config5"))),
                                 ("nci1_like_wl", lambda: device_step_summary(eng, Workload("nci1")))):
                    try:
                        out["extra"][name] = fn()
                    except Exception as e:
                        out["extra"][name] = {"error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

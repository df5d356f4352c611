#!/usr/bin/env python3
"""ShortestPath kernel (BASELINE config 4 stand-in: 4110 NCI1-like graphs) on one MI355X.

Prints one JSON line: wall of fit_transform from packed CSR resident in HBM (K left in HBM),
per-phase HIP-event times, the Floyd-Warshall kernel's min-plus rate and the f64 Gram rate.
Reference (this container, real grakel 0.1.11): 15.2 s adjacency/Floyd-Warshall route,
21.8 s dict/Dijkstra route, i.e. 1.1e6 / 7.8e5 graph-pairs/s.
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grakel_amd.batch import sp_batch_from_input
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import nci1_like

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4110
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
G = nci1_like(N, 0, as_adj=True)
t0 = time.perf_counter()
gb, _ = sp_batch_from_input(G, True)
t_ingest = time.perf_counter() - t0
eng = get_engine()
for _o in os.environ.get("GK_TOOL_OPTS", "").split():          # e.g. GK_TOOL_OPTS="sp.hist_no_batch=1"
    eng.set_option(_o.split("=")[0], int(_o.split("=")[1]))
db = eng.upload(gb)
sizes = np.diff(gb.graph_ptr).astype(np.float64)


def step(profile=False):
    pb = eng.sp_build(db, None, True)
    feat = eng.features(pb, 1)
    eng.gram(feat, 0, to_host=False)
    info = dict(n_pairs=pb.n_nodes, n_keys=pb.label_counts[0], dense=feat.n_cols, rare=feat.n_cols_low,
                dtype=("i8", "f64")[feat.dtype], max_count=feat.max_count, gram=eng.gram_stats(feat))
    feat.close(); pb.close()
    return info


for _ in range(2):
    info = step()
eng.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    info = step()
eng.synchronize()
dt = (time.perf_counter() - t0) / steps
eng.profile(True)
step()
phases = {k: round(eng.profile_get(k)[0], 4) for k in ("sp", "features", "gram")}
eng.profile(False)
flops, gemm_ms = info["gram"]
print(json.dumps({
    "workload": "NCI1-like stand-in, %d graphs, ShortestPath(with_labels), packed CSR resident in HBM" % N,
    "ms_per_fit_transform": dt * 1e3, "graph_pairs_per_s": N * N / dt, "phases_ms": phases,
    "host_ingest_s": t_ingest, "fw_minplus_ops": float((sizes ** 3).sum()),
    "fw_Gops_per_s_incl_emit_and_dictionary": float((sizes ** 3).sum()) / (phases["sp"] * 1e-3) / 1e9,
    "pairs": info["n_pairs"], "features": info["n_keys"], "dense_columns": info["dense"],
    "rare_columns": info["rare"], "gram_dtype": info["dtype"], "max_count": info["max_count"],
    "gram_kernel_ms": gemm_ms, "gram_TFLOPs": flops / (gemm_ms * 1e-3) / 1e12,
    "reference_cpu_s": {"floyd_warshall_route": 15.2, "dijkstra_route": 21.8},
}))

#!/usr/bin/env python3
"""One stand-in for a published TU dataset (grakel_amd/synthetic.py PUBLISHED_LIKE) through the device path:

    python tools/published_like.py SET wl|sp|wl_e2e [steps]

wl      WL-subtree h=5 device step (packed CSR in HBM -> float64 K in HBM): ms, phases, relabel route, operand, the Gram
        kernel against both roofs, the matrix against the reference's checksums (tests/golden/pub_SET.npz)
sp      ShortestPath(with_labels) fit_transform on the FULL set from the packed CSR in HBM: ms, phases, all-pairs rate
wl_e2e  packed CSR on the host -> float64 K on the host, unnormalised and normalised, entry-wise against the golden

Prints one JSON line.  (Each mode is its own process so that tools/profile_published.sh can put a timeout and a
rocprofv3 kernel trace around it.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def sp_summary(eng, wl, steps):
    gb = wl.batch
    db = eng.upload(gb)
    sizes = np.diff(gb.graph_ptr).astype(np.float64)
    N = wl.N

    def step():
        pb = eng.sp_build(db, None, True)
        feat = eng.features(pb, 1)
        eng.gram(feat, 0, to_host=False)
        info = dict(n_pairs=pb.n_nodes, n_keys=pb.label_counts[0], dense=feat.n_cols, rare=feat.n_cols_low,
                    max_count=feat.max_count, operand=feat.operand, gram=eng.gram_stats(feat))
        return info, feat, pb

    t0 = time.perf_counter()
    info, feat, pb = step()
    eng.synchronize()
    first = time.perf_counter() - t0
    feat.close(), pb.close()
    t0 = time.perf_counter()
    for _ in range(steps):
        info, feat, pb = step()
        if _ + 1 < steps:
            feat.close(), pb.close()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    s, tr, asym = eng.gram_checksum(feat)
    feat.close(), pb.close()
    eng.profile(True)
    info2, feat, pb = step()
    feat.close(), pb.close()
    ph = {k: round(eng.profile_get(k)[0], 4) for k in ("sp", "sp_fw", "features", "gram")}
    eng.profile(False)
    ops = float((sizes ** 3).sum())
    return {"workload": wl.describe.replace("WL-subtree h=5", "ShortestPath(with_labels)"), "graphs": N,
            "first_call_ms": first * 1e3, "ms_per_fit_transform": dt * 1e3, "graph_pairs_per_s": N * N / dt, "phases_ms": ph,
            "sum_n3_minplus": ops, "sum_n_times_m_bfs": float((sizes * np.diff(gb.row_ptr).sum() / max(gb.n_nodes, 1)).sum()),
            "fw_G_minplus_per_s": ops / max(ph["sp_fw"], 1e-6) / 1e6, "pairs": info["n_pairs"], "features": info["n_keys"],
            "dense_columns": info["dense"], "rare_columns": info["rare"], "max_count": info["max_count"],
            "operand": info["operand"], "gram_kernel_ms": info["gram"][1],
            "K_sum": s, "K_trace": tr, "max_abs_K_minus_KT": asym,
            "reference_publishes": wl.published}


def main():
    name, mode = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    import torch  # noqa: F401
    from grakel_amd.engine import get_engine
    wl = bench.Workload(name)
    eng = get_engine()
    for o in os.environ.get("GK_TOOL_OPTS", "").split():
        eng.set_option(o.split("=")[0], int(o.split("=")[1]))
    if mode == "wl":
        out = bench.device_step_summary(eng, wl, steps)
    elif mode == "sp":
        out = sp_summary(eng, wl, steps)
    else:
        out, Ku = bench.host_to_host(eng, wl, steps, 2)
        if wl.golden is not None:
            err, n = bench.check_host_matrix(Ku, wl.golden)
            out["gram_max_abs_err"], out["entries_checked"] = err, n
    print(json.dumps(out))


if __name__ == "__main__":
    main()

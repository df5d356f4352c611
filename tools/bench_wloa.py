"""WL-OA timing on the GPU: python tools/bench_wloa.py [config2|config3].
Inputs resident in HBM, float64 K left in HBM; HIP-event time of relabel + features + Gram."""
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from grakel_amd import GraphBatch                      # noqa: E402
from grakel_amd.engine import get_engine              # noqa: E402
from grakel_amd.synthetic import er_dataset_csr       # noqa: E402

CONFIGS = {"config2": (1000, 50, 0.1, 5, 0, 3), "config3": (10000, 100, 0.05, 5, 0, 5)}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "config2"
    N, n, p, L, seed, h = CONFIGS[tag]
    eng = get_engine()
    db = eng.upload(GraphBatch(*er_dataset_csr(N, n, p, L, seed), L))
    times = []
    for it in range(8):
        eng.timer_start()
        eng.wl_relabel(db, h)
        feat = eng.features(db, h + 1, kind=1)
        eng.gram(feat, 0, to_host=False)
        times.append(eng.timer_stop_ms())
        info = (feat.n_cols, feat.n_cols_low, feat.max_count)
        if it == 7:
            K = eng.gram(feat, 0)
        feat.close()
    ms = float(np.median(times[3:]))
    print(json.dumps({"workload": "WL-OA " + tag, "n_graphs": N, "h": h, "ms": round(ms, 3),
                      "graph_pairs_per_s": N * N / ms * 1e3, "unary_dense_cols": info[0],
                      "rare_cols": info[1], "max_count": info[2], "K_sum": int(K.sum()),
                      "K_trace": int(np.trace(K))}))


if __name__ == "__main__":
    main()

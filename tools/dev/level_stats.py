"""Per WL level of config 3 (or `N n p`): labels, shared classes (>= 2 nodes), nodes in shared classes, (label, graph) entries of
shared classes, classes by df."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grakel_amd import GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
a = sys.argv[1:]
N, n, p = (int(a[0]), int(a[1]), float(a[2])) if len(a) >= 3 else (10000, 100, 0.05)
gp, rp, ci, lab = er_dataset_csr(N, n, p, 5, 0)
eng = get_engine()
db = eng.upload(GraphBatch(gp, rp, ci, lab, 5))
eng.wl_relabel(db, 5)
graph_of = np.repeat(np.arange(N), np.diff(gp))
for l in range(6):
    L = eng.wl_labels(db, l) if hasattr(eng, "wl_labels") else eng.wl_get_labels(db, l)
    u, inv, cnt = np.unique(L, return_inverse=True, return_counts=True)
    shared = cnt[inv] >= 2
    pairs = np.unique(np.stack([inv[shared], graph_of[shared]]), axis=1)
    df = np.bincount(pairs[0], minlength=len(u))
    print("level %d: labels %d shared classes %d nodes in them %d entries %d | classes with df>=2: %d, df>=16: %d, max df %d" % (
        l, len(u), int((cnt >= 2).sum()), int(shared.sum()), pairs.shape[1], int((df >= 2).sum()), int((df >= 16).sum()), int(df.max())))

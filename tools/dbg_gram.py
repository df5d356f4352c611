"""Gram kernel vs the host product of its own operand: python tools/dbg_gram.py [N n]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GK_LOW_DF"] = "2"
from grakel_amd import GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = get_engine()
gp, rp, ci, lab = er_dataset_csr(N, n, 0.15, 3, 0)
for env in ({}, {"GK_GRAM_NO_FP4": "1"}):
    os.environ.pop("GK_GRAM_NO_FP4", None)
    os.environ.update(env)
    db = eng.upload(GraphBatch(gp, rp, ci, lab, 3))
    eng.wl_relabel(db, 2)
    feat = eng.features(db, 3)
    phi = eng.debug_phi(feat)
    K = eng.gram(feat, 0)
    R = phi @ phi.T
    np.fill_diagonal(R, eng.selfk(feat))
    bad = np.argwhere(K != R)
    print(env, "cols", feat.n_cols, "low", feat.n_cols_low, "max", feat.max_count, "mismatches", len(bad), "of", N * N)
    if len(bad):
        print("  first:", [(int(i), int(j), K[i, j], R[i, j]) for i, j in bad[:8]])
        print("  rows with errors:", np.unique(bad[:, 0])[:40], "cols:", np.unique(bad[:, 1])[:40])
        d = (K - R)
        print("  diff stats: min %g max %g" % (d.min(), d.max()))
    feat.close(); db.close()

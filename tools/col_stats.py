"""Distribution of the largest count per dense Gram column (decides the operand classes of gram.hip):
python tools/col_stats.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grakel_amd import GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
eng = get_engine()
for name, (N, n, p, h) in {"config3": (10000, 100, 0.05, 5), "config5_10k": (10000, 30, 0.1, 5), "config2": (1000, 50, 0.1, 3)}.items():
    gp, rp, ci, lab = er_dataset_csr(N, n, p, 5, 0)
    db = eng.upload(GraphBatch(gp, rp, ci, lab, 5))
    eng.wl_relabel(db, h)
    feat = eng.features(db, h + 1)
    phi = eng.debug_phi(feat)
    mx = phi.max(axis=0)
    df = (phi > 0).sum(axis=0)
    hist = np.bincount(np.minimum(mx, 20).astype(int), minlength=21)
    print(name, "dense cols", feat.n_cols, "rare", feat.n_cols_low, "max count hist (0..19, >=20):", hist.tolist())
    print("   cols with max<=1: %d, <=3: %d, <=4: %d, <=7: %d, <=15: %d" % tuple((mx <= t).sum() for t in (1, 3, 4, 7, 15)))
    print("   density of Phi_s: %.4f, mean df %.1f, median df %.0f" % ((phi > 0).mean(), df.mean(), np.median(df)))
    feat.close(); db.close()

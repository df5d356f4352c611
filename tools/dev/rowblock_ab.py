#!/usr/bin/env python3
"""Row-block Gram A/B (round 6): the tile-kernel forms on a row block of a large job -- the mode the multi-GPU path and
config 6 run in (no mirrored tile: every stored tile costs a full operand pass).

    python tools/dev/rowblock_ab.py [N=50000] [rows=6250] [opt=val[,opt=val] ...]

Builds the WL h=5 features of N Erdos-Renyi graphs of 30 vertices once, then times gk_gram_rows(0, rows) (matrix left in HBM)
under each option set; prints kernel ms (HIP events), algorithmic TB/s = (8 rows N + operand bytes) / ms, checksum."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from grakel_amd import GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
args = sys.argv[1:]
N = int(args[0]) if args and args[0].isdigit() else 50000
rows = int(args[1]) if len(args) > 1 and args[1].isdigit() else N // 8
nv = int(os.environ.get("GK_AB_VERTICES", "30"))        # 30: the config 5 / 6 family; 100 (p = 0.05): config 3's
gp, rp, ci, lab = er_dataset_csr(N, nv, 0.1 if nv == 30 else 0.05, 5, 0)
eng = get_engine()
db = eng.upload(GraphBatch(gp, rp, ci, lab, 5))
eng.wl_relabel(db, 5)
feat = eng.features(db, 6)
print("N", N, "vertices", nv, "rows", rows, "dense cols", feat.n_cols, "rare", feat.n_cols_low, "operand", feat.operand)
variants = [dict()] + [dict((x.split("=")[0], int(x.split("=")[1])) for x in v.split(",")) for v in args if "=" in v]
ref = None
for opts in variants:
    with eng.options(**opts):
        ms = []
        for it in range(5):
            eng.gram(feat, 0, rows=(0, rows), to_host=False)
            ms.append(eng.gram_stats(feat)[1])
        chk = eng.gram_checksum(feat)
        if ref is None:
            ref = chk
        m = min(ms)
        print(opts, "kernel ms min %.3f med %.3f" % (m, sorted(ms)[2]), "| %.2f TB/s of K bytes" % (8.0 * rows * N / m / 1e9),
              "chk", chk[0], "OK" if chk == ref else "MISMATCH")

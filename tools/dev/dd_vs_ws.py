"""gram_dd (direct-store form) against gram_ws (parked tile + store waves) per job: K-steps by operand type and both timings."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import torch  # noqa: F401
from grakel_amd import GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
eng = get_engine()
jobs = [("nci1-like", bench.Workload("nci1").batch)]
for N, n, p in ((1000, 50, 0.1), (2000, 100, 0.05), (4000, 100, 0.05), (6000, 100, 0.05), (4000, 30, 0.1)):
    jobs.append(("ER N=%d n=%d" % (N, n), GraphBatch(*er_dataset_csr(N, n, p, 5, 0), 5)))
for name, gb in jobs:
    db = eng.upload(gb)
    eng.wl_relabel(db, 5)
    feat = eng.features(db, 6)
    fp4, k1, k8, nw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
    eng.lib.gk_features_operand(feat.handle, ctypes.byref(fp4), ctypes.byref(k1), ctypes.byref(k8), ctypes.byref(nw))
    res = {}
    for form, opt in (("dd", 1), ("ws", 2)):
        with eng.options(**{"gram.dd": opt}):
            ms = []
            for _ in range(6):
                eng.gram(feat, 0, to_host=False)
                ms.append(eng.gram_stats(feat)[1])
            res[form] = min(ms[1:])
    T = (gb.n_graphs + 127) // 128
    print("%-18s tiles %5d fp4 %d k1 %3d k8 %3d dense %5d  dd %.4f ms  ws %.4f ms  -> %s" % (
        name, T * (T + 1) // 2, fp4.value, k1.value, k8.value, feat.n_cols, res["dd"], res["ws"], "dd" if res["dd"] < res["ws"] else "ws"))
    feat.close(); db.close()

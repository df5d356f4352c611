"""Config 3 Gram phase by rare-pair route: python tools/dev/fold_check.py  (fold 0 = default choice, 1 = fold in, 2 = atomics afterwards)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grakel_amd import GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
eng = get_engine()
db = eng.upload(GraphBatch(*er_dataset_csr(10000, 100, 0.05, 5, 0), 5))
eng.wl_relabel(db, 5)
for norm in (0, 2):
    for fold, binglobal in ((2, 0), (1, 0), (0, 0)):
        with eng.options(**{"gram.fold": fold}):
            ts = []
            for _ in range(6):
                f = eng.features(db, 6)                   # the bins are built once per feature job: a fresh job per step, as in the bench
                eng.timer_start()
                eng.gram(f, norm, to_host=False)
                ts.append(eng.timer_stop_ms())
                chk = eng.gram_checksum(f)
                f.close()
            print("normalize %d fold %d bin_global %d: gram phase min %.4f med %.4f ms  checksum %s" % (norm, fold, binglobal, min(ts), sorted(ts)[3], chk[:2]))

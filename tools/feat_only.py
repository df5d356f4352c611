"""Feature-builder A/B on the config-3 batch: python tools/feat_only.py [opt=val[,opt=val] ...]
(context options as in gk_set_option, e.g. feat.gm_no_priv=1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grakel_amd import GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
eng = get_engine()
db = eng.upload(GraphBatch(*er_dataset_csr(10000, 100, 0.05, 5, 0), 5))
eng.wl_relabel(db, 5)
variants = [dict()] + [dict((x.split("=")[0], int(x.split("=")[1])) for x in v.split(",")) for v in sys.argv[1:]]
for opts in variants:
    with eng.options(**opts):
        ms = []
        for it in range(5):
            eng.profile(True)
            f = eng.features(db, 6)
            ms.append(eng.profile_get("features")[0])
            eng.profile(False)
            info = (f.n_cols, f.n_cols_low, f.nnz, f.max_count)
            f.close()
    print(opts, "features ms min %.3f med %.3f" % (min(ms), sorted(ms)[2]), info)

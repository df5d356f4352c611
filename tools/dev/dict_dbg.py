import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grakel_amd.batch import GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
eng = get_engine()
gp, rp, ci, lab = er_dataset_csr(10000, 100, 0.05, 5, 0)
db = eng.upload(GraphBatch(gp, rp, ci, lab, 5))
for _ in range(3):
    eng.wl_relabel(db, 5)
eng.set_option("wl.debug", 2)
eng.wl_relabel(db, 5)

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from grakel_amd import GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
eng = get_engine()
for (N, n, p) in ((4096, 30, 0.1), (4096, 60, 0.05), (512, 110, 0.03), (4096, 110, 0.03), (256, 110, 0.03)):
    gp, rp, ci, lab = er_dataset_csr(N, n, p, 5, 0)
    db = eng.upload(GraphBatch(gp, rp, ci, lab, 5))
    for mode in ("0", "1"):
        os.environ["GK_SP_DBG"] = mode
        eng.profile(True)
        for _ in range(3):
            pb = eng.sp_build(db, None, True); pb.close()
        ms, _ = eng.profile_get("sp")
        eng.profile(False)
        print("N=%d n=%d dbg=%s sp phase ms %.3f  (n^3 total %.2e)" % (N, n, mode, ms / 3, N * n ** 3))

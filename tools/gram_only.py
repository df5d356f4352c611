"""Gram-kernel A/B on the config-3 features: python tools/gram_only.py [N]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grakel_amd import GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
gp, rp, ci, lab = er_dataset_csr(N, 100, 0.05, 5, 0)
eng = get_engine()
db = eng.upload(GraphBatch(gp, rp, ci, lab, 5))
eng.wl_relabel(db, 5)
feat = eng.features(db, 6)
print("cols", feat.n_cols, "dtype", feat.dtype)
ref = None
variants = [dict()]
variants += [dict(x.split("=") for x in v.split(",")) for v in sys.argv[2:]]
for env in variants:
    for k in list(os.environ):
        if k.startswith("GK_GRAM") or k == "GK_LOW_DF": del os.environ[k]
    os.environ.update(env)
    if any(k in env for k in ("GK_LOW_DF", "GK_GRAM_NO_FP4")) or getattr(feat, "_special", False):
        feat.close(); feat = eng.features(db, 6)
        feat._special = any(k in env for k in ("GK_LOW_DF", "GK_GRAM_NO_FP4"))
        print("   features: dense cols", feat.n_cols, "low_df", env.get("GK_LOW_DF"))
    ms, tot = [], []
    for it in range(6):
        eng.timer_start()
        eng.gram(feat, 0, to_host=False)
        tot.append(eng.timer_stop_ms())
        ms.append(eng.gram_stats(feat)[1])
    fl = eng.gram_stats(feat)[0]
    K = eng.gram(feat, 0)
    chk = (int(K.sum()), int(np.trace(K)), bool(np.array_equal(K, K.T)))
    if ref is None: ref = chk
    print(env, "gemm ms min %.3f med %.3f | gram total min %.3f" % (min(ms), sorted(ms)[len(ms)//2], min(tot)), "TOP/s %.0f" % (fl / min(ms) / 1e9), "chk", chk, "OK" if chk == ref else "MISMATCH")
if N == 10000: print("golden sum 200604613570 trace 25874190")

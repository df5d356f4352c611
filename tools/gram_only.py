"""Gram-kernel A/B on the config-3 features: python tools/gram_only.py [N] [opt=val[,opt=val] ...] [--abl]

Each further argument is one variant: comma separated context options (gk_set_option names, e.g.
`gram.no_fp4=1`, `feat.low_df=16,gram.no_patch=1`).  `--abl` loads the tools' build of the library
(`make -C grakel_amd/csrc abl` -> libgk_hip_abl.so), whose Gram kernel honours GK_GRAM_ABL=<nostore|nok|noload|
nomfma|mfmaonly|nobarrier|puremfma>: timing ablations with WRONG results; the shipped library has none of that."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from grakel_amd import GraphBatch, _lib
args = [a for a in sys.argv[1:] if a != "--abl"]
if "--abl" in sys.argv:
    _lib.LIB_PATH = os.path.join(ROOT, "grakel_amd", "libgk_hip_abl.so")
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
N = int(args[0]) if args and args[0].isdigit() else 10000
gp, rp, ci, lab = er_dataset_csr(N, 100, 0.05, 5, 0)
eng = get_engine()
db = eng.upload(GraphBatch(gp, rp, ci, lab, 5))
eng.wl_relabel(db, 5)
feat = eng.features(db, 6)
print("cols", feat.n_cols, "operand", feat.operand)
ref = None
variants = [dict()]
variants += [dict((x.split("=")[0], int(x.split("=")[1])) for x in v.split(",")) for v in args if "=" in v]
for opts in variants:
    with eng.options(**opts):
        rebuilt = any(k.startswith("feat.") or k == "gram.no_fp4" for k in opts)
        f = eng.features(db, 6) if rebuilt else feat
        if rebuilt:
            print("   features: dense cols", f.n_cols, "rare", f.n_cols_low, "operand", f.operand)
        ms, tot = [], []
        for it in range(6):
            eng.timer_start()
            eng.gram(f, 0, to_host=False)
            tot.append(eng.timer_stop_ms())
            ms.append(eng.gram_stats(f)[1])
        fl = eng.gram_stats(f)[0]
        K = eng.gram(f, 0)
        chk = (int(K.sum()), int(np.trace(K)), bool(np.array_equal(K, K.T)))
        if ref is None: ref = chk
        print(opts, os.environ.get("GK_GRAM_ABL", ""), "gemm ms min %.3f med %.3f | gram total min %.3f" % (min(ms), sorted(ms)[len(ms)//2], min(tot)),
              "TOP/s %.0f" % (fl / min(ms) / 1e9), "chk", chk, "OK" if chk == ref else "MISMATCH")
        if "--abl" in sys.argv:
            import ctypes
            t = (ctypes.c_ulonglong * 12)()
            eng.lib.gk_debug_ws_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            if eng.lib.gk_debug_ws_times(eng.handle, t) == 0:
                for r, name in enumerate(("multiply", "store", "load")):
                    tot, bar, body, walk = (t[4 * r + k] for k in range(4))
                    print("      wg0 %-8s cycles: total %9d  barrier waits %9d  body %9d  other %9d" % (name, tot, bar, body, walk))
        if rebuilt:
            f.close()
if N == 10000: print("golden sum 200604613570 trace 25874190")

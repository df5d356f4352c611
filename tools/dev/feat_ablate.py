"""Feature-builder A/B on config 3 (or `N n p`): per-phase HIP-event time of the feature build, GK_GM_ABL ablations with --abl
(tools' build of the library, WRONG results by construction).  python tools/dev/feat_ablate.py [--abl] [N n p]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from grakel_amd import GraphBatch, _lib
args = [a for a in sys.argv[1:] if a != "--abl"]
if "--abl" in sys.argv:
    _lib.LIB_PATH = os.path.join(ROOT, "grakel_amd", "libgk_hip_abl.so")
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset_csr
N, n, p = (int(args[0]), int(args[1]), float(args[2])) if len(args) >= 3 else (10000, 100, 0.05)
eng = get_engine()
db = eng.upload(GraphBatch(*er_dataset_csr(N, n, p, 5, 0), 5))
eng.wl_relabel(db, 5)
for _ in range(3):
    eng.features(db, 6).close()
ts = []
for _ in range(8):
    eng.timer_start()
    f = eng.features(db, 6)
    ts.append(eng.timer_stop_ms())
    f.close()
print("GK_GM_ABL=%s features min %.4f med %.4f ms" % (os.environ.get("GK_GM_ABL", "-"), min(ts), sorted(ts)[len(ts) // 2]))

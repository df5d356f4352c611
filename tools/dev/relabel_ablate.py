"""Stream relabel A/B on config 3 (or `N n p`): HIP-event time of gk_wl_relabel, GK_SR_ABL ablations with --abl (tools' build,
WRONG results by construction).  python tools/dev/relabel_ablate.py [--abl] [N n p]"""
import os, sys
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
ts = []
for _ in range(10):
    eng.timer_start()
    try:
        eng.wl_relabel(db, 5)
    except Exception as e:
        print("relabel raised:", str(e)[:80])
    ts.append(eng.timer_stop_ms())
print("GK_SR_ABL=%s relabel min %.4f med %.4f ms  route %s" % (os.environ.get("GK_SR_ABL", "-"), min(ts), sorted(ts)[5], getattr(db, "stream_route", None)))

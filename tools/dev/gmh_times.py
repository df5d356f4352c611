"""Cycles per phase of gm_pairs_huge_kernel summed over the huge graphs of a published-like set (tools' build:
make -C grakel_amd/csrc abl):  python tools/dev/gmh_times.py reddit|dd"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import grakel_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "grakel_amd", "libgk_hip_abl.so")
import bench
from grakel_amd.engine import get_engine
wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "reddit")
eng = get_engine()
db = eng.upload(wl.batch)
t = (ctypes.c_ulonglong * 8)()
eng.lib.gk_debug_gmh_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
for rep in range(2):
    feat, _ = eng.wl_fit_transform(db, 5, to_host=False); feat.close()
    eng.lib.gk_debug_gmh_times(eng.handle, t, 1)
g = max(1, t[3])
print("%d huge graphs | cycles per graph (six levels): clear %d  insert %d  entries + statistics %d | scan of the other graphs %d per workgroup-graph" %
      (t[3], t[0] // g, t[1] // g, t[2] // g, t[4] // g))

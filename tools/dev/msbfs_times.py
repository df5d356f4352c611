"""Cycles per phase of sp_msbfs_kernel, summed over the workgroups of every size class, on a published-like set
(tools' build: make -C grakel_amd/csrc abl):  python tools/dev/msbfs_times.py reddit|dd|collab"""
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
t = (ctypes.c_ulonglong * 48)()
eng.lib.gk_debug_spb_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
for rep in range(2):
    pb = eng.sp_build(db, None, True); pb.close()
    eng.lib.gk_debug_spb_times(eng.handle, t, 1)
names = ["staging", "set-up", "pull", "update", "epilogue"]
for c in range(5):
    r = t[c * 8:(c + 1) * 8]
    if r[5]:
        print("class %d: %6d workgroups, %.1f levels each | cycles per workgroup: " % (c, r[5], r[6] / r[5]) +
              "  ".join("%s %d" % (n, r[i] // r[5]) for i, n in enumerate(names)) + "  | total %d" % (sum(r[:5]) // r[5]))

"""Cycles per phase of sp_hist_kernel's workgroup 0 at BASELINE config 4 (tools' build: make -C grakel_amd/csrc abl)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import grakel_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "grakel_amd", "libgk_hip_abl.so")
from grakel_amd.batch import sp_batch_from_input
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import nci1_like
eng = get_engine()
gb, _ = sp_batch_from_input(nci1_like(4110, 0, as_adj=True), True)
db = eng.upload(gb)
for opt in (0, 1):
    eng.set_option("sp.hist_no_batch", opt)
    for _ in range(3):
        pb = eng.sp_build(db, None, True); feat = eng.features(pb, 1); feat.close(); pb.close()
    t = (ctypes.c_ulonglong * 8)()
    eng.lib.gk_debug_sph_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert eng.lib.gk_debug_sph_times(eng.handle, t) == 0
    names = ["select", "clear+labels", "walk", "void check", "compaction", "finish", "rounds", "graphs"]
    print("no_batch=%d " % opt + "  ".join("%s %d" % (n, v) for n, v in zip(names, t)), "total cycles", sum(t[:6]))

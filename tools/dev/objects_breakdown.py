"""Where an estimator call from Python objects spends its wall, call by call (development)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import torch  # noqa: F401
import grakel_amd
from grakel_amd.engine import get_engine
from grakel_amd.batch import wl_batch_from_input

wl = bench.Workload("config3")
eng = get_engine()
X = wl.objects()
hold = []
if len(sys.argv) > 1 and sys.argv[1] == "hold":          # keep two earlier matrices alive, as bench.py does (Ku and a released one)
    out, Ku = bench.host_to_host(eng, wl, 5, 2)
    hold.append(Ku)
    print("h2h", {k: round(v["ms_per_step"], 2) for k, v in out.items() if isinstance(v, dict)})
est = grakel_amd.WeisfeilerLehman(n_iter=5)
est.fit_transform(X[:50])
K = est.fit_transform(X)
for i in range(8):
    K = None
    t0 = time.perf_counter()
    gb, _ = wl_batch_from_input(X)
    t1 = time.perf_counter()
    db = eng.upload(gb)
    eng.synchronize()
    t2 = time.perf_counter()
    feat, K = eng.wl_fit_transform(db, 5, to_host=True)
    t3 = time.perf_counter()
    d = eng.selfk(feat)
    feat.close(); db.close()
    t4 = time.perf_counter()
    K = None
    t5 = time.perf_counter()
    K = est.fit_transform(X)
    t6 = time.perf_counter()
    print("call %d: ingest %.2f upload %.2f fused+copy %.2f selfk+close %.2f | estimator %.2f ms" % (
        i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t6 - t5) * 1e3), flush=True)
print(open("/sys/fs/cgroup/cpu.stat").read() if os.path.exists("/sys/fs/cgroup/cpu.stat") else "")

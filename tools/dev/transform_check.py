"""transform by look-up (csrc/wl_transform.hip) against the joint route and the oracle; timings (development)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import grakel_amd
from grakel_amd.synthetic import er_dataset
from grakel_amd.engine import get_engine
from oracle import grakel_oracle as O

def case(name, X, Y, h, normalize=False):
    ref = O.WLOracle(n_iter=h, normalize=normalize)
    ref.fit_transform(X)
    Kref = ref.transform(Y)
    out = {}
    for route in ("lookup", "joint"):
        est = grakel_amd.WeisfeilerLehman(n_iter=h, normalize=normalize)
        est.transform_route = route
        est.fit(X)
        K = est.transform(Y)
        xd, yd = est.diagonal()
        out[route] = (K, xd, yd)
    cmp = (lambda a, b: np.allclose(a, b, rtol=1e-12, atol=0)) if normalize else np.array_equal
    ok = cmp(out["lookup"][0], Kref) and cmp(out["joint"][0], Kref)
    okd = np.array_equal(out["lookup"][1], out["joint"][1]) and np.array_equal(out["lookup"][2], out["joint"][2])
    print("%-34s lookup==oracle %s joint==oracle %s diagonals equal %s  (K sum %s)" % (
        name, cmp(out["lookup"][0], Kref), cmp(out["joint"][0], Kref), okd, float(Kref.sum())), flush=True)
    return ok and okd

X = er_dataset(300, 40, 0.08, 4, 1)
good = True
good &= case("targets = fitted graphs", X, X[:20], 4)
good &= case("fresh targets", X, er_dataset(25, 40, 0.08, 4, 99), 4)
good &= case("normalized", X, er_dataset(25, 40, 0.08, 4, 98), 4, normalize=True)
Yu = er_dataset(10, 40, 0.08, 6, 5)          # labels 4, 5 never seen in the fit
good &= case("unseen input labels", X, Yu, 3)
good &= case("isolated + tiny targets", er_dataset(200, 30, 0.05, 3, 7), [[{0: []}, {0: 1}], [{0: [1], 1: [0]}, {0: 0, 1: 2}]] + er_dataset(5, 30, 0.05, 3, 8), 5)
good &= case("one target", X, er_dataset(1, 40, 0.08, 4, 77), 5)
print("ALL GOOD" if good else "FAILURES", flush=True)

# timing: config-3-like fit, few targets
eng = get_engine()
Xb = er_dataset(10000, 100, 0.05, 5, 0)
for route in ("lookup", "joint"):
    est = grakel_amd.WeisfeilerLehman(n_iter=5)
    est.transform_route = route
    est.fit(Xb)
    for nt in (1, 100, 1000):
        Y = er_dataset(nt, 100, 0.05, 5, 1234)
        est.transform(Y); est.transform(Y)
        t0 = time.perf_counter()
        for _ in range(5):
            K = est.transform(Y)
        dt = (time.perf_counter() - t0) / 5
        eng.profile(True)
        est.transform(Y)
        ph = {k: round(eng.profile_get(k)[0], 4) for k in ("relabel", "features", "gram", "transform")}
        eng.profile(False)
        print(route, "targets", nt, "wall ms %.3f" % (dt * 1e3), "device phases", ph, "K sum", float(K.sum()), flush=True)

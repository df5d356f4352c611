"""Stream relabel route (wl_stream.hip) against the host-driven route and the oracle: per-level partitions, label counts,
Gram matrices; then timings of both routes at config-3 size.  Development tool (gpurun)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grakel_amd.batch import wl_batch_from_input, GraphBatch
from grakel_amd.engine import get_engine
from grakel_amd.synthetic import er_dataset, er_dataset_csr
from oracle import grakel_oracle as O

eng = get_engine()


def canon(a):
    return np.array(O.canonical_partition(a.tolist()))


def case(name, X, h):
    gb, _ = wl_batch_from_input(X)
    out = {}
    for route in ("stream", "host"):
        eng.set_option("wl.no_stream", 0 if route == "stream" else 1)
        eng.set_option("wl.debug", 1 if route == "stream" else 0)
        db = eng.upload(gb)
        counts = eng.wl_relabel(db, h)
        labs = [eng.wl_labels(db, l) for l in range(h + 1)]
        feat = eng.features(db, h + 1)
        K = eng.gram(feat)
        out[route] = (counts, labs, K, feat.n_cols, feat.n_cols_low)
        feat.close(); db.close()
    eng.set_option("wl.no_stream", 0); eng.set_option("wl.debug", 0)
    ok = out["stream"][0] == out["host"][0]
    for l in range(h + 1):
        same = np.array_equal(canon(out["stream"][1][l]), canon(out["host"][1][l]))
        dense = sorted(set(out["stream"][1][l].tolist())) == list(range(out["stream"][0][l])) if l > 0 else True
        if not (same and dense):
            print("   level %d partition same %s dense ids %s" % (l, same, dense))
        ok = ok and same and dense
    kk = np.array_equal(out["stream"][2], out["host"][2])
    print("%-28s counts %s  partitions+counts %s  K equal %s  cols %s/%s vs %s/%s" % (
        name, out["stream"][0], ok, kk, out["stream"][3], out["stream"][4], out["host"][3], out["host"][4]), flush=True)
    return ok and kk


good = True
good &= case("er 64x30 h3", er_dataset(64, 30, 0.1, 4, 1), 3)
good &= case("er 500x40 sparse h7", er_dataset(500, 40, 0.07, 3, 17), 7)
good &= case("er 300x100 h5", er_dataset(300, 100, 0.05, 5, 3), 5)
good &= case("only isolated", [[{i: [] for i in range(5)}, {i: i % 3 for i in range(5)}] for _ in range(7)], 3)
good &= case("17 labels (hashed level 1)", er_dataset(200, 30, 0.1, 17, 5), 4)
good &= case("er 2000x100 h5", er_dataset(2000, 100, 0.05, 5, 0), 5)
print("ALL GOOD" if good else "FAILURES", flush=True)

# ---- timing at config-3 size
gp, rp, ci, lab = er_dataset_csr(10000, 100, 0.05, 5, 0)
full = GraphBatch(gp, rp, ci, lab, 5)
db = eng.upload(full)
for route in ("host", "stream"):
    eng.set_option("wl.no_stream", 0 if route == "stream" else 1)
    for _ in range(3):
        c = eng.wl_relabel(db, 5); f = eng.features(db, 6); eng.gram(f, 0, to_host=False); f.close()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        c = eng.wl_relabel(db, 5); f = eng.features(db, 6); eng.gram(f, 0, to_host=False)
        if _ < 19:
            f.close()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / 20
    s = eng.gram_checksum(f)
    f.close()
    eng.profile(True)
    c = eng.wl_relabel(db, 5); f = eng.features(db, 6); eng.gram(f, 0, to_host=False); f.close()
    ph = {k: round(eng.profile_get(k)[0], 4) for k in ("relabel", "features", "gram")}
    eng.profile(False)
    print(route, "ms/step %.4f" % (dt * 1e3), "counts", c, "K sum/trace", s[:2], "phases", ph, flush=True)

"""Host ingestion of config 3's 10 k dict graphs by thread count (run on the GPU box's host; no device work)."""
import time
from grakel_amd.synthetic import er_dataset
from grakel_amd import batch
import grakel_amd._gk_ingest as I
import os
print("cores", os.cpu_count())
X = er_dataset(10000, 100, 0.05, 5, 0)
ref = I.wl_ingest(X, 2, False, 0, 1)
for nt in (1, 2, 4, 8, 16, 24, 32, 48, 64, 0):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = I.wl_ingest(X, 2, False, 0, nt); ts.append(time.perf_counter() - t0)
    same = all(bytes(a) == bytes(b) for a, b in zip(r, ref))
    print("threads %2d: C walk %.1f ms (min of 5)  same=%s" % (nt, min(ts) * 1e3, same))
for nt in (1, 0):
    batch.INGEST_THREADS = nt
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); batch.wl_batch_from_input(X); ts.append(time.perf_counter() - t0)
    print("wl_batch_from_input, INGEST_THREADS=%d: %.1f ms" % (nt, min(ts) * 1e3))

# round 5: the other forms on the same machinery
off, Xs, Xa = 1, [], []
import numpy as np
for ed, lab in X:
    n = len(lab)
    Xs.append([{(u + off, v + off) for u, l in ed.items() for v in l}, {u + off: x for u, x in lab.items()}])
    A = np.zeros((n, n), dtype=np.int64)
    for u, l in ed.items():
        A[u, l] = 1
    Xa.append([A, lab])
    off += n
for name, Z in (("tuple sets", Xs), ("adjacency matrices", Xa)):
    for nt in (1, 8, 16, 32, 64):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); r = I.wl_ingest(Z, 2, False, 0, nt); ts.append(time.perf_counter() - t0)
        print("%s threads %2d: C walk %.1f ms" % (name, nt, min(ts) * 1e3))

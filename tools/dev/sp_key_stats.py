#!/usr/bin/env python3
"""Distinct ShortestPath feature keys (label_u, label_v, distance) per graph of a published-dataset stand-in -- CPU only
(scipy), the numbers DESIGN.md §3 "ShortestPath on large graphs" quotes:

    python tools/dev/sp_key_stats.py dd|reddit|collab|nci1

Prints the job's alphabet, largest distance, distinct keys in the job (= width Q of a counter row), pairs, the sum of the
per-graph distinct keys, and how many graphs (and pairs) exceed an LDS table of 6 144 / 12 288 / 16 384 / 32 768 keys."""
import os
import sys
import time

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import shortest_path

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from grakel_amd import synthetic as S  # noqa: E402


def main():
    name = sys.argv[1]
    graphs = S.PUBLISHED_LIKE[name][0]()
    labs = np.unique(np.concatenate([g[3] for g in graphs]))
    L = len(labs)
    t = time.time()
    maxd, allk, per = 0, set(), []
    for n, eu, ev, lab in graphs:
        A = csr_matrix((np.ones(len(eu) * 2), (np.r_[eu, ev], np.r_[ev, eu])), shape=(n, n))
        D = shortest_path(A, unweighted=True)
        m = np.isfinite(D) & ~np.eye(n, dtype=bool)
        l = np.searchsorted(labs, lab)
        i, j = np.nonzero(m)
        d = D[i, j].astype(np.int64)
        key = (l[i] * L + l[j]) * 4096 + d
        u = np.unique(key)
        maxd = max(maxd, int(d.max()) if len(d) else 0)
        per.append((n, len(key), len(u)))
        allk.update(u.tolist())
    per = np.array(per)
    print(name, "labels", L, "max distance", maxd, "key space", L * L * (maxd + 1), "distinct keys in the job", len(allk),
          "pairs", int(per[:, 1].sum()), "sum of per-graph distinct keys", int(per[:, 2].sum()), "(%.0f s)" % (time.time() - t))
    print("largest number of distinct keys in one graph:", int(per[:, 2].max()))
    for th in (6144, 12288, 16384, 32768):
        over = per[:, 2] > th
        print("graphs above %5d distinct keys: %4d of %d, holding %d pairs" % (th, int(over.sum()), len(per), int(per[over, 1].sum())))


if __name__ == "__main__":
    main()

"""CPU oracle, vectorised form of the ShortestPath kernel for UNIT-WEIGHT graphs -- TEST INFRASTRUCTURE ONLY.

Nothing in ``grakel_amd/`` may import this module (same rule as ``grakel_oracle.py``): only ``tests/``,
``tests/golden/make_golden.py`` and the assert legs of ``tools/published_like.py`` / ``bench.py`` use it, as the checker.

What it restates (all citations into /root/reference, ysig/GraKeL v0.1.11):

* ``Graph.build_shortest_path_matrix`` -> ``floyd_warshall`` (``grakel/graph.py:588-687,1767-1794``) on an adjacency
  matrix of 0/1 entries: the result is the hop distance, ``inf`` for unreachable pairs -- here one breadth-first search
  per source (``scipy.sparse.csgraph.shortest_path(unweighted=True)``), which gives the same integers;
* ``ShortestPath.parse_input`` pair walk (``grakel/kernels/shortest_path.py:468-490``): every ordered pair ``u != v`` with a
  finite distance counts one occurrence of the key ``(label_u, label_v, distance)`` (``lhash_labels``, ``:510-511``);
* ``fit_transform`` (``:370-410``): ``K = Phi . Phi^T`` over the job's distinct keys.  The column ORDER (first seen) does not
  enter K, so the keys are numbered in sorted order here.

Parity status: PINNED -- ``tests/test_oracle.py::test_fast_sp_oracle_*`` checks this file against the literal
``grakel_oracle.SPOracle`` and against matrices the REAL reference produced (``tests/golden/pub_*.npz[sp_K]``,
``pub_*_sp_big.npz``: the largest graphs of the D&D-/REDDIT-like sets through grakel 0.1.11).

Why it exists: the literal oracle walks pairs in Python (the D&D-like set has 172 M of them, the REDDIT-like set 643 M);
this form does the full sets in minutes, so that full-size device runs can be checked entry by entry.
"""
import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import shortest_path

DIST_SPAN = 1 << 12       # distances of a unit-weight graph are below its vertex count; the stand-ins stay below 4096


def graph_key_counts(n, eu, ev, lab_ids, n_labels, with_labels=True):
    """One graph (n vertices, undirected edges eu < ev, dense label ids) -> (sorted distinct keys, counts, pairs)."""
    if n == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), 0
    A = csr_matrix((np.ones(2 * len(eu), np.int8), (np.r_[eu, ev], np.r_[ev, eu])), shape=(n, n))
    D = shortest_path(A, unweighted=True, directed=False)
    finite = np.isfinite(D)
    np.fill_diagonal(finite, False)                                    # shortest_path.py:471: u == v is skipped
    i, j = np.nonzero(finite)
    d = D[i, j].astype(np.int64)
    del D, finite
    assert d.size == 0 or int(d.max()) < DIST_SPAN
    if with_labels:
        key = (lab_ids[i] * np.int64(n_labels) + lab_ids[j]) * DIST_SPAN + d
    else:
        key = d
    keys, counts = np.unique(key, return_counts=True)
    return keys, counts.astype(np.int64), int(key.size)


def sp_unit_gram(graphs, with_labels=True, progress=None):
    """`graphs`: list of (n, eu, ev, labels) as grakel_amd.synthetic emits them.  Returns (K int64 [N, N], number of
    distinct keys in the job = len(ShortestPath._enum), number of pairs)."""
    labs = np.unique(np.concatenate([np.asarray(g[3]) for g in graphs])) if graphs else np.zeros(0, np.int64)
    L = max(len(labs), 1)
    rows, cols, vals, pairs = [], [], [], 0
    for g, (n, eu, ev, lab) in enumerate(graphs):
        k, c, p = graph_key_counts(n, eu, ev, np.searchsorted(labs, lab).astype(np.int64), L, with_labels)
        rows.append(np.full(k.size, g, np.int64)), cols.append(k), vals.append(c)
        pairs += p
        if progress and g % progress == 0:
            print("   sp_fast: graph", g, "of", len(graphs), flush=True)
    cols = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    uniq, cid = np.unique(cols, return_inverse=True)
    phi = csr_matrix((np.concatenate(vals), (np.concatenate(rows), cid)), shape=(len(graphs), max(uniq.size, 1)), dtype=np.int64)
    K = (phi @ phi.T).toarray().astype(np.int64)                       # shortest_path.py:404 (np.dot of the dense Phi)
    return K, int(uniq.size), pairs

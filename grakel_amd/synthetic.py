"""Deterministic synthetic graph sets behind BASELINE.json's configs.

These are input *specifications* (SURVEY.md 8d / Appendix A): the checksums
in BASELINE.md are only reproducible with exactly this draw order
(``numpy.random.RandomState``, one shared stream; per graph the edge mask is
drawn first, then the labels).

Two emitters produce the same graphs:

* ``er_dataset`` / ``nci1_like`` -- the grakel input form
  ``[edge_dict_or_adjacency, node_labels]`` (python objects);
* ``er_dataset_csr`` -- the packed CSR batch arrays directly, for benchmarks
  that must not spend minutes building python dicts.
"""
import numpy as np


def er_dataset(N, n, p, L, seed):
    """N Erdos-Renyi graphs G(n, p), L discrete labels, grakel input form.

    config 2: er_dataset(1000, 50, 0.1, 5, 0);  config 3: (10000, 100, 0.05, 5, 0);
    config 5: (50000, 30, 0.1, 5, 0).
    """
    rs = np.random.RandomState(seed)
    iu = np.triu_indices(n, 1)
    out = []
    for _ in range(N):
        mask = rs.rand(len(iu[0])) < p
        u, v = iu[0][mask], iu[1][mask]
        ed = {i: [] for i in range(n)}
        for a, b in zip(u.tolist(), v.tolist()):
            ed[a].append(b)
            ed[b].append(a)
        labels = dict(enumerate(rs.randint(0, L, n).tolist()))
        out.append([ed, labels])
    return out


def er_dataset_csr(N, n, p, L, seed):
    """Same graphs as ``er_dataset`` as (graph_ptr, row_ptr, col_idx, labels) int32 arrays.

    Node ids are global (graph g owns [g*n, (g+1)*n)); col_idx holds global ids;
    neighbour lists are ascending (the WL signature sorts neighbour *labels*, so
    list order is immaterial).
    """
    rs = np.random.RandomState(seed)
    iu0, iu1 = np.triu_indices(n, 1)
    deg = np.zeros(N * n, dtype=np.int64)
    srcs, dsts = [], []
    labels = np.empty(N * n, dtype=np.int32)
    for g in range(N):
        mask = rs.rand(len(iu0)) < p
        u = iu0[mask] + g * n
        v = iu1[mask] + g * n
        srcs.append(u), srcs.append(v)
        dsts.append(v), dsts.append(u)
        labels[g * n:(g + 1) * n] = rs.randint(0, L, n)
    src = np.concatenate(srcs) if srcs else np.zeros(0, np.int64)
    dst = np.concatenate(dsts) if dsts else np.zeros(0, np.int64)
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    deg = np.bincount(src, minlength=N * n)
    row_ptr = np.zeros(N * n + 1, dtype=np.int64)
    np.cumsum(deg, out=row_ptr[1:])
    graph_ptr = (np.arange(N + 1, dtype=np.int64) * n)
    return (graph_ptr.astype(np.int32), row_ptr.astype(np.int32),
            dst.astype(np.int32), labels)


def nci1_like(N, seed, as_adj):
    """Config-4 stand-in with NCI1's published statistics (real NCI1 is not bundled).

    Random recursive tree with short back-links + a few ring-closing edges,
    37 skewed labels.  ``as_adj`` -> adjacency input (Floyd-Warshall route),
    else dict-of-lists (Dijkstra route).
    """
    rs = np.random.RandomState(seed)
    pl = 1.0 / np.arange(1, 38) ** 1.5
    pl /= pl.sum()
    out = []
    for _ in range(N):
        n = int(np.clip(round(rs.gamma(4.5, 29.87 / 4.5)), 3, 111))
        edges = set()
        for v in range(1, n):
            u = rs.randint(max(0, v - 3), v)
            edges.add((u, v))
        for _k in range(rs.binomial(n, 0.09)):
            a, b = rs.randint(0, n, 2)
            if a != b:
                edges.add((min(a, b), max(a, b)))
        labels = dict(enumerate(rs.choice(37, n, p=pl).tolist()))
        if as_adj:
            A = np.zeros((n, n), dtype=int)
            for a, b in edges:
                A[a, b] = A[b, a] = 1
            out.append([A, labels])
        else:
            ed = {i: [] for i in range(n)}
            for a, b in edges:
                ed[a].append(b)
                ed[b].append(a)
            out.append([ed, labels])
    return out


def random_labelled_graphs(N, nmin, nmax, p, L, seed, directed=False, fmt="dict"):
    """Small ragged test sets in several grakel input forms (used by tests only)."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(N):
        n = int(rs.randint(nmin, nmax + 1))
        A = (rs.rand(n, n) < p).astype(int)
        np.fill_diagonal(A, 0)
        if not directed:
            A = np.triu(A, 1)
            A = A + A.T
        lab = dict(enumerate(rs.randint(0, L, n).tolist()))
        if fmt == "adj":
            out.append([A, lab])
        elif fmt == "dict":
            out.append([{i: np.nonzero(A[i])[0].tolist() for i in range(n)}, lab])
        elif fmt == "tuples":
            ii, jj = np.nonzero(A)
            es = {(int(a), int(b)): 1.0 for a, b in zip(ii, jj)}
            if not es:          # an edgeless graph has no tuple form; fall back to adjacency
                out.append([A, lab])
            else:
                verts = {a for e in es for a in e}
                out.append([es, {v: lab[v] for v in verts}])
        else:
            raise ValueError(fmt)
    return out

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


# ------------------------------------------------------------------------------------------------------
# Stand-ins for the TU datasets the reference PUBLISHES its running times on (doc/benchmarks/evaluation.rst:
# 19-73, comparison.rst:22-39).  The real files cannot be downloaded here (datasets/base.py:78-79), so these
# generators reproduce the published statistics -- number of graphs, vertex-count range and mean, edge
# density, label alphabet -- and the structural trait that makes each set hard for a WL / ShortestPath engine
# (D&D: graphs of thousands of vertices; REDDIT-BINARY: hubs of degree > 1 000 and, with the degree labels
# `fetch_dataset(..., produce_labels_nodes=True)` gives unlabelled sets (datasets/base.py:243-245), an input
# alphabet of hundreds of labels; COLLAB: near-cliques, mean degree ~ 60; NCI1: small molecules whose WL
# labels are shared by most graphs).  Every generator returns `Graphs`: a list of (n, eu, ev, labels) with
# eu < ev the undirected edges, emitted as grakel input objects (`as_grakel`) or as a packed CSR batch
# (`as_csr`) -- the same graphs either way.
# ------------------------------------------------------------------------------------------------------
def _finish_edges(n, pairs):
    """unique undirected edges (u < v) of a list of int arrays [2, k]"""
    if not pairs:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    e = np.concatenate(pairs, axis=1)
    lo, hi = np.minimum(e[0], e[1]), np.maximum(e[0], e[1])
    keep = lo != hi
    key = np.unique(lo[keep] * np.int64(n) + hi[keep])
    return key // n, key % n


def _degree_labels(n, eu, ev):
    return np.bincount(np.concatenate([eu, ev]), minlength=n).astype(np.int64)


def nci1_like_graphs(N=4110, seed=0):
    """The graphs of ``nci1_like`` (same draw order, same graphs) in the (n, eu, ev, labels) form."""
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
        labels = rs.choice(37, n, p=pl).astype(np.int64)
        e = np.array(sorted(edges), np.int64).reshape(-1, 2)
        out.append((n, e[:, 0].copy(), e[:, 1].copy(), labels))
    return out


def dd_like_graphs(N=1178, seed=0, giant=5748):
    """D&D stand-in (evaluation.rst:19: WL-VH 5 m 53 s, SP 55 m 59 s): 1 178 protein contact graphs, 30..5 748
    vertices (mean 284), 2.5 edges per vertex, 82 labels.  Backbone chain + contacts to the 2nd..4th predecessor
    + a few long-range contacts; sizes log-normal, ONE graph of `giant` vertices (as the real set has)."""
    rs = np.random.RandomState(seed)
    pl = 1.0 / np.arange(1, 83) ** 1.2
    pl /= pl.sum()
    out = []
    for g in range(N):
        n = int(np.clip(round(rs.lognormal(5.44, 0.6)), 30, 3000))
        if giant and g == N // 2:
            n = int(giant)
        v = np.arange(1, n)
        pairs = [np.stack([v - 1, v])]
        for back, prob in ((2, 0.6), (3, 0.5), (4, 0.3)):
            w = np.arange(back, n)
            w = w[rs.rand(w.size) < prob]
            pairs.append(np.stack([w - back, w]))
        k = rs.binomial(n, 0.12)
        pairs.append(rs.randint(0, n, (2, k)))
        eu, ev = _finish_edges(n, pairs)
        out.append((n, eu, ev, rs.choice(82, n, p=pl).astype(np.int64)))
    return out


def reddit_like_graphs(N=2000, seed=0, degree_labels=True):
    """REDDIT-BINARY stand-in (evaluation.rst:63: WL-VH 16 m 3 s, SP 4 h 48 m): 2 000 discussion threads, 6..3 782
    vertices (mean 430), 1.16 edges per vertex, unlabelled.  A tree in which three vertices out of four answer one of
    1..3 hub users (hub degrees of several hundred to > 1 000) + a few cross links.  Labels: the vertex degrees
    (`produce_labels_nodes=True`, datasets/base.py:243-245) or one constant label."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(N):
        n = int(np.clip(round(rs.lognormal(5.744, 0.8)), 6, 3782))
        hubs = 1 + int(rs.randint(0, 3))
        v = np.arange(1, n)
        to_hub = rs.rand(n - 1) < 0.75
        parent = np.where(to_hub, rs.randint(0, hubs, n - 1), (rs.rand(n - 1) * v).astype(np.int64))
        parent = np.minimum(parent, v - 1)
        k = rs.binomial(n, 0.16)
        eu, ev = _finish_edges(n, [np.stack([parent, v]), rs.randint(0, n, (2, k))])
        lab = _degree_labels(n, eu, ev) if degree_labels else np.zeros(n, np.int64)
        out.append((n, eu, ev, lab))
    return out


def collab_like_graphs(N=5000, seed=0, degree_labels=True):
    """COLLAB stand-in (evaluation.rst:63: WL-VH 38 m 42 s, SP 1 h 9 m): 5 000 ego networks of co-authorship, 32..492
    vertices (mean 74.5), ~2 450 edges per graph (mean degree ~ 60), unlabelled.  The ego is adjacent to everybody, the
    others form 1 + Poisson(n / 90) overlapping cliques (papers / groups)."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(N):
        n = int(np.clip(32 + round(rs.lognormal(3.2, 1.0)), 32, 492))
        groups = 1 + int(rs.poisson(n / 90.0))
        member = rs.randint(0, groups, n)
        second = np.where(rs.rand(n) < 0.25, rs.randint(0, groups, n), -1)
        pairs = [np.stack([np.zeros(n - 1, np.int64), np.arange(1, n)])]
        for c in range(groups):
            m = np.flatnonzero((member == c) | (second == c))
            if m.size > 1:
                a, b = np.triu_indices(m.size, 1)
                pairs.append(np.stack([m[a], m[b]]))
        eu, ev = _finish_edges(n, pairs)
        lab = _degree_labels(n, eu, ev) if degree_labels else np.zeros(n, np.int64)
        out.append((n, eu, ev, lab))
    return out


PUBLISHED_LIKE = {          # name -> (generator, what the reference publishes for the real set: evaluation.rst:19-73)
    "nci1": (nci1_like_graphs, {"graphs": 4110, "WL-VH": "7 m 5 s", "SP": "1 m 10 s"}),
    "dd": (dd_like_graphs, {"graphs": 1178, "WL-VH": "5 m 53 s", "SP": "55 m 59 s"}),
    "reddit": (reddit_like_graphs, {"graphs": 2000, "WL-VH": "16 m 3 s", "SP": "4 h 48 m"}),
    "collab": (collab_like_graphs, {"graphs": 5000, "WL-VH": "38 m 42 s", "SP": "1 h 9 m"}),
}


def as_grakel(graphs, adjacency=False):
    """`Graphs` -> grakel input objects `[{u: [v, ...]}, {u: label}]` (or `[ndarray adjacency, {u: label}]`)."""
    out = []
    for n, eu, ev, lab in graphs:
        labels = dict(enumerate(lab.tolist()))
        if adjacency:
            A = np.zeros((n, n), dtype=int)
            A[eu, ev] = 1
            A[ev, eu] = 1
            out.append([A, labels])
            continue
        src = np.concatenate([eu, ev])
        dst = np.concatenate([ev, eu])
        order = np.lexsort((dst, src))
        src, dst = src[order], dst[order]
        cut = np.searchsorted(src, np.arange(n + 1))
        d = dst.tolist()
        out.append([{u: d[cut[u]:cut[u + 1]] for u in range(n)}, labels])
    return out


def as_csr(graphs):
    """`Graphs` -> (graph_ptr, row_ptr, col_idx, node_label ids, n_labels): the packed batch of the same graphs, labels
    compressed to dense ids in sorted order (what the estimator's ingestion does, weisfeiler_lehman.py:199-210)."""
    sizes = np.array([g[0] for g in graphs], np.int64)
    gp = np.zeros(len(graphs) + 1, np.int64)
    np.cumsum(sizes, out=gp[1:])
    srcs, dsts = [], []
    for (n, eu, ev, _), off in zip(graphs, gp[:-1].tolist()):
        srcs.append(eu + off), srcs.append(ev + off)
        dsts.append(ev + off), dsts.append(eu + off)
    src = np.concatenate(srcs) if srcs else np.zeros(0, np.int64)
    dst = np.concatenate(dsts) if dsts else np.zeros(0, np.int64)
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    V = int(gp[-1])
    rp = np.zeros(V + 1, np.int64)
    np.cumsum(np.bincount(src, minlength=V), out=rp[1:])
    lab = np.concatenate([g[3] for g in graphs])
    uniq, ids = np.unique(lab, return_inverse=True)
    return gp.astype(np.int32), rp.astype(np.int32), dst.astype(np.int32), ids.astype(np.int32), int(uniq.size)

"""CPU oracle: a restatement of GraKeL's WL / VertexHistogram / ShortestPath path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``grakel_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` use it, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  ``tests/golden/make_golden.py`` runs the real
reference (v0.1.11 built from ``/root/reference``) and this oracle on the same
inputs and stores the reference's outputs under ``tests/golden``;
``tests/test_oracle.py`` checks this file against those fixtures and against
the reference's own known-answer vectors (4x4 APSP matrix
``grakel/tests/test_graph.py:61-74``, the H2O/H3O doctest scalars
``doc/documentation/introduction.rst:313-343`` and
``doc/documentation/creating_kernels.rst:104-109``).

Everything here is deliberately written the slow, literal way the reference
computes it (string credentials, first-seen column enumeration, one dense
N x N matrix per WL iteration) so that (a) the integers it produces are the
reference's integers and (b) timing it is an honest stand-in for the
reference's CPU cost.  All ``file:line`` citations are into /root/reference.
"""
from collections import Counter
import heapq
import numbers

import numpy as np
from scipy.sparse import csr_matrix, issparse

INF = float("inf")


# --------------------------------------------------------------------------
# Input normalisation (grakel/graph.py)
# --------------------------------------------------------------------------
class ParsedGraph(object):
    """What ``grakel.Graph`` boils an input down to, for this path only.

    kind      : "adjacency" or "dictionary" (which family the input was)
    vertices  : sorted list of vertex symbols      (graph.py:894-907)
    edges     : {u: {v: w}} two level edge dict    (graph.py:963-965,1613-1705)
    labels    : node-label dict as the user gave it (keyed by symbol / index)
    adjacency : n x n float array or None
    """

    def __init__(self, kind, vertices, edges, labels, adjacency):
        self.kind, self.vertices, self.edges = kind, vertices, edges
        self.labels, self.adjacency = labels, adjacency


def _as_adjacency(g):
    # graph.py:1542-1585 (is_adjacency)
    if isinstance(g, np.ndarray) and g.ndim == 2:
        return g
    if issparse(g):
        return np.asarray(g.todense())
    if type(g) is list and all(
            isinstance(r, list) and all(isinstance(i, numbers.Number) for i in r) for r in g):
        return np.array(g)
    return None


def _as_edge_dict(g):
    """graph.py:1588-1709 (is_edge_dictionary with transform=True)."""
    def close(ed, keys, vals):
        for v in vals - keys:          # sinks get an empty adjacency
            ed[v] = dict()
        return keys | vals, ed

    if type(g) is dict:
        if all(type(k) is tuple and len(k) == 2 and isinstance(w, numbers.Number)
               for k, w in g.items()):
            ed, ks, vs = dict(), set(), set()
            for (a, b), w in g.items():
                ks.add(a), vs.add(b)
                ed.setdefault(a, dict())[b] = w
            return close(ed, ks, vs)
        if all(isinstance(d, list) for d in g.values()):
            ed, ks, vs = dict(), set(), set()
            for a, lst in g.items():
                ks.add(a)
                vs |= set(lst)
                for b in lst:
                    ed.setdefault(a, dict())[b] = 1.
            return close(ed, ks, vs)
        if all(isinstance(d, dict) and all(isinstance(w, numbers.Number) for w in d.values())
               for d in g.values()):
            ed = {a: dict(d) for a, d in g.items()}
            ks = set(ed.keys())
            vs = {b for a in ed for b in ed[a]}
            return close(ed, ks, vs)
    try:
        items = list(g)
    except TypeError:
        return None
    if all(type(t) is tuple and len(t) == 2 for t in items):
        ed, ks, vs = dict(), set(), set()
        for a, b in items:
            ks.add(a), vs.add(b)
            ed.setdefault(a, dict())[b] = 1.
        return close(ed, ks, vs)
    if all(type(t) is tuple and len(t) == 3 for t in items):
        ed, ks, vs = dict(), set(), set()
        for a, b, w in items:
            ks.add(a), vs.add(b)
            ed.setdefault(a, dict())[b] = w
        return close(ed, ks, vs)
    return None


def parse_graph(obj, labels):
    """graph.py:167-232 build_graph: adjacency is tried first, then edge dict."""
    A = _as_adjacency(obj)
    if A is not None:
        if A.shape[0] != A.shape[1]:
            raise ValueError('input matrix must be squared')          # graph.py:943-944
        n = A.shape[0]
        edges = {i: dict() for i in range(n)}                          # graph.py:960-965
        ii, jj = np.where(A > 0)
        for i, j in zip(ii.tolist(), jj.tolist()):
            edges[i][j] = A[i, j]
        return ParsedGraph("adjacency", list(range(n)), edges, labels, np.array(A, dtype=float))
    r = _as_edge_dict(obj)
    if r is None:
        raise ValueError('Unsupported input type.')                   # graph.py:207-211
    verts, edges = r
    return ParsedGraph("dictionary", sorted(verts), edges, labels, None)


def _elements(X, allowed_len, what):
    """Common iteration/validation shell of every ``parse_input``
    (weisfeiler_lehman.py:142-194, vertex_histogram.py:75-101, shortest_path.py:441-466)."""
    import warnings
    from collections.abc import Iterable
    if not isinstance(X, Iterable):
        raise TypeError('input must be an iterable\n')
    out = []
    for idx, x in enumerate(iter(X)):
        if not isinstance(x, Iterable):
            raise TypeError('each element of X must be ' + what)
        x = list(x)
        if len(x) == 0:
            warnings.warn('Ignoring empty element on index: ' + str(idx))
            continue
        if not allowed_len(len(x)):
            raise TypeError('each element of X must be ' + what)
        out.append(x)
    if len(out) == 0:
        raise ValueError('parsed input is empty')
    return out


# --------------------------------------------------------------------------
# VertexHistogram (grakel/kernels/vertex_histogram.py)
# --------------------------------------------------------------------------
def vh_features(label_dicts, columns):
    """vertex_histogram.py:103-137: Counter per graph, first-seen column ids."""
    rows, cols, data = [], [], []
    for gi, L in enumerate(label_dicts):
        for label, freq in Counter(L.values()).items():
            c = columns.get(label)
            if c is None:
                c = len(columns)
                columns[label] = c
            rows.append(gi), cols.append(c), data.append(freq)
    return csr_matrix((data, (rows, cols)), shape=(len(label_dicts), len(columns)),
                      dtype="float64")


def _normalize(K, dr, dc, nan_to_num):
    with np.errstate(divide='ignore', invalid='ignore'):
        K = K / np.sqrt(np.outer(dr, dc))
    return np.nan_to_num(K) if nan_to_num else K


class VHOracle(object):
    """VertexHistogram.fit_transform / transform (kernel.py:167-204,123-165)."""

    def __init__(self, normalize=False):
        self.normalize = normalize

    def fit_transform(self, X):
        els = _elements(X, lambda n: n in (2, 3), 'a list with a graph and node labels')
        self.columns = dict()
        self.phi = vh_features([x[1] for x in els], self.columns)
        K = (self.phi @ self.phi.T).toarray()                         # vertex_histogram.py:176-182
        self.x_diag = np.diagonal(K).copy()
        return _normalize(K, self.x_diag, self.x_diag, False) if self.normalize else K

    def transform(self, Y):
        els = _elements(Y, lambda n: n in (2, 3), 'a list with a graph and node labels')
        cols = dict(self.columns)                                     # vertex_histogram.py:84
        phi_y = vh_features([x[1] for x in els], cols)
        K = (phi_y[:, :self.phi.shape[1]] @ self.phi.T).toarray()     # vertex_histogram.py:179
        self.y_diag = np.asarray(phi_y.multiply(phi_y).sum(axis=1)).ravel()
        return _normalize(K, self.y_diag, self.x_diag, False) if self.normalize else K


class EHOracle(VHOracle):
    """EdgeHistogram (grakel/kernels/edge_histogram.py:57-175): the VertexHistogram machinery
    over the VALUES of the edge-label dictionary x[2]; inputs must have exactly 3 elements."""

    def fit_transform(self, X):
        els = _elements(X, lambda n: n == 3, 'a list with a graph, node labels and edge labels')
        return VHOracle.fit_transform(self, [[x[0], x[2]] for x in els])

    def transform(self, Y):
        els = _elements(Y, lambda n: n == 3, 'a list with a graph, node labels and edge labels')
        return VHOracle.transform(self, [[x[0], x[2]] for x in els])


# --------------------------------------------------------------------------
# Weisfeiler-Lehman (grakel/kernels/weisfeiler_lehman.py)
# --------------------------------------------------------------------------
def _wl_ingest(X):
    els = _elements(X, lambda n: n >= 2, 'a list with at least a graph and node labels')
    graphs = [parse_graph(x[0], x[1]) for x in els]                   # weisfeiler_lehman.py:157-173
    return [g.edges for g in graphs], [dict(g.labels) for g in graphs]


def _credential(L, ed, v):
    # weisfeiler_lehman.py:235-239 -- the STRING is the dictionary key
    return str(L[v]) + "," + str(sorted([L[n] for n in ed.get(v, dict()).keys()]))


def _level_gram(phi):
    return (phi @ phi.T).toarray()                                     # vertex_histogram.py:156-184


class WLOracle(object):
    """WeisfeilerLehman(base_graph_kernel=VertexHistogram)."""

    def __init__(self, n_iter=5, normalize=False):
        self.n_iter, self.normalize = n_iter, normalize

    def fit_transform(self, X, keep_levels=False, n_jobs=None):
        """n_jobs > 1: the per-level base-kernel products run in worker processes, as the reference hands
        ``efit_transform(base_graph_kernel[i], level i)`` to joblib (weisfeiler_lehman.py:271-283); the relabel
        loop itself stays sequential there too (it is the generator joblib consumes)."""
        eds, L = _wl_ingest(X)
        n_lev = self.n_iter + 1                                        # weisfeiler_lehman.py:114
        inv0 = {dv: i for i, dv in enumerate(sorted({l for d in L for l in d.values()}))}
        self.inv_labels = {0: inv0}                                    # weisfeiler_lehman.py:199-210
        count = len(inv0)
        L = [{k: inv0[v] for k, v in d.items()} for d in L]
        self.vh, mats, self.levels = [], [], []
        for i in range(n_lev):
            if i > 0:                                                  # weisfeiler_lehman.py:223-258
                creds = [{v: _credential(Lj, ed, v) for v in Lj} for Lj, ed in zip(L, eds)]
                inv = dict()
                for c in sorted({c for d in creds for c in d.values()}):
                    inv[c] = count
                    count += 1
                L = [{v: inv[c] for v, c in d.items()} for d in creds]
                self.inv_labels[i] = inv
            if keep_levels:
                self.levels.append([dict(d) for d in L])
            cols = dict()
            phi = vh_features(L, cols)                                 # weisfeiler_lehman.py:269
            self.vh.append((phi, cols))
            if not n_jobs or n_jobs <= 1:
                mats.append((phi @ phi.T).toarray())
        if n_jobs and n_jobs > 1:
            import multiprocessing
            with multiprocessing.get_context("fork").Pool(min(int(n_jobs), n_lev)) as pool:
                mats = pool.map(_level_gram, [phi for phi, _ in self.vh])
        K = np.sum(mats, axis=0)                                       # weisfeiler_lehman.py:270
        self.label_counts = [len(self.inv_labels[i]) for i in range(n_lev)]
        self.x_diag = np.diagonal(K).copy()
        return _normalize(K, self.x_diag, self.x_diag, True) if self.normalize else K

    def label_counts_only(self, X):
        """The relabel loop alone (weisfeiler_lehman.py:199-258), for inputs whose N x N matrices do not fit
        (config 5: the reference itself cannot run it): number of distinct labels per level."""
        eds, L = _wl_ingest(X)
        inv0 = {dv: i for i, dv in enumerate(sorted({l for d in L for l in d.values()}))}
        counts = [len(inv0)]
        L = [{k: inv0[v] for k, v in d.items()} for d in L]
        for i in range(1, self.n_iter + 1):
            creds = [{v: _credential(Lj, ed, v) for v in Lj} for Lj, ed in zip(L, eds)]
            inv = {c: j for j, c in enumerate(sorted({c for d in creds for c in d.values()}))}
            L = [{v: inv[c] for v, c in d.items()} for d in creds]
            counts.append(len(inv))
        return counts

    def transform(self, Y, keep_levels=False):
        eds, L = _wl_ingest(Y)
        inv0 = self.inv_labels[0]
        nl = len(inv0)                                                 # weisfeiler_lehman.py:417-418
        fresh = sorted({l for d in L for l in d.values() if l not in inv0})
        new0 = {dv: i for i, dv in enumerate(fresh, nl)}
        L = [{k: (inv0[v] if v in inv0 else new0[v]) for k, v in d.items()} for d in L]
        mats, ydiag = [], 0
        self.y_levels = []
        for i in range(self.n_iter + 1):
            if i > 0:                                                  # weisfeiler_lehman.py:435-476
                nl += len(self.inv_labels[i])
                creds = [{v: _credential(Lj, ed, v) for v in Lj} for Lj, ed in zip(L, eds)]
                inv = self.inv_labels[i]
                unseen = sorted({c for d in creds for c in d.values() if c not in inv})
                new = {c: nl + k for k, c in enumerate(unseen)}
                L = [{v: (inv[c] if c in inv else new[c]) for v, c in d.items()} for d in creds]
            if keep_levels:
                self.y_levels.append([dict(d) for d in L])
            phi_x, cols_x = self.vh[i]
            cols = dict(cols_x)
            phi_y = vh_features(L, cols)
            mats.append((phi_y[:, :phi_x.shape[1]] @ phi_x.T).toarray())
            ydiag = ydiag + np.asarray(phi_y.multiply(phi_y).sum(axis=1)).ravel()
        K = np.sum(mats, axis=0)
        self.y_diag = ydiag
        return _normalize(K, ydiag, self.x_diag, True) if self.normalize else K


class WLOAOracle(object):
    """WeisfeilerLehmanOptimalAssignment (grakel/kernels/weisfeiler_lehman_optimal_assignment.py).

    WL relabelling as above, but (a) from level 1 on only vertices WITH AN ENTRY in the edge
    dictionary are relabelled (`for v in Gs_ed[j].keys()`, :176), (b) every label remembers its
    parent label, (c) a graph's histogram counts, for each of its surviving vertices, the final
    label and all its ancestors (:201-206) and (d) K is the histogram INTERSECTION
    sum_l min(H_i[l], H_j[l]) (:268-279).
    """

    def __init__(self, n_iter=5, normalize=False):
        self.n_iter, self.normalize = n_iter, normalize

    @staticmethod
    def _hist(L, parent, width):
        H = np.zeros((len(L), width))
        for j, d in enumerate(L):
            for lab in d.values():
                while lab is not None:
                    H[j, lab] += 1
                    lab = parent[lab]
        return H

    @staticmethod
    def _intersection(A, B):
        K = np.zeros((A.shape[0], B.shape[0]))
        for i in range(A.shape[0]):
            K[i] = np.minimum(A[i][None, :], B).sum(axis=1)
        return K

    def fit_transform(self, X):
        eds, L = _wl_ingest(X)
        inv0 = {dv: i for i, dv in enumerate(sorted({l for d in L for l in d.values()}))}
        self.inv_labels = {0: inv0}
        self.parent = {i: None for i in range(len(inv0))}          # children of the root
        count = len(inv0)
        L = [{k: inv0[v] for k, v in d.items()} for d in L]
        for i in range(1, self.n_iter + 1):
            creds = [{v: _credential(Lj, ed, v) for v in ed.keys()} for Lj, ed in zip(L, eds)]
            pairs = sorted({(c, Lj[v]) for d, Lj in zip(creds, L) for v, c in d.items()}, key=lambda t: t[0])
            inv = dict()
            for c, prev in pairs:
                inv[c] = count
                self.parent[count] = prev
                count += 1
            L = [{v: inv[c] for v, c in d.items()} for d in creds]
            self.inv_labels[i] = inv
        self.H = self._hist(L, self.parent, count)
        K = self._intersection(self.H, self.H)
        self.x_diag = np.diagonal(K).copy()
        return _normalize(K, self.x_diag, self.x_diag, True) if self.normalize else K

    def transform(self, Y):
        eds, L = _wl_ingest(Y)
        parent = dict(self.parent)
        count = sum(len(self.inv_labels[i]) for i in range(len(self.inv_labels)))
        inv0 = self.inv_labels[0]
        new0 = dict()
        for dv in sorted({l for d in L for l in d.values() if l not in inv0}):
            new0[dv] = count
            parent[count] = None
            count += 1
        L = [{k: (inv0[v] if v in inv0 else new0[v]) for k, v in d.items()} for d in L]
        for i in range(1, self.n_iter + 1):
            creds = [{v: _credential(Lj, ed, v) for v in ed.keys()} for Lj, ed in zip(L, eds)]
            inv = self.inv_labels[i]
            unseen = sorted({(c, Lj[v]) for d, Lj in zip(creds, L) for v, c in d.items() if c not in inv},
                            key=lambda t: t[0])
            new = dict()
            for c, prev in unseen:
                new[c] = count
                parent[count] = prev
                count += 1
            L = [{v: (inv[c] if c in inv else new[c]) for v, c in d.items()} for d in creds]
        Hy = self._hist(L, parent, count)
        K = self._intersection(Hy[:, :self.H.shape[1]], self.H)
        self.y_diag = Hy.sum(axis=1)
        return _normalize(K, self.y_diag, self.x_diag, True) if self.normalize else K


# --------------------------------------------------------------------------
# Shortest paths (grakel/graph.py:588-687,1712-1794) and the SP kernel
# --------------------------------------------------------------------------
def floyd_warshall(A):
    """graph.py:1767-1794. Zero entries mean 'no edge'; weights are honoured."""
    n = A.shape[0]
    dist = np.array(A, copy=True).astype(float)
    dist[dist == 0] = INF
    np.fill_diagonal(dist, 0)
    for k in range(n):
        dist = np.minimum(dist, dist[:, k:k + 1] + dist[k:k + 1, :])
    return dist


def dijkstra_all(edges, vertices):
    """graph.py:660-671 + 1712-1764: one Dijkstra per source over the edge dict."""
    index = {v: i for i, v in enumerate(vertices)}
    n = len(vertices)
    S = np.full((n, n), INF)
    for src in vertices:
        done, heap, tick = dict(), [(0, 0, src)], 1
        while heap:
            d, _, v = heapq.heappop(heap)
            if v in done:
                continue
            done[v] = d
            for w, wt in edges[v].items():      # KeyError for {v: []} inputs: graph.py:1754
                if w not in done:
                    heapq.heappush(heap, (d + wt, tick, w))
                    tick += 1
        for v, d in done.items():
            S[index[src], index[v]] = d
    return S


def sp_matrix(g, algorithm_type="auto"):
    """Graph.build_shortest_path_matrix (graph.py:588-687) -> (S, index labels)."""
    if algorithm_type == "auto":
        algorithm_type = "floyd_warshall" if g.kind == "adjacency" else "dijkstra"
    n = len(g.vertices)
    if g.kind == "adjacency":
        A, lab = g.adjacency, g.labels
    else:
        index = {v: i for i, v in enumerate(g.vertices)}
        A = np.zeros((n, n))                                           # graph.py:1028-1036
        for a in g.edges:
            for b, w in g.edges[a].items():
                A[index[a], index[b]] = w
        lab = None if not g.labels else {i: g.labels[v] for i, v in enumerate(g.vertices)}
    if algorithm_type == "floyd_warshall":
        S = floyd_warshall(A)
    elif algorithm_type == "dijkstra":
        S = dijkstra_all(g.edges, g.vertices)
    else:
        raise ValueError('Unsupported "algorithm_type"')              # shortest_path.py:251-252
    return S, lab


class SPOracle(object):
    """ShortestPath.fit_transform / transform (shortest_path.py:264-318,370-499)."""

    def __init__(self, normalize=False, with_labels=True, algorithm_type="auto"):
        self.normalize, self.with_labels, self.algorithm_type = normalize, with_labels, algorithm_type

    def _counts(self, X, enum, frozen):
        ok = (lambda n: n in (2, 3)) if self.with_labels else (lambda n: n in (1, 2, 3))
        els = _elements(X, ok, 'a list with at least one and at most 3 elements')
        counts = []
        for x in els:
            g = parse_graph(x[0], x[1] if len(x) > 1 else {})
            S, lab = sp_matrix(g, self.algorithm_type)
            if self.with_labels and not lab:
                raise ValueError('Graph does not have any labels for vertices.')
            c = dict()
            n = S.shape[0]
            for u in range(n):                                         # shortest_path.py:469-490
                for v in range(n):
                    if u == v or S[u, v] == INF:
                        continue
                    key = (lab[u], lab[v], S[u, v]) if self.with_labels else S[u, v]
                    if frozen is not None and key in frozen:
                        idx = frozen[key]
                    else:
                        if key not in enum:
                            enum[key] = len(enum) + (len(frozen) if frozen is not None else 0)
                        idx = enum[key]
                    c[idx] = c.get(idx, 0) + 1
            counts.append(c)
        return counts

    @staticmethod
    def _dense(counts, width):
        phi = np.zeros((len(counts), width))
        for i, c in enumerate(counts):
            for j, v in c.items():
                phi[i, j] = v
        return phi

    def fit_transform(self, X):
        self.enum = dict()
        self.phi_x = self._dense(self._counts(X, self.enum, None), len(self.enum))
        K = np.dot(self.phi_x, self.phi_x.T)                           # shortest_path.py:404
        self.x_diag = np.diagonal(K).copy()
        return _normalize(K, self.x_diag, self.x_diag, False) if self.normalize else K

    def transform(self, Y):
        y_enum = dict()
        counts = self._counts(Y, y_enum, self.enum)
        phi_y = self._dense(counts, len(self.enum) + len(y_enum))
        K = np.dot(phi_y[:, :len(self.enum)], self.phi_x.T)            # shortest_path.py:312
        self.y_diag = np.sum(np.square(phi_y), axis=1)
        return _normalize(K, self.y_diag, self.x_diag, False) if self.normalize else K


class WLSPOracle(object):
    """WeisfeilerLehman(base_graph_kernel=ShortestPath): the framework fits one base kernel per
    level on ``(graph, level labels)`` and sums the matrices (weisfeiler_lehman.py:260-270,
    transform :478-494); normalisation happens once, on the sum (:323-328)."""

    def __init__(self, n_iter=5, normalize=False, with_labels=True):
        self.n_iter, self.normalize, self.with_labels = n_iter, normalize, with_labels

    def fit_transform(self, X):
        els = _elements(X, lambda n: n >= 2, 'a list with at least a graph and node labels')
        self.wl = WLOracle(self.n_iter)
        self.wl.fit_transform(X, keep_levels=True)
        self.sp, K = [], 0
        for lev in self.wl.levels:
            sp = SPOracle(with_labels=self.with_labels)
            K = K + sp.fit_transform([[x[0], d] for x, d in zip(els, lev)])
            self.sp.append(sp)
        self.x_diag = np.diagonal(K).copy()
        return _normalize(K, self.x_diag, self.x_diag, True) if self.normalize else K

    def transform(self, Y):
        els = _elements(Y, lambda n: n >= 2, 'a list with at least a graph and node labels')
        self.wl.transform(Y, keep_levels=True)
        K, self.y_diag = 0, 0
        for sp, lev in zip(self.sp, self.wl.y_levels):
            K = K + sp.transform([[y[0], d] for y, d in zip(els, lev)])
            self.y_diag = self.y_diag + sp.y_diag
        return _normalize(K, self.y_diag, self.x_diag, True) if self.normalize else K


# --------------------------------------------------------------------------
# Core framework (grakel/kernels/core_framework.py)
# --------------------------------------------------------------------------
def core_numbers(A):
    """k-core number of every vertex of the undirected graph with adjacency A
    (core_framework.py:376-416 computes the same numbers with the bin walk)."""
    n = A.shape[0]
    nbrs = [set(np.nonzero(A[v] > 0)[0].tolist()) - {v} for v in range(n)]
    deg = [len(x) for x in nbrs]
    core, alive, k = [0] * n, set(range(n)), 0
    while alive:
        peel = [v for v in alive if deg[v] <= k]
        if not peel:
            k += 1
            continue
        for v in peel:
            core[v] = k
            alive.discard(v)
            for u in nbrs[v]:
                if u in alive:
                    deg[u] -= 1
    return core


def _to_adjacency(x):
    g = parse_graph(x[0], x[1] if len(x) > 1 else {})
    if g.adjacency is not None:
        return np.array(g.adjacency, dtype=float), dict(g.labels)
    verts = list(g.vertices)
    pos = {v: i for i, v in enumerate(verts)}
    A = np.zeros((len(verts), len(verts)))
    for a, d in g.edges.items():
        for b, w in d.items():
            A[pos[a], pos[b]] = w
    return A, {pos[v]: l for v, l in g.labels.items()}


class CoreOracle(object):
    """CoreFramework: K = sum over core levels i of the base kernel on the subgraphs induced by
    the vertices of core number >= i; graphs without such vertices sit out of level i
    (core_framework.py:163-212).  ``make_base`` returns a fresh base oracle (fit_transform /
    transform / x_diag / y_diag)."""

    def __init__(self, make_base, normalize=False):
        self.make_base, self.normalize = make_base, normalize

    @staticmethod
    def _levels(X):
        els = _elements(X, lambda n: n >= 1, 'a list with at least a graph')
        graphs = [_to_adjacency(x) for x in els]
        cores = [core_numbers(A) for A, _ in graphs]
        return graphs, cores, max(max(c) for c in cores)

    @staticmethod
    def _subgraphs(graphs, cores, i):
        subs, idx = [], []
        for j, ((A, lab), c) in enumerate(zip(graphs, cores)):
            keep = [v for v in range(A.shape[0]) if c[v] >= i]
            if keep:
                subs.append([A[np.ix_(keep, keep)], {k: lab[v] for k, v in enumerate(keep)}])
                idx.append(j)
        return subs, np.array(idx, dtype=int)

    def fit_transform(self, X):
        graphs, cores, self.max_core = self._levels(X)
        n = len(graphs)
        K = np.zeros((n, n))
        self.base, self.fit_idx = dict(), dict()
        for i in range(self.max_core, -1, -1):
            subs, idx = self._subgraphs(graphs, cores, i)
            self.fit_idx[i] = idx
            if len(idx):
                self.base[i] = self.make_base()
                K[np.ix_(idx, idx)] += self.base[i].fit_transform(subs)
        self.x_diag = np.diagonal(K).copy()
        return _normalize(K, self.x_diag, self.x_diag, True) if self.normalize else K

    def transform(self, Y):
        graphs, cores, t_max = self._levels(Y)
        K = np.zeros((len(graphs), len(self.x_diag)))
        self.y_diag = np.zeros(len(graphs))
        for i in range(t_max, -1, -1):
            subs, idx = self._subgraphs(graphs, cores, i)
            if not len(idx):
                continue
            if self.max_core < i or not len(self.fit_idx[i]):
                dummy = self.make_base()                              # :196-200: diagonal only
                dummy.fit_transform(subs)
                self.y_diag[idx] += dummy.x_diag
            else:
                K[np.ix_(idx, self.fit_idx[i])] += self.base[i].transform(subs)
                self.y_diag[idx] += self.base[i].y_diag
        return _normalize(K, self.y_diag, self.x_diag, True) if self.normalize else K


# --------------------------------------------------------------------------
# Partition helper used by the parity tests ("bit-exact integer WL labels" ==
# same partition of the nodes per level)
# --------------------------------------------------------------------------
def canonical_partition(labels):
    """Map a label sequence to first-occurrence ids so two labelings can be compared."""
    seen, out = dict(), np.empty(len(labels), dtype=np.int64)
    for i, l in enumerate(labels):
        out[i] = seen.setdefault(l, len(seen))
    return out

"""Test-local drivers that call a base kernel class exactly as the reference's FRAMEWORKS do (SURVEY 8b "Callers").

TEST INFRASTRUCTURE: restatements of the *calling sequence* of ``grakel/kernels/hadamard_code.py:130-260`` and
``grakel/kernels/core_framework.py:119-234`` (which elements a base kernel is handed per level, which of
``fit`` / ``fit_transform`` / ``transform`` / ``diagonal`` is called, how the level matrices are combined), so that the GPU
box -- which has no reference -- can drive the accelerated classes through the protocol.  ``tests/test_host.py`` checks on the
CPU box, where the real grakel is importable, that the REAL frameworks hand the accelerated classes the same batches as
these drivers do.  Citations are into /root/reference.
"""
from math import ceil, log2

import numpy as np
from scipy.linalg import hadamard

from oracle import grakel_oracle as O


class GraphLike(object):
    """What a base kernel sees of a ``grakel.Graph`` in dictionary format (graph.py:1182 ``get_edge_dictionary``, :689
    ``get_labels``, :277 ``desired_format``): CoreFramework hands its base kernel such objects (core_framework.py:176-191)."""
    _format = "dictionary"

    def __init__(self, edges, labels):
        self._edges, self._labels = edges, labels

    def get_edge_dictionary(self):
        return self._edges

    def get_labels(self, purpose="dictionary", label_type="vertex", return_none=False):
        return self._labels if label_type == "vertex" else (None if return_none else {})

    def desired_format(self, graph_format, warn=False):
        assert graph_format in ("dictionary", "auto", "all")

    def get_vertices(self, purpose="dictionary"):
        return set(self._edges.keys())


def _edge_dict(x):
    """element -> ({u: {v: w}} over ALL vertices, {u: label}) -- the reference's Graph in dictionary format"""
    g = O.parse_graph(x[0], x[1] if len(x) > 1 else {})
    if g.adjacency is not None:
        A = np.asarray(g.adjacency)
        n = A.shape[0]
        return {u: {int(v): A[u, v] for v in np.nonzero(A[u])[0]} for u in range(n)}, dict(g.labels)
    ed = {v: dict() for v in g.vertices}
    for a, d in g.edges.items():
        ed[a].update(d)
    return ed, dict(g.labels)


# ------------------------------------------------------------------------------------------------
# HadamardCode (hadamard_code.py:130-260)
# ------------------------------------------------------------------------------------------------
class HadamardCaller(object):
    """``HadamardCode(n_iter, base_graph_kernel=base_cls, normalize)`` as a sequence of calls on ``make_base()`` objects."""

    def __init__(self, make_base, n_iter, normalize=False):
        self.make_base, self.n_iter, self.normalize = make_base, n_iter, normalize

    def _levels(self, X, fit):
        inp, neighbors, labels = [], [], []
        if fit:
            self.enum = dict()
        enum = self.enum if fit else dict(self.enum)
        for x in X:
            ed, lab = _edge_dict(x)
            inp.append(x[0]), neighbors.append(ed), labels.append(lab)
            for v in set(lab.values()):                                  # :174-177 first-seen enumeration of the labels
                if v not in enum:
                    enum[v] = len(enum)
        H = hadamard(int(2 ** (ceil(log2(len(enum))))))                  # :183
        cur = [{k: H[enum[v], :] for k, v in lab.items()} for lab in labels]
        yield [(obj, {k: tuple(c) for k, c in code.items()}) for obj, code in zip(inp, cur)]      # :189-199
        for _ in range(1, self.n_iter):                                  # :201-214
            nxt = []
            for nb, old in zip(neighbors, cur):
                new = dict()
                for k, ns in nb.items():
                    new[k] = old[k]
                    for q in ns:
                        new[k] = np.add(new[k], old[q])
                nxt.append(new)
            cur = nxt
            yield [(obj, {k: tuple(c) for k, c in code.items()}) for obj, code in zip(inp, cur)]

    def level_inputs(self, X, fit=True):
        return list(self._levels(X, fit))

    def fit_transform(self, X):
        self.base = {i: self.make_base() for i in range(self.n_iter)}    # :216-217
        K = np.sum([self.base[i].fit_transform(g) for i, g in enumerate(self._levels(X, True))], axis=0)   # :224-229
        self.x_diag = np.diagonal(K).copy()
        return np.nan_to_num(K / np.sqrt(np.outer(self.x_diag, self.x_diag))) if self.normalize else K

    def transform(self, Y):
        K = np.sum([self.base[i].transform(g) for i, g in enumerate(self._levels(Y, False))], axis=0)      # :233-234
        if not self.normalize:
            return K
        xd = yd = 0                                                      # diagonal(): :336-389 -- every base kernel's diagonal()
        for i in range(self.n_iter):
            x, y = self.base[i].diagonal()
            xd, yd = xd + x, yd + y
        return np.nan_to_num(K / np.sqrt(np.outer(yd, xd)))


# ------------------------------------------------------------------------------------------------
# CoreFramework (core_framework.py:119-234)
# ------------------------------------------------------------------------------------------------
class CoreCaller(object):
    """``CoreFramework(base_graph_kernel=base_cls, normalize)`` as a sequence of calls on ``make_base()`` objects; the
    subgraphs are handed over as graph OBJECTS in dictionary format (core_framework.py:40,153,176-191)."""

    def __init__(self, make_base, normalize=False):
        self.make_base, self.normalize = make_base, normalize

    @staticmethod
    def _parse(X):
        graphs, cores = [], []
        for x in X:
            ed, lab = _edge_dict(x)
            verts = sorted(ed)
            pos = {v: i for i, v in enumerate(verts)}
            A = np.zeros((len(verts), len(verts)))
            for a, d in ed.items():
                for b in d:
                    A[pos[a], pos[b]] = 1
            c = O.core_numbers(A)                                         # :157 core_number(x)
            graphs.append((ed, lab)), cores.append({v: c[pos[v]] for v in verts})
        return graphs, cores, max(max(c.values()) for c in cores)

    @staticmethod
    def _subgraphs(graphs, cores, i):
        subs, idx = [], []
        for j, ((ed, lab), cn) in enumerate(zip(graphs, cores)):
            keep = {k for k, v in cn.items() if v >= i}                   # :176
            if keep:                                                      # :177-191 g.get_subgraph(vertices)
                subs.append(GraphLike({v: {w: wt for w, wt in ed[v].items() if w in keep} for v in keep},
                                      {v: lab[v] for v in keep}))
                idx.append(j)
        return subs, np.array(idx, dtype=int)

    def level_inputs(self, X):
        graphs, cores, mx = self._parse(X)
        return {i: self._subgraphs(graphs, cores, i) for i in range(mx, -1, -1)}

    def fit_transform(self, X):
        graphs, cores, self.max_core = self._parse(X)
        K = np.zeros((len(graphs), len(graphs)))
        self.base, self.fit_idx = dict(), dict()
        for i in range(self.max_core, -1, -1):                            # :173
            subs, idx = self._subgraphs(graphs, cores, i)
            self.fit_idx[i] = idx
            if len(idx):                                                  # :203-208
                self.base[i] = self.make_base()
                M = self.base[i].fit_transform(subs)
                for j in range(len(idx)):
                    K[idx[j], idx] += M[j, :]
        self.x_diag = np.diagonal(K).copy()
        return np.nan_to_num(K / np.sqrt(np.outer(self.x_diag, self.x_diag))) if self.normalize else K

    def transform(self, Y):
        graphs, cores, t_max = self._parse(Y)
        K = np.zeros((len(graphs), len(self.x_diag)))
        yd = np.zeros(len(graphs))
        for i in range(t_max, -1, -1):
            subs, idx = self._subgraphs(graphs, cores, i)
            if not len(idx):
                continue
            if self.max_core < i or not len(self.fit_idx[i]):             # :209-214 a dummy kernel for the diagonal
                dummy = self.make_base()
                dummy.fit(subs)
                yd[idx] += dummy.diagonal()
            else:                                                         # :215-219
                M = self.base[i].transform(subs)
                for j in range(len(idx)):
                    K[idx[j], self.fit_idx[i]] += M[j, :]
                yd[idx] += self.base[i].diagonal()[1]                     # :339-357 diagonal() after a transform: (X_diag, Y_diag)
        self.y_diag = yd
        return np.nan_to_num(K / np.sqrt(np.outer(yd, self.x_diag))) if self.normalize else K

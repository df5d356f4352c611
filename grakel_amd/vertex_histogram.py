"""VertexHistogram kernel on MI355X (drop-in for ``grakel.VertexHistogram``,
``grakel/kernels/vertex_histogram.py:23``)."""
import numpy as np
from sklearn.utils.validation import check_is_fitted

from .batch import vh_batch_from_input
from .kernel import Kernel, NORM_NONE, NORM_PLAIN


def count_matrix(node_graph, node_col, n_graphs, n_cols):
    """scipy CSR [n_graphs x n_cols] of how often column id ``node_col[v]`` occurs in graph ``node_graph[v]``
    (vertex_histogram.py:118-137: the reference's fitted ``X``)."""
    from scipy.sparse import csr_matrix
    keep = node_col >= 0
    key = node_graph[keep].astype(np.int64) * max(int(n_cols), 1) + node_col[keep]
    uniq, cnt = np.unique(key, return_counts=True)
    rows, cols = uniq // max(int(n_cols), 1), uniq % max(int(n_cols), 1)
    return csr_matrix((cnt.astype(np.float64), (rows, cols)), shape=(int(n_graphs), int(n_cols)))


def first_seen_columns(ids):
    """{id: column} in first-seen order over the nodes (graph after graph, vertex after vertex): the
    reference's ``_labels`` enumeration (vertex_histogram.py:109-116)."""
    uniq, first = np.unique(ids, return_index=True)
    order = np.argsort(first, kind="stable")
    return {int(uniq[i]): c for c, i in enumerate(order)}


class FittedFeatures(object):
    """The reference's fitted ``X`` (the N x D label-count matrix, vertex_histogram.py:118-137) without its
    cost: the device works on its own column-compacted operand, so the host matrix is only built when
    somebody reads it.  ``shape`` is free; any other attribute (``toarray``, ``dot``, ``nnz``, indexing ...)
    materialises a ``scipy.sparse.csr_matrix`` once and forwards to it."""

    def __init__(self, n_graphs, n_labels, build=None):
        self.shape = (n_graphs, n_labels)
        self._build, self._m = build, None

    def materialize(self):
        if self._m is None:
            if self._build is None:
                raise AttributeError("this fitted feature matrix was not kept (no builder)")
            self._m = self._build()
            self.shape = self._m.shape
        return self._m

    def __getattr__(self, name):
        if name.startswith("__") or name in ("_build", "_m", "shape"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __array__(self, dtype=None, copy=None):
        a = self.materialize().toarray()
        return a if dtype is None else a.astype(dtype)

    def __getstate__(self):
        return dict(shape=self.shape, _build=None, _m=self._m)

    def __setstate__(self, st):
        self.__dict__.update(st)

    def __repr__(self):
        return "FittedFeatures(shape=%r, %s)" % (self.shape, "lazy" if self._m is None else "materialized")


class VertexHistogram(Kernel):
    """K[i,j] = <label histogram of G_i, label histogram of G_j>.

    Parameters as the reference (vertex_histogram.py:26-41): n_jobs, normalize, verbose,
    sparse ('auto' | bool; only affects host storage in the reference, accepted and ignored).
    """

    def __init__(self, n_jobs=None, normalize=False, verbose=False, sparse='auto'):
        super(VertexHistogram, self).__init__(n_jobs=n_jobs, normalize=normalize, verbose=verbose)
        self.sparse = sparse
        self._initialized.update({'sparse': True})

    def _ingest(self, X, fitted):
        return vh_batch_from_input(X, fitted)

    def _prepare(self, engine, dbatch):
        engine.wl_relabel(dbatch, 0)        # level 0 only: group nodes by label
        return dbatch, 1

    def fit(self, X, y=None):
        """kernel.py:86-121."""
        self._is_transformed = False
        self._method_calling = 1
        self.initialize()
        if X is None:
            raise ValueError('`fit` input cannot be None')
        self._fit_host(X)
        gb = self._fit_batch
        cols = first_seen_columns(gb.node_label)                    # vertex_histogram.py:109-116
        if self._label_map is not None:
            inv = {i: k for k, i in self._label_map.items()}
            self._labels = {inv[i]: c for i, c in cols.items()}
        else:                                                       # packed input: the ids are the labels
            self._labels = dict(cols)
        self.sparse_ = bool(self.sparse) if self.sparse != 'auto' else True

        def build(gb=gb, cols=cols):
            lut = np.full(gb.n_labels, -1, np.int64)
            for i, c in cols.items():
                lut[i] = c
            node_graph = np.repeat(np.arange(gb.n_graphs), np.diff(gb.graph_ptr))
            return count_matrix(node_graph, lut[gb.node_label], gb.n_graphs, len(cols))

        self.X = FittedFeatures(self._nx, len(cols), build)
        return self

    def fit_transform(self, X, y=None):
        """kernel.py:167-204."""
        self._method_calling = 2
        self.fit(X)
        eng, feat = self._gram_fit()
        if self.normalize:
            self._warn_unnormalizable(self._X_diag)
        return eng.gram(feat, NORM_PLAIN if self.normalize else NORM_NONE)

    def transform(self, X):
        """kernel.py:123-165; output [n_targets, n_fitted]; unseen labels are dropped
        (vertex_histogram.py:179) but count in the targets' own diagonal."""
        self._method_calling = 3
        check_is_fitted(self, ['X'])
        if X is None:
            raise ValueError('`transform` input cannot be None')
        eng, feat = self._gram_transform(X)
        self._is_transformed = True
        if self.normalize:
            self._warn_unnormalizable(self._X_diag, self._Y_diag)
        return eng.gram(feat, NORM_PLAIN if self.normalize else NORM_NONE)


class EdgeHistogram(VertexHistogram):
    """K[i,j] = <edge-label histogram of G_i, edge-label histogram of G_j> (drop-in for
    ``grakel.EdgeHistogram``, ``grakel/kernels/edge_histogram.py:23``): the same
    label-count features and Gram path as VertexHistogram, fed with the values of the edge-label
    dictionary ``x[2]`` (inputs must have three elements, edge_histogram.py:88-96)."""

    def initialize(self):
        """edge_histogram.py:46-55."""
        if not self._initialized["n_jobs"]:
            if self.n_jobs is not None:
                import warnings
                warnings.warn('no implemented parallelization for EdgeHistogram')
            self._parallel = None
            self._initialized["n_jobs"] = True

    def _ingest(self, X, fitted):
        return vh_batch_from_input(X, fitted, edge_labels=True)

"""Shortest-path kernel on MI355X (drop-in for ``grakel.ShortestPath``,
``grakel/kernels/shortest_path.py:167``)."""
import warnings

import numpy as np
from sklearn.utils.validation import check_is_fitted

from .batch import sp_batch_from_input
from .kernel import Kernel, NORM_NONE, NORM_PLAIN


class _EnumStub(object):
    """Stand-in for the reference's ``_enum`` ((l_u,l_v,d) -> column) dictionary: the
    dictionary lives on the device; the host keeps its size."""

    def __init__(self, n):
        self._n = int(n)

    def __len__(self):
        return self._n

    def __repr__(self):
        return "<%d shortest-path features on device>" % self._n


class ShortestPath(Kernel):
    """K[i,j] = <histogram of (label_u, label_v, d(u,v)) over ordered pairs of G_i, same of G_j>.

    Parameters as the reference (shortest_path.py:224-228): n_jobs, normalize, verbose,
    with_labels=True, algorithm_type in {"auto", "dijkstra", "floyd_warshall"} (validated; the
    device always runs the batched Floyd-Warshall / row-relaxation kernels, which give the
    same distances as either host algorithm for positive integer weights).
    """

    def __init__(self, n_jobs=None, normalize=False, verbose=False, with_labels=True,
                 algorithm_type="auto"):
        super(ShortestPath, self).__init__(n_jobs=n_jobs, normalize=normalize, verbose=verbose)
        self.with_labels = with_labels
        self.algorithm_type = algorithm_type
        self._initialized.update({"with_labels": False, "algorithm_type": False})

    def initialize(self):
        """shortest_path.py:237-262."""
        if not self._initialized["n_jobs"]:
            if self.n_jobs is not None:
                warnings.warn('no implemented parallelization for ShortestPath')
            self._initialized["n_jobs"] = True
        if not self._initialized["algorithm_type"]:
            if self.algorithm_type not in ("auto", "floyd_warshall", "dijkstra"):
                raise ValueError('Unsupported "algorithm_type"')
            self._initialized["algorithm_type"] = True
        if not self._initialized["with_labels"]:
            self._lt = "vertex" if self.with_labels else "none"
            self._initialized["with_labels"] = True

    def _ingest(self, X, fitted):
        return sp_batch_from_input(X, bool(self.with_labels), fitted)

    def _prepare(self, engine, dbatch):
        gb = self._cur_batch
        w = gb.edge_weight
        if w is not None and (w.size == 0 or np.all(w == 1)):
            w = None
        pb = engine.sp_build(dbatch, w, bool(self.with_labels))
        pb._parent = dbatch
        return pb, 1

    # the base flows upload ``_fit_batch`` / the union; remember which one for edge weights
    def _gram_fit(self):
        self._cur_batch = self._fit_batch
        return super(ShortestPath, self)._gram_fit()

    def _gram_transform(self, Y):
        from .batch import GraphBatch
        ybatch, _ = self._ingest(Y, self._label_map if self._label_map is not None else {})
        self._ny = ybatch.n_graphs
        self._cur_batch = GraphBatch.concat(self._fit_batch, ybatch)
        eng = self._engine()
        db = eng.upload(self._cur_batch)
        fb, n_levels = self._prepare(eng, db)
        feat = eng.features(fb, n_levels, n_fit=self._nx)
        selfk = eng.selfk(feat)
        self._X_diag = selfk[:self._nx]
        self._Y_diag = selfk[self._nx:]
        return eng, feat

    def fit(self, X, y=None):
        """kernel.py:86-121."""
        self._is_transformed = False
        self._method_calling = 1
        self.initialize()
        if X is None:
            raise ValueError('`fit` input cannot be None')
        self._fit_host(X)
        self.X = {i: None for i in range(self._nx)}
        self._enum = _EnumStub(0)
        return self

    def fit_transform(self, X, y=None):
        """shortest_path.py:370-410 (normalisation divides silently, :407-408)."""
        self._method_calling = 2
        self.fit(X)
        eng, feat = self._gram_fit()
        self._enum = _EnumStub(self._last_info["label_counts"][0])
        with np.errstate(divide='ignore', invalid='ignore'):
            return eng.gram(feat, NORM_PLAIN if self.normalize else NORM_NONE)

    def transform(self, X):
        """shortest_path.py:264-318."""
        self._method_calling = 3
        check_is_fitted(self, ['X', '_nx', '_enum'])
        if X is None:
            raise ValueError('transform input cannot be None')
        eng, feat = self._gram_transform(X)
        self._is_transformed = True
        return eng.gram(feat, NORM_PLAIN if self.normalize else NORM_NONE)

"""VertexHistogram kernel on MI355X (drop-in for ``grakel.VertexHistogram``,
``grakel/kernels/vertex_histogram.py:23``)."""
import numpy as np
from sklearn.utils.validation import check_is_fitted

from .batch import vh_batch_from_input
from .kernel import Kernel, NORM_NONE, NORM_PLAIN


class FittedFeatures(object):
    """Stand-in for the reference's fitted ``X`` (the N x D label-count matrix): the matrix
    itself lives column-compacted in HBM; the host keeps its shape only."""

    def __init__(self, n_graphs, n_labels):
        self.shape = (n_graphs, n_labels)

    def __repr__(self):
        return "FittedFeatures(shape=%r, on_device)" % (self.shape,)


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
        if self._label_map is not None:
            # first-seen column order like vertex_histogram.py:109-116
            ids = self._fit_batch.node_label
            _, first = np.unique(ids, return_index=True)
            order = np.argsort(first, kind="stable")
            inv = {i: k for k, i in self._label_map.items()}
            self._labels = {inv[int(i)]: c for c, i in enumerate(order)}
        self.sparse_ = bool(self.sparse) if self.sparse != 'auto' else True
        self.X = FittedFeatures(self._nx, self._fit_batch.n_labels)
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

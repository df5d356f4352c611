"""Weisfeiler-Lehman optimal-assignment kernel on MI355X (drop-in for
``grakel.WeisfeilerLehmanOptimalAssignment``,
``grakel/kernels/weisfeiler_lehman_optimal_assignment.py:19``).

K[i,j] = sum over the WL label hierarchy of min(H_i[l], H_j[l]) where H_g[l] counts the vertices
of g whose level-i label is l, i = 0..n_iter (:201-206 walks every final label up to the root, so
each vertex adds one to its label of every level; :268-279 is the histogram intersection).  The
device path is the WL relabel of ``grakel_amd.weisfeiler_lehman`` plus the same integer Gram
kernel fed with unary-expanded counts (``GK_FEAT_MINSUM``, include/gk_hip.h)."""
from sklearn.utils.validation import check_is_fitted

from .batch import wloa_batch_from_input
from .kernel import Kernel, NORM_NONE, NORM_NAN_TO_NUM
from .vertex_histogram import FittedFeatures


class WeisfeilerLehmanOptimalAssignment(Kernel):
    """Parameters as the reference (:53-61): n_jobs, verbose, normalize, n_iter=5, sparse=False
    (``sparse`` picks the host container of the histograms in the reference; they stay in HBM
    here, so it is accepted and has no effect)."""

    _graph_format = "dictionary"
    _norm_mode = NORM_NAN_TO_NUM
    _feature_kind = 1          # GK_FEAT_MINSUM

    def __init__(self, n_jobs=None, verbose=False, normalize=False, n_iter=5, sparse=False):
        super(WeisfeilerLehmanOptimalAssignment, self).__init__(
            n_jobs=n_jobs, verbose=verbose, normalize=normalize)
        self.n_iter = n_iter
        self.sparse = sparse
        self._initialized.update({"n_iter": False, "sparse": True})

    def initialize(self):
        """:63-75."""
        super(WeisfeilerLehmanOptimalAssignment, self).initialize()
        if not self._initialized["n_iter"]:
            if type(self.n_iter) is not int or self.n_iter <= 0:
                raise TypeError("'n_iter' must be a positive integer")
            self._n_iter = self.n_iter + 1
            self._initialized["n_iter"] = True
        if not self._initialized["sparse"]:
            self._initialized["sparse"] = True

    def _ingest(self, X, fitted):
        return wloa_batch_from_input(X, fitted, fit=fitted is None)

    def _prepare(self, engine, dbatch):
        engine.wl_relabel(dbatch, self._n_iter - 1)
        return dbatch, self._n_iter

    def _after_fit(self):
        self._inv_labels = {0: dict(self._label_map) if self._label_map is not None else {}}
        self._hierarchy = None                      # lives on the device as the per-level labels
        self.X = FittedFeatures(self._nx, None)

    def fit(self, X, y=None):
        """kernel.py:86-121 with :77-234 as parse_input."""
        self._is_transformed = False
        self._method_calling = 1
        self.initialize()
        if X is None:
            raise ValueError('`fit` input cannot be None')
        self._fit_host(X)
        self._after_fit()
        return self

    def fit_transform(self, X, y=None):
        """:236-283."""
        self._method_calling = 2
        self._is_transformed = False
        self.initialize()
        if X is None:
            raise ValueError('transform input cannot be None')
        self._fit_host(X)
        self._after_fit()
        if self._n_iter <= 48:                      # one library call (kernel.Kernel._fit_transform_fused)
            K = self._fit_transform_fused(self._n_iter - 1, NORM_NAN_TO_NUM if self.normalize else NORM_NONE)
            self.X = FittedFeatures(self._nx, sum(self._last_info["label_counts"]))
            return K
        eng, feat = self._gram_fit()
        self.X = FittedFeatures(self._nx, sum(self._last_info["label_counts"]))
        return eng.gram(feat, NORM_NAN_TO_NUM if self.normalize else NORM_NONE)

    def transform(self, X):
        """:285-416 (dense branch).  Targets are relabelled jointly with the fitted graphs; labels
        no fitted graph carries have an all-zero fitted column, so they add nothing to
        K[targets, fitted] but do count in the targets' own diagonal -- the reference's
        ``Hs[i, :self.X.shape[1]]`` cut and its ``diagonal()`` over the uncut ``self.Y``."""
        self._method_calling = 3
        check_is_fitted(self, ['X', '_nx', '_inv_labels'])
        if X is None:
            raise ValueError('transform input cannot be None')
        eng, feat = self._gram_transform(X)
        self._is_transformed = True
        return eng.gram(feat, NORM_NAN_TO_NUM if self.normalize else NORM_NONE)

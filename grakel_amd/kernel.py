"""``Kernel`` base class: the drop-in boundary (mirrors ``grakel/kernels/kernel.py:23``).

Same constructor parameters, lazy ``initialize`` / ``set_params`` protocol
(``kernel.py:386-433``), ``fit`` / ``transform`` / ``fit_transform`` / ``diagonal``
contract and error behaviour as the reference, for the three kernels on the hot path.
The Gram matrices themselves come from libgk_hip.so through ``grakel_amd.engine``.
"""
import copy
import warnings

import numpy as np
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.exceptions import NotFittedError
from sklearn.utils.validation import check_is_fitted

from .batch import GraphBatch
from .engine import get_engine

# normalisation modes of gk_gram (include/gk_hip.h)
NORM_NONE, NORM_PLAIN, NORM_NAN_TO_NUM = 0, 1, 2

_DEVICE_ATTRS = ("_dev_fit", "_dev_last", "_dev_wlfit")


class Kernel(BaseEstimator, TransformerMixin):
    """Base of the MI355X-backed kernels.

    Parameters (``kernel.py:27-38``): n_jobs (accepted and validated, the GPU path has no
    use for joblib), normalize, verbose.
    """

    X = None
    _graph_format = "dictionary"
    _method_calling = 0
    _norm_mode = NORM_PLAIN
    _feature_kind = 0          # GK_FEAT_DOT; WL-OA overrides with GK_FEAT_MINSUM

    def __init__(self, n_jobs=None, normalize=False, verbose=False):
        self.verbose = verbose
        self.n_jobs = n_jobs
        self.normalize = normalize
        self._initialized = dict(n_jobs=False)

    # -- sklearn plumbing ------------------------------------------------------------------
    def initialize(self):
        """kernel.py:386-399."""
        if not self._initialized["n_jobs"]:
            if type(self.n_jobs) is not int and self.n_jobs is not None:
                raise ValueError('n_jobs parameter must be an int indicating the number of '
                                 'jobs as in joblib or None')
            self._parallel = None
            self._initialized["n_jobs"] = True

    def set_params(self, **params):
        """kernel.py:417-433: changing a parameter re-arms its lazy initialisation."""
        if len(self._initialized):
            params = copy.deepcopy(params)
            for key in params:
                key, delim, sub_key = key.partition('__')
                if delim:
                    if sub_key in self._initialized:
                        self._initialized[sub_key] = False
                elif key in self._initialized:
                    self._initialized[key] = False
        return super(Kernel, self).set_params(**params)

    def __getstate__(self):
        state = super(Kernel, self).__getstate__() if hasattr(BaseEstimator, "__getstate__") \
            else dict(self.__dict__)
        state = dict(state)
        for k in _DEVICE_ATTRS:       # device handles never leave the process
            state.pop(k, None)
        return state

    # -- device helpers ----------------------------------------------------------------------
    def _engine(self):
        return get_engine()

    def _drop_device_state(self):
        for k in _DEVICE_ATTRS:
            if k in self.__dict__:
                del self.__dict__[k]

    # hooks -------------------------------------------------------------------------------------
    def _ingest(self, X, fitted):
        raise NotImplementedError

    def _prepare(self, engine, dbatch):
        """Device preparation of an uploaded batch -> (batch to featurise, n_levels)."""
        raise NotImplementedError

    # -- shared flows ----------------------------------------------------------------------------
    def _fit_host(self, X):
        batch, mapping = self._ingest(X, None)
        self._fit_batch = batch
        self._label_map = mapping
        self._nx = batch.n_graphs
        self._drop_device_state()
        for attr in ("_X_diag", "_Y_diag"):
            if hasattr(self, attr):
                delattr(self, attr)
        self._is_transformed = False

    def _fitted_on_device(self, eng):
        """The fitted batch in HBM, uploaded once and kept between ``fit_transform`` / ``transform`` /
        ``diagonal`` calls (never pickled).  Relabelling adds level arrays to it, the CSR itself is immutable."""
        d = self.__dict__.get("_dev_fit")
        if d is None or d.handle is None or d.engine is not eng or d.engine.handle is None:
            d = self._dev_fit = eng.upload(self._fit_batch)
        return d

    def _union_on_device(self, eng, ybatch):
        """Fitted graphs + targets as one device batch; only the targets are uploaded."""
        yb = eng.upload(ybatch)
        try:
            return eng.concat(self._fitted_on_device(eng), yb, max(self._fit_batch.n_labels, ybatch.n_labels))
        finally:
            yb.close()

    def _gram_fit(self):
        eng = self._engine()
        db = self._fitted_on_device(eng)
        fb, n_levels = self._prepare(eng, db)
        feat = eng.features(fb, n_levels, kind=self._feature_kind)
        self._X_diag = eng.selfk(feat)
        self._last_info = dict(n_cols=feat.n_cols, n_cols_low=feat.n_cols_low, nnz=feat.nnz, max_count=feat.max_count,
                               dtype=feat.operand, label_counts=fb.label_counts)
        return eng, feat

    def _fit_transform_fused(self, n_iter, norm):
        """fit_transform of a WL-type kernel in one library call (``gk_wl_fit_transform``: the relabel is queued without a
        host round trip, the feature builder runs behind it on device-side counts).  Sets what ``_gram_fit`` sets; returns
        the host matrix."""
        eng = self._engine()
        db = self._fitted_on_device(eng)
        feat, K = eng.wl_fit_transform(db, n_iter, kind=self._feature_kind, normalize=norm)
        self._X_diag = eng.selfk(feat)
        self._last_info = dict(n_cols=feat.n_cols, n_cols_low=feat.n_cols_low, nnz=feat.nnz, max_count=feat.max_count,
                               dtype=feat.operand, label_counts=db.label_counts)
        feat.close()
        return K

    def _gram_transform(self, Y):
        ybatch, _ = self._ingest(Y, self._label_map if self._label_map is not None else {})
        self._ny = ybatch.n_graphs
        eng = self._engine()
        db = self._union_on_device(eng, ybatch)
        fb, n_levels = self._prepare(eng, db)
        feat = eng.features(fb, n_levels, n_fit=self._nx, kind=self._feature_kind)
        selfk = eng.selfk(feat)
        self._X_diag = selfk[:self._nx]
        self._Y_diag = selfk[self._nx:]
        return eng, feat

    def diagonal(self):
        """kernel.py:298-336 contract: X_diag, or (X_diag, Y_diag) once transformed."""
        check_is_fitted(self, ['X'])
        if not hasattr(self, "_X_diag"):
            eng, feat = self._gram_fit()
            feat.close()
        if getattr(self, "_is_transformed", False):
            return self._X_diag, self._Y_diag
        return self._X_diag

    def _warn_unnormalizable(self, *diagonals):
        """kernel.py:206-234 (the zero-diagonal branch is the only reachable one here)."""
        for d in diagonals:
            if np.any(np.asarray(d) == 0):
                warnings.warn(
                    type(self).__name__ + ' has zero self similarities, so normalizing it '
                    'yields NaNs: those graphs have no features this kernel can see. Either '
                    'drop them or pass normalize=False.', RuntimeWarning)
                break


__all__ = ["Kernel", "NotFittedError", "check_is_fitted"]


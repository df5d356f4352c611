"""Shortest-path kernel on MI355X (drop-in for ``grakel.ShortestPath``,
``grakel/kernels/shortest_path.py:167``)."""
import warnings

import numpy as np
from sklearn.utils.validation import check_is_fitted

from .batch import sp_batch_from_input
from .kernel import Kernel, NORM_NONE, NORM_PLAIN


class _LazyDict(dict):
    """A dictionary of the reference's fitted state (``_enum``, ``X``, ``_Y_enum``) that is rebuilt from the
    device's all-pairs distances the first time it is read; ``len`` of ``_enum`` is known without that."""

    def __init__(self, fill, known_len=None):
        dict.__init__(self)
        self._fill, self._known_len = fill, known_len

    def _ensure(self):
        f = self._fill
        if f is not None:
            filled = f()                     # may raise (no GPU, out of memory): the next read tries again
            self._fill = None
            dict.update(self, filled)

    def get(self, key, default=None):
        self._ensure()
        return dict.get(self, key, default)

    def copy(self):
        self._ensure()
        return dict(dict.items(self))

    def __eq__(self, other):
        self._ensure()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        self._ensure()
        return dict.__repr__(self)

    def __len__(self):
        if self._fill is not None and self._known_len is not None:
            return self._known_len
        self._ensure()
        return dict.__len__(self)

    def __missing__(self, key):
        if self._fill is None:
            raise KeyError(key)
        self._ensure()
        return dict.__getitem__(self, key)

    def __iter__(self):
        self._ensure()
        return dict.__iter__(self)

    def __contains__(self, key):
        self._ensure()
        return dict.__contains__(self, key)

    def keys(self):
        self._ensure()
        return dict.keys(self)

    def items(self):
        self._ensure()
        return dict.items(self)

    def values(self):
        self._ensure()
        return dict.values(self)

    def get(self, key, default=None):
        self._ensure()
        return dict.get(self, key, default)

    def __reduce__(self):          # pickles as what has been rebuilt so far (ShortestPath.__setstate__ re-arms it)
        return (dict, (dict(dict.items(self)),))


def sp_weight_args(gb, algorithm_type):
    """What ``Engine.sp_build`` needs for the edge weights of ``gb``: (int32 weights or None, float64 weights or None,
    per-graph algorithm flags or None).  General float weights (``GraphBatch.float_weight``): the device reproduces the
    reference's float distances, and which of its two algorithms it runs matters then -- "auto" follows the element's
    format (graph.py:652-656: dictionary -> dijkstra, adjacency -> floyd_warshall), a named algorithm converts every
    element (shortest_path.py:244-250).  Integer / power-of-two-multiple weights: both algorithms give the same exact
    distances."""
    fw = getattr(gb, "float_weight", None)
    if fw is not None:
        if algorithm_type == "auto":
            fd = getattr(gb, "from_dict", None)
            algo = fd if fd is not None else np.zeros(gb.n_graphs, np.uint8)
        else:
            algo = np.full(gb.n_graphs, 1 if algorithm_type == "dijkstra" else 0, np.uint8)
        return None, fw, algo
    w = gb.edge_weight
    if w is not None and (w.size == 0 or np.all(w == 1)):
        w = None
    return w, None, None


class ShortestPath(Kernel):
    """K[i,j] = <histogram of (label_u, label_v, d(u,v)) over ordered pairs of G_i, same of G_j>.

    Parameters as the reference (shortest_path.py:224-228): n_jobs, normalize, verbose,
    with_labels=True, algorithm_type in {"auto", "dijkstra", "floyd_warshall"} (validated; the
    device always runs the batched Floyd-Warshall / row-relaxation kernels, which give the
    same distances as either host algorithm).  Edge weights: positive integers below 2**20, or floats that
    are integer multiples of one common power of two (0.5, 1.25, ...): distances are then counted exactly in
    that unit (batch.quantise_weights) and ``_enum`` keys are the reference's float distances.  Other positive
    float weights (0.1, ...): the reference's features then depend on how it rounds, so the device reproduces its
    float distances bit for bit -- its floyd_warshall for adjacency input, its dijkstra for dictionary input, as
    ``algorithm_type`` says (sp.hip: gk_sp_build_f64; graphs above 143 vertices work on their float64 matrix in HBM: slower, no limit).
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
        w, fw, algo = sp_weight_args(self._cur_batch, self.algorithm_type)
        pb = engine.sp_build(dbatch, w, bool(self.with_labels), float_weights=fw, graph_algo=algo)
        pb._parent = dbatch
        return pb, 1

    # the base flows upload ``_fit_batch`` / the union; remember which one for edge weights
    def _gram_fit(self):
        self._cur_batch = self._fit_batch
        return super(ShortestPath, self)._gram_fit()

    def _gram_transform(self, Y):
        from .batch import GraphBatch
        ybatch, ymap = self._ingest(Y, self._label_map if self._label_map is not None else {})
        self._ny = ybatch.n_graphs
        self._cur_y_batch, self._cur_y_map = ybatch, ymap
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
        self.__dict__.pop("_state", None)
        self._arm_lazy_state()
        return self

    def _arm_lazy_state(self):
        self.X = _LazyDict(lambda: self._fitted_state()["X"], self._nx)
        self._enum = _LazyDict(lambda: self._fitted_state()["enum"])

    def __setstate__(self, state):
        super(ShortestPath, self).__setstate__(state)
        if "_fit_batch" in self.__dict__ and not isinstance(self.__dict__.get("_enum"), _LazyDict):
            filled_enum, filled_X = self.__dict__.get("_enum") or {}, self.__dict__.get("X") or {}
            if filled_enum and filled_X:
                self._state = dict(enum=dict(filled_enum), X=dict(filled_X))
            self._arm_lazy_state()

    # ---- the reference's fitted state (shortest_path.py:396-406,412-499), rebuilt on demand ------------
    def _pair_features(self, gb, n_first, enum, new_enum):
        """(l_u, l_v, d) counts of the graphs of ``gb`` in the reference's enumeration order: graph after
        graph, u then v in vertex order; keys not in ``enum`` are appended to ``new_enum`` (first seen
        first).  Distances come from the device (gk_sp_debug_apsp), labels are the original values."""
        eng = self._engine()
        db = eng.upload(gb)
        w, fw, algo = sp_weight_args(gb, self.algorithm_type)
        inv = None
        if self.with_labels and self._label_map is not None:
            inv = {i: k for k, i in self._all_label_ids.items()}
        counts = dict()
        for g in range(gb.n_graphs):
            v0, v1 = int(gb.graph_ptr[g]), int(gb.graph_ptr[g + 1])
            n = v1 - v0
            if fw is not None:
                S = eng.sp_debug_apsp_f64(db, fw, algo, g, n) if n > 0 else np.zeros((0, 0))
            else:
                S = eng.sp_debug_apsp(db, w, g, n) if n > 0 else np.zeros((0, 0), np.int32)
            lab = gb.node_label[v0:v1].tolist()
            row = dict()
            us, vs = np.nonzero((S >= 0) & ~np.eye(n, dtype=bool))
            step = 1.0 if fw is not None else getattr(gb, "weight_step", 1.0)   # dyadic weights: device distances count this unit
            for u, v, d in zip(us.tolist(), vs.tolist(), S[us, vs].tolist()):
                d = float(d) * step
                if self.with_labels:
                    key = (inv[lab[u]], inv[lab[v]], d) if inv is not None else (lab[u], lab[v], d)
                else:
                    key = d
                idx = enum.get(key)
                if idx is None:
                    idx = new_enum.get(key)
                    if idx is None:
                        idx = n_first + len(new_enum)
                        new_enum[key] = idx
                row[idx] = row.get(idx, 0) + 1
            counts[g] = row
        db.close()
        return counts

    def _fitted_state(self):
        st = self.__dict__.get("_state")
        if st is None:
            self._all_label_ids = dict(self._label_map) if self._label_map is not None else {}
            enum = dict()
            counts = self._pair_features(self._fit_batch, 0, dict(), enum)
            st = self._state = dict(enum=enum, X=counts)
        return st

    @staticmethod
    def _dense(counts, n_rows, n_cols):
        phi = np.zeros((n_rows, n_cols))
        for i, row in counts.items():
            for j, c in row.items():
                phi[i, j] = c
        return phi

    def _transform_state(self):
        """Counts and the extension ``_Y_enum`` of the last transform's targets (shortest_path.py:472-489)."""
        if "_Y_batch" not in self.__dict__:
            raise AttributeError("no transform has been called")
        ys = self.__dict__.get("_Y_state")
        if ys is None:
            st = self._fitted_state()
            self._all_label_ids = dict(self._label_map) if self._label_map is not None else {}
            self._all_label_ids.update(self._cur_y_map or {})
            y_enum = dict()
            counts = self._pair_features(self._Y_batch, len(st["enum"]), st["enum"], y_enum)
            ys = self._Y_state = dict(enum=y_enum, X=counts)
        return ys

    @property
    def _phi_X(self):
        """shortest_path.py:396-403 (fit_transform) / :292-301 (as wide as ``_enum`` + ``_Y_enum`` after a transform)."""
        if "_fit_batch" not in self.__dict__:
            raise AttributeError("_phi_X")
        st = self._fitted_state()
        width = len(st["enum"])
        if getattr(self, "_is_transformed", False):
            width += len(self._transform_state()["enum"])
        return self._dense(st["X"], self._nx, width)

    @property
    def _phi_Y(self):
        """shortest_path.py:292-301 of the last transform."""
        ys = self._transform_state()
        return self._dense(ys["X"], self._ny, len(self._fitted_state()["enum"]) + len(ys["enum"]))

    @property
    def _Y_enum(self):
        return self._transform_state()["enum"]

    def __getstate__(self):
        state = super(ShortestPath, self).__getstate__()
        for k in ("_state", "_Y_state", "_cur_y_batch", "_all_label_ids"):
            state.pop(k, None)
        return state

    def fit_transform(self, X, y=None):
        """shortest_path.py:370-410 (normalisation divides silently, :407-408)."""
        self._method_calling = 2
        self.fit(X)
        eng, feat = self._gram_fit()
        self._enum._known_len = self._last_info["label_counts"][0]
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
        self.__dict__.pop("_Y_state", None)
        self._Y_batch = self._cur_y_batch
        return eng.gram(feat, NORM_PLAIN if self.normalize else NORM_NONE)

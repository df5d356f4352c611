"""Weisfeiler-Lehman subtree kernel on MI355X (drop-in for ``grakel.WeisfeilerLehman``,
``grakel/kernels/weisfeiler_lehman.py:20``)."""
import numpy as np
from sklearn.utils.validation import check_is_fitted

from .batch import GraphBatch, _is_graph_object, iter_elements, sp_batch_from_input, wl_batch_from_input
from .kernel import Kernel, NORM_NONE, NORM_NAN_TO_NUM
from .shortest_path import ShortestPath
from .vertex_histogram import EdgeHistogram, VertexHistogram, FittedFeatures, count_matrix, first_seen_columns


MAX_LEVELS = 48        # FEAT_MAX_LEVELS of csrc/features.hip


def _accelerated(base):
    """The reference's own ``grakel.VertexHistogram`` / ``grakel.ShortestPath`` classes (callers that only
    swapped the WeisfeilerLehman import) map to the accelerated classes of the same name."""
    if type(base) is type and getattr(base, "__module__", "").split(".")[0] == "grakel":
        return {"VertexHistogram": VertexHistogram, "ShortestPath": ShortestPath,
                "EdgeHistogram": EdgeHistogram}.get(base.__name__, base)
    return base


def _is_kernel_class(base):
    """A base kernel class: one of ours, one of the reference's (`grakel.kernels.Kernel` subclasses: callers that only swapped
    the WeisfeilerLehman import), or anything with the estimator methods the framework calls (weisfeiler_lehman.py:260-285,
    :493, :519-541)."""
    if type(base) is not type and not isinstance(base, type):
        return False
    return issubclass(base, Kernel) or all(callable(getattr(base, m, None)) for m in ("fit", "fit_transform", "transform", "diagonal"))


class _LazyInvLabels(dict):
    """``_inv_labels`` (weisfeiler_lehman.py:199-210,257).  Level 0 -- the input label map -- is there from
    ``fit`` on; the dictionaries of the levels >= 1 hold the reference's credential strings, which only the
    host pass ``WeisfeilerLehman.inv_labels()`` can rebuild, so they are filled in the first time anything but
    level 0 is looked at.  Pickles as a plain dict of what is filled in."""

    def __init__(self, level0, fill):
        dict.__init__(self, {0: level0})
        self._fill = fill

    def _ensure(self):
        f = self._fill
        if f is not None:
            filled = f()                     # may raise (no GPU, out of memory): the next read tries again
            self._fill = None
            for k, v in filled.items():
                dict.__setitem__(self, k, v)

    def __missing__(self, key):
        self._ensure()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if key != 0:
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
        self._ensure()
        return dict.__len__(self)

    def __iter__(self):
        self._ensure()
        return dict.__iter__(self)

    def __contains__(self, key):
        if key != 0:
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

    def __reduce__(self):
        return (dict, (dict(dict.items(self)),))


class FittedLevel(object):
    """``WeisfeilerLehman.X[i]``: the base kernel the reference fits on level i (weisfeiler_lehman.py:260-285).
    ``X`` (graphs x labels of the level) and ``_labels`` ({reference label id: column}, first seen first) are
    rebuilt on the host from the reference-identical label ids when somebody reads them."""

    def __init__(self, owner, level, n_graphs, n_labels):
        self._owner, self._level = owner, level
        self.X = FittedFeatures(n_graphs, n_labels, self._build_X)
        self.sparse_ = True
        self._cols = None

    def _columns(self):
        if self._cols is None:
            self._cols = first_seen_columns(self._owner._level_reference_ids(self._level))
        return self._cols

    @property
    def _labels(self):
        return dict(self._columns())

    def _build_X(self):
        gb = self._owner._fit_batch
        ids = self._owner._level_reference_ids(self._level)
        cols = self._columns()
        uniq = np.fromiter(cols.keys(), np.int64, len(cols))
        col_of = np.fromiter(cols.values(), np.int64, len(cols))
        order = np.argsort(uniq)
        pos = np.searchsorted(uniq[order], ids)
        node_graph = np.repeat(np.arange(gb.n_graphs), np.diff(gb.graph_ptr))
        return count_matrix(node_graph, col_of[order][pos], gb.n_graphs, len(cols))

    def __getstate__(self):
        return dict(_owner=None, _level=self._level, X=self.X, sparse_=True, _cols=self._cols)

    def __setstate__(self, st):
        self.__dict__.update(st)


class WeisfeilerLehman(Kernel):
    """Sum over WL levels 0..n_iter of the vertex-histogram kernel of the relabelled graphs.

    Parameters as the reference (weisfeiler_lehman.py:58-65): n_jobs, verbose, normalize,
    n_iter=5, base_graph_kernel=VertexHistogram.  Base kernels on the accelerated path:
    ``VertexHistogram`` (the WL-subtree kernel) and ``ShortestPath`` (class or
    ``(ShortestPath, {params})``): one APSP, then per level the pairs keyed by
    that level's WL labels, all levels concatenated into one Gram product.
    """

    _graph_format = "dictionary"
    _norm_mode = NORM_NAN_TO_NUM
    _generic = False           # base kernel outside the accelerated set: relabel on the device, base kernels on the host

    def __init__(self, n_jobs=None, verbose=False, normalize=False, n_iter=5,
                 base_graph_kernel=VertexHistogram):
        super(WeisfeilerLehman, self).__init__(n_jobs=n_jobs, verbose=verbose, normalize=normalize)
        self.n_iter = n_iter
        self.base_graph_kernel = base_graph_kernel
        self._initialized.update({"n_iter": False, "base_graph_kernel": False})
        self._base_graph_kernel = None

    def initialize(self):
        """weisfeiler_lehman.py:74-115."""
        super(WeisfeilerLehman, self).initialize()
        if not self._initialized["base_graph_kernel"]:
            base = self.base_graph_kernel
            if base is None:
                base, params = VertexHistogram, dict()
            elif _is_kernel_class(_accelerated(base)):
                base, params = _accelerated(base), dict()
            else:
                try:
                    base, params = base
                except Exception:
                    raise TypeError('Base kernel was not formulated in the correct way. '
                                    'Check documentation.')
                base = _accelerated(base)
                if not _is_kernel_class(base):
                    raise TypeError('The first argument must be a valid grakel.kernel.kernel Object')
                if type(params) is not dict:
                    raise ValueError('If the second argument of base kernel exists, it must be a '
                                     'dictionary between parameters names and values')
                params.pop("normalize", None)
            if base is ShortestPath:
                probe = ShortestPath(**{k: v for k, v in params.items() if k != "normalize"})
                probe.initialize()                       # validates algorithm_type like the reference
                self._sp_with_labels = bool(probe.with_labels)
                self._sp_algorithm_type = probe.algorithm_type
            elif base is EdgeHistogram:
                EdgeHistogram(**{k: v for k, v in params.items() if k != "normalize"}).initialize()
            # any OTHER base kernel (weisfeiler_lehman.py:77-109 accepts every grakel.Kernel; the reference's own regression
            # test uses NeighborhoodSubgraphPairwiseDistance, grakel/tests/test_kernels.py:82-106): the relabelling runs on
            # the MI355X, every level's relabelled graphs go back to the HOST base kernel exactly as the reference hands
            # them over (`_generic_*` below).  The drop-in contract holds; only the relabel loop is accelerated.
            self._generic = base not in (VertexHistogram, ShortestPath, EdgeHistogram)
            params["normalize"] = False
            params["verbose"] = self.verbose
            params["n_jobs"] = None
            self._base_graph_kernel = base
            self._params = params
            self._initialized["base_graph_kernel"] = True
        if not self._initialized["n_iter"]:
            if type(self.n_iter) is not int or self.n_iter <= 0:
                raise TypeError("'n_iter' must be a positive integer")
            self._n_iter = self.n_iter + 1
            self._initialized["n_iter"] = True

    # ---- base kernel EdgeHistogram: the edge labels x[2] reach the base kernel untouched at every level
    # (weisfeiler_lehman.py:186-190: extras = x[2:], :235-258 relabels the NODES only), so every level's matrix is the
    # EdgeHistogram matrix and their sum is (n_iter + 1) times it.  The WL ingestion still runs (it validates the node
    # labels exactly like the reference does before the base kernels see anything).
    def _eh_inner(self):
        return EdgeHistogram(**dict(self._params))

    def _eh_scale(self, K, dr, dc):
        K = K * float(self._n_iter)
        if self.normalize:
            with np.errstate(divide="ignore", invalid="ignore"):
                K = np.nan_to_num(np.divide(K, np.sqrt(np.outer(dr, dc))))
        return K

    def _ingest(self, X, fitted):
        not_iter = TypeError if fitted is None else ValueError      # :143-144 vs :358-359
        if self._base_graph_kernel is ShortestPath:
            # the SP node set (every vertex of the graph object, with edge weights); WL itself
            # always needs the labels, whatever the base kernel's with_labels says
            return sp_batch_from_input(X, True, fitted, len_ok=lambda n: n >= 2, not_iterable=not_iter)
        return wl_batch_from_input(X, fitted, min_len=2, not_iterable=not_iter)

    def _prepare(self, engine, dbatch):
        engine.wl_relabel(dbatch, self._n_iter - 1)
        if self._base_graph_kernel is ShortestPath:
            from .shortest_path import sp_weight_args
            # the reference's WL keeps every graph in its dictionary format (weisfeiler_lehman.py:56) and hands the base
            # kernel edge dictionaries, so "auto" means dijkstra for every element there (matters for general float weights)
            algo_type = "dijkstra" if self._sp_algorithm_type == "auto" else self._sp_algorithm_type
            w, fw, algo = sp_weight_args(self._cur_batch, algo_type)
            pb = engine.sp_build(dbatch, w, self._sp_with_labels, n_levels=self._n_iter, float_weights=fw, graph_algo=algo)
            pb._parent = dbatch
            pb.label_counts, pb.pair_key_counts = dbatch.label_counts, pb.label_counts
            return pb, self._n_iter
        return dbatch, self._n_iter

    def _gram_fit(self):
        self._cur_batch = self._fit_batch
        return super(WeisfeilerLehman, self)._gram_fit()

    def _gram_transform(self, Y):
        ybatch, _ = self._ingest(Y, self._label_map if self._label_map is not None else {})
        self._ny = ybatch.n_graphs
        eng = self._engine()
        if self._base_graph_kernel is ShortestPath:        # edge weights travel with the host batch
            self._cur_batch = GraphBatch.concat(self._fit_batch, ybatch)
            db = eng.upload(self._cur_batch)
        else:
            db = self._union_on_device(eng, ybatch)
        fb, n_levels = self._prepare(eng, db)
        feat = eng.features(fb, n_levels, n_fit=self._nx)
        selfk = eng.selfk(feat)
        self._X_diag = selfk[:self._nx]
        self._Y_diag = selfk[self._nx:]
        return eng, feat

    def _after_fit(self):
        self.__dict__.pop("_reference_labels", None)
        self.__dict__.pop("_lookup_declined", None)
        lvl0 = dict(self._label_map) if self._label_map is not None else \
            {int(i): int(i) for i in np.unique(self._fit_batch.node_label).tolist()}
        self._inv_labels = _LazyInvLabels(lvl0, self._all_inv_labels)
        self.X = {i: FittedLevel(self, i, self._nx, None) for i in range(self._n_iter)}

    def __setstate__(self, state):
        super(WeisfeilerLehman, self).__setstate__(state)
        d = self.__dict__.get("_inv_labels")
        if isinstance(d, dict) and not isinstance(d, _LazyInvLabels):      # pickled as a plain dict
            lazy = _LazyInvLabels(d.get(0, {}), self._all_inv_labels if len(d) <= 1 else None)
            for k, v in d.items():
                dict.__setitem__(lazy, k, v)
            self._inv_labels = lazy
        if isinstance(self.__dict__.get("X"), dict):
            for lvl in self.X.values():
                if isinstance(lvl, FittedLevel):
                    lvl._owner = self

    def _level_reference_ids(self, level):
        """Reference label id of every node of the fit batch at ``level`` (the values the reference's
        level-``level`` base kernel sees as labels)."""
        if "_reference_labels" not in self.__dict__:
            self._inv_labels._ensure()
        if "_reference_labels" not in self.__dict__:      # the dictionaries came from a pickle: replay
            self._all_inv_labels()
        return self._reference_labels[level]

    # ---- any other base kernel (host side) -------------------------------------------------------------------------
    _GENERIC_MSG = ('each element of X must be either a graph object or a list with at least a graph '
                    'like object and node labels dict \n')

    def _generic_elements(self, X, fitting):
        """The validated elements as (graph object, label dict, extras) -- what weisfeiler_lehman.py:142-190 keeps per
        graph (Gs_ed[j], L[j], extras[j]); the graph object itself is handed on (the base kernel parses it as it would
        have parsed the reference's edge dictionary)."""
        els = []
        for x in iter_elements(X, lambda n: n >= 2, self._GENERIC_MSG, TypeError if fitting else ValueError):
            if _is_graph_object(x):
                if hasattr(x, "desired_format"):
                    x.desired_format("dictionary")
                els.append((x.get_edge_dictionary(), x.get_labels(purpose="dictionary"), tuple()))
            else:
                els.append((x[0], x[1], tuple(x[2:])))
        return els

    @staticmethod
    def _generic_graphs(els, graph_ptr, ids):
        gp = graph_ptr.tolist()
        out = []
        for j, (g, lab, extra) in enumerate(els):
            out.append((g, dict(zip(lab.keys(), ids[gp[j]:gp[j + 1]].tolist()))) + extra)
        return out

    def _generic_fit(self, X, want_matrix):
        """weisfeiler_lehman.py:212-290: WL levels by the device relabel (reference-identical label ids: `_inv_labels`),
        one host base kernel per level fitted on the relabelled graphs; K = sum of their matrices."""
        els = self._generic_elements(X, True)
        self._fit_host([[g, lab] for g, lab, _ in els])
        self._after_fit()
        base, K = dict(), None
        for l in range(self._n_iter):
            graphs = self._generic_graphs(els, self._fit_batch.graph_ptr, self._level_reference_ids(l))
            base[l] = self._base_graph_kernel(**self._params)
            if want_matrix:
                Kl = base[l].fit_transform(graphs)
                K = Kl if K is None else K + Kl
            else:
                base[l].fit(graphs)
        self.X = base
        return K

    def _generic_transform(self, X):
        """weisfeiler_lehman.py:330-500: the targets are relabelled jointly with the fitted graphs on the device; a target
        class that also occurs among the fitted graphs gets the fitted (reference-identical) id, every other class an id
        above all fitted ones (the reference numbers those by sorted credential: any distinct ids give the same matrices)."""
        els = self._generic_elements(X, False)
        ybatch, _ = wl_batch_from_input([[g, lab] for g, lab, _ in els], self._label_map if self._label_map is not None else {})
        self._ny = ybatch.n_graphs
        eng = self._engine()
        db = self._union_on_device(eng, ybatch)
        eng.wl_relabel(db, self._n_iter - 1)
        Vf = self._fit_batch.n_nodes
        fresh = 1 + max(int(self._level_reference_ids(l).max()) for l in range(self._n_iter)) + int(ybatch.n_labels)
        K = None
        for l in range(self._n_iter):
            if l == 0:
                ids = ybatch.node_label.astype(np.int64)           # the fit's level-0 ids, unseen values above them
            else:
                dev = eng.wl_labels(db, l).astype(np.int64)
                lut = np.full(int(dev.max()) + 1, -1, np.int64)
                lut[dev[:Vf]] = self._level_reference_ids(l)
                ids = lut[dev[Vf:]]
                unseen = ids < 0
                ids[unseen] = fresh + dev[Vf:][unseen]
            Kl = self.X[l].transform(self._generic_graphs(els, ybatch.graph_ptr, ids))
            K = Kl if K is None else K + Kl
        db.close()
        self._is_transformed = True
        if self.normalize:
            X_diag, Y_diag = self.diagonal()
            with np.errstate(divide="ignore", invalid="ignore"):
                K = np.nan_to_num(np.divide(K, np.sqrt(np.outer(Y_diag, X_diag))))
        return K

    def _generic_diagonal(self):
        """weisfeiler_lehman.py:502-555: the sum of the base kernels' diagonals."""
        check_is_fitted(self, ['X'])
        if getattr(self, "_is_transformed", False):
            xs, ys = zip(*(self.X[i].diagonal() for i in range(self._n_iter)))
            if not hasattr(self, "_X_diag"):
                self._X_diag = np.sum(xs, axis=0)
            return self._X_diag, np.sum(ys, axis=0)
        if not hasattr(self, "_X_diag"):
            self._X_diag = np.sum([self.X[i].diagonal() for i in range(self._n_iter)], axis=0)
        return self._X_diag

    def fit(self, X, y=None):
        """kernel.py:86-121 with weisfeiler_lehman.py:117-290 as parse_input."""
        self._is_transformed = False
        self._method_calling = 1
        self.initialize()
        if X is None:
            raise ValueError('`fit` input cannot be None')
        if self._generic:
            self._generic_fit(X, False)
            return self
        X = list(X) if self._base_graph_kernel is EdgeHistogram and not isinstance(X, (list, tuple)) else X
        self._fit_host(X)
        self._after_fit()
        if self._base_graph_kernel is EdgeHistogram:
            self._eh = self._eh_inner().fit(X)
        return self

    def fit_transform(self, X, y=None):
        """weisfeiler_lehman.py:292-328."""
        self._method_calling = 2
        self._is_transformed = False
        self.initialize()
        if X is None:
            raise ValueError('transform input cannot be None')
        if self._generic:
            K = self._generic_fit(X, True)
            self._X_diag = np.diagonal(K).copy()
            if self.normalize:
                with np.errstate(divide="ignore", invalid="ignore"):
                    K = np.nan_to_num(np.divide(K, np.sqrt(np.outer(self._X_diag, self._X_diag))))
            return K
        X = list(X) if self._base_graph_kernel is EdgeHistogram and not isinstance(X, (list, tuple)) else X
        self._fit_host(X)
        self._after_fit()
        if self._base_graph_kernel is EdgeHistogram:
            self._eh = self._eh_inner()
            K = self._eh.fit_transform(X)
            self._X_diag = self._eh.diagonal() * float(self._n_iter)
            return self._eh_scale(K, self._X_diag, self._X_diag)
        if self._base_graph_kernel is VertexHistogram and self._n_iter <= MAX_LEVELS:
            K = self._fit_transform_fused(self._n_iter - 1, NORM_NAN_TO_NUM if self.normalize else NORM_NONE)
            for i, c in enumerate(self._last_info["label_counts"]):
                self.X[i].X.shape = (self._nx, c)
            return K
        eng, feat = self._gram_fit()
        for i, c in enumerate(self._last_info["label_counts"]):
            self.X[i].X.shape = (self._nx, c)
        return eng.gram(feat, NORM_NAN_TO_NUM if self.normalize else NORM_NONE)

    def transform(self, X):
        """weisfeiler_lehman.py:330-500.  The targets are relabelled JOINTLY with the fitted
        graphs: the WL partition of the union restricted to the fitted graphs is the fitted
        partition, so K[targets, fitted] equals the reference's dictionary look-up path."""
        self._method_calling = 3
        check_is_fitted(self, ['X', '_nx'])
        if X is None:
            raise ValueError('transform input cannot be None')
        if self._generic:
            return self._generic_transform(X)
        if self._base_graph_kernel is EdgeHistogram:
            X = list(X) if not isinstance(X, (list, tuple)) else X
            ybatch, _ = self._ingest(X, self._label_map if self._label_map is not None else {})     # WL's own input checks
            self._ny = ybatch.n_graphs
            K = self._eh.transform(X)
            xd, yd = self._eh.diagonal()
            self._X_diag, self._Y_diag = xd * float(self._n_iter), yd * float(self._n_iter)
            self._is_transformed = True
            return self._eh_scale(K, self._Y_diag, self._X_diag)
        # ONE ingestion for both routes (a one-shot iterable is exhausted by the first walk; a list would pay the host
        # walk twice): `_ingest` hands a ready GraphBatch back unchanged
        ybatch, _ = self._ingest(X, self._label_map if self._label_map is not None else {})
        K = self._transform_lookup(ybatch)
        if K is not None:
            self._is_transformed = True
            return K
        eng, feat = self._gram_transform(ybatch)
        self._is_transformed = True
        return eng.gram(feat, NORM_NAN_TO_NUM if self.normalize else NORM_NONE)

    # which transform route: "auto" = look-up while the targets are at most 1/32 of the fitted nodes (its work is
    # proportional to the targets; the joint relabel of fitted graphs + targets is the MFMA route for large target sets),
    # "lookup" / "joint" force one (tests).  Not a constructor parameter: get_params() stays the reference's.
    transform_route = "auto"

    def _transform_lookup(self, X):
        """weisfeiler_lehman.py:435-498 the reference's way: relabel the targets alone and look their signatures up in the
        fitted dictionaries kept on the device (csrc/wl_transform.hip).  None: take the joint route."""
        if self._base_graph_kernel is not VertexHistogram or type(self)._feature_kind != 0 or self.transform_route == "joint":
            return None
        if self._n_iter > MAX_LEVELS or self.__dict__.get("_lookup_declined"):
            return None
        ybatch, _ = self._ingest(X, self._label_map if self._label_map is not None else {})
        if self.transform_route != "lookup" and ybatch.n_nodes * 32 > self._fit_batch.n_nodes:
            return None                               # measured at 10 000 fitted graphs: look-up wins up to a few hundred targets
        eng = self._engine()
        h = self._n_iter - 1
        dfit = self._fitted_on_device(eng)
        wf = self.__dict__.get("_dev_wlfit")
        for attempt in range(2):
            if wf is None or wf.handle is None or wf.batch is not dfit:
                if getattr(dfit, "label_counts", None) is None or len(dfit.label_counts) != self._n_iter:
                    eng.wl_relabel(dfit, h)
                wf = eng.wl_fitted(dfit, h)
                if wf is None:                        # hash collision among the fitted classes / a hub: joint route from now on
                    self._lookup_declined = True
                    return None
                self._dev_wlfit = wf
            yb = eng.upload(ybatch)
            try:
                eng.wl_relabel(yb, h)
                out = eng.wl_transform(wf, yb, NORM_NAN_TO_NUM if self.normalize else NORM_NONE)
            finally:
                yb.close()
            if out is None:                           # a target graph too large / a hub among the targets: joint route for this call
                return None
            if out != "stale":
                break
            wf.close()                                # the fitted batch was relabelled since (diagonal(), a second fit_transform)
            wf = None
            if attempt:
                return None
        K, ydiag = out
        self._ny = ybatch.n_graphs
        if not hasattr(self, "_X_diag"):
            self._X_diag = eng.wl_fitted_selfk(wf)
        self._Y_diag = ydiag
        return K

    def diagonal(self):
        """weisfeiler_lehman.py:502-555."""
        if getattr(self, "_generic", False):
            return self._generic_diagonal()
        if getattr(self, "_base_graph_kernel", None) is EdgeHistogram:
            check_is_fitted(self, ['X'])
            if not hasattr(self, "_X_diag"):
                self._X_diag = self._eh.diagonal() * float(self._n_iter)
            return (self._X_diag, self._Y_diag) if getattr(self, "_is_transformed", False) else self._X_diag
        return super(WeisfeilerLehman, self).diagonal()

    def _all_inv_labels(self):
        """Reference-identical ``_inv_labels`` for every level.

        The device uses arbitrary dense ids per level (the Gram matrix does not depend on
        them).  The reference numbers the labels of level i by the LEXICOGRAPHIC order of the
        credential strings ``str(own) + "," + str(sorted(neighbour ids))`` with a counter that
        continues across levels (weisfeiler_lehman.py:235-246).  This host pass rebuilds exactly
        that from one representative node per device label: cost O(#labels * degree) in Python,
        so it is on demand only.  Returns the dict and stores it in ``self._inv_labels``.
        """
        check_is_fitted(self, ['X', '_nx'])
        gb = self._fit_batch
        eng = self._engine()
        db = eng.upload(gb)
        eng.wl_relabel(db, self._n_iter - 1)
        ref_prev = gb.node_label.astype(np.int64)          # level-0 ids are already the reference's
        out = {0: dict(dict.__getitem__(self._inv_labels, 0))}
        count = len(out[0])
        self._reference_labels = [ref_prev]
        for i in range(1, self._n_iter):
            dev = eng.wl_labels(db, i)
            uniq, first = np.unique(dev, return_index=True)     # uniq == 0..L-1
            creds = []
            for v in first.tolist():
                nb = ref_prev[gb.col_idx[gb.row_ptr[v]:gb.row_ptr[v + 1]]]
                creds.append(str(int(ref_prev[v])) + "," + str(sorted(nb.tolist())))
            ids = np.empty(len(creds), np.int64)
            inv = dict()
            for c in sorted(range(len(creds)), key=creds.__getitem__):
                inv[creds[c]] = count
                ids[c] = count
                count += 1
            out[i] = inv
            ref_prev = ids[dev]
            self._reference_labels.append(ref_prev)
        db.close()
        return out

    def inv_labels(self):
        """Reference-identical ``_inv_labels`` for every level, as a plain dict (also what reading
        ``self._inv_labels[i]`` for i >= 1 triggers)."""
        check_is_fitted(self, ['X', '_nx'])
        self._inv_labels._ensure()
        return dict(dict.items(self._inv_labels))

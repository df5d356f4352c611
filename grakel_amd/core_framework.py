"""Core-variant framework on MI355X (drop-in for ``grakel.CoreFramework``,
``grakel/kernels/core_framework.py:19``).

K = sum over core levels i = max_core .. 0 of the base kernel on the subgraphs induced by the
vertices of core number >= i; a graph without such vertices sits out of level i
(core_framework.py:163-212).  The k-core numbers come from ``gk_core_numbers`` (one workgroup per
graph, peeling in LDS); the induced sub-batches are cut out of the packed CSR on the host (index
arithmetic only) and every level runs the accelerated base kernel on its sub-batch.
"""
import numpy as np
from sklearn.utils.validation import check_is_fitted

from .batch import GraphBatch, sp_batch_from_input
from .kernel import Kernel
from .shortest_path import ShortestPath
from .vertex_histogram import EdgeHistogram


def induced_subbatch(gb, keep):
    """Sub-batch of the vertices flagged in ``keep`` (bool[n_nodes]) with the edges among them;
    graphs left without a vertex are dropped.  Returns (GraphBatch, indices of the kept graphs)."""
    N, V = gb.n_graphs, gb.n_nodes
    node_graph = np.repeat(np.arange(N), np.diff(gb.graph_ptr))
    sizes = np.bincount(node_graph[keep], minlength=N)
    gidx = np.nonzero(sizes)[0]
    new_id = np.cumsum(keep) - 1
    src = np.repeat(np.arange(V), np.diff(gb.row_ptr))
    ekeep = keep[src] & keep[gb.col_idx]
    n_kept = int(keep.sum())
    row_ptr = np.zeros(n_kept + 1, np.int64)
    np.cumsum(np.bincount(new_id[src[ekeep]], minlength=n_kept), out=row_ptr[1:])
    graph_ptr = np.zeros(len(gidx) + 1, np.int64)
    np.cumsum(sizes[gidx], out=graph_ptr[1:])
    ew = gb.edge_weight[ekeep] if gb.edge_weight is not None and gb.edge_weight.size else gb.edge_weight
    fw = getattr(gb, "float_weight", None)
    if fw is not None:
        # general float weights travel with the edges.  The reference turns every element into its dictionary format
        # (core_framework.py:40,153) and cuts the subgraphs out in that format (graph.py:1379), so the base ShortestPath's
        # "auto" runs dijkstra on every subgraph (graph.py:652-656): all flags 1
        return GraphBatch(graph_ptr, row_ptr, new_id[gb.col_idx[ekeep]], gb.node_label[keep], gb.n_labels, None,
                          1.0, fw[ekeep], np.ones(len(gidx), np.uint8)), gidx
    return GraphBatch(graph_ptr, row_ptr, new_id[gb.col_idx[ekeep]], gb.node_label[keep], gb.n_labels, ew,
                      getattr(gb, "weight_step", 1.0)), gidx


class CoreFramework(Kernel):
    """Parameters as the reference (core_framework.py:42-50): n_jobs, verbose, normalize,
    min_core=-1, base_graph_kernel=None (-> ShortestPath).  Like the reference, ``min_core`` is
    stored as -1 whatever is passed (:48), so every core level down to 0 is used.  Base kernels:
    any accelerated kernel that takes packed batches (ShortestPath, VertexHistogram,
    WeisfeilerLehman, WeisfeilerLehmanOptimalAssignment), as a class or ``(class, params)``."""

    _graph_format = "adjacency"

    def __init__(self, n_jobs=None, verbose=False, normalize=False, min_core=-1, base_graph_kernel=None):
        super(CoreFramework, self).__init__(n_jobs=n_jobs, verbose=verbose, normalize=normalize)
        self.min_core = -1
        self.base_graph_kernel = base_graph_kernel
        self._initialized.update({"min_core": False, "base_graph_kernel": False})

    def initialize(self):
        """core_framework.py:52-93."""
        super(CoreFramework, self).initialize()
        if not self._initialized["base_graph_kernel"]:
            base = self.base_graph_kernel
            if base is None:
                base, params = ShortestPath, dict()
            elif type(base) is type and issubclass(base, Kernel):
                params = dict()
            else:
                try:
                    base, params = base
                except Exception:
                    raise TypeError('Base kernel was not formulated in the correct way. Check documentation.')
                if not (type(base) is type and issubclass(base, Kernel)):
                    raise TypeError('The first argument must be a valid grakel.kernel.kernel Object')
                if type(params) is not dict:
                    raise ValueError('If the second argument of base kernel exists, it must be a '
                                     'dictionary between parameters names and values')
                params = dict(params)
                params.pop("normalize", None)
            if issubclass(base, (EdgeHistogram, CoreFramework)):
                raise NotImplementedError('CoreFramework on MI355X needs a base kernel that takes packed '
                                          'node-labelled batches (SP, VH, WL, WL-OA)')
            params["normalize"] = False
            params["verbose"] = self.verbose
            params["n_jobs"] = None
            self.base_graph_kernel_, self.params_ = base, params
            self._initialized["base_graph_kernel"] = True
        if not self._initialized["min_core"]:
            if type(self.min_core) is not int or self.min_core < -1:
                raise TypeError("'min_core' must be an integer bigger than -1")
            self._initialized["min_core"] = True

    # -- ingestion + core numbers ---------------------------------------------------------------
    def _ingest(self, X, fitted):
        base_unlabelled = self.base_graph_kernel_ is ShortestPath and self.params_.get("with_labels", True) is False
        return sp_batch_from_input(X, not base_unlabelled, fitted, len_ok=lambda n: n >= 1)

    def _levels(self, gb):
        eng = self._engine()
        db = eng.upload(gb)
        core = eng.core_numbers(db)
        db.close()
        return core, int(core.max()) if core.size else 0

    def _new_base(self):
        return self.base_graph_kernel_(**self.params_)

    def _fit_levels(self, X, want_matrix):
        gb, mapping = self._ingest(X, None)
        self._fit_batch, self._label_map, self._nx = gb, mapping, gb.n_graphs
        core, self._max_core_number = self._levels(gb)
        if self._max_core_number <= self.min_core:
            raise ValueError('The maximum core equals the min_core boundary set in init.')
        K = np.zeros((self._nx, self._nx)) if want_matrix else None
        self.X, self._fit_indexes = dict(), dict()
        x_diag = np.zeros(self._nx)
        for i in range(self._max_core_number, self.min_core, -1):
            sub, idx = induced_subbatch(gb, core >= i)
            self._fit_indexes[i] = idx
            if not len(idx):
                continue
            self.X[i] = self._new_base()
            if want_matrix:
                K[np.ix_(idx, idx)] += self.X[i].fit_transform(sub)
            else:
                self.X[i].fit(sub)
            x_diag[idx] += self.X[i].diagonal()
        self._X_diag = x_diag
        return K

    def fit(self, X, y=None):
        self._method_calling = 1
        self._is_transformed = False
        self.initialize()
        if X is None:
            raise ValueError('`fit` input cannot be None')
        self._fit_levels(X, False)
        return self

    def fit_transform(self, X, y=None):
        """core_framework.py:262-297."""
        self._method_calling = 2
        self._is_transformed = False
        self.initialize()
        if X is None:
            raise ValueError('transform input cannot be None')
        km = self._fit_levels(X, True)
        if self.normalize:
            with np.errstate(divide='ignore', invalid='ignore'):
                km = np.nan_to_num(np.divide(km, np.sqrt(np.outer(self._X_diag, self._X_diag))))
        return km

    def transform(self, X):
        """core_framework.py:226-260 with :186-205: levels the fitted data never reached only feed
        the targets' diagonal (the reference's "dummy" kernels)."""
        self._method_calling = 3
        check_is_fitted(self, ['X'])
        if X is None:
            raise ValueError('transform input cannot be None')
        gb, _ = self._ingest(X, self._label_map if self._label_map is not None else {})
        core, t_max = self._levels(gb)
        if t_max <= self.min_core:
            raise ValueError('The maximum core equals the min_core boundary set in init.')
        km = np.zeros((gb.n_graphs, self._nx))
        y_diag = np.zeros(gb.n_graphs)
        for i in range(t_max, self.min_core, -1):
            sub, idx = induced_subbatch(gb, core >= i)
            if not len(idx):
                continue
            if self._max_core_number < i or not len(self._fit_indexes[i]):
                dummy = self._new_base()
                dummy.fit(sub)
                y_diag[idx] += dummy.diagonal()
            else:
                km[np.ix_(idx, self._fit_indexes[i])] += self.X[i].transform(sub)
                y_diag[idx] += self.X[i].diagonal()[1]
        self._Y_diag, self._t_nx = y_diag, gb.n_graphs
        self._is_transformed = True
        if self.normalize:
            with np.errstate(divide='ignore', invalid='ignore'):
                km = np.nan_to_num(km / np.sqrt(np.outer(y_diag, self._X_diag)))
        return km

    def diagonal(self):
        """core_framework.py:299-373."""
        check_is_fitted(self, ['X'])
        if getattr(self, "_is_transformed", False):
            return self._X_diag, self._Y_diag
        return self._X_diag

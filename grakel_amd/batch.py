"""Host-side ingestion: grakel input objects -> packed CSR ``GraphBatch``.

Replaces, for the WL / VH / SP path only, what the reference does per graph with
``grakel.Graph`` objects (``grakel/graph.py:147-232``, format detection
``:1542-1709``) and the ``parse_input`` loops of the three kernels
(``weisfeiler_lehman.py:142-194``, ``vertex_histogram.py:75-121``,
``shortest_path.py:441-466``).  No ``Graph`` objects are built: every accepted
input form is flattened straight into int32 arrays.

Accepted element forms (same as the reference, ``SURVEY.md`` 8b): an iterable
``[graph_obj, node_labels(, edge_labels(, *extras))]`` where ``graph_obj`` is an
adjacency matrix (ndarray / list of lists / scipy.sparse) or an edge dictionary
(``{(u,v): w}``, ``{u: [v..]}``, ``{u: {v: w}}``, iterable of 2-/3-tuples), or any
object exposing ``get_edge_dictionary()`` / ``get_labels()`` (a ``grakel.Graph``).
"""
import numbers
import warnings
from collections.abc import Iterable
from itertools import chain

import numpy as np
from scipy.sparse import issparse

try:                                   # optional C fast path of the dict-of-lists / dict-of-dicts walk
    from . import _gk_ingest           # (csrc/ingest.c, built by `make -C grakel_amd/csrc`); host logic only
except ImportError:                    # not built: the Python path below does everything
    _gk_ingest = None

# host threads of the C walk over `[{u: [v, ...]}, {u: label}]` elements (csrc/ingest.c: wl_ingest_threads):
# 0 = one per host core (at most 32; 64 for the tuple-set form), 1 = the calling thread only.  The result does not depend on it.
INGEST_THREADS = 0


class GraphBatch(object):
    """A set of graphs packed as CSR (int32), the unit the HIP library consumes.

    graph_ptr[n_graphs+1], row_ptr[n_nodes+1], col_idx[n_edges] (global node ids),
    node_label[n_nodes] (dense level-0 ids), n_labels (ids are < n_labels),
    edge_weight[n_edges] or None (ShortestPath only): positive integers; the weight of an edge
    is edge_weight * weight_step (weight_step is 1.0 unless the input had float weights that are
    integer multiples of a common power of two, see quantise_weights()).
    float_weight[n_edges] (float64) or None: ShortestPath input with OTHER positive float weights; the device then
    reproduces the reference's float distances bit for bit (sp.hip: gk_sp_build_f64) and needs to know per graph which of
    its two algorithms the reference's "auto" runs: from_dict[n_graphs] (uint8), 1 = the element was an edge dictionary
    (dijkstra), 0 = an adjacency matrix (floyd_warshall).  edge_weight is None in that mode.
    """

    def __init__(self, graph_ptr, row_ptr, col_idx, node_label, n_labels, edge_weight=None, weight_step=1.0,
                 float_weight=None, from_dict=None):
        self.graph_ptr = np.ascontiguousarray(graph_ptr, dtype=np.int32)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        self.node_label = np.ascontiguousarray(node_label, dtype=np.int32)
        self.n_labels = int(n_labels)
        self.edge_weight = None if edge_weight is None else np.ascontiguousarray(edge_weight, np.int32)
        self.weight_step = float(weight_step)
        self.float_weight = None if float_weight is None else np.ascontiguousarray(float_weight, np.float64)
        self.from_dict = None if from_dict is None else np.ascontiguousarray(from_dict, np.uint8)
        if self.float_weight is not None and (self.float_weight.shape[0] != self.col_idx.shape[0] or self.edge_weight is not None):
            raise ValueError("GraphBatch: float_weight needs one value per edge and excludes edge_weight")
        if self.row_ptr.shape[0] != self.node_label.shape[0] + 1:
            raise ValueError("GraphBatch: row_ptr must have n_nodes+1 entries")
        if int(self.graph_ptr[-1]) != self.n_nodes or int(self.row_ptr[-1]) != self.n_edges:
            raise ValueError("GraphBatch: inconsistent pointer arrays")
        # O(V) sanity of what the kernels index with (the neighbour ranges are checked on the device,
        # gk_batch_create -> GK_ERR_ARG): monotone pointers from 0, label ids inside [0, n_labels)
        if self.graph_ptr[0] != 0 or self.row_ptr[0] != 0 or np.any(np.diff(self.graph_ptr) < 0) \
                or np.any(np.diff(self.row_ptr) < 0):
            raise ValueError("GraphBatch: graph_ptr / row_ptr must start at 0 and never decrease")
        if self.n_nodes and (int(self.node_label.min()) < 0 or int(self.node_label.max()) >= max(self.n_labels, 1)):
            raise ValueError("GraphBatch: node_label ids must lie in [0, n_labels)")

    n_graphs = property(lambda self: self.graph_ptr.shape[0] - 1)
    n_nodes = property(lambda self: self.node_label.shape[0])
    n_edges = property(lambda self: self.col_idx.shape[0])

    def slice_graphs(self, lo, hi):
        """Sub-batch of graphs [lo, hi) (multi-GPU sharding of the input)."""
        v0, v1 = int(self.graph_ptr[lo]), int(self.graph_ptr[hi])
        e0, e1 = int(self.row_ptr[v0]), int(self.row_ptr[v1])
        ew = None if self.edge_weight is None else self.edge_weight[e0:e1]
        fw = getattr(self, "float_weight", None)
        fd = getattr(self, "from_dict", None)
        return GraphBatch(self.graph_ptr[lo:hi + 1] - v0, self.row_ptr[v0:v1 + 1] - e0,
                          self.col_idx[e0:e1] - v0, self.node_label[v0:v1], self.n_labels, ew,
                          getattr(self, "weight_step", 1.0), None if fw is None else fw[e0:e1],
                          None if fd is None else fd[lo:hi])

    @staticmethod
    def concat(a, b):
        """Union batch: graphs of ``a`` first, then graphs of ``b`` (transform = fit + targets)."""
        ew, step = None, 1.0
        fa, fb = getattr(a, "float_weight", None), getattr(b, "float_weight", None)
        da, db_ = getattr(a, "from_dict", None), getattr(b, "from_dict", None)
        fd = None
        if da is not None or db_ is not None:
            fd = np.concatenate([da if da is not None else np.zeros(a.n_graphs, np.uint8),
                                 db_ if db_ is not None else np.zeros(b.n_graphs, np.uint8)])
        if fa is not None or fb is not None:
            # one side has general float weights: the union counts in float64 (the other side's exact weights as floats)
            def as_float(x, f):
                if f is not None:
                    return f
                if x.edge_weight is None:
                    return np.ones(x.n_edges, np.float64)
                return x.edge_weight.astype(np.float64) * getattr(x, "weight_step", 1.0)
            return GraphBatch(
                np.concatenate([a.graph_ptr, b.graph_ptr[1:] + a.n_nodes]),
                np.concatenate([a.row_ptr, b.row_ptr[1:] + a.n_edges]),
                np.concatenate([a.col_idx, b.col_idx + a.n_nodes]),
                np.concatenate([a.node_label, b.node_label]),
                max(a.n_labels, b.n_labels), None, 1.0, np.concatenate([as_float(a, fa), as_float(b, fb)]), fd)
        if a.edge_weight is not None and b.edge_weight is not None:
            # one weight unit for the union: the finer of the two steps (both are powers of two)
            sa, sb = getattr(a, "weight_step", 1.0), getattr(b, "weight_step", 1.0)
            step = min(sa, sb)
            wa = a.edge_weight.astype(np.int64) * int(round(sa / step))
            wb = b.edge_weight.astype(np.int64) * int(round(sb / step))
            for w in (wa, wb):
                if w.size and int(w.max()) >= MAX_EDGE_WEIGHT:
                    raise NotImplementedError('edge weights of the fitted and the target graphs need more than '
                                              '20 bits at their common power-of-two step')
            ew = np.concatenate([wa, wb])
        return GraphBatch(
            np.concatenate([a.graph_ptr, b.graph_ptr[1:] + a.n_nodes]),
            np.concatenate([a.row_ptr, b.row_ptr[1:] + a.n_edges]),
            np.concatenate([a.col_idx, b.col_idx + a.n_nodes]),
            np.concatenate([a.node_label, b.node_label]),
            max(a.n_labels, b.n_labels), ew, step, None, fd)


MAX_EDGE_WEIGHT = 2 ** 20        # int32 distances: sp.hip guards (n - 1) * max weight < SP_INF per batch
MAX_FLOAT_WEIGHT_NODES = 143     # general float weights: up to here the float64 distance matrix of a graph lives in LDS, larger
                                 # graphs work on it in HBM, one workgroup per graph (sp.hip: sp_f64_big_kernel) -- slower, no limit


def quantise_weights(weights):
    """Positive edge weights of a whole batch (list of arrays) -> (list of int64 arrays, step).

    The reference keys ShortestPath features by the float distance itself (shortest_path.py:389,
    graph.py:1767-1794: float sums of the weights along a path), so two pairs share a feature iff their
    float sums are equal.  When every weight is an integer multiple of one power of two ``step`` and the
    multiples stay below 2**20, every path sum is exact in float64 and equals ``step`` times the integer
    sum: integer distances on the device then give exactly the reference's equalities (and
    ``d * step`` is the reference's key).  Integral inputs keep step 1.  Anything else (0.1, 1/3, ...:
    the reference's own sums then depend on rounding and on the order of addition) is declined.
    """
    flat = np.concatenate([np.asarray(w, np.float64).ravel() for w in weights]) if weights else np.zeros(0)
    if flat.size == 0:
        return [np.zeros(0, np.int64) for _ in weights], 1.0
    if not np.all(np.isfinite(flat)) or flat.min() <= 0:
        raise NotImplementedError('ShortestPath on MI355X supports positive finite edge weights only')
    if np.array_equal(np.rint(flat), flat):
        step_exp = 0
    else:
        m, e = np.frexp(flat)                                 # flat = m * 2**e, 0.5 <= m < 1
        M = np.ldexp(m, 53).astype(np.int64)                  # the 53-bit mantissa as an integer
        tz = np.log2((M & -M).astype(np.float64)).astype(np.int64)
        step_exp = int((e.astype(np.int64) - 53 + tz).min())  # exponent of the lowest set bit of any weight
    out = []
    for w in weights:
        k = np.ldexp(np.asarray(w, np.float64), -step_exp)
        ki = np.rint(k).astype(np.int64)
        if k.size and (not np.array_equal(ki, k) or ki.max() >= MAX_EDGE_WEIGHT):
            raise NotImplementedError(
                'ShortestPath on MI355X supports edge weights that are integers below 2**20, or float weights '
                'that are integer multiples (below 2**20) of one common power of two (0.5, 1.25, ...); other '
                'float weights make the reference use rounded float sums as dictionary keys (SURVEY.md 7.5c)')
        out.append(ki)
    return out, float(np.ldexp(1.0, step_exp))


# ------------------------------------------------------------------------------------------
# element validation shared by the three kernels
# ------------------------------------------------------------------------------------------
def _is_graph_object(x):
    return hasattr(x, "get_edge_dictionary") and hasattr(x, "get_labels")


def iter_elements(X, len_ok, type_msg, not_iterable=TypeError, element_error=TypeError):
    """Yield validated elements; empty elements warn and are skipped (reference behaviour)."""
    if not isinstance(X, Iterable):
        raise not_iterable('input must be an iterable\n')
    n = 0
    for idx, x in enumerate(iter(X)):
        if _is_graph_object(x):
            n += 1
            yield x
            continue
        if not isinstance(x, Iterable):
            raise element_error(type_msg)
        x = list(x)
        if len(x) == 0:
            warnings.warn('Ignoring empty element on index: ' + str(idx))
            continue
        if not len_ok(len(x)):
            raise element_error(type_msg)
        n += 1
        yield x
    if n == 0:
        raise ValueError('parsed input is empty')


# ------------------------------------------------------------------------------------------
# graph object -> (sources, destinations[, weights]) in the graph's own vertex symbols
# ------------------------------------------------------------------------------------------
def _adjacency_array(g):
    if isinstance(g, np.ndarray) and g.ndim == 2:
        return g
    if issparse(g):
        return np.asarray(g.todense())
    if type(g) is list and all(isinstance(r, list) and
                               all(isinstance(i, numbers.Number) for i in r) for r in g):
        return np.array(g)
    return None


def _edge_lists(g):
    """Normalise an edge-dictionary style object to {u: {v: w}} (weights kept for SP).

    Detection order follows grakel/graph.py:1613-1705.  Returns (vertices_set, nested dict,
    entries) or None when the object is not a supported edge dictionary; ``entries`` is the key
    set of the reference's edge dictionary (a ``{v: []}`` item of a dict of lists leaves no
    entry unless v is somebody's neighbour only, graph.py:1640-1660).
    """
    nested = None
    listy = False
    if type(g) is dict:
        if all(type(k) is tuple and len(k) == 2 and isinstance(w, numbers.Number)
               for k, w in g.items()):
            nested = dict()
            for (a, b), w in g.items():
                nested.setdefault(a, dict())[b] = w
            keys = {k[0] for k in g}
        elif all(isinstance(d, list) for d in g.values()):
            nested = {a: dict.fromkeys(lst, 1.) for a, lst in g.items() if len(lst)}
            keys = set(g.keys())
            listy = True
        elif all(isinstance(d, dict) and all(isinstance(w, numbers.Number) for w in d.values())
                 for d in g.values()):
            nested = {a: d for a, d in g.items()}
            keys = set(g.keys())
    if nested is None:
        try:
            items = list(g)
        except TypeError:
            return None
        if all(type(t) is tuple and len(t) == 2 for t in items):
            nested = dict()
            for a, b in items:
                nested.setdefault(a, dict())[b] = 1.
        elif all(type(t) is tuple and len(t) == 3 for t in items):
            nested = dict()
            for a, b, w in items:
                nested.setdefault(a, dict())[b] = w
        else:
            return None
        keys = set(nested.keys())
    vertices = set(keys)
    for d in nested.values():
        vertices.update(d.keys())
    entries = (set(nested.keys()) | (vertices - keys)) if listy else vertices
    return vertices, nested, entries


def _unsupported():
    return ValueError('Unsupported input type. For more information check the documentation, '
                      'concerning valid input types for graph type object.')


# ------------------------------------------------------------------------------------------
# WL / VH ingestion: nodes are the LABELLED vertices (weisfeiler_lehman.py:234: `for v in L[j]`)
# ------------------------------------------------------------------------------------------
def _wl_graph_arrays(gobj, labels):
    """-> (label_values list, src idx array, dst idx array) with node index = position in labels."""
    if not isinstance(labels, dict):
        raise TypeError('node labels must be a dictionary')
    if not labels:
        raise ValueError('Graph does not have any labels for vertices.')     # graph.py:737-738
    n = len(labels)
    keys = list(labels.keys())
    identity = keys == list(range(n))
    A = _adjacency_array(gobj)
    if A is not None:
        if A.shape[0] != A.shape[1]:
            raise ValueError('input matrix must be squared')
        ii, jj = np.nonzero(A > 0)                       # graph.py:963-965
        if identity and A.shape[0] == n:
            return list(labels.values()), ii, jj
        pos = {k: i for i, k in enumerate(keys)}
        keep = np.fromiter((i in pos for i in ii.tolist()), bool, len(ii))
        ii, jj = ii[keep], jj[keep]
        try:
            jmap = np.fromiter((pos[j] for j in jj.tolist()), np.int64, len(jj))
        except KeyError as e:
            raise KeyError(e.args[0])                    # unlabelled neighbour: weisfeiler_lehman.py:238
        imap = np.fromiter((pos[i] for i in ii.tolist()), np.int64, len(ii))
        return list(labels.values()), imap, jmap
    if type(gobj) is dict and identity and len(gobj) and all(type(l) is list for l in gobj.values()):
        # fast path: {u: [v, ...]} over vertices 0..n-1 (what graph generators emit)
        try:
            gkeys = np.fromiter(gobj.keys(), np.int64, len(gobj))
            lens = np.fromiter(map(len, gobj.values()), np.int64, len(gobj))
            flat = np.fromiter(chain.from_iterable(gobj.values()), np.int64, int(lens.sum()))
            ok = (gkeys.min() >= 0 and gkeys.max() < n and
                  (flat.size == 0 or (flat.min() >= 0 and flat.max() < n)))
        except (TypeError, ValueError):
            ok = False
        if ok:
            return list(labels.values()), np.repeat(gkeys, lens), flat
    if _is_graph_object(gobj):
        nested = gobj.get_edge_dictionary()
    else:
        r = _edge_lists(gobj)
        if r is None:
            raise _unsupported()
        nested = r[1]
    pos = {k: i for i, k in enumerate(keys)}
    src, dst = [], []
    for v, i in pos.items():
        d = nested.get(v)
        if d:
            for nb in d.keys():
                src.append(i)
                dst.append(pos[nb])                       # KeyError like the reference
    return list(labels.values()), np.asarray(src, np.int64), np.asarray(dst, np.int64)


def _wloa_entry_mask(gobj, labels):
    """Positions (in label order) of the vertices WL-OA keeps: those with an entry in the
    reference's edge dictionary (weisfeiler_lehman_optimal_assignment.py:176 iterates
    ``Gs_ed[j].keys()``).  An entry without a label is the reference's KeyError (:177)."""
    pos = {k: i for i, k in enumerate(labels.keys())}
    A = _adjacency_array(gobj)
    if A is not None:
        entries = range(A.shape[0])                        # graph.py:960: every index has an entry
    elif _is_graph_object(gobj):
        entries = gobj.get_edge_dictionary().keys()
    else:
        entries = _edge_lists(gobj)[2]
    mask = np.zeros(len(pos), bool)
    for v in entries:
        mask[pos[v]] = True                                # KeyError like the reference
    return mask


def wloa_batch_from_input(X, fitted_labels=None, fit=True):
    """WL-OA ingestion: the WL batch restricted to the vertices that own an edge-dictionary entry
    (labelled vertices without one never reach the histogram, :176,:201-206).  Level-0 ids are
    still numbered over ALL label values like the reference (:147-155).  A ready GraphBatch is
    taken as adjacency-style input: every vertex owns an entry."""
    if isinstance(X, GraphBatch):
        return X, None
    if _gk_ingest is not None and type(X) in (list, tuple):
        # plain `[edge dict, label dict, ...]` elements: the walk and the entry flags in C (csrc/ingest.c);
        # the restriction to the vertices with an entry is then one vectorised pass over the batch
        r = _gk_ingest.wl_ingest(X, 2, True, 0 if fit else 3)
        if r is not None:
            sizes, row_ptr, col, values, mask = r
            ids, mapping = compress_labels(values, fitted_labels)
            n_labels = (len(fitted_labels) + len(mapping)) if fitted_labels is not None else len(mapping)
            sizes = np.frombuffer(sizes, dtype=np.int32)
            row_ptr = np.frombuffer(row_ptr, dtype=np.int32)
            col = np.frombuffer(col, dtype=np.int32)
            mask = np.frombuffer(mask, dtype=np.uint8).astype(bool)
            graph_ptr = np.zeros(len(sizes) + 1, dtype=np.int64)
            np.cumsum(sizes, out=graph_ptr[1:])
            if not mask.all():
                new = np.cumsum(mask, dtype=np.int64) - 1
                src = np.repeat(np.arange(len(mask), dtype=np.int64), np.diff(row_ptr))
                kept = np.concatenate([[0], np.cumsum(mask, dtype=np.int64)])
                graph_ptr = kept[graph_ptr]                     # vertices with an entry before each graph
                row_ptr, col, _ = _csr_from_global(int(graph_ptr[-1]), new[src], new[col.astype(np.int64)])
                ids = ids[mask]
            return GraphBatch(graph_ptr, row_ptr, col, np.ascontiguousarray(ids), max(n_labels, 1)), mapping
    if fit:
        len_ok, err = (lambda n: n >= 2), TypeError
        msg = ('each element of X must be either a graph object or a list with at least a graph '
               'like object and node labels dict \n')
    else:
        len_ok, err = (lambda n: n in (2, 3)), ValueError
        msg = 'each element of X must have at least one and at most 3 elements\n'
    sizes, srcs, dsts, values, masks = [], [], [], [], []
    for x in iter_elements(X, len_ok, msg, err, element_error=err):
        if _is_graph_object(x):
            if hasattr(x, "desired_format"):
                x.desired_format("dictionary")
            gobj, labels = x, x.get_labels(purpose="dictionary")
        else:
            gobj, labels = x[0], x[1]
        vals, s, d = _wl_graph_arrays(gobj, labels)
        mask = _wloa_entry_mask(gobj, labels)
        if not mask.all():
            new = np.cumsum(mask) - 1
            s, d = new[s], new[d]                          # edges only join vertices with entries
        sizes.append(int(mask.sum())), srcs.append(s), dsts.append(d)
        values.extend(vals), masks.append(mask)
    ids, mapping = compress_labels(values, fitted_labels)
    ids = ids[np.concatenate(masks)] if masks else ids
    n_labels = (len(fitted_labels) + len(mapping)) if fitted_labels is not None else len(mapping)
    graph_ptr, row_ptr, col, _ = _pack_csr(sizes, srcs, dsts)
    return GraphBatch(graph_ptr, row_ptr, col, np.ascontiguousarray(ids), max(n_labels, 1)), mapping


def compress_labels(values, fitted=None):
    """Level-0 label compression (weisfeiler_lehman.py:199-210, transform :417-418).

    values: flat python list of label objects.  fit: ids follow ``sorted(distinct)``;
    transform: fitted ids are reused, unseen values get ids >= len(fitted) in sorted order.
    Returns (int32 ids, mapping dict label -> id used for FIT (or the extension for transform)).
    """
    arr = None
    if isinstance(values, (bytearray, bytes)):          # csrc/ingest.c: every label is an exact int64
        arr = np.frombuffer(values, dtype=np.int64)
        values = arr
    elif len(values) and isinstance(values[0], (int, np.integer)) and not isinstance(values[0], bool):
        # (a list that starts with anything else cannot become an integer array: skip the million-element conversion)
        try:
            cand = np.asarray(values)
            if cand.ndim == 1 and cand.dtype.kind in "iu" and len(values) == cand.shape[0]:
                arr = cand
        except Exception:
            arr = None
    if arr is not None and fitted is not None and not all(type(k) is int for k in fitted):
        values = arr.tolist()                            # the generic look-up below wants Python objects
    if fitted is None:
        if arr is not None:
            uniq, inv = _unique_inverse(arr)
            return inv.astype(np.int32), {int(u): i for i, u in enumerate(uniq.tolist())}
        mapping = {dv: i for i, dv in enumerate(sorted(set(values)))}
        return np.fromiter(map(mapping.__getitem__, values), np.int32, len(values)), mapping
    nl = len(fitted)
    if arr is not None and arr.size and all(type(k) is int for k in fitted):
        # integer labels: the look-up as a sorted search instead of a million dictionary probes
        keys = np.fromiter(fitted.keys(), np.int64, nl) if nl else np.zeros(0, np.int64)
        kid = np.fromiter(fitted.values(), np.int64, nl) if nl else np.zeros(0, np.int64)
        order = np.argsort(keys, kind="stable")
        keys, kid = keys[order], kid[order]
        a64 = arr.astype(np.int64, copy=False)
        pos = np.searchsorted(keys, a64)
        pos_c = np.minimum(pos, max(nl - 1, 0))
        seen = (keys[pos_c] == a64) if nl else np.zeros(a64.shape, bool)
        ids = (kid[pos_c] if nl else np.zeros(a64.shape[0], np.int64)).astype(np.int32)
        ext = {}
        if not seen.all():
            unseen = ~seen
            fresh, inv = _unique_inverse(a64[unseen])         # sorted distinct unseen values
            ids[unseen] = nl + inv
            ext = {int(u): nl + i for i, u in enumerate(fresh.tolist())}
        return ids, ext
    fresh = sorted(set(values).difference(fitted))
    ext = {dv: nl + i for i, dv in enumerate(fresh)}
    both = dict(fitted)
    both.update(ext)                                          # one C-level look-up per value
    ids = np.fromiter(map(both.__getitem__, values), np.int32, len(values))
    return ids, ext


def _unique_inverse(arr):
    """np.unique(arr, return_inverse=True) for an integer array; a counting pass instead of a sort when the
    values span a small non-negative range (label alphabets usually do)."""
    if arr.size:
        lo, hi = int(arr.min()), int(arr.max())
        if lo >= 0 and hi < (1 << 22):
            present = np.zeros(hi + 1, bool)
            present[arr] = True
            uniq = np.flatnonzero(present)
            rank = np.cumsum(present) - 1
            return uniq.astype(arr.dtype, copy=False), rank[arr]
    return np.unique(arr, return_inverse=True)


def _pack_csr(n_nodes_per_graph, srcs, dsts, weights=None):
    """Concatenate per-graph local (src,dst) arrays into one global, de-duplicated CSR."""
    sizes = np.asarray(n_nodes_per_graph, dtype=np.int64)
    graph_ptr = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=graph_ptr[1:])
    V = int(graph_ptr[-1])
    if V >= 2 ** 31 - 1:
        raise ValueError("batch too large for int32 node indices")
    offs = np.repeat(graph_ptr[:-1], [len(s) for s in srcs]) if srcs else np.zeros(0, np.int64)
    src = (np.concatenate(srcs) if srcs else np.zeros(0, np.int64)) + offs
    dst = (np.concatenate(dsts) if dsts else np.zeros(0, np.int64)) + offs
    w = np.concatenate(weights) if weights is not None and weights else None
    row_ptr, dst, w = _csr_from_global(V, src, dst, w)
    return graph_ptr, row_ptr, dst, w


def _csr_from_global(V, src, dst, w=None):
    """Global (src, dst[, w]) edge arrays over V nodes -> (row_ptr, col_idx, w), sorted and de-duplicated."""
    key = src * np.int64(V if V > 0 else 1) + dst
    if key.size and not np.all(key[1:] > key[:-1]):
        # not already (src,dst)-sorted & unique: sort and collapse duplicates (dict semantics:
        # a repeated edge is one key; for weights the LAST occurrence wins like a dict assignment)
        order = np.argsort(key, kind="stable")
        key, src, dst = key[order], src[order], dst[order]
        last = np.ones(key.size, bool)
        last[:-1] = key[1:] != key[:-1]
        src, dst = src[last], dst[last]
        if w is not None:
            w = w[order][last]
    row_ptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=V), out=row_ptr[1:])
    return row_ptr, dst, w


def wl_batch_from_input(X, fitted_labels=None, min_len=2, not_iterable=TypeError):
    """Ingest the WL / VH input -> (GraphBatch, label mapping)."""
    if isinstance(X, GraphBatch):
        return X, None
    if _gk_ingest is not None and type(X) in (list, tuple):
        # plain `[edge dict, label dict, ...]` elements: the same walk in C; anything it does not
        # recognise makes it return None and the Python path below takes the whole input
        r = _gk_ingest.wl_ingest(X, int(min_len), False, 0, int(INGEST_THREADS))
        if r is not None:
            sizes, row_ptr, col, values = r
            ids, mapping = compress_labels(values, fitted_labels)
            n_labels = (len(fitted_labels) + len(mapping)) if fitted_labels is not None else len(mapping)
            sizes = np.frombuffer(sizes, dtype=np.int32)
            graph_ptr = np.zeros(len(sizes) + 1, dtype=np.int64)
            np.cumsum(sizes, out=graph_ptr[1:])
            return GraphBatch(graph_ptr, np.frombuffer(row_ptr, dtype=np.int32), np.frombuffer(col, dtype=np.int32),
                              ids, max(n_labels, 1)), mapping
    msg = ('each element of X must be either a graph object or a list with at least a graph '
           'like object and node labels dict \n')
    sizes, srcs, dsts, values = [], [], [], []
    for x in iter_elements(X, lambda n: n >= min_len, msg, not_iterable):
        if _is_graph_object(x):
            if hasattr(x, "desired_format"):
                x.desired_format("dictionary")
            gobj, labels = x, x.get_labels(purpose="dictionary")
        else:
            gobj, labels = x[0], x[1]
        vals, s, d = _wl_graph_arrays(gobj, labels)
        sizes.append(len(vals)), srcs.append(s), dsts.append(d)
        values.extend(vals)
    ids, mapping = compress_labels(values, fitted_labels)
    n_labels = (len(fitted_labels) + len(mapping)) if fitted_labels is not None else len(mapping)
    graph_ptr, row_ptr, col, _ = _pack_csr(sizes, srcs, dsts)
    return GraphBatch(graph_ptr, row_ptr, col, ids, max(n_labels, 1)), mapping


def vh_batch_from_input(X, fitted_labels=None, edge_labels=False):
    """VertexHistogram only reads x[1].values() (vertex_histogram.py:96,107): no edges.
    ``edge_labels``: EdgeHistogram reads x[2].values() of 3-element inputs instead
    (edge_histogram.py:93-96); the "nodes" of the batch are then the labelled edges."""
    if isinstance(X, GraphBatch):
        return X, None
    msg = ('each element of X must be either a graph object or a list with at least a graph '
           'like object and node labels dict \n')
    sizes, values = [], []
    len_ok = (lambda n: n == 3) if edge_labels else (lambda n: n in (2, 3))
    for x in iter_elements(X, len_ok, msg):
        if _is_graph_object(x):
            L = x.get_labels(purpose="any", label_type="edge") if edge_labels else x.get_labels(purpose="any")
        else:
            L = x[2] if edge_labels else x[1]
        vals = list(L.values())
        sizes.append(len(vals))
        values.extend(vals)
    ids, mapping = compress_labels(values, fitted_labels)
    n_labels = (len(fitted_labels) + len(mapping)) if fitted_labels is not None else len(mapping)
    gp = np.zeros(len(sizes) + 1, np.int64)
    np.cumsum(sizes, out=gp[1:])
    V = int(gp[-1])
    return GraphBatch(gp, np.zeros(V + 1, np.int64), np.zeros(0, np.int64), ids, max(n_labels, 1)), mapping


# ------------------------------------------------------------------------------------------
# ShortestPath ingestion: nodes are ALL vertices of the graph, in the order the reference
# indexes them (sorted symbols for edge dictionaries, graph.py:894-907; 0..n-1 for matrices)
# ------------------------------------------------------------------------------------------
def _sp_graph_arrays(gobj, labels, with_labels):
    """-> (n, label_values list or None, src, dst, weight, is_dict) with local vertex indices; is_dict: the reference
    holds this element in its dictionary format (and runs dijkstra on it under algorithm_type="auto", graph.py:652-656)."""
    A = _adjacency_array(gobj)
    is_dict = _sp_from_dict_flag(gobj)
    if A is not None:
        if A.shape[0] != A.shape[1]:
            raise ValueError('input matrix must be squared')
        if np.any(A < 0):
            raise NotImplementedError('ShortestPath on MI355X needs non-negative edge weights')
        n = A.shape[0]
        ii, jj = np.nonzero(A)                      # floyd_warshall: zero == no edge (graph.py:1785)
        ww = A[ii, jj]
        verts = range(n)
    else:
        if _is_graph_object(gobj):
            nested = gobj.get_edge_dictionary()
            vertices = set(nested.keys())
            for d in nested.values():
                vertices.update(d.keys())
        else:
            r = _edge_lists(gobj)
            if r is None:
                raise _unsupported()
            vertices, nested = r[0], r[1]
        verts = sorted(vertices)
        n = len(verts)
        pos = {v: i for i, v in enumerate(verts)}
        src, dst, wl = [], [], []
        for a, d in nested.items():
            ia = pos[a]
            for b, w in d.items():
                src.append(ia), dst.append(pos[b]), wl.append(w)
        ii, jj, ww = np.asarray(src, np.int64), np.asarray(dst, np.int64), np.asarray(wl)
    try:                                             # integers come later, per batch (quantise_weights)
        ww = np.asarray(ww, np.float64) if ww.size else np.zeros(0, np.float64)
    except (TypeError, ValueError):
        raise NotImplementedError('ShortestPath on MI355X needs numeric edge weights')
    vals = None
    if with_labels:
        if not labels:
            raise ValueError('Graph does not have any labels for vertices.')   # graph.py:737-738
        vals = [labels[v] for v in verts]            # KeyError when a vertex has no label
    return n, vals, ii, jj, ww, is_dict


def _sp_from_dict_flag(g):
    """1 when the reference holds this graph object in its dictionary format (and runs dijkstra on it under
    algorithm_type="auto", graph.py:652-656), 0 for the adjacency forms -- ONE rule for every ingestion path."""
    if isinstance(g, np.ndarray) or issparse(g) or type(g) is bytes:
        return 0
    if _is_graph_object(g):
        return 0 if getattr(g, "_format", None) == "adjacency" else 1
    if type(g) is list and len(g) and type(g[0]) is list:          # rows of numbers: a matrix (graph.py:1564-1580)
        return 0 if _adjacency_array(g) is not None else 1
    return 1


def _sp_first_shows_weights(X):
    """Does the FIRST element already show edge weights other than 1?  Then the unit-weight walk below would decline after
    a pass over (part of) the input and the weighted walk would start over: decide the form once."""
    try:
        g = X[0][0]
        if isinstance(g, np.ndarray):
            return bool(g.size) and (g.dtype.kind not in "biu" or int(g.max()) > 1 or int(g.min()) < 0)
        if issparse(g):                                            # CSR matrices of 0 / 1 entries take the walk (round 6)
            d = g.data
            return g.format != "csr" or (bool(d.size) and (d.dtype.kind not in "biu" or int(d.max()) > 1 or int(d.min()) < 0))
        if type(g) is dict:
            for k, v in g.items():
                if type(v) is dict:
                    return any(w != 1 for w in v.values()) or type(next(iter(v.values()), 1)) is float
                if type(k) is tuple:
                    return any(w != 1 for w in g.values())
                break
    except Exception:
        pass
    return False


def sp_batch_from_input(X, with_labels, fitted_labels=None, len_ok=None, not_iterable=TypeError):
    if isinstance(X, GraphBatch):
        return X, None
    if _gk_ingest is not None and with_labels and type(X) in (list, tuple) and len(X) and not _sp_first_shows_weights(X):
        # round 5: unit-weight graphs whose vertex set IS the label keys in their (sorted) order -- dict of neighbour lists over
        # 0 .. n-1, `(u, v)`-tuple sets / lists / dicts (the fetch_dataset form), 0/1 adjacency matrices -- take the threaded
        # walks of the WL ingestion (csrc/ingest.c, sp_mode); None: weights, unlabelled or unsorted vertices, ... -> below
        try:
            r = _gk_ingest.wl_ingest(X, 2, False, 0 if len_ok is not None else 3, int(INGEST_THREADS), 1)
        except TypeError:                      # an older build of the module without the sp_mode argument
            r = None
        if r is not None:
            sizes, row_ptr, col, values = r
            ids, mapping = compress_labels(values, fitted_labels)
            n_labels = (len(fitted_labels) + len(mapping)) if fitted_labels is not None else len(mapping)
            sizes = np.frombuffer(sizes, dtype=np.int32)
            graph_ptr = np.zeros(len(sizes) + 1, dtype=np.int64)
            np.cumsum(sizes, out=graph_ptr[1:])
            col = np.frombuffer(col, dtype=np.int32)
            fd = np.fromiter((_sp_from_dict_flag(x[0]) for x in X), np.uint8, len(X))
            return GraphBatch(graph_ptr, np.frombuffer(row_ptr, dtype=np.int32), col, ids, max(n_labels, 1),
                              edge_weight=np.ones(col.shape[0], np.int32), from_dict=fd), mapping
    if _gk_ingest is not None and len_ok is None and type(X) in (list, tuple) and hasattr(_gk_ingest, "sp_ingest"):
        # adjacency arrays / int-keyed edge dictionaries with integer weights: the same walk in C
        # (csrc/ingest.c); None = not recognised, the Python path below takes the whole input
        r = _gk_ingest.sp_ingest(X, bool(with_labels), 2 if with_labels else 1, 3)
        if r is not None:
            sizes, row_ptr, col, w, values = r
            sizes = np.frombuffer(sizes, dtype=np.int32)
            graph_ptr = np.zeros(len(sizes) + 1, dtype=np.int64)
            np.cumsum(sizes, out=graph_ptr[1:])
            if with_labels:
                ids, mapping = compress_labels(values, fitted_labels)
                n_labels = (len(fitted_labels) + len(mapping)) if fitted_labels is not None else len(mapping)
            else:
                ids, mapping, n_labels = np.zeros(int(graph_ptr[-1]), np.int32), {}, 1
            fd = np.fromiter((_sp_from_dict_flag(x[0]) for x in X), np.uint8, len(X))
            return GraphBatch(graph_ptr, np.frombuffer(row_ptr, dtype=np.int32), np.frombuffer(col, dtype=np.int32), ids,
                              max(n_labels, 1), edge_weight=np.frombuffer(w, dtype=np.int32), from_dict=fd), mapping
    msg = 'each element of X must have at least one and at most 3 elements\n'
    ok = (lambda n: n in (2, 3)) if with_labels else (lambda n: n in (1, 2, 3))
    if len_ok is not None:
        ok = len_ok
    sizes, srcs, dsts, wts, values, dicts = [], [], [], [], [], []
    for x in iter_elements(X, ok, msg, not_iterable):
        if _is_graph_object(x):
            gobj, labels = x, (x.get_labels(purpose="dictionary") if with_labels else {})
        else:
            gobj, labels = x[0], (x[1] if len(x) > 1 else {})
        n, vals, s, d, w, is_dict = _sp_graph_arrays(gobj, labels, with_labels)
        sizes.append(n), srcs.append(s), dsts.append(d), wts.append(w), dicts.append(is_dict)
        if with_labels:
            values.extend(vals)
    if with_labels:
        ids, mapping = compress_labels(values, fitted_labels)
        n_labels = (len(fitted_labels) + len(mapping)) if fitted_labels is not None else len(mapping)
    else:
        ids, mapping, n_labels = np.zeros(int(np.sum(sizes)), np.int32), {}, 1
    fd = np.asarray(dicts, np.uint8)
    try:
        qw, step = quantise_weights(wts)
    except NotImplementedError:
        # general positive float weights: the device reproduces the reference's float distances (GraphBatch.float_weight)
        flat = np.concatenate(wts) if wts else np.zeros(0)
        if flat.size and (not np.all(np.isfinite(flat)) or flat.min() <= 0 or flat.max() >= 1e300):
            raise
        graph_ptr, row_ptr, col, w = _pack_csr(sizes, srcs, dsts, wts)
        return GraphBatch(graph_ptr, row_ptr, col, ids, max(n_labels, 1), float_weight=w, from_dict=fd), mapping
    graph_ptr, row_ptr, col, w = _pack_csr(sizes, srcs, dsts, qw)
    if w is None:
        w = np.zeros(0, np.int64)
    return GraphBatch(graph_ptr, row_ptr, col, ids, max(n_labels, 1), edge_weight=w, weight_step=step, from_dict=fd), mapping

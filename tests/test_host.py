"""CPU tests: host ingestion, estimator plumbing, and the C-ABI surface (no compute)."""
import os
import pickle
import re

import numpy as np
import pytest

from conftest import ROOT, reference_available
from golden.small_sets import SMALL_SETS
from oracle import grakel_oracle as O
import grakel_amd
from grakel_amd import GraphBatch, ShortestPath, VertexHistogram, WeisfeilerLehman
from grakel_amd.batch import (compress_labels, sp_batch_from_input, vh_batch_from_input,
                              wl_batch_from_input)
from grakel_amd.synthetic import er_dataset, er_dataset_csr, random_labelled_graphs


def _adjacency_sets(gb):
    out = []
    for v in range(gb.n_nodes):
        out.append(sorted(gb.col_idx[gb.row_ptr[v]:gb.row_ptr[v + 1]].tolist()))
    return out


@pytest.mark.parametrize("name", [n for n, _ in SMALL_SETS])
def test_wl_ingestion_matches_reference_semantics(name):
    """Nodes = labelled vertices, neighbours = keys of the edge dictionary (oracle.parse_graph
    restates grakel/graph.py); label ids = rank in sorted(distinct labels)."""
    X = random_labelled_graphs(**dict(SMALL_SETS)[name])
    gb, mapping = wl_batch_from_input(X)
    assert gb.n_graphs == len(X)
    base = 0
    want_adj, want_lab = [], []
    for x in X:
        g = O.parse_graph(x[0], x[1])
        keys = list(x[1].keys())
        pos = {k: base + i for i, k in enumerate(keys)}
        for k in keys:
            want_adj.append(sorted(pos[nb] for nb in g.edges.get(k, {}).keys()))
            want_lab.append(x[1][k])
        base += len(keys)
    assert _adjacency_sets(gb) == want_adj
    ranks = {l: i for i, l in enumerate(sorted(set(want_lab)))}
    assert gb.node_label.tolist() == [ranks[l] for l in want_lab]
    assert mapping == ranks


def test_all_edge_dictionary_forms_give_the_same_batch():
    lab = {0: 'x', 1: 'y', 2: 'x', 3: 'z'}
    forms = [
        {0: [1, 2], 1: [0], 2: [0, 3], 3: [2]},
        {0: {1: 1.0, 2: 1.0}, 1: {0: 1.0}, 2: {0: 1.0, 3: 1.0}, 3: {2: 1.0}},
        {(0, 1): 1, (0, 2): 1, (1, 0): 1, (2, 0): 1, (2, 3): 1, (3, 2): 1},
        [(0, 1), (0, 2), (1, 0), (2, 0), (2, 3), (3, 2)],
        [(0, 1, 2.0), (0, 2, 1.0), (1, 0, 1), (2, 0, 1), (2, 3, 1), (3, 2, 1)],
        np.array([[0, 1, 1, 0], [1, 0, 0, 0], [1, 0, 0, 1], [0, 0, 1, 0]]),
        [[0, 1, 1, 0], [1, 0, 0, 0], [1, 0, 0, 1], [0, 0, 1, 0]],
    ]
    from scipy.sparse import csr_matrix
    forms.append(csr_matrix(forms[-2]))
    ref = None
    for f in forms:
        gb, _ = wl_batch_from_input([[f, lab]])
        cur = (_adjacency_sets(gb), gb.node_label.tolist())
        ref = ref or cur
        assert cur == ref
    # duplicated neighbours collapse (dictionary keys), symbols may be anything hashable
    gb, _ = wl_batch_from_input([[{'a': ['b', 'b', 'c'], 'b': ['a'], 'c': ['a']},
                                  {'a': 1, 'b': 2, 'c': 2}]])
    assert _adjacency_sets(gb) == [[1, 2], [0], [0]]
    with pytest.raises(KeyError):                       # unlabelled neighbour: reference KeyError
        wl_batch_from_input([[{0: [1], 1: [0]}, {0: 'a'}]])
    with pytest.raises(ValueError):
        wl_batch_from_input([["nonsense", {0: 'a'}]])


def _both_paths(X, **kw):
    """wl_batch_from_input through the C fast path (csrc/ingest.c) and through the Python path."""
    from grakel_amd import batch as B
    if B._gk_ingest is None:                        # fresh checkout: build the optional module (gcc + Python.h)
        import importlib
        import subprocess
        subprocess.call(["make", "-C", os.path.join(ROOT, "grakel_amd", "csrc"), "ingest"])
        try:
            B._gk_ingest = importlib.import_module("grakel_amd._gk_ingest")
        except ImportError:
            pytest.skip("the C ingestion module cannot be built here (no Python.h?)")

    def run():
        try:
            gb, m = B.wl_batch_from_input(X, **kw)
            return ("ok", gb.graph_ptr.tolist(), gb.row_ptr.tolist(), gb.col_idx.tolist(), gb.node_label.tolist(),
                    gb.n_labels, m)
        except Exception as e:                      # noqa: BLE001 -- the two paths must raise alike
            return ("raise", type(e), e.args)
    fast = run()
    saved, B._gk_ingest = B._gk_ingest, None
    try:
        slow = run()
    finally:
        B._gk_ingest = saved
    return fast, slow


def test_c_ingestion_fast_path_equals_the_python_path():
    rs = np.random.RandomState(5)
    cases = {}
    for name, kw in SMALL_SETS:
        cases[name] = random_labelled_graphs(**kw)
    cases["er"] = er_dataset(40, 30, 0.15, 5, 3)                       # identity numbering, lists
    X = er_dataset(25, 12, 0.3, 3, 9)
    cases["dict_of_dicts"] = [[{u: {v: 1.0 for v in nb} for u, nb in g.items()}, lab] for g, lab in X]
    sym = [[{"v%d" % u: ["v%d" % v for v in nb] for u, nb in g.items()}, {"v%d" % u: l for u, l in lab.items()}]
           for g, lab in X]
    cases["string_vertices"] = sym
    cases["shuffled_label_order"] = [[g, dict(sorted(lab.items(), key=lambda kv: -kv[0]))] for g, lab in X]
    cases["duplicates_and_hubs"] = [[{0: [1, 1, 2] + list(range(3, 60)) * 2, **{i: [0, 0] for i in range(1, 60)}},
                                     {i: int(rs.randint(0, 4)) for i in range(60)}]]
    cases["unlabelled_source_is_ignored"] = [[{0: [1], 1: [0], 9: [0]}, {0: 'a', 1: 'b'}]]
    cases["vertex_without_entry"] = [[{0: [1], 1: [0]}, {0: 'a', 1: 'b', 2: 'c'}]]
    cases["extras_and_tuples"] = [(g, lab, {}, "extra") for g, lab in X]
    cases["mixed_forms_decline"] = X[:3] + [[np.eye(3), {0: 1, 1: 1, 2: 2}]] + sym[:2]
    cases["numpy_int_neighbours"] = [[{0: [np.int64(1)], 1: [np.int64(0)]}, {0: 'a', 1: 'b'}]]
    cases["float_neighbours"] = [[{0: [1.0], 1: [0.0]}, {0: 'a', 1: 'b'}]]
    cases["bool_weights"] = [[{0: {1: True}, 1: {0: True}}, {0: 'a', 1: 'b'}]]
    cases["unlabelled_neighbour"] = [[{0: [1, 7], 1: [0]}, {0: 'a', 1: 'b'}]]
    cases["unlabelled_neighbour_symbols"] = [[{'a': ['b', 'zz'], 'b': ['a']}, {'a': 1, 'b': 2}]]
    cases["empty_labels"] = [[{0: [1], 1: [0]}, {}]]
    cases["empty_element"] = [[{0: [1], 1: [0]}, {0: 1, 1: 2}], []]
    cases["labels_not_a_dict"] = [[{0: [1], 1: [0]}, [1, 2]]]
    cases["negative_neighbour"] = [[{0: [-1], 1: [0]}, {0: 'a', 1: 'b'}]]
    import warnings
    from grakel_amd import batch as B

    def oa(X):
        try:
            gb, m = B.wloa_batch_from_input(X)
            return ("ok", gb.graph_ptr.tolist(), gb.row_ptr.tolist(), gb.col_idx.tolist(), gb.node_label.tolist(), gb.n_labels, m)
        except Exception as e:                      # noqa: BLE001
            return ("raise", type(e), e.args)
    for name, X in cases.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fast, slow = _both_paths(X)
            assert fast == slow, name
            fast = oa(X)                            # WL-OA: same walk + the entry mask
            saved, B._gk_ingest = B._gk_ingest, None
            try:
                slow = oa(X)
            finally:
                B._gk_ingest = saved
            assert fast == slow, "wloa " + name
    fitted = {'a': 0, 'b': 1}
    fast, slow = _both_paths(cases["unlabelled_source_is_ignored"] + [[{0: [1], 1: [0]}, {0: 'b', 1: 'q'}]],
                             fitted_labels=fitted)
    assert fast == slow and fast[0] == "ok"
    assert _both_paths(cases["unlabelled_neighbour"])[0][:2] == ("raise", KeyError)


def test_threaded_ingestion_equals_the_one_thread_walk_and_the_python_path():
    """csrc/ingest.c: wl_ingest_threads takes inputs of 256 or more elements of the one common form (identity numbering,
    neighbour lists, small int labels); anything else falls back to the one-thread walk, which falls back to Python.
    Whatever the route, the batch -- or the exception -- is the same."""
    from grakel_amd import batch as B
    rs = np.random.RandomState(11)
    base = er_dataset(400, 12, 0.3, 4, 2)
    cases = {"common_form": base}
    messy = [[{u: list(nb) for u, nb in g.items()}, dict(lab)] for g, lab in base]
    for g, _ in messy[::7]:
        for u in g:
            g[u] = g[u][::-1] + g[u][:1]                      # descending, with a duplicate: sorted and deduplicated
    cases["unsorted_duplicates"] = messy
    cases["string_label_at_the_end"] = base[:-1] + [[base[-1][0], {u: "x%d" % l for u, l in base[-1][1].items()}]]
    cases["big_and_negative_labels"] = [[g, {u: (l - 2) * 2 ** 40 for u, l in lab.items()}] for g, lab in base]
    cases["dict_of_dicts_in_the_middle"] = base[:200] + [[{u: {v: 1 for v in nb} for u, nb in base[200][0].items()}, base[200][1]]] + base[201:]
    cases["tuple_elements_with_extras"] = [(g, lab, None) for g, lab in base]
    cases["unlabelled_neighbour_near_the_end"] = base[:-2] + [[{0: [1, 77], 1: [0]}, {0: 1, 1: 2}]] + base[-2:]
    cases["vertex_without_entry"] = base[:300] + [[{0: [1], 1: [0]}, {0: 1, 1: 2, 2: 3}]] + base[300:]
    cases["shuffled_label_order"] = base[:50] + [[g, dict(sorted(lab.items(), key=lambda kv: -kv[0]))] for g, lab in base[50:60]] + base[60:]
    cases["numpy_neighbours"] = base[:399] + [[{0: [np.int64(1)], 1: [np.int64(0)]}, {0: 3, 1: 1}]]
    saved = B.INGEST_THREADS
    try:
        for name, X in cases.items():
            B.INGEST_THREADS = 4
            fast, slow = _both_paths(X)
            assert fast == slow, name
            B.INGEST_THREADS = 1
            one, _ = _both_paths(X[:1])                       # (builds the module if needed)
            one = _both_paths(X)[0]
            assert one == fast, name
        # the threaded walk itself against the one-thread walk, array by array (the common form is taken by the threads)
        r4 = B._gk_ingest.wl_ingest(base, 2, False, 0, 4)
        r1 = B._gk_ingest.wl_ingest(base, 2, False, 0, 1)
        assert len(r4) == 4 and all(bytes(a) == bytes(b) for a, b in zip(r4, r1))
        rm4 = B._gk_ingest.wl_ingest(messy, 2, False, 0, 3)
        rm1 = B._gk_ingest.wl_ingest(messy, 2, False, 0, 1)
        assert all(bytes(a) == bytes(b) for a, b in zip(rm4, rm1))
    finally:
        B.INGEST_THREADS = saved


def test_pair_and_matrix_ingestion_in_c_equals_the_python_path():
    """Round 5, csrc/ingest.c: the forms real data arrives as -- a set / list of `(u, v)` tuples or a dict keyed by them with
    labels keyed by arbitrary (global) vertex ids, which is what `fetch_dataset` / `read_data` produce
    (datasets/base.py:273-279), and numpy adjacency matrices -- take the threaded C walk; everything unusual (an unlabelled
    neighbour = the reference's KeyError, weighted 3-tuples, string ids, a matrix that is not square or not C-contiguous,
    labels that are not keyed 0 .. n-1) still gets the Python path's batch or exception."""
    from grakel_amd import batch as B
    base = er_dataset(330, 14, 0.3, 4, 6)
    off, sets, lists, dicts, mats = 1, [], [], [], []
    for ed, lab in base:
        n = len(lab)
        es = {(u + off, v + off) for u, l in ed.items() for v in l}
        L = {u + off: lab[u] for u in sorted(lab, key=lambda u: (u * 7) % n)}          # label order != id order
        sets.append([es, L, {e: 0 for e in es}])
        lists.append([sorted(es) + sorted(es)[:2], L])                                 # duplicates collapse
        dicts.append([{e: 2.0 for e in es}, L])
        A = np.zeros((n, n), int)
        for u, l in ed.items():
            A[u, l] = 1
        mats.append([A, lab])
        off += n
    cases = {"sets_global_ids": sets, "frozensets": [[frozenset(g), L] for g, L, _ in sets], "lists": lists,
             "tuples_of_tuples": [(tuple(g), L) for g, L in lists], "dict_of_pairs": dicts, "few_elements": sets[:5],
             "unlabelled_neighbour": sets[:100] + [[{(1, 2), (2, 1), (2, 3)}, {1: 0, 2: 1}]] + sets[100:],
             "unlabelled_source_is_ignored": sets[:3] + [[{(1, 2), (2, 1), (3, 1)}, {1: 0, 2: 1}]],
             "weighted_triples_in_the_middle": lists[:40] + [[[(1, 2, 0.5), (2, 1, 0.5)], {1: 0, 2: 1}]] + lists[40:],
             "string_vertices": sets[:20] + [[{("a", "b"), ("b", "a")}, {"a": 1, "b": 2}]],
             "string_labels": [[g, {k: "L%d" % v for k, v in L.items()}] for g, L, _ in sets[:30]],
             "big_ids": [[{(10 ** 12, 5), (5, 10 ** 12)}, {10 ** 12: 1, 5: 2}]] + sets[:10],
             "empty_edge_set": sets[:4] + [[set(), {1: 0}]],
             "matrices_int64": mats, "matrices_bool": [[a.astype(bool), l] for a, l in mats],
             "matrices_float32_weights_and_negatives": [[(a * 2.5 - (1 - a)).astype(np.float32), l] for a, l in mats],
             "matrices_uint8": [[a.astype(np.uint8), l] for a, l in mats],
             "matrix_fortran_order": mats[:10] + [[np.asfortranarray(mats[10][0]), mats[10][1]]],
             "matrix_not_square": mats[:5] + [[np.ones((3, 4), int), {0: 1, 1: 1, 2: 1}]],
             "matrix_labels_not_identity": mats[:5] + [[mats[5][0], dict(reversed(list(mats[5][1].items())))]],
             "matrix_fewer_labels": mats[:5] + [[np.ones((3, 3), int), {0: 1, 1: 1}]]}
    # round 6: scipy.sparse adjacency matrices (graph.py:1564-1580 takes them as adjacency matrices).  CSR matrices take the
    # threaded walk over indptr / indices / data; other formats, duplicate (non-canonical) entries, weights, explicit zeros,
    # non-square shapes and a mixed input get the Python path's batch or exception
    import scipy.sparse as sps
    csr = [[sps.csr_matrix(a), l] for a, l in mats]
    dup = sps.csr_matrix((np.array([1, 1, 1]), np.array([1, 1, 0]), np.array([0, 2, 3])), shape=(2, 2))      # (0, 1) twice
    zero = sps.csr_matrix(mats[0][0] * 1.0)
    zero.data[0] = 0.0                                                                                       # an explicit zero is no edge
    cases.update({"csr_int64": csr, "csr_float32": [[sps.csr_matrix(a.astype(np.float32)), l] for a, l in mats],
                  "csr_bool": [[sps.csr_matrix(a.astype(bool)), l] for a, l in mats],
                  "csr_int32_indices_int8_data": [[sps.csr_matrix(a.astype(np.int8)), l] for a, l in mats[:64]],
                  "csc_and_coo": [[sps.csc_matrix(a), l] for a, l in mats[:20]] + [[sps.coo_matrix(a), l] for a, l in mats[20:40]],
                  "csr_weights_and_negatives": [[sps.csr_matrix(a * 2.5 - (1 - a)), l] for a, l in mats[:40]],
                  "csr_with_duplicate_entries": csr[:70] + [[dup, {0: 1, 1: 2}]],
                  "csr_with_an_explicit_zero": [[zero, mats[0][1]]] + csr[1:70],
                  "csr_not_square": csr[:5] + [[sps.csr_matrix(np.ones((3, 4), int)), {0: 1, 1: 1, 2: 1}]],
                  "csr_then_dense": csr[:70] + mats[70:80]})
    saved = B.INGEST_THREADS
    try:
        for threads in (0, 1, 3):
            B.INGEST_THREADS = threads
            for name, X in cases.items():
                fast, slow = _both_paths(X)
                assert fast == slow, (name, threads)
        B.INGEST_THREADS = 0
        # ... and they are the graphs of the dict-of-lists form
        ref = _both_paths(base)[0]
        plain_sets = [[{(u + 5, v + 5) for u, l in ed.items() for v in l}, {u + 5: x for u, x in lab.items()}] for ed, lab in base]
        assert _both_paths(plain_sets)[0][1:5] == ref[1:5] and _both_paths(mats)[0][1:5] == ref[1:5]
        # the C module really took them (None = declined)
        assert B._gk_ingest.wl_ingest(sets, 2, False, 0, 0) is not None and B._gk_ingest.wl_ingest(mats, 2, False, 0, 0) is not None
        assert B._gk_ingest.wl_ingest(cases["string_vertices"], 2, False, 0, 0) is None
        assert B._gk_ingest.wl_ingest(csr, 2, False, 0, 0) is not None and _both_paths(csr)[0][1:5] == ref[1:5]
        assert B._gk_ingest.wl_ingest(cases["csc_and_coo"], 2, False, 0, 0) is None
        assert B._gk_ingest.wl_ingest(cases["csr_with_duplicate_entries"], 2, False, 0, 0) is None
        # ShortestPath's ingestion of unit-weight CSR matrices takes the same walk (sp_mode)
        gs, _ = B.sp_batch_from_input(csr, True)
        gd, _ = B.sp_batch_from_input(mats, True)
        assert np.array_equal(gs.col_idx, gd.col_idx) and np.array_equal(gs.row_ptr, gd.row_ptr) and np.array_equal(gs.from_dict, gd.from_dict)
    finally:
        B.INGEST_THREADS = saved


def test_c_sp_ingestion_fast_path_equals_the_python_path():
    from grakel_amd import batch as B
    from grakel_amd.synthetic import nci1_like
    _both_paths([[{0: [1], 1: [0]}, {0: 1, 1: 2}]])            # builds / loads the C module if needed

    def run(X, with_labels, **kw):
        try:
            gb, m = B.sp_batch_from_input(X, with_labels, **kw)
            return ("ok", gb.graph_ptr.tolist(), gb.row_ptr.tolist(), gb.col_idx.tolist(), gb.node_label.tolist(),
                    gb.edge_weight.tolist(), gb.n_labels, m)
        except Exception as e:                      # noqa: BLE001
            return ("raise", type(e), e.args)

    def both(X, with_labels=True, **kw):
        fast = run(X, with_labels, **kw)
        saved, B._gk_ingest = B._gk_ingest, None
        try:
            slow = run(X, with_labels, **kw)
        finally:
            B._gk_ingest = saved
        return fast, slow
    rs = np.random.RandomState(2)
    A = (rs.rand(6, 6) < 0.4).astype(np.int64)
    np.fill_diagonal(A, 0)
    lab6 = {i: "abc"[i % 3] for i in range(6)}
    cases = {
        "nci1_adjacency": nci1_like(60, 0, as_adj=True),
        "nci1_dicts": nci1_like(60, 0, as_adj=False),
        "weights_int64": [[A * 3, lab6]],
        "weights_float64_integral": [[(A * 2).astype(np.float64), lab6]],
        "bool_matrix": [[A.astype(bool), lab6]],
        "uint8_int32": [[A.astype(np.uint8), lab6], [A.astype(np.int32), lab6]],
        "float32_declines": [[A.astype(np.float32), lab6]],
        "fortran_order_declines": [[np.asfortranarray(A), lab6]],
        "non_square": [[np.ones((2, 3), np.int64), {0: 1, 1: 2}]],
        "negative_weight": [[-A, lab6]],
        "fractional_weight": [[A * 0.5, lab6]],
        "huge_weight": [[A * (2 ** 20), lab6]],
        "dict_of_dicts_weights": [[{5: {9: 2, 7: 3.0}, 9: {5: 2}, 7: {}}, {5: 'x', 7: 'y', 9: 'x'}]],
        "zero_weight_in_dict": [[{0: {1: 0}, 1: {0: 1}}, {0: 1, 1: 1}]],
        "float_weight_in_dict": [[{0: {1: 0.5}, 1: {0: 1}}, {0: 1, 1: 1}]],
        "vertex_only_as_neighbour": [[{3: [8, 8, 1]}, {1: 'a', 3: 'b', 8: 'c'}]],
        "missing_label": [[{0: [1], 1: [0]}, {0: 'a'}]],
        "missing_label_matrix": [[A, {0: 'a'}]],
        "empty_labels": [[A, {}]],
        "string_vertices_decline": [[{'a': ['b'], 'b': ['a']}, {'a': 1, 'b': 2}]],
        "tuple_keys_decline": [[{(0, 1): 1, (1, 0): 1}, {0: 1, 1: 2}]],
        "list_of_lists_declines": [[A.tolist(), lab6]],
        "extras": [(A, lab6, {}), (A * 2, lab6)],
        "too_long": [(A, lab6, {}, "x")],
        "mixed": nci1_like(5, 1, as_adj=True) + nci1_like(5, 2, as_adj=False),
        # round 5: the threaded walks of the WL ingestion take unit-weight inputs whose vertex set is the label keys in sorted
        # order (csrc/ingest.c sp_mode); everything else must still end where it ended before
        "tuple_sets_global_ids": [[{(10, 11), (11, 10), (11, 12), (12, 11)}, {10: 'a', 11: 'b', 12: 'a'}],
                                  [[(20, 21), (21, 20)], {20: 1, 21: 1}, {(20, 21): 0}]],
        "tuple_set_int_labels": [[{(10, 11), (11, 10), (11, 12), (12, 11)}, {10: 3, 11: 4, 12: 3}]] * 3,
        "tuple_set_unsorted_label_keys": [[{(10, 11), (11, 10)}, {11: 3, 10: 4}]],
        "tuple_set_labelled_vertex_without_edge": [[{(10, 11), (11, 10)}, {10: 3, 11: 4, 12: 5}]],
        "tuple_set_unlabelled_source": [[{(10, 11), (11, 10), (12, 10)}, {10: 3, 11: 4}]],
        "tuple_dict_unit_weights": [[{(1, 2): 1, (2, 1): 1.0}, {1: 0, 2: 1}]],
        "tuple_dict_weights": [[{(1, 2): 2, (2, 1): 2}, {1: 0, 2: 1}]],
        "er_dict_of_lists_with_isolated_vertices": er_dataset(30, 12, 0.15, 3, 4),
        "matrix_01_int_labels": [[A, {i: i % 2 for i in range(6)}], [A.astype(np.uint8), {i: 1 for i in range(6)}]],
    }
    for name, X in cases.items():
        fast, slow = both(X)
        assert fast == slow, name
    saved_threads = B.INGEST_THREADS
    try:
        for B.INGEST_THREADS in (1, 3):
            for name in ("tuple_set_int_labels", "er_dict_of_lists_with_isolated_vertices", "matrix_01_int_labels", "nci1_dicts"):
                fast, slow = both(cases[name] * 40)
                assert fast == slow and fast[0] == "ok", name
    finally:
        B.INGEST_THREADS = saved_threads
    assert B._gk_ingest.wl_ingest(cases["tuple_set_int_labels"], 2, False, 3, 0, 1) is not None       # taken by the threaded walk
    assert B._gk_ingest.wl_ingest(cases["tuple_set_unsorted_label_keys"], 2, False, 3, 0, 1) is None
    for name in ("nci1_adjacency", "nci1_dicts", "missing_label_matrix", "empty_labels"):
        X = [[x[0]] for x in cases[name]]                     # unlabelled: one-element inputs are fine
        assert both(X, False)[0] == both(X, False)[1], name
        assert both(cases[name], False)[0] == both(cases[name], False)[1], name
    assert both(cases["nci1_adjacency"], True, fitted_labels={0: 0, 1: 1})[0][0] == "ok"
    fast, slow = both(cases["nci1_dicts"], True, fitted_labels={0: 0, 1: 1})
    assert fast == slow
    assert both(cases["missing_label"])[0][:2] == ("raise", KeyError)


def test_c_ingestion_never_diverges_from_the_python_path_on_random_inputs():
    """Property test: whatever mixture of vertex symbols, edge-dictionary forms, weights and label
    dictionaries, the C helper either declines or produces exactly what the Python path produces --
    the same batch or the same exception."""
    hyp = pytest.importorskip("hypothesis")
    st = hyp.strategies
    import warnings
    from grakel_amd import batch as B
    _both_paths([[{0: [1], 1: [0]}, {0: 1, 1: 2}]])            # builds / loads the C module if needed
    symbols = st.sampled_from([0, 1, 2, 3, 4, 7, -1, 10 ** 12, True, 1.0, 2.5, "a", "b", (1, 2), np.int64(3), np.int32(1)])
    weights = st.sampled_from([1, 2, 3, 2.0, 1.0, 0, 0.5, -1, 2 ** 20, True, 7])
    labels_v = st.sampled_from([0, 1, 2, "x", "y", (1,), 3.5])

    @st.composite
    def element(draw):
        verts = draw(st.lists(symbols, min_size=1, max_size=6, unique_by=lambda v: (hash(v), v == v)))
        form = draw(st.sampled_from(["lists", "dicts"]))
        g = {}
        for v in draw(st.lists(st.sampled_from(verts), max_size=6)):
            nbrs = draw(st.lists(st.sampled_from(verts + [99]), max_size=4))
            g[v] = list(nbrs) if form == "lists" else {nb: draw(weights) for nb in nbrs}
        if draw(st.booleans()) and all(isinstance(v, int) and not isinstance(v, bool) and 0 <= v < 8 for v in verts):
            lab = {i: draw(labels_v) for i in range(max(verts) + 1)}      # identity numbering 0..n-1
        else:
            lab = {v: draw(labels_v) for v in draw(st.lists(st.sampled_from(verts + [99]), max_size=7))}
        extra = draw(st.sampled_from([(), ({},), ({}, "z")]))
        return draw(st.sampled_from([list, tuple]))((g, lab) + extra)

    def sp_run(X, with_labels):
        try:
            gb, m = B.sp_batch_from_input(X, with_labels)
            return ("ok", gb.graph_ptr.tolist(), gb.row_ptr.tolist(), gb.col_idx.tolist(), gb.node_label.tolist(),
                    gb.edge_weight.tolist(), gb.n_labels, m)
        except Exception as e:                      # noqa: BLE001
            return ("raise", type(e), e.args)

    def oa_run(X, fit):
        try:
            gb, m = B.wloa_batch_from_input(X, fitted_labels=None if fit else {0: 0, "x": 1}, fit=fit)
            return ("ok", gb.graph_ptr.tolist(), gb.row_ptr.tolist(), gb.col_idx.tolist(), gb.node_label.tolist(),
                    gb.n_labels, m)
        except Exception as e:                      # noqa: BLE001
            return ("raise", type(e), e.args)

    @hyp.settings(max_examples=400, deadline=None, suppress_health_check=list(hyp.HealthCheck))
    @hyp.given(st.lists(element(), min_size=1, max_size=4), st.booleans())
    def check(X, with_labels):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fast, slow = _both_paths(X)
            assert fast == slow
            fast = sp_run(X, with_labels)
            saved, B._gk_ingest = B._gk_ingest, None
            try:
                slow = sp_run(X, with_labels)
            finally:
                B._gk_ingest = saved
            assert fast == slow
            for fit in (True, False):                      # WL-OA: the batch restricted to the vertices with an entry
                fast = oa_run(X, fit)
                saved, B._gk_ingest = B._gk_ingest, None
                try:
                    slow = oa_run(X, fit)
                finally:
                    B._gk_ingest = saved
                assert fast == slow
    check()


def test_label_compression_fit_and_transform():
    ids, m = compress_labels(['b', 'a', 'c', 'a'])
    assert ids.tolist() == [1, 0, 2, 0] and m == {'a': 0, 'b': 1, 'c': 2}
    ids2, ext = compress_labels(['c', 'zz', 'a', 'd'], m)      # unseen ids continue past the fitted
    assert ids2.tolist() == [2, 4, 0, 3] and ext == {'d': 3, 'zz': 4}
    ids, m = compress_labels([5, 3, 5, 9])
    assert ids.tolist() == [1, 0, 1, 2] and m == {3: 0, 5: 1, 9: 2}


def test_csr_emitter_describes_the_same_graphs_as_the_object_form():
    X = er_dataset(30, 20, 0.2, 4, 7)
    gb, _ = wl_batch_from_input(X)
    gp, rp, ci, lab = er_dataset_csr(30, 20, 0.2, 4, 7)
    assert np.array_equal(gb.graph_ptr, gp) and np.array_equal(gb.row_ptr, rp)
    assert _adjacency_sets(gb) == _adjacency_sets(GraphBatch(gp, rp, ci, lab, 4))
    assert np.array_equal(gb.node_label, lab)


def test_concat_and_slice():
    X = random_labelled_graphs(12, 3, 9, 0.4, 3, 2)
    a, _ = wl_batch_from_input(X[:7])
    b, _ = wl_batch_from_input(X[7:])
    u = GraphBatch.concat(a, b)
    full, _ = wl_batch_from_input(X)
    assert np.array_equal(u.graph_ptr, full.graph_ptr) and np.array_equal(u.row_ptr, full.row_ptr)
    assert np.array_equal(u.col_idx, full.col_idx)
    s = full.slice_graphs(7, 12)
    assert np.array_equal(s.graph_ptr, b.graph_ptr) and np.array_equal(s.col_idx, b.col_idx)


def test_sp_ingestion_vertex_order_weights_and_errors():
    g = [{'c': {'a': 2}, 'a': {'b': 1}}, {'a': 0, 'b': 1, 'c': 0}]
    gb, m = sp_batch_from_input([g], True)
    assert gb.n_nodes == 3 and gb.col_idx.tolist() == [1, 0] and gb.edge_weight.tolist() == [1, 2]
    assert gb.node_label.tolist() == [0, 1, 0]
    with pytest.raises(ValueError):                      # no labels with with_labels=True
        sp_batch_from_input([[{0: [1], 1: [0]}, {}]], True)
    # not a multiple of a power of two: the batch keeps the float64 weights and which algorithm the reference's "auto" runs
    gb, _ = sp_batch_from_input([[{(0, 1): 0.1}, {0: 1, 1: 1}], [np.array([[0, 0.3], [0.3, 0]]), {0: 1, 1: 1}]], True)
    assert gb.edge_weight is None and gb.float_weight.tolist() == [0.1, 0.3, 0.3] and gb.from_dict.tolist() == [1, 0]
    ints, _ = sp_batch_from_input([[np.array([[0, 2], [2, 0]]), {0: 1, 1: 1}]], True)
    u = GraphBatch.concat(ints, gb)                      # an integer-weight fit + float-weight targets: the union counts in floats
    assert u.edge_weight is None and u.float_weight.tolist() == [2.0, 2.0, 0.1, 0.3, 0.3] and u.from_dict.tolist() == [0, 1, 0]
    assert u.slice_graphs(1, 3).float_weight.tolist() == [0.1, 0.3, 0.3]
    big = np.zeros((150, 150)); big[0, 1] = big[1, 0] = 0.1
    gbig, _ = sp_batch_from_input([[big, {i: 0 for i in range(150)}]], True)      # round 4: no size limit for general floats any more
    assert gbig.float_weight.tolist() == [0.1, 0.1] and gbig.n_nodes == 150
    with pytest.raises(NotImplementedError):
        sp_batch_from_input([[np.array([[0, -0.1], [-0.1, 0]]), {0: 1, 1: 1}]], True)
    gb, _ = sp_batch_from_input([[np.array([[0, 3], [0, 0]])]], False)
    assert gb.edge_weight.tolist() == [3] and gb.n_labels == 1


def test_float_edge_weights_are_quantised_exactly_or_declined():
    """quantise_weights: one power-of-two unit per batch, multiples below 2**20, exact or NotImplementedError."""
    from grakel_amd.batch import quantise_weights, GraphBatch
    ints, step = quantise_weights([np.array([0.5, 1.25, 3.0]), np.array([2.0])])
    assert step == 0.25 and [a.tolist() for a in ints] == [[2, 5, 12], [8]]
    assert quantise_weights([np.array([2.0, 4.0])])[1] == 1.0             # integral input keeps the unit 1
    assert quantise_weights([np.array([2.0 ** -20, 1.0 - 2.0 ** -20])])[1] == 2.0 ** -20
    for bad in ([0.1], [1.0 / 3], [2.0 ** 20], [2.0 ** -30, 1.0], [0.0], [-0.5], [np.inf], [np.nan]):
        with pytest.raises(NotImplementedError):
            quantise_weights([np.array(bad)])
    # every path sum of such weights is exact: the integer distances times the step are the float distances
    rs = np.random.RandomState(5)
    w = rs.randint(1, 2 ** 12, size=200) / 64.0
    ints, step = quantise_weights([w])
    for _ in range(50):
        idx = rs.randint(0, 200, size=30)
        assert float(np.sum(w[idx])) == float(ints[0][idx].sum()) * step
    # the union of a fitted and a target batch counts in the finer unit
    A = np.array([[0, 0.5, 0], [0.5, 0, 1.25], [0, 1.25, 0]])
    a, _ = sp_batch_from_input([[A, {0: 'a', 1: 'b', 2: 'a'}]], True)
    b, _ = sp_batch_from_input([[np.array([[0, 3], [3, 0]]), {0: 'a', 1: 'b'}]], True)
    assert a.weight_step == 0.25 and a.edge_weight.tolist() == [2, 2, 5, 5] and b.weight_step == 1.0
    u = GraphBatch.concat(a, b)
    assert u.weight_step == 0.25 and u.edge_weight.tolist() == [2, 2, 5, 5, 12, 12]
    assert a.slice_graphs(0, 1).weight_step == 0.25
    big, _ = sp_batch_from_input([[np.array([[0, 2 ** 19], [2 ** 19, 0]]), {0: 'a', 1: 'b'}]], True)
    with pytest.raises(NotImplementedError):
        GraphBatch.concat(a, big)


def test_vh_reads_only_the_label_dict():
    gb, m = vh_batch_from_input([["anything at all", {0: 'a', 1: 'b'}], [None, {7: 'b'}]])
    assert gb.n_graphs == 2 and gb.n_edges == 0 and gb.node_label.tolist() == [0, 1, 1]


def test_estimator_parameters_and_lazy_initialisation():
    # get_params must match the reference exactly
    assert sorted(WeisfeilerLehman().get_params()) == ['base_graph_kernel', 'n_iter', 'n_jobs',
                                                       'normalize', 'verbose']
    assert sorted(ShortestPath().get_params()) == ['algorithm_type', 'n_jobs', 'normalize',
                                                   'verbose', 'with_labels']
    assert sorted(VertexHistogram().get_params()) == ['n_jobs', 'normalize', 'sparse', 'verbose']
    wl = WeisfeilerLehman(n_iter=3)
    wl.initialize()
    assert wl._n_iter == 4 and wl._initialized["n_iter"]
    wl.set_params(n_iter=7)
    assert not wl._initialized["n_iter"]
    wl.initialize()
    assert wl._n_iter == 8
    for bad in (0, -1, 2.0, "3"):
        with pytest.raises(TypeError):
            WeisfeilerLehman(n_iter=bad).initialize()
    with pytest.raises(ValueError):
        WeisfeilerLehman(n_jobs="4").initialize()
    with pytest.raises(TypeError):
        WeisfeilerLehman(base_graph_kernel=3).initialize()
    WeisfeilerLehman(base_graph_kernel=(VertexHistogram, {"sparse": False})).initialize()
    wsp = WeisfeilerLehman(base_graph_kernel=(ShortestPath, {"with_labels": False}))
    wsp.initialize()                                   # SP is an accelerated base kernel
    assert wsp._base_graph_kernel is ShortestPath and wsp._sp_with_labels is False
    from grakel_amd import EdgeHistogram, WeisfeilerLehmanOptimalAssignment
    weh = WeisfeilerLehman(base_graph_kernel=EdgeHistogram)
    weh.initialize()                                   # round 3: EdgeHistogram is an accelerated base kernel too
    assert weh._base_graph_kernel is EdgeHistogram
    wany = WeisfeilerLehman(base_graph_kernel=WeisfeilerLehmanOptimalAssignment)
    wany.initialize()                                  # round 5: any other kernel class: device relabel + host base kernels
    assert wany._generic and wany._base_graph_kernel is WeisfeilerLehmanOptimalAssignment and not weh._generic
    with pytest.raises(ValueError):
        WeisfeilerLehman(base_graph_kernel=(ShortestPath, {"algorithm_type": "bfs"})).initialize()
    with pytest.raises(ValueError):
        ShortestPath(algorithm_type="bfs").initialize()
    from sklearn.base import clone
    assert clone(WeisfeilerLehman(n_iter=2, normalize=True)).get_params()["n_iter"] == 2


def test_fit_is_host_only_and_picklable():
    X = random_labelled_graphs(10, 3, 8, 0.4, 3, 1)
    wl = WeisfeilerLehman(n_iter=2).fit(X)               # no GPU needed to fit (ingestion only)
    wl2 = pickle.loads(pickle.dumps(wl))
    assert wl2._nx == 10 and wl2._inv_labels[0] == wl._inv_labels[0]      # level 0 = the input label map
    # the dictionaries of the levels >= 1 come from the device: without a GPU reading them fails loudly (and a
    # later read tries again -- the lazy fill is only disarmed by a successful fill)
    from grakel_amd._lib import GkError
    for _ in range(2):
        with pytest.raises(GkError):
            wl2._inv_labels == wl._inv_labels
        with pytest.raises(GkError):
            wl2._inv_labels.get(1)
    assert wl2._inv_labels.get(0) == wl._inv_labels[0]
    assert np.array_equal(wl2._fit_batch.col_idx, wl._fit_batch.col_idx)
    vh = pickle.loads(pickle.dumps(VertexHistogram().fit(X)))
    assert vh._labels == VertexHistogram().fit(X)._labels
    pickle.loads(pickle.dumps(ShortestPath().fit(X)))


def test_c_abi_exports_every_declared_symbol():
    """libgk_hip.so must load here (no GPU) and export exactly what include/gk_hip.h declares."""
    from grakel_amd import _lib
    header = open(os.path.join(ROOT, "include", "gk_hip.h")).read()
    declared = set(re.findall(r"\b(gk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gk_version().startswith(b"gk_hip")
    if _lib.device_count() == 0:                         # product path fails loudly without a GPU
        with pytest.raises(_lib.GkError):
            WeisfeilerLehman(n_iter=1).fit_transform(random_labelled_graphs(3, 3, 5, 0.5, 2, 0))


def test_every_context_option_is_documented_in_the_header():
    """The route / tuning options of a context (api.hip: the name table of gk_set_option) are part of the C ABI's contract:
    include/gk_hip.h names every one of them, and names none that the library does not know."""
    src = open(os.path.join(ROOT, "grakel_amd", "csrc", "api.hip")).read()
    known = set(re.findall(r'\{"([a-z0-9_.]+)",\s*&gk_opts::', src))
    assert len(known) > 40
    header = open(os.path.join(ROOT, "include", "gk_hip.h")).read()
    documented = set(re.findall(r'"((?:wl|feat|gram|sp|scan|transform|debug|sort)\.[a-z0-9_]+|no_mailbox)"', header))
    assert known - documented == set(), sorted(known - documented)
    assert documented - known == set(), sorted(documented - known)


def test_graph_kernel_dispatcher():
    """grakel/graph_kernels.py:452-554 for the accelerated kernels."""
    from grakel_amd import GraphKernel
    gk = GraphKernel(kernel=[{"name": "weisfeiler_lehman", "n_iter": 3}, {"name": "vertex_histogram"}],
                     normalize=True)
    gk.initialize()
    assert type(gk.kernel_) is WeisfeilerLehman and gk.kernel_.n_iter == 3 and gk.kernel_.normalize
    gk.kernel_.initialize()
    assert gk.kernel_._base_graph_kernel is VertexHistogram
    for name, cls in (("WL", WeisfeilerLehman), ("VH", VertexHistogram), ("ST-WL", VertexHistogram),
                      ("SP", ShortestPath), ("shortest_path", ShortestPath)):
        g = GraphKernel(kernel=name)
        g.initialize()
        assert type(g.kernel_) is cls
    g = GraphKernel(kernel={"name": "shortest_path", "with_labels": False})
    g.initialize()
    assert g.kernel_.with_labels is False
    for bad in ("random_walk", "HC", {"name": "SP", "as_attributes": True}):
        with pytest.raises(NotImplementedError):
            GraphKernel(kernel=bad).initialize()
    with pytest.raises(ValueError):
        GraphKernel(kernel="nope").initialize()
    with pytest.raises(NotImplementedError):
        GraphKernel(kernel="WL", Nystroem=20).initialize()
    assert sorted(GraphKernel().get_params()) == ['Nystroem', 'kernel', 'n_jobs', 'normalize',
                                                  'random_state', 'verbose']


def _write_mutag_tu(tmp_path, z):
    d = os.path.join(str(tmp_path), "MUTAG")
    os.makedirs(d)
    np.savetxt(os.path.join(d, "MUTAG_graph_indicator.txt"), z["node_graph"] + 1, fmt="%d")
    np.savetxt(os.path.join(d, "MUTAG_node_labels.txt"), z["node_label"], fmt="%d")
    with open(os.path.join(d, "MUTAG_A.txt"), "w") as f:
        for a, b in zip(z["edge_src"].tolist(), z["edge_dst"].tolist()):
            f.write("%d, %d\n" % (a, b))
    np.savetxt(os.path.join(d, "MUTAG_graph_labels.txt"), np.arange(188) % 2 * 2 - 1, fmt="%d")
    return str(tmp_path)


def test_tu_loader_gives_the_same_batch_as_the_object_path(tmp_path, mutag_graphs):
    """files -> CSR without per-graph Python objects (grakel/datasets/base.py:135-290)."""
    from grakel_amd.datasets import read_tu
    G, z = mutag_graphs
    batch, classes = read_tu(_write_mutag_tu(tmp_path, z), "MUTAG")
    ref, mapping = wl_batch_from_input(G)
    assert classes.shape == (188,) and batch.n_graphs == 188 and batch.label_map == mapping
    assert np.array_equal(batch.graph_ptr, ref.graph_ptr) and np.array_equal(batch.node_label, ref.node_label)
    assert _adjacency_sets(batch) == _adjacency_sets(ref)


def test_wloa_ingestion_keeps_only_vertices_with_an_edge_dictionary_entry():
    """weisfeiler_lehman_optimal_assignment.py:176: from level 1 on only ``Gs_ed[j].keys()`` are
    relabelled, and the histogram is built from those vertices alone (:201-206)."""
    from grakel_amd.batch import wloa_batch_from_input
    g = [[{0: [1], 1: [0], 2: []}, {0: 'a', 1: 'b', 2: 'a'}],            # {2: []} leaves no entry
         [{0: [1, 2], 1: [0], 2: [0]}, {0: 'a', 1: 'b', 2: 'c'}]]
    b, mapping = wloa_batch_from_input(g)
    assert b.graph_ptr.tolist() == [0, 2, 5] and mapping == {'a': 0, 'b': 1, 'c': 2}
    assert b.node_label.tolist() == [0, 1, 0, 1, 2]
    assert b.col_idx.tolist() == [1, 0, 3, 4, 2, 2]
    sink = [[{0: [1], 1: [0, 2]}, {0: 'a', 1: 'b', 2: 'a'}]]              # a pure sink gets an entry
    assert wloa_batch_from_input(sink)[0].graph_ptr.tolist() == [0, 3]
    A = np.array([[0, 1, 0], [1, 0, 0], [0, 0, 0]])                       # adjacency: all indices
    assert wloa_batch_from_input([[A, {0: 'a', 1: 'b', 2: 'c'}]])[0].graph_ptr.tolist() == [0, 3]
    with pytest.raises(KeyError):
        wloa_batch_from_input([[A, {0: 'a', 1: 'b'}]])                    # entry without a label
    with pytest.raises(ValueError):
        wloa_batch_from_input([[A]], fitted_labels={}, fit=False)
    with pytest.raises(TypeError):
        wloa_batch_from_input([[A]])


def test_induced_subbatch_and_core_framework_parameters():
    from grakel_amd import CoreFramework, GraphKernel
    from grakel_amd.batch import sp_batch_from_input
    from grakel_amd.core_framework import induced_subbatch
    A = np.array([[0, 2, 1, 0], [2, 0, 1, 0], [1, 1, 0, 3], [0, 0, 3, 0]])
    gb, _ = sp_batch_from_input([[A, {0: 'a', 1: 'b', 2: 'a', 3: 'c'}], [np.zeros((2, 2)), {0: 'a', 1: 'a'}]], True)
    sub, idx = induced_subbatch(gb, np.array([1, 1, 1, 0, 0, 0], bool))      # the triangle of graph 0
    assert idx.tolist() == [0] and sub.graph_ptr.tolist() == [0, 3]
    assert sub.row_ptr.tolist() == [0, 2, 4, 6] and sub.col_idx.tolist() == [1, 2, 0, 2, 0, 1]
    assert sub.edge_weight.tolist() == [2, 1, 2, 1, 1, 1] and sub.node_label.tolist() == [0, 1, 0]
    sub, idx = induced_subbatch(gb, np.ones(6, bool))
    assert idx.tolist() == [0, 1] and sub.n_edges == gb.n_edges
    cf = CoreFramework(min_core=3)
    assert cf.min_core == -1                               # core_framework.py:48 ignores the argument
    assert sorted(cf.get_params()) == ['base_graph_kernel', 'min_core', 'n_jobs', 'normalize', 'verbose']
    cf.initialize()
    assert cf.base_graph_kernel_ is ShortestPath           # default base (core_framework.py:61-62)
    with pytest.raises(TypeError):
        CoreFramework(base_graph_kernel=3).initialize()
    g = GraphKernel(kernel=[{"name": "CORE"}, {"name": "WL", "n_iter": 2}])
    g.initialize()
    assert type(g.kernel_).__name__ == "CoreFramework" and g.kernel_.base_graph_kernel[0] is WeisfeilerLehman


def test_limits_fail_early_and_reference_base_classes_are_accepted():
    """n_iter is not capped; a malformed packed batch fails in GraphBatch; the reference's own base-kernel classes map to the accelerated ones."""
    from grakel_amd import GraphBatch
    from grakel_amd.weisfeiler_lehman_optimal_assignment import WeisfeilerLehmanOptimalAssignment
    # any positive n_iter, like the reference (weisfeiler_lehman.py:112-114): beyond 48 levels the features are built
    # in chunks (gk_features_build_range) and the chunks' matrices add up
    WeisfeilerLehman(n_iter=48).initialize()
    WeisfeilerLehmanOptimalAssignment(n_iter=60).initialize()
    WeisfeilerLehman(n_iter=47).initialize()
    gp, rp, ci, lab = np.array([0, 2, 3]), np.array([0, 1, 2, 2]), np.array([1, 0]), np.array([0, 1, 0])
    assert GraphBatch(gp, rp, ci, lab, 2).n_graphs == 2
    for bad in (dict(rp=np.array([0, 2, 1, 2])), dict(gp=np.array([0, 4, 3])), dict(lab=np.array([0, 2, 0])),
                dict(lab=np.array([0, -1, 0]))):
        a = dict(gp=gp, rp=rp, ci=ci, lab=lab)
        a.update(bad)
        with pytest.raises(ValueError):
            GraphBatch(a["gp"], a["rp"], a["ci"], a["lab"], 2)
    # a class that only LOOKS like the reference's (module "grakel....", same name) is enough for the mapping
    fake = type("VertexHistogram", (object,), {"__module__": "grakel.kernels.vertex_histogram"})
    wl = WeisfeilerLehman(base_graph_kernel=fake)
    wl.initialize()
    assert wl._base_graph_kernel is VertexHistogram
    fake_sp = type("ShortestPath", (object,), {"__module__": "grakel.kernels.shortest_path"})
    wl = WeisfeilerLehman(base_graph_kernel=(fake_sp, {"with_labels": True}))
    wl.initialize()
    assert wl._base_graph_kernel is ShortestPath


def test_c_header_is_plain_c_and_the_multi_gpu_stub_type_checks():
    """include/gk_hip.h must be consumable by a C compiler (the boundary is a C ABI), and INTEGRATION.md's section C --
    kept as tests/c_abi/multi_gpu_stub.c -- must match the declared signatures (gcc -fsyntax-only; nothing runs)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = os.path.join(ROOT, "tests", "c_abi", "multi_gpu_stub.c")
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    # the calls of INTEGRATION.md's section C are the calls of the stub
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## C. Multi-GPU without Python"):text.index("## Build / deploy")]
    stub = open(src).read()
    for call in re.findall(r"\b(gk_[a-z0-9_]+)\(", section.split("```c")[1].split("```")[0]):
        assert call + "(" in stub, call


def _same_batch(a, b):
    """two fitted batches describe the same graphs with the same label partition (ids may be numbered differently)"""
    return (np.array_equal(a.graph_ptr, b.graph_ptr) and np.array_equal(a.row_ptr, b.row_ptr) and
            _adjacency_sets(a) == _adjacency_sets(b) and a.n_labels == b.n_labels and
            np.array_equal(O.canonical_partition(a.node_label.tolist()), O.canonical_partition(b.node_label.tolist())))


@pytest.mark.skipif(not reference_available(), reason="needs the real grakel (oracle/build_ref.sh): build container only")
def test_the_reference_frameworks_drive_the_accelerated_base_classes():
    """SURVEY 8b "Callers": the REAL grakel.HadamardCode (hadamard_code.py:19,189), CoreFramework (core_framework.py:199-204)
    and WeisfeilerLehman (weisfeiler_lehman.py:260-285) are constructed over the accelerated classes
    (grakel_amd.for_grakel: the same classes with the reference's Kernel among their bases, which is what the frameworks'
    issubclass check asks for), initialised, cloned, pickled -- and FITTED: `fit` is host-only here, so the frameworks'
    own per-level calls (grakel.Graph objects, tuple-valued Hadamard labels, extras) run through the accelerated
    ingestion.  Every level's fitted batch must equal the one the test-local restatement of the calling sequence
    (tests/caller_protocols.py -- what the GPU box, which has no reference, drives the device with) produces."""
    import sys
    ref = os.environ.get("GK_REF_BUILD", "/tmp/grakel_oracle")
    if ref not in sys.path:
        sys.path.insert(0, ref)
    import grakel
    from sklearn.base import clone
    from grakel_amd import for_grakel as FG
    from caller_protocols import CoreCaller, HadamardCaller
    for cls in (FG.VertexHistogram, FG.ShortestPath, FG.WeisfeilerLehman, FG.EdgeHistogram, FG.WeisfeilerLehmanOptimalAssignment):
        assert type(cls) is type and issubclass(cls, grakel.kernels.Kernel) and issubclass(cls, grakel_amd.Kernel)
        assert cls.__mro__[1].__module__.startswith("grakel_amd.")            # every method: the accelerated class first
        k = cls(normalize=False, verbose=False, n_jobs=None)                   # what every framework passes (hadamard_code.py:92-94)
        assert clone(k).get_params() == k.get_params() and type(pickle.loads(pickle.dumps(k))) is cls
    G = random_labelled_graphs(**dict(SMALL_SETS)["adj_u"])
    tr = G[:26]
    # HadamardCode over VertexHistogram (its default base kernel) and over ShortestPath
    for base, n_iter in ((FG.VertexHistogram, 3), ((FG.ShortestPath, {"with_labels": True}), 2)):
        hc = grakel.HadamardCode(base_graph_kernel=base, n_iter=n_iter)
        hc.fit(tr)
        assert sorted(hc.X) == list(range(n_iter))
        bcls = base if type(base) is type else base[0]
        mine = HadamardCaller(None, n_iter).level_inputs(tr)
        for i in range(n_iter):
            assert type(hc.X[i]) is bcls and hc.X[i].get_params()["normalize"] is False
            assert _same_batch(hc.X[i]._fit_batch, bcls().fit(mine[i])._fit_batch), (base, i)
        hc2 = pickle.loads(pickle.dumps(hc))                                   # a fitted framework holds the class AND fitted bases
        assert hc2.base_graph_kernel_[0] is bcls and _same_batch(hc2.X[0]._fit_batch, hc.X[0]._fit_batch)
    # CoreFramework over ShortestPath (its default base kernel) and VertexHistogram
    for base in (FG.ShortestPath, FG.VertexHistogram):
        cf = grakel.CoreFramework(base_graph_kernel=base)
        cf.fit(tr)
        mine = CoreCaller(None).level_inputs(tr)
        assert sorted(cf.X) == sorted(i for i, (subs, idx) in mine.items() if len(idx))
        for i, k in cf.X.items():
            subs, idx = mine[i]
            assert np.array_equal(cf._fit_indexes[i], idx)
            assert _same_batch(k._fit_batch, base().fit(subs)._fit_batch), (base, i)
        pickle.loads(pickle.dumps(cf))
    # the reference's WeisfeilerLehman framework over the accelerated VertexHistogram: level inputs are
    # (edge dictionary, level labels) lists (weisfeiler_lehman.py:220,257)
    wl = grakel.WeisfeilerLehman(base_graph_kernel=FG.VertexHistogram, n_iter=2)
    wl.fit(tr)
    ref = O.WLOracle(n_iter=2)
    ref.fit_transform(tr, keep_levels=True)
    assert [wl.X[i]._fit_batch.n_labels for i in range(3)] == [len(ref.inv_labels[i]) for i in range(3)]

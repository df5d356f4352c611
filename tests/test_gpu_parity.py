"""GPU parity tests (run with ``-m gpu`` on the MI355X box): the HIP path, called through
the C ABI, against the reference's golden fixtures and the CPU oracle.

Bars (BASELINE.json): integer WL labels -> identical node partition per level; Gram matrices
are integer valued -> compared EXACTLY as float64; normalised matrices within 1e-5 relative
(they are in fact required to match to 1e-13 here).
"""
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from golden.small_sets import SMALL_SETS, split, sp_inputs
from oracle import grakel_oracle as O
from grakel_amd.synthetic import (er_dataset, er_dataset_csr, nci1_like, random_labelled_graphs)

pytestmark = pytest.mark.gpu

REL_TOL = 1e-13      # normalised Gram: far inside BASELINE.json's 1e-5 relative bar


@pytest.fixture(scope="module")
def gk():
    import grakel_amd
    from grakel_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libgk_hip.so not built"
    assert _lib.device_count() > 0, "no GPU visible"
    return grakel_amd


@pytest.fixture
def gkopt(gk):
    """Route / capacity options of the process-wide engine for ONE test (``gk_set_option``): restored afterwards."""
    from grakel_amd.engine import get_engine
    eng = get_engine()
    old = {}

    def set_option(name, value=1):
        old.setdefault(name, eng.get_option(name))
        eng.set_option(name, value)

    yield set_option
    for name, value in old.items():
        eng.set_option(name, value)


def _mix64(z):
    z = z.astype(np.uint64)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xbf58476d1ce4e5b9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94d049bb133111eb)
    z ^= z >> np.uint64(31)
    return z


def _signature_numpy(gb, lab_prev, seed):
    """numpy restatement of wl_signature_* (grakel_amd/csrc/wl.hip) for the kernel-level test."""
    with np.errstate(over='ignore'):
        seed = np.uint64(seed)
        deg = np.diff(gb.row_ptr).astype(np.uint64)
        nb = lab_prev[gb.col_idx].astype(np.uint64)
        elem = _mix64((nb + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + seed)
        acc = np.zeros(gb.n_nodes, np.uint64)
        src = np.repeat(np.arange(gb.n_nodes), np.diff(gb.row_ptr))
        np.add.at(acc, src, elem)
        own = lab_prev.astype(np.uint64)
        head = _mix64(_mix64(own + np.uint64(0x632BE59BD9B4E019) * (seed | np.uint64(1))) ^
                      (deg * np.uint64(0xD6E8FEB86659FD93)))
        h = _mix64(head + acc)
    sorted_nb = np.empty_like(gb.col_idx)
    for v in range(gb.n_nodes):
        s, e = gb.row_ptr[v], gb.row_ptr[v + 1]
        sorted_nb[s:e] = np.sort(lab_prev[gb.col_idx[s:e]])
    return h, sorted_nb


def _oracle_levels(X, n_iter):
    wl = O.WLOracle(n_iter=n_iter)
    K = wl.fit_transform(X, keep_levels=True)
    flat = [np.array([l for d in lev for l in d.values()]) for lev in wl.levels]
    return wl, K, flat


def same_partition(a, b):
    return np.array_equal(O.canonical_partition(a.tolist()), O.canonical_partition(b.tolist()))


# ------------------------------------------------------------------------------------------
def test_signature_kernel_matches_numpy(gk):
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    G = random_labelled_graphs(60, 2, 25, 0.3, 4, 5, fmt="dict")
    # hubs of degree 40 .. 20000 (> WL_DEG_SMALL)
    # degree 33..1024: the wave-per-node kernel (1, 2, 4, 8, 16 registers per lane: every boundary), beyond: the workgroup one
    for hub in (40, 64, 65, 128, 129, 300, 512, 513, 1000, 1024, 1025, 20000):
        ed = {0: list(range(1, hub + 1))}
        ed.update({i: [0] for i in range(1, hub + 1)})
        G.append([ed, {i: (i * 7) % 5 for i in range(hub + 1)}])
    gb, _ = wl_batch_from_input(G)
    eng = get_engine()
    db = eng.upload(gb)
    for seed in (0, 12345678901234567):
        h, s = eng.wl_debug_signature(db, 1, seed)
        h_ref, s_ref = _signature_numpy(gb, gb.node_label, seed)
        assert np.array_equal(s, s_ref)
        assert np.array_equal(h, h_ref)


@pytest.mark.parametrize("name", [n for n, _ in SMALL_SETS])
def test_wl_label_partitions_match_oracle(gk, name):
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    X = random_labelled_graphs(**dict(SMALL_SETS)[name])
    wl, _, levels = _oracle_levels(X, 4)
    gb, _ = wl_batch_from_input(X)
    eng = get_engine()
    db = eng.upload(gb)
    counts = eng.wl_relabel(db, 4)
    assert counts == wl.label_counts
    for lvl in range(5):
        assert same_partition(eng.wl_labels(db, lvl), levels[lvl]), "level %d" % lvl


def _graphs_with_isolated_vertices(kind, seed):
    """Graphs whose connected part freezes quickly while many isolated vertices keep sharing labels:
    the active-set levels then carry the isolated classes (gk_batch::iso_info) instead of sorting them."""
    rs = np.random.RandomState(seed)
    X = []
    for g in range(60):
        if kind == "paths":            # a path with pairwise distinct labels: every class is a singleton at once
            m = int(rs.randint(3, 9))
            ed = {i: [j for j in (i - 1, i + 1) if 0 <= j < m] for i in range(m)}
            lab = {i: 100 * g + i for i in range(m)}
        else:                          # a small random connected-ish part with few labels (classes keep splitting)
            m = int(rs.randint(4, 10))
            ed = {i: [] for i in range(m)}
            for i in range(1, m):
                j = int(rs.randint(0, i))
                ed[i].append(j), ed[j].append(i)
            lab = {i: int(rs.randint(0, 2)) for i in range(m)}
        for q in range(int(rs.randint(15, 40))):      # isolated vertices: 3 shared labels + a few unique ones
            v = m + q
            ed[v] = []
            lab[v] = int(rs.randint(0, 3)) if rs.rand() < 0.9 else 10 ** 6 + 1000 * g + q
        X.append([ed, lab])
    return X


@pytest.mark.parametrize("kind", ["paths", "trees"])
def test_isolated_vertices_are_carried_through_active_set_levels(gk, kind):
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    X = _graphs_with_isolated_vertices(kind, 3)
    wl, K, levels = _oracle_levels(X, 5)
    gb, _ = wl_batch_from_input(X)
    assert int((np.diff(gb.row_ptr) == 0).sum()) * 4 > 3 * gb.n_nodes      # mostly isolated: the active path runs
    eng = get_engine()
    db = eng.upload(gb)
    counts = eng.wl_relabel(db, 5)
    assert counts == wl.label_counts
    for lvl in range(6):
        assert same_partition(eng.wl_labels(db, lvl), levels[lvl]), "level %d" % lvl
    feat = eng.features(db, 6)
    assert np.array_equal(eng.gram(feat), K)
    est = gk.WeisfeilerLehman(n_iter=5, normalize=True)
    Kn = est.fit_transform(X)
    Kref = O.WLOracle(n_iter=5, normalize=True)
    assert np.allclose(Kn, Kref.fit_transform(X), rtol=REL_TOL, atol=0)
    Y = _graphs_with_isolated_vertices(kind, 4)[:17]
    assert np.allclose(est.transform(Y), Kref.transform(Y), rtol=REL_TOL, atol=0)
    oa = gk.WeisfeilerLehmanOptimalAssignment(n_iter=4)
    assert np.array_equal(oa.fit_transform(X), O.WLOAOracle(n_iter=4).fit_transform(X))


def _cycle_and_path_graphs(n_labels, seed, with_isolated):
    rs = np.random.RandomState(seed)
    X = []
    for g in range(80):
        m = int(rs.randint(3, 12))
        cyc = rs.rand() < 0.5
        ed = {i: [j for j in ((i - 1) % m if cyc else i - 1, (i + 1) % m if cyc else i + 1) if 0 <= j < m and j != i]
              for i in range(m)}
        ed = {i: sorted(set(v)) for i, v in ed.items()}
        lab = {i: int(rs.randint(0, n_labels)) for i in range(m)}
        if with_isolated:
            for q in range(int(rs.randint(0, 4))):
                ed[m + q] = []
                lab[m + q] = int(rs.randint(0, n_labels))
        X.append([ed, lab])
    return X


@pytest.mark.parametrize("n_labels,with_isolated", [(16, False), (16, True), (17, True), (2, True)])
def test_level1_exact_signature_codes_and_their_limits(gk, n_labels, with_isolated):
    """Max degree 2: 16 labels still fit the exact 32-bit level-1 code (16 * 3^16 < 2^32), 17 do not
    (the library then hashes); both must give the reference partition.  n_iter 0 and 1 included."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    X = _cycle_and_path_graphs(n_labels, 11, with_isolated)
    gb, _ = wl_batch_from_input(X)
    eng = get_engine()
    for h in (0, 1, 4):
        wl, K, levels = _oracle_levels(X, h)
        db = eng.upload(gb)
        assert eng.wl_relabel(db, h) == wl.label_counts
        for lvl in range(h + 1):
            assert same_partition(eng.wl_labels(db, lvl), levels[lvl]), "h %d level %d" % (h, lvl)
        assert np.array_equal(eng.gram(eng.features(db, h + 1)), K)
    only_isolated = [[{i: [] for i in range(5)}, {i: i % 3 for i in range(5)}] for _ in range(7)]
    assert np.array_equal(gk.WeisfeilerLehman(n_iter=3).fit_transform(only_isolated),
                          O.WLOracle(n_iter=3).fit_transform(only_isolated))


# the host-driven relabel route (wl.hip) and its per-level choices: every entry also switches the stream route off
_HOST_ROUTES = [
    (), ("wl.no_tiny",), ("wl.no_listscan",), ("wl.no_iso",), ("wl.no_split",), ("wl.no_exact1",), ("wl.no_active_set",),
    # round 2: sort-free dictionary, graph-major features, level-0 histogram, where the singleton flags travel,
    # register sort of the neighbour lists, workgroup-private df histograms
    ("wl.no_bucket_dict",), ("feat.no_gm",), ("wl.no_hist0",), ("wl.frozen_words",), ("wl.flag_bytes",), ("wl.sig_no_regs",),
    ("feat.gm_no_priv",), ("feat.gm_rows_wg",), ("no_mailbox",), ("sort.buckets", 1), ("sort.buckets", 2),
    # combinations that meet in real jobs: label-major features on top of a sort-free relabel is impossible by
    # construction (feat.no_gm turns both off), the words + no list scan pair is the round-1 data flow
    ("wl.frozen_words", "wl.no_listscan"), ("wl.no_hist0", "wl.no_exact1", "wl.no_bucket_dict"),
    # round 5: the fused scans (scan_fn.h) with a scanned tile-sum array, the form jobs above 8 M items take, with the
    # sorting dictionary and the label-major feature builder (their users)
    ("scan.direct_max",), ("scan.direct_max", "wl.no_bucket_dict", "feat.no_gm"),
    # round 6: the operand sizes read back after the column scan (rounds 2-5) instead of posted from its tile sums
    ("feat.gm_no_early_post",),
]
# round 4: the relabel route without host round trips (wl_stream.hip) is the default; its own switches
ROUTE_OPTIONS = [(), ("wl.no_exact1",), ("wl.sig_no_regs",), ("feat.gm_no_priv",), ("feat.gm_rows_wg",), ("no_mailbox",),
                 ("feat.gm_no_early_post",)] + \
    [r + ("wl.no_stream",) for r in _HOST_ROUTES]


def _apply_route(gkopt, route):
    it = iter(route)
    for name in it:
        if name == "sort.buckets":
            gkopt(name, next(it))
        else:
            gkopt(name, 1)


_route_cache = {}


def _route_case():
    """One oracle run for all the route tests."""
    if "case" not in _route_cache:
        X = er_dataset(500, 40, 0.07, 3, 17)          # sparse: isolated vertices, and the refinement needs ~6 levels
        _route_cache["case"] = (X,) + _oracle_levels(X, 7)
    return _route_cache["case"]


@pytest.mark.parametrize("route", ROUTE_OPTIONS, ids=lambda r: "+".join(str(x) for x in r) or "default")
def test_every_relabel_path_gives_the_reference_partition(gk, route, gkopt):
    """The relabel loop picks among several equivalent routes per level (single-workgroup tail levels,
    active list from the previous list or from all nodes, carried isolated classes, split listing, exact
    level-1 codes, active sets at all, sort-free or sorting dictionary, flag transport) and the feature
    builder between two forms; each option removes or forces one of them, the result must not move."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    _apply_route(gkopt, route)
    X, wl, K, levels = _route_case()
    gb, _ = wl_batch_from_input(X)
    eng = get_engine()
    db = eng.upload(gb)
    assert eng.wl_relabel(db, 7) == wl.label_counts
    for lvl in range(8):
        assert same_partition(eng.wl_labels(db, lvl), levels[lvl]), "level %d" % lvl
    assert np.array_equal(eng.gram(eng.features(db, 8)), K)
    oa = gk.WeisfeilerLehmanOptimalAssignment(n_iter=3)      # the min-sum features take the same routes
    if "oa" not in _route_cache:
        _route_cache["oa"] = O.WLOAOracle(n_iter=3).fit_transform(X[:120])
    assert np.array_equal(oa.fit_transform(X[:120]), _route_cache["oa"])


def _few_labels_many_isolated(seed):
    rs = np.random.RandomState(seed)
    X = []
    for g in range(70):
        m = int(rs.randint(4, 12))
        ed = {i: [] for i in range(m)}
        for i in range(1, m):
            j = int(rs.randint(0, i))
            ed[i].append(j), ed[j].append(i)
        lab = {i: int(rs.randint(0, 3)) for i in range(m)}
        for q in range(int(rs.randint(10, 30))):       # isolated vertices, three labels (one of them rare)
            ed[m + q] = []
            lab[m + q] = 2 if rs.rand() < 0.02 else int(rs.randint(0, 2))
        X.append([ed, lab])
    return X


@pytest.mark.parametrize("case", ["er", "sparse", "hashed1", "isolated", "deep"])
def test_stream_relabel_route_against_the_oracle(gk, gkopt, case):
    """The relabel route without host round trips (csrc/wl_stream.hip): it is the one that runs for these jobs, its
    label ids are dense per level, partitions / label counts / K are the oracle's (levels that walk all nodes and levels
    that walk the active list, exact level-1 codes and hashed level 1, isolated vertices carried from level 1 on, a
    hierarchy deeper than the refinement), and a table overflow hands the job to the host-driven route."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    X, h = {"er": (er_dataset(300, 100, 0.05, 5, 3), 5), "sparse": (er_dataset(500, 40, 0.07, 3, 17), 7),
            "hashed1": (er_dataset(200, 30, 0.1, 17, 5), 4), "isolated": (_few_labels_many_isolated(7), 5),
            "deep": (er_dataset(40, 12, 0.3, 2, 9), 20)}[case]
    wl, K, levels = _oracle_levels(X, h)
    gb, _ = wl_batch_from_input(X)
    eng = get_engine()
    db = eng.upload(gb)
    assert eng.wl_relabel(db, h) == wl.label_counts and db.stream_route
    for lvl in range(h + 1):
        lab = eng.wl_labels(db, lvl)
        assert same_partition(lab, levels[lvl]), "level %d" % lvl
        if lvl:
            assert np.array_equal(np.unique(lab), np.arange(wl.label_counts[lvl])), "ids of level %d are not dense" % lvl
    assert np.array_equal(eng.gram(eng.features(db, h + 1)), K)
    gkopt("wl.bd_slots", 1)                     # every bucket overflows: the job is relabelled by the host-driven route
    db2 = eng.upload(gb)
    assert eng.wl_relabel(db2, h) == wl.label_counts and not db2.stream_route
    assert np.array_equal(eng.gram(eng.features(db2, h + 1)), K)


@pytest.mark.parametrize("normalize", [False, True])
def test_transform_by_lookup_against_the_oracle_and_the_joint_route(gk, normalize):
    """WeisfeilerLehman.transform looks target signatures up in the fitted dictionaries kept on the device
    (csrc/wl_transform.hip; weisfeiler_lehman.py:435-498): same matrix and diagonals as the joint relabel of fitted graphs
    + targets and as the oracle -- targets that ARE fitted graphs, fresh ones, input labels the fit never saw, isolated
    vertices, a one-vertex graph; the state survives diagonal() / a second fit_transform relabelling the fitted batch."""
    X = er_dataset(300, 40, 0.08, 4, 1)
    cases = {"fitted": X[:20], "fresh": er_dataset(25, 40, 0.08, 4, 99), "unseen labels": er_dataset(10, 40, 0.08, 6, 5),
             "tiny": [[{0: []}, {0: 1}], [{0: [1], 1: [0]}, {0: 0, 1: 2}]] + er_dataset(5, 30, 0.05, 3, 8)}
    ref = O.WLOracle(n_iter=4, normalize=normalize)
    ref.fit_transform(X)
    same = (lambda a, b: np.allclose(a, b, rtol=REL_TOL, atol=0)) if normalize else np.array_equal
    est = gk.WeisfeilerLehman(n_iter=4, normalize=normalize)
    est.transform_route = "lookup"
    est.fit(X)
    joint = gk.WeisfeilerLehman(n_iter=4, normalize=normalize)
    joint.transform_route = "joint"
    joint.fit(X)
    for name, Y in cases.items():
        K = est.transform(Y)
        assert "_dev_wlfit" in est.__dict__, "the look-up route did not run"
        assert same(K, ref.transform(Y)), name
        assert same(K, joint.transform(Y)), name
        assert all(np.array_equal(a, b) for a, b in zip(est.diagonal(), joint.diagonal())), name
    # diagonal() after a fresh fit relabels the fitted batch: the fitted state is rebuilt, not used stale
    est.fit(X)
    est.transform(cases["fresh"])
    wf = est._dev_wlfit
    del est._X_diag
    est._is_transformed = False
    est.diagonal()
    assert same(est.transform(cases["fresh"]), ref.transform(cases["fresh"])) and est._dev_wlfit is not wf
    # default policy: a handful of targets against a larger fit take the look-up, a large target set the joint route
    auto = gk.WeisfeilerLehman(n_iter=4, normalize=normalize).fit(X)
    assert same(auto.transform(cases["fresh"][:3]), ref.transform(cases["fresh"][:3])) and "_dev_wlfit" in auto.__dict__
    auto2 = gk.WeisfeilerLehman(n_iter=4, normalize=normalize).fit(X)
    assert same(auto2.transform(X[:100]), ref.transform(X[:100])) and "_dev_wlfit" not in auto2.__dict__
    import pickle
    again = pickle.loads(pickle.dumps(est))
    assert same(again.transform(cases["fresh"]), ref.transform(cases["fresh"]))


def test_transform_lookup_declines_a_hub_and_takes_the_joint_route(gk):
    """A class representative with more than 64 neighbours is outside the look-up's in-register signature comparison: the
    fitted state declines (GK_ERR_UNSUPPORTED) and transform silently takes the joint route."""
    star = {0: list(range(1, 80))}
    star.update({i: [0] for i in range(1, 80)})
    X = er_dataset(40, 20, 0.2, 3, 3) + [[star, {i: i % 3 for i in range(80)}]]
    Y = er_dataset(3, 20, 0.2, 3, 4)
    est = gk.WeisfeilerLehman(n_iter=3)
    est.transform_route = "lookup"
    ref = O.WLOracle(n_iter=3)
    ref.fit_transform(X)
    assert np.array_equal(est.fit(X).transform(Y), ref.transform(Y)) and est._lookup_declined


@pytest.mark.parametrize("slots", [1, 8, 40])
def test_bucket_dictionary_overflow_takes_the_sorting_path(gk, gkopt, slots):
    """A bucket of the sort-free dictionary that holds more distinct keys than its table (forced here by shrinking
    the table, ``wl.bd_slots``) raises the overflow flag; the whole job is then relabelled again on the sorting
    path.  Partitions, label counts and K must be the oracle's, and a later feature build on the label-major path
    must find its label-grouped orders."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    X, wl, K, levels = _route_case()
    gb, _ = wl_batch_from_input(X)
    eng = get_engine()
    gkopt("wl.bd_slots", slots)
    db = eng.upload(gb)
    assert eng.wl_relabel(db, 7) == wl.label_counts
    for lvl in range(8):
        assert same_partition(eng.wl_labels(db, lvl), levels[lvl]), "level %d" % lvl
    assert np.array_equal(eng.gram(eng.features(db, 8)), K)
    gkopt("feat.gm_row_lds_max", 64)                     # ... and the label-major builder on the same batch
    assert np.array_equal(eng.gram(eng.features(db, 8)), K)


@pytest.mark.parametrize("kind", ["dot", "minsum"])
def test_graph_major_builder_declines_and_the_label_major_builder_takes_over(gk, gkopt, kind):
    """The graph-major builder assembles an operand row in LDS and declines rows wider than that
    (GK_ERR_UNSUPPORTED inside the library); the relabel had skipped the label-grouped orders for it, so the
    label-major builder first rebuilds them (gk_batch_rebuild_order).  Forced by ``feat.gm_row_lds_max``."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    X, wl, K, levels = _route_case()
    eng = get_engine()
    if kind == "dot":
        gb, _ = wl_batch_from_input(X)
        db = eng.upload(gb)
        eng.wl_relabel(db, 7)
        gkopt("feat.gm_row_lds_max", 64)                      # an operand row is at least one 128-byte K-step
        assert np.array_equal(eng.gram(eng.features(db, 8)), K)
        est = gk.WeisfeilerLehman(n_iter=7)
        assert np.array_equal(est.fit_transform(X), K)
        assert np.array_equal(est.transform(X[:9]), K[:9])
    else:
        gkopt("feat.gm_row_lds_max", 64)
        want = O.WLOAOracle(n_iter=3)
        Kw = want.fit_transform(X[:150])
        oa = gk.WeisfeilerLehmanOptimalAssignment(n_iter=3)
        assert np.array_equal(oa.fit_transform(X[:150]), Kw)
        assert np.array_equal(oa.transform(X[150:170]), want.transform(X[150:170]))


def test_natural_fallbacks_many_input_labels_and_a_large_graph(gk):
    """No option set: more than 256 input labels (level 0 goes through the sorting dictionary instead of the
    per-graph histogram) and a graph above 1024 nodes (the graph-major builder and the sort-free dictionary do
    not apply to the batch) -- the routes real inputs of that shape take."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    rs = np.random.RandomState(5)
    X = random_labelled_graphs(60, 5, 30, 0.2, 400, 31, fmt="dict")       # 400 possible input labels
    assert len({l for g in X for l in g[1].values()}) > 256
    eng = get_engine()
    wl, K, levels = _oracle_levels(X, 3)
    gb, _ = wl_batch_from_input(X)
    db = eng.upload(gb)
    assert eng.wl_relabel(db, 3) == wl.label_counts
    for lvl in range(4):
        assert same_partition(eng.wl_labels(db, lvl), levels[lvl]), "level %d" % lvl
    assert np.array_equal(eng.gram(eng.features(db, 4)), K)
    # one graph of 1300 nodes (a long path with chords) among small ones
    n = 1300
    ed = {i: [] for i in range(n)}
    for i in range(n - 1):
        ed[i].append(i + 1), ed[i + 1].append(i)
    for a, b in zip(rs.randint(0, n, 200).tolist(), rs.randint(0, n, 200).tolist()):
        if a != b and b not in ed[a]:
            ed[a].append(b), ed[b].append(a)
    big = [ed, {i: int(rs.randint(0, 3)) for i in range(n)}]
    Y = [big] + er_dataset(80, 25, 0.1, 3, 9)
    wl, K, levels = _oracle_levels(Y, 4)
    gb, _ = wl_batch_from_input(Y)
    db = eng.upload(gb)
    assert eng.wl_relabel(db, 4) == wl.label_counts
    for lvl in range(5):
        assert same_partition(eng.wl_labels(db, lvl), levels[lvl]), "level %d" % lvl
    assert np.array_equal(eng.gram(eng.features(db, 5)), K)
    est = gk.WeisfeilerLehman(n_iter=4, normalize=True)
    ref = O.WLOracle(n_iter=4, normalize=True)
    assert np.allclose(est.fit_transform(Y), ref.fit_transform(Y), rtol=REL_TOL, atol=0)
    assert np.allclose(est.transform(Y[:5]), ref.transform(Y[:5]), rtol=REL_TOL, atol=0)


@pytest.mark.parametrize("bits", [3, 6, 10])
def test_forced_hash_collisions_are_resolved_exactly(gk, bits):
    """Truncated hashes collide massively; the verify + refine loop must still be exact."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    X = random_labelled_graphs(50, 3, 20, 0.3, 3, 21, fmt="dict")
    wl, K, levels = _oracle_levels(X, 3)
    gb, _ = wl_batch_from_input(X)
    eng = get_engine()
    db = eng.upload(gb)
    counts = eng.wl_relabel(db, 3, hash_bits=bits)
    assert db.refine_rounds > 0                      # the exact path really ran
    assert counts == wl.label_counts
    for lvl in range(4):
        assert same_partition(eng.wl_labels(db, lvl), levels[lvl])
    feat = eng.features(db, 4)
    assert np.array_equal(eng.gram(feat), K)


@pytest.mark.parametrize("name", [n for n, _ in SMALL_SETS])
def test_small_sets_against_reference_goldens(gk, name):
    z = load_golden("small_sets.npz")
    kw = dict(SMALL_SETS)[name]
    tr, te = split(random_labelled_graphs(**kw))
    for h in (1, 3):
        wl = gk.WeisfeilerLehman(n_iter=h)
        assert np.array_equal(wl.fit_transform(tr), z["%s/wl%d_fit" % (name, h)])
        assert np.array_equal(wl.transform(te), z["%s/wl%d_tr" % (name, h)])
        assert [x.X.shape[1] for x in wl.X.values()] == z["%s/wl%d_counts" % (name, h)].tolist()
    wln = gk.WeisfeilerLehman(n_iter=2, normalize=True)
    assert np.allclose(wln.fit_transform(tr), z[name + "/wl2n_fit"], rtol=REL_TOL, atol=0)
    assert np.allclose(wln.transform(te), z[name + "/wl2n_tr"], rtol=REL_TOL, atol=0)
    vh = gk.VertexHistogram()
    assert np.array_equal(vh.fit_transform(tr), z[name + "/vh_fit"])
    assert np.array_equal(vh.transform(te), z[name + "/vh_tr"])
    vhn = gk.VertexHistogram(normalize=True)
    assert np.allclose(vhn.fit_transform(tr), z[name + "/vhn_fit"], rtol=REL_TOL, atol=0, equal_nan=True)
    assert np.allclose(vhn.transform(te), z[name + "/vhn_tr"], rtol=REL_TOL, atol=0, equal_nan=True)
    trs, tes = sp_inputs(kw, tr), sp_inputs(kw, te)
    sp = gk.ShortestPath()
    assert np.array_equal(sp.fit_transform(trs), z[name + "/sp_fit"])
    assert np.array_equal(sp.transform(tes), z[name + "/sp_tr"])
    spn = gk.ShortestPath(normalize=True)
    assert np.allclose(spn.fit_transform(trs), z[name + "/spn_fit"], rtol=REL_TOL, atol=0, equal_nan=True)
    assert np.allclose(spn.transform(tes), z[name + "/spn_tr"], rtol=REL_TOL, atol=0, equal_nan=True)
    spu = gk.ShortestPath(with_labels=False)
    assert np.array_equal(spu.fit_transform(trs), z[name + "/spu_fit"])
    assert np.array_equal(spu.transform(tes), z[name + "/spu_tr"])


def test_doc_known_answers(gk):
    H2O = [{'a': ['b', 'c'], 'b': ['a'], 'c': ['a']}, {'a': 'O', 'b': 'H', 'c': 'H'}]
    H3O = [{'a': ['b', 'c', 'd'], 'b': ['a'], 'c': ['a'], 'd': ['a']},
           {'a': 'O', 'b': 'H', 'c': 'H', 'd': 'H'}]
    sp = gk.ShortestPath()
    assert sp.fit_transform([H2O]).tolist() == [[12.0]]          # introduction.rst:325
    assert sp.transform([H3O]).tolist() == [[24.0]]
    spn = gk.ShortestPath(normalize=True)
    spn.fit([H2O])
    assert abs(spn.transform([H3O])[0, 0] - 0.94280904) < 1e-8
    vh = gk.VertexHistogram(normalize=True)
    vh.fit([H2O])
    assert vh.transform([H3O])[0, 0] == pytest.approx(0.9899494936611665, abs=1e-15)
    wl = gk.WeisfeilerLehman(n_iter=5)
    assert wl.fit_transform([H2O, H3O]).tolist() == [[30, 13], [13, 60]]
    assert wl._inv_labels[0] == {'H': 0, 'O': 1}


def test_mutag_against_reference_goldens(gk, mutag_graphs):
    G, z = mutag_graphs
    assert np.array_equal(gk.VertexHistogram().fit_transform(G), z["K_vh"])
    wl = gk.WeisfeilerLehman(n_iter=5)
    K = wl.fit_transform(G)
    assert np.array_equal(K, z["K_wl5"])
    assert [x.X.shape[1] for x in wl.X.values()] == z["wl5_label_counts"].tolist()
    assert np.array_equal(wl.diagonal(), np.diagonal(z["K_wl5"]))
    assert np.linalg.eigvalsh(K).min() > -1e-5            # grakel/tests/test_kernels.py:516-520
    assert np.array_equal(gk.ShortestPath().fit_transform(G), z["K_sp"])
    wl3 = gk.WeisfeilerLehman(n_iter=3)
    wl3.fit(G[:120])
    assert np.array_equal(wl3.transform(G[120:]), z["K_wl3_tr"])
    wl3n = pickle.loads(pickle.dumps(gk.WeisfeilerLehman(n_iter=3, normalize=True).fit(G[:120])))
    assert np.allclose(wl3n.transform(G[120:]), z["K_wl3_tr_norm"], rtol=REL_TOL, atol=0)
    xd, yd = wl3n.diagonal()
    assert xd.shape == (120,) and yd.shape == (68,)
    sp = gk.ShortestPath()
    sp.fit(G[:120])
    assert np.array_equal(sp.transform(G[120:]), z["K_sp_tr"])


def test_wl_sink_node_regression(gk):
    """grakel/tests/test_kernels.py:61-79: a vertex absent from the edge dict keeps its label."""
    g1 = [{(0, 1): 1, (1, 2): 1}, {0: 'a', 1: 'b', 2: 'c'}]
    g2 = [{(0, 1): 1, (2, 1): 1}, {0: 'a', 1: 'b', 2: 'a'}]
    wl = gk.WeisfeilerLehman(n_iter=2, normalize=True)
    K = wl.fit_transform([g1, g2])
    Ko = O.WLOracle(n_iter=2, normalize=True).fit_transform([g1, g2])
    assert K.shape == (2, 2) and np.allclose(np.diagonal(K), 1.0)
    assert np.allclose(K, Ko, rtol=REL_TOL, atol=0)
    wl.fit([g1])
    assert wl.transform([g2]).shape == (1, 1)


def test_er_sets_against_reference_goldens(gk):
    for tag in ("n200", "config2"):
        z = load_golden("er_%s.npz" % tag)
        N, n, L, seed, h = z["params"].tolist()
        p = float(z["p"][0])
        wl = gk.WeisfeilerLehman(n_iter=h)
        K = wl.fit_transform(er_dataset(N, n, p, L, seed))
        assert [x.X.shape[1] for x in wl.X.values()] == z["label_counts"].tolist()
        assert int(K.sum()) == int(z["K_sum"][0]) and int(np.trace(K)) == int(z["K_trace"][0])
        assert np.array_equal(K[:64, :64], z["K_block"])
        assert np.array_equal(K[z["samp_i"], z["samp_j"]], z["samp_v"])
        assert np.array_equal(K.sum(axis=1), z["row_sums"])
        assert np.array_equal(K, K.T)
        # the packed-CSR emitter describes the same graphs
        gp, rp, ci, lab = er_dataset_csr(N, n, p, L, seed)
        K2 = gk.WeisfeilerLehman(n_iter=h).fit_transform(gk.GraphBatch(gp, rp, ci, lab, L))
        assert np.array_equal(K, K2)


def test_config3_full_size_against_reference_checksums(gk):
    """BASELINE config 3 (10k graphs, n=100, h=5) at full size: checksums of the real
    reference's 93 s run (tests/golden/er_config3.npz)."""
    path = os.path.join(os.path.dirname(__file__), "golden", "er_config3.npz")
    if not os.path.exists(path):
        pytest.skip("config-3 golden not generated")
    z = np.load(path)
    N, n, L, seed, h = z["params"].tolist()
    gp, rp, ci, lab = er_dataset_csr(N, n, float(z["p"][0]), L, seed)
    wl = gk.WeisfeilerLehman(n_iter=h)
    K = wl.fit_transform(gk.GraphBatch(gp, rp, ci, lab, L))
    assert [x.X.shape[1] for x in wl.X.values()] == z["label_counts"].tolist()
    assert int(K.sum()) == int(z["K_sum"][0])
    assert int(np.trace(K)) == int(z["K_trace"][0]) and int(K.max()) == int(z["K_max"][0])
    assert np.array_equal(np.diagonal(K), z["diag"])
    assert np.array_equal(K[:64, :64], z["K_block"])
    assert np.array_equal(K[z["samp_i"], z["samp_j"]], z["samp_v"])
    assert np.array_equal(K.sum(axis=1), z["row_sums"])
    assert np.array_equal(K, K.T)


@pytest.mark.parametrize("n,form", [(40, "uint16"), (110, "int32")])
def test_compact_host_copy_is_the_plain_copy(gk, gkopt, n, form):
    """An integer-valued matrix crosses PCIe as uint16 (bound below 2^16) or int32 (below 2^31) and is widened by host
    threads (gram.hip: gram_copy_out); the float64 array the caller gets must be the plain copy's, for every thread count,
    for a row range that does not start at a chunk boundary, and for a normalised (float) matrix, which goes plain."""
    from grakel_amd import GraphBatch
    from grakel_amd.engine import get_engine
    eng = get_engine()
    N = 2300                                                  # 5.3 M entries: above the compact path's 4 Mi threshold
    db = eng.upload(GraphBatch(*er_dataset_csr(N, n, 0.08, 3, 5), 3))
    eng.wl_relabel(db, 5)
    feat = eng.features(db, 6)
    assert (6 * n * n < 65536) == (form == "uint16")
    gkopt("gram.no_compact", 1)
    plain = eng.gram(feat, 0).copy()
    plain_rows = eng.gram(feat, 0, rows=(7, N - 3)).copy()
    norm = eng.gram(feat, 2).copy()
    gkopt("gram.no_compact", 0)
    # the whole symmetric matrix takes the TRIANGLE form (round 5: only the 256 x 256 blocks on / above the diagonal cross
    # PCIe, the host threads widen and mirror them); gram.no_tri = 1 keeps the rectangular form of rounds 3-4
    for no_tri in (0, 1):
        gkopt("gram.no_tri", no_tri)
        for threads in (0, 1, 3, 7):
            gkopt("gram.copy_threads", threads)
            K = eng.gram(feat, 0)
            assert np.array_equal(K, plain), (no_tri, threads)
            del K
        assert np.array_equal(eng.gram(feat, 0, rows=(7, N - 3)), plain_rows)
    assert np.array_equal(eng.gram(feat, 2), norm)            # rectangular form: a normalised matrix goes plain
    # triangle form, normalised: the device keeps the exact integer matrix, the widening threads apply
    # rs[i] * rs[j] (rs = 1 / sqrt(K_ii)) -- exactly symmetric, diagonal exactly 1, within 2 ulp of the device's own
    gkopt("gram.no_tri", 0)
    for threads, no_avx2 in ((0, 0), (5, 0), (3, 1)):
        gkopt("gram.copy_threads", threads)
        gkopt("gram.no_avx2", no_avx2)                       # the SSE2 rows of a CPU without AVX2
        assert np.array_equal(eng.gram(feat, 0), plain)
        Kn = eng.gram(feat, 2)
        assert np.array_equal(Kn, Kn.T) and np.all(np.diagonal(Kn) == 1.0)
        assert np.abs(Kn - norm).max() <= 4 * np.finfo(np.float64).eps
        d = np.sqrt(np.diagonal(plain))
        assert np.abs(Kn - plain / np.outer(d, d)).max() <= 4 * np.finfo(np.float64).eps
        del Kn
    assert plain.max() < (65536 if form == "uint16" else 2 ** 31) and np.array_equal(plain, plain.T)


def test_triangle_copy_edge_blocks_and_graphs_without_features(gk, gkopt):
    """The triangle form of the compact copy on a matrix whose size is NOT a multiple of its 256-entry blocks, with graphs
    that have no ShortestPath feature at all (one vertex: self similarity 0 -> the reference's 0 / 0): NaN rows and columns
    under ShortestPath's plain normalisation, zeros under WL's nan_to_num, everything else as the plain copy's."""
    from grakel_amd.batch import sp_batch_from_input
    from grakel_amd.engine import get_engine
    eng = get_engine()
    rs = np.random.RandomState(11)
    X = []
    for g in range(2085):                                     # 8 full blocks + 37 rows
        n = 1 if g % 97 == 5 else int(rs.randint(2, 7))
        A = np.triu((rs.rand(n, n) < 0.6).astype(int), 1)
        X.append([A + A.T, dict(enumerate(rs.randint(0, 3, n).tolist()))])
    gb, _ = sp_batch_from_input(X, True)
    db = eng.upload(gb)
    pb = eng.sp_build(db, None, True)
    feat = eng.features(pb, 1)
    gkopt("gram.no_compact", 1)
    plain = eng.gram(feat, 0).copy()
    n1 = eng.gram(feat, 1).copy()
    n2 = eng.gram(feat, 2).copy()
    gkopt("gram.no_compact", 0)
    assert np.array_equal(eng.gram(feat, 0), plain)
    for mode, ref in ((1, n1), (2, n2)):
        K = eng.gram(feat, mode)
        assert np.array_equal(np.isnan(K), np.isnan(ref)) and np.isnan(ref).any() == (mode == 1)
        ok = ~np.isnan(ref)
        assert np.abs(K[ok] - ref[ok]).max() <= 4 * np.finfo(np.float64).eps
        assert np.array_equal(np.nan_to_num(K), np.nan_to_num(K).T)
    empty = np.flatnonzero(np.diagonal(plain) == 0)
    assert len(empty) >= 20 and np.all(n2[empty] == 0) and np.all(np.isnan(n1[empty]))


@pytest.mark.parametrize("opts", [(), ("gram.no_fp4",), ("feat.low_df=200",), ("feat.low_df=200", "gram.no_fp4"), ("kind=1",),
                                  ("feat.low_df=200", "gram.pair_cap=8"), ("gram.pair_cap=1", "kind=1")],
                         ids=lambda o: "+".join(o) or "default")
def test_rare_pair_updates_inside_the_tile_kernel_equal_the_atomic_updates(gk, gkopt, opts):
    """A full symmetric job of the warp-specialised tile kernel takes the rare labels' pair updates INTO its parked tiles
    (binned per tile, LDS atomics by the wave that parked the quadrant) instead of float64 atomics afterwards, and then
    normalises in its own epilogue.  Same matrix as the atomic route (which the oracle tests pin), plain and normalised,
    fp4 and int8 operands, dot and min-sum features, also when nearly every column is rare (thousands of pairs per tile,
    several rounds per wave), on diagonal tiles, and when the per-tile buckets are far too small (``gram.pair_cap``: the
    overflow list then carries most pairs; its normalised contributions are added separately, hence 1e-13)."""
    from grakel_amd import GraphBatch
    from grakel_amd.engine import get_engine
    eng = get_engine()
    kind = 0
    gkopt("gram.dd", 2)                                        # small jobs would take the direct-store form, which does not fold
    gkopt("gram.fold", 1)                                      # also for the unnormalised matrix (by default only where it pays)
    for o in opts:
        name, _, val = o.partition("=")
        if name == "kind":
            kind = int(val)
        else:
            gkopt(name, int(val) if val else 1)
    N = 700
    db = eng.upload(GraphBatch(*er_dataset_csr(N, 30, 0.1, 4, 3), 4))
    eng.wl_relabel(db, 3)
    feat = eng.features(db, 4, kind=kind)
    assert feat.n_cols_low > 0
    folded, folded_n = eng.gram(feat, 0).copy(), eng.gram(feat, 2).copy()
    gkopt("gram.fold", 2)
    atomics, atomics_n = eng.gram(feat, 0).copy(), eng.gram(feat, 2).copy()
    assert np.array_equal(folded, atomics) and np.array_equal(folded, folded.T)
    assert np.allclose(folded_n, atomics_n, rtol=1e-13, atol=0)
    if kind == 0:
        wl = O.WLOracle(n_iter=3)
        from grakel_amd.synthetic import er_dataset
        assert np.array_equal(folded[:60, :60], wl.fit_transform(er_dataset(N, 30, 0.1, 4, 3)[:60]))


def test_counts_above_127_take_the_f64_path(gk):
    """A label occurring > 127 times in one graph cannot be an int8 operand."""
    rs = np.random.RandomState(3)
    G = []
    for n in (300, 150, 40, 260):
        A = np.triu((rs.rand(n, n) < 0.02).astype(int), 1)
        A = A + A.T
        G.append([A, {i: int(rs.rand() < 0.1) for i in range(n)}])
    wl = gk.WeisfeilerLehman(n_iter=2)
    K = wl.fit_transform(G)
    assert wl._last_info["max_count"] > 127          # such columns go to the float64 side operand
    assert np.array_equal(K, O.WLOracle(n_iter=2).fit_transform(G))
    assert np.array_equal(gk.VertexHistogram().fit_transform(G), O.VHOracle().fit_transform(G))


@pytest.mark.parametrize("n,n_labels,parts", [(300, 2, 2), (900, 3, 3)])
def test_counts_of_128_to_381_are_split_into_int8_digit_columns(gk, gkopt, n, n_labels, parts):
    """A label with counts above 127 but at most 3 x 127 stays in the int8 GEMM: parts^2 columns holding the digits of
    the count, digit p on the left and digit r on the right at column (p, r), so that left . right^T is the exact product.
    The dense term against the host product of the two operands, then the public API against the oracle with and
    without the split (option gram.no_split8 = the float64 side operand), normalised and as fit + transform."""
    from grakel_amd import GraphBatch
    from grakel_amd.engine import get_engine
    gkopt("feat.low_df", 2)
    eng = get_engine()
    N = 70
    db = eng.upload(GraphBatch(*er_dataset_csr(N, n, 2.0 / n, n_labels, 5), n_labels))
    eng.wl_relabel(db, 1)
    feat = eng.features(db, 2)
    assert 127 * (parts - 1) < feat.max_count <= 127 * parts and feat.n_cols_low == 0
    left = eng.debug_phi(feat)
    right, got_parts = eng.debug_phi_right(feat)
    assert got_parts == parts and left.max() <= 127 and right.max() <= 127 and not np.array_equal(left, right)
    assert eng.lib.gk_features_operand is not None and "f64" not in feat.operand
    K = eng.gram(feat, 0)
    R = left @ right.T
    assert np.array_equal(R, R.T)
    np.fill_diagonal(R, eng.selfk(feat))
    assert np.array_equal(K, R)
    d = eng.selfk(feat)
    Kn = eng.gram(feat, 2, rows=(3, N - 2))
    assert np.allclose(Kn, R[3:N - 2] / np.sqrt(np.outer(d[3:N - 2], d)), rtol=1e-13, atol=0)
    gkopt("gram.no_split8", 1)
    feat2 = eng.features(db, 2)
    assert "f64" in feat2.operand and np.array_equal(eng.gram(feat2, 0), K)
    gkopt("gram.no_split8", 0)
    # every form of the tile kernel takes the two operands (left rows x right rows), also without the symmetric shortcut
    for form in (("gram.no_ws",), ("gram.dd",), ("gram.dd=2",), ("gram.no_sym",), ("gram.no_patch",), ("gram.no_ws", "gram.no_sym")):
        for k in form:
            gkopt(k.split("=")[0], int(k.split("=")[1]) if "=" in k else 1)
        assert np.array_equal(eng.gram(feat, 0), K), form
        assert np.array_equal(eng.gram(feat, 0, rows=(5, N - 9)), K[5:N - 9]), form
        for k in form:
            gkopt(k.split("=")[0], 0)
    # public API
    rs = np.random.RandomState(n)
    G = []
    for k in range(24):
        m = n - 10 * (k % 5)
        A = np.triu((rs.rand(m, m) < 2.0 / m).astype(int), 1)
        G.append([A + A.T, {i: int(rs.randint(n_labels)) for i in range(m)}])
    for norm in (False, True):
        want_fit = O.WLOracle(n_iter=2, normalize=norm)
        Kf = want_fit.fit_transform(G[:16])
        Kt = want_fit.transform(G[16:])
        for no_split in (0, 1):
            gkopt("gram.no_split8", no_split)
            wl = gk.WeisfeilerLehman(n_iter=2, normalize=norm)
            got = wl.fit_transform(G[:16])
            assert ("f64" in wl._last_info["dtype"]) == bool(no_split)
            if norm:
                assert np.allclose(got, Kf, rtol=1e-13, atol=0) and np.allclose(wl.transform(G[16:]), Kt, rtol=1e-13, atol=0)
            else:
                assert np.array_equal(got, Kf) and np.array_equal(wl.transform(G[16:]), Kt)
    gkopt("gram.no_split8", 0)
    sp_graphs = [[g[0], {i: 0 for i in range(len(g[1]))}] for g in G[:6]]          # one label: pair counts far above 381 stay float64
    assert np.array_equal(gk.ShortestPath().fit_transform(sp_graphs), O.SPOracle().fit_transform(sp_graphs))


def test_apsp_known_answer_and_large_graphs(gk):
    from grakel_amd.batch import sp_batch_from_input
    from grakel_amd.engine import get_engine
    # grakel/tests/test_graph.py:61-74 (weighted, directed)
    A = np.array([[0, 1, 0, 3], [1, 0, 0, 2], [2, 3, 0, 1], [1, 0, 0, 0]])
    rs = np.random.RandomState(9)
    n = 330                                             # > LDS Floyd-Warshall cap: relax kernel
    B = np.triu((rs.rand(n, n) < 0.01).astype(int) * rs.randint(1, 4, (n, n)), 1)
    B = B + B.T
    G = [[A, {i: 0 for i in range(4)}], [B, {i: i % 3 for i in range(n)}]]
    gb, _ = sp_batch_from_input(G, True)
    eng = get_engine()
    db = eng.upload(gb)
    S = eng.sp_debug_apsp(db, gb.edge_weight, 0, 4)
    assert S.tolist() == [[0, 1, -1, 3], [1, 0, -1, 2], [2, 3, 0, 1], [1, 2, -1, 0]]
    S1 = eng.sp_debug_apsp(db, gb.edge_weight, 1, n).astype(float)
    S1[S1 < 0] = np.inf
    assert np.array_equal(S1, O.floyd_warshall(B))
    assert np.array_equal(gk.ShortestPath().fit_transform(G), O.SPOracle().fit_transform(G))


def test_nci1_like_sp_against_reference_goldens(gk):
    z = load_golden("nci1_like_sp_300.npz")
    sp = gk.ShortestPath()
    K = sp.fit_transform(nci1_like(300, 0, as_adj=True))
    assert len(sp._enum) == int(z["n_features"][0])
    assert int(K.sum()) == int(z["K_sum"][0]) and int(K.max()) == int(z["K_max"][0])
    assert np.array_equal(K[:64, :64], z["K_block"])
    assert np.array_equal(K, gk.ShortestPath().fit_transform(nci1_like(300, 0, as_adj=False)))
    z = load_golden("nci1_like_sp_4110.npz")               # BASELINE config 4 stand-in, full size
    sp = gk.ShortestPath()
    K = sp.fit_transform(nci1_like(4110, 0, as_adj=True))
    assert len(sp._enum) == int(z["n_features"][0])
    assert int(K.sum()) == int(z["K_sum"][0]) and int(K.max()) == int(z["K_max"][0])
    assert np.array_equal(np.diagonal(K), z["diag"])
    assert np.array_equal(K[z["samp_i"], z["samp_j"]], z["samp_v"])
    assert np.array_equal(K.sum(axis=1), z["row_sums"])


def _sp_rank_worker(rank, world, port, out_dir, n_graphs, weighted):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from grakel_amd.batch import sp_batch_from_input
    from grakel_amd.dist import ShardedSP, shard_bounds
    from grakel_amd.engine import get_engine
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full, _ = sp_batch_from_input(_sp_shard_input(n_graphs, weighted), True)
        b = shard_bounds(full.n_graphs, world)
        sp = ShardedSP(get_engine(0))
        for _ in range(2):
            K, info = sp.step(full.slice_graphs(b[rank], b[rank + 1]), to_host=True)
        assert info["rows"] == (b[rank], b[rank + 1])
        np.save(os.path.join(out_dir, "Ksp_%d.npy" % rank), K)
        np.save(os.path.join(out_dir, "nsp_%d.npy" % rank), np.array([info["n_keys"], info["n_pairs"]]))
        sp.close()
    finally:
        dist.destroy_process_group()


def _sp_shard_input(n_graphs, weighted):
    if weighted == "float":                                   # general float weights, edge dictionaries, directed matrices
        import sys
        sys.path.insert(0, GOLDEN)
        from small_sets import sp_float_graphs
        return sp_float_graphs(n_graphs)
    G = nci1_like(n_graphs, 0, as_adj=True)
    if weighted:                                              # integer edge weights 1..3 on the adjacency matrices
        rs = np.random.RandomState(3)
        for g in G:
            A = g[0]
            W = np.triu(rs.randint(1, 4, A.shape), 1)
            g[0] = A * (W + W.T)
    return G


@pytest.mark.parametrize("n_graphs,weighted", [(4110, False), (300, True), (40, "float")])
def test_sharded_shortest_path_two_processes_on_one_gpu(gk, tmp_path, n_graphs, weighted):
    """ShardedSP with two ranks (both on cuda:0, gloo): shard -> all-gather (CSR + edge weights) -> distances, pair
    dictionary and features on the global batch -> the rank's Gram rows.  The stacked row blocks are the
    single-process matrix; at 4110 graphs that is BASELINE config 4's stand-in, checked against the real reference's
    golden (tests/golden/nci1_like_sp_4110.npz); the third case has general float edge weights (float64 weights and the
    per-graph dictionary flags travel with the shards, the reference's float distances are reproduced on every rank)."""
    import torch.multiprocessing as mp
    port = 29000 + (os.getpid() % 2000)
    mp.spawn(_sp_rank_worker, args=(2, port, str(tmp_path), n_graphs, weighted), nprocs=2, join=True)
    K = np.vstack([np.load(os.path.join(str(tmp_path), "Ksp_%d.npy" % r)) for r in range(2)])
    nk = np.load(os.path.join(str(tmp_path), "nsp_0.npy"))
    if not weighted:
        z = load_golden("nci1_like_sp_4110.npz")
        assert int(nk[0]) == int(z["n_features"][0])
        assert int(K.sum()) == int(z["K_sum"][0]) and int(K.max()) == int(z["K_max"][0])
        assert np.array_equal(np.diagonal(K), z["diag"]) and np.array_equal(K.sum(axis=1), z["row_sums"])
        assert np.array_equal(K[z["samp_i"], z["samp_j"]], z["samp_v"])
    else:
        G = _sp_shard_input(n_graphs, weighted)
        assert np.array_equal(K, O.SPOracle().fit_transform(G))
        assert np.array_equal(K, gk.ShortestPath().fit_transform(G))


@pytest.mark.parametrize("route", [(), ("sp.no_hist",), ("sp.no_pk",), ("sp.no_reg",), ("sp.no_hist", "sp.no_reg"),
                                   ("feat.gm_row_lds_max",), ("feat.gm_no_priv",), ("sp.no_hist", "scan.direct_max"),
                                   ("sp.no_rows",), ("sp.no_bfs",), ("sp.bfs_no_lds_cols",), ("gram.no_split64",), ("gram.no_sym",), ("sp.rows_all",), ("sp.rows_all", "sp.hist_unit=1"),
                                   ("sp.rows_all", "sp.hist_unit=300", "sp.hist_slots=16"), ("sp.hist_slots=32", "feat.gm_no_priv"), ("sp.no_prep",), ("sp.bfs_one_stream",), ("sp.bfs_no_bytes",), ("sp.hist_no_batch",), ("feat.gm_rows_256",), ("sp.static_type",), ("sp.static_type=2",), ("sp.no_fused_mark",), ("sp.rows_all", "sp.rows_no_merge"), ("sp.rows_all", "sp.rows_no_merge=2"), ("sp.rows_all", "sp.rows_no_merge=4"), ("sp.rows_all", "sp.hist_unit=300", "sp.hist_slots=16", "sp.rows_no_merge=3"),
                                   ("sp.no_hist", "sp.bfs_no_bytes")],
                         ids=lambda r: "+".join(r) or "default")
def test_every_shortest_path_route_gives_the_reference_matrix(gk, gkopt, route):
    """ShortestPath picks among equivalent routes: all-pairs distances in 16-bit packed registers (one wave per graph up to
    64 vertices, a four-wave workgroup up to 128), in 32-bit registers, or in the LDS workgroup kernel; pair features as
    per-graph histograms of the distance matrices or through explicit pair items, the sorting dictionary and the
    label-major builder (also reached when the histogram builder declines: feat.gm_row_lds_max); the histograms of graphs
    above 6 144 pairs through counter rows in HBM (sp.rows_all: every graph; one matrix row per counting workgroup; an LDS
    table of 16 slots, i.e. nearly every key spilled to the row with a global atomic) or not (sp.no_rows).  Every route must
    give the reference's matrix (fit_transform and transform), with unit, integer and dyadic float weights."""
    import sys
    sys.path.insert(0, GOLDEN)
    from small_sets import sp_dyadic_graphs
    for name in route:
        name, _, val = name.partition("=")
        gkopt(name, int(val) if val else (64 if name == "feat.gm_row_lds_max" else 1))
    z = load_golden("nci1_like_sp_300.npz")
    G = nci1_like(300, 0, as_adj=True)
    sp = gk.ShortestPath()
    K = sp.fit_transform(G)
    assert int(K.sum()) == int(z["K_sum"][0]) and int(K.max()) == int(z["K_max"][0])
    assert np.array_equal(K[:64, :64], z["K_block"])
    assert np.array_equal(sp.transform(G[100:140]), K[100:140])
    spn = gk.ShortestPath(normalize=True)
    ref = O.SPOracle(normalize=True)
    assert np.allclose(spn.fit_transform(G[:60]), ref.fit_transform(G[:60]), rtol=REL_TOL, atol=0)
    assert np.allclose(spn.transform(G[60:75]), ref.transform(G[60:75]), rtol=REL_TOL, atol=0)
    W = _sp_shard_input(80, True)                              # integer weights 1..3
    assert np.array_equal(gk.ShortestPath().fit_transform(W), O.SPOracle().fit_transform(W))
    D = sp_dyadic_graphs()
    zd = load_golden("sp_dyadic.npz")
    spd = gk.ShortestPath()
    assert np.array_equal(spd.fit_transform(D[:16]), zd["K_fit_auto"]) and np.array_equal(spd.transform(D[16:]), zd["K_tr_auto"])
    big = random_labelled_graphs(5, 140, 210, 0.02, 3, 3, fmt="adj") + G[:20]     # graphs above 128 vertices and above the LDS cap
    assert np.array_equal(gk.ShortestPath().fit_transform(big), O.SPOracle().fit_transform(big))


@pytest.mark.parametrize("no_bfs", [0, 1, 2, 3, 4, 5])
def test_large_unit_weight_graphs_directed_hubs_and_unreachable_pairs(gk, gkopt, no_bfs):
    """Graphs above 128 vertices with unit weights take the bit-parallel breadth-first search (sp.hip: sp_msbfs_kernel;
    1: the row relaxation instead, 2: the search with the adjacency entries read from HBM instead of LDS, 3: the search
    leaving 32-bit matrices instead of byte matrices, 4 / 5: byte matrices read by the pair-item route / by the LDS-table
    histograms of every graph that fits one).  What the
    REDDIT- / D&D-like goldens do not hold (tests/golden/small_sets.py: sp_large_unit_graphs; golden from the real
    reference in sp_large_unit.npz): DIRECTED adjacency matrices (d[u][v] follows the out-edges of u), a hub above 32 and
    one above 1 024 neighbours next to vertices nothing leads to, more than 1 024 vertices (a thread owns several),
    isolated vertices -- and a path of 300 vertices: distances up to 299, the search keeps its levels in bytes, reports
    the overflow and the job is repeated with the row relaxation."""
    import sys
    sys.path.insert(0, GOLDEN)
    from small_sets import sp_large_unit_graphs, sp_large_unit_paths
    gkopt("sp.no_bfs", 1 if no_bfs == 1 else 0)
    gkopt("sp.bfs_no_lds_cols", 1 if no_bfs == 2 else 0)
    gkopt("sp.bfs_no_bytes", 1 if no_bfs == 3 else 0)
    gkopt("sp.no_hist", 1 if no_bfs == 4 else 0)
    gkopt("sp.no_rows", 1 if no_bfs == 5 else 0)
    z = load_golden("sp_large_unit.npz")
    G, P = sp_large_unit_graphs(), sp_large_unit_paths()
    assert np.array_equal(gk.ShortestPath().fit_transform(G), z["K"])
    assert np.array_equal(gk.ShortestPath(with_labels=False).fit_transform(G), z["K_nolabels"])
    assert np.array_equal(gk.ShortestPath().fit_transform(P), z["K_paths"])
    sp = gk.ShortestPath()
    sp.fit(G[:3])
    assert np.array_equal(sp.transform(G[3:] + P[2:]), z["K_tr"])


def test_sp_histogram_table_overflow_in_a_job_of_small_graphs(gk):
    """No graph above 128 vertices: the job counts with one workgroup and one LDS table per graph and skips the counter rows
    (features_gm.hip).  120 vertices with a label each are ~14 000 pairs with nearly as many distinct (label, label,
    distance) keys -- more than the table's 6 144: the builder reports the overflow and the job is repeated with counter
    rows instead of leaving for the pair items."""
    rs = np.random.RandomState(11)
    G = []
    for n in (120, 100, 30):
        A = (rs.rand(n, n) < 0.06).astype(np.int64)
        A = np.triu(A, 1)
        A = A + A.T
        G.append([A, dict(enumerate(rs.permutation(n).tolist()))])
    assert np.array_equal(gk.ShortestPath().fit_transform(G), O.SPOracle().fit_transform(G))


def test_sp_float_weights_against_reference_goldens(gk):
    """Float edge weights that are integer multiples of a power of two (here 1/8): integer distances in that
    unit on the device, the reference's matrices and float-keyed ``_enum`` (graph.py:1767-1794,
    shortest_path.py:389).  General float weights: test_sp_general_float_weights_against_reference_goldens."""
    import sys
    sys.path.insert(0, GOLDEN)
    from small_sets import sp_dyadic_graphs
    z = load_golden("sp_dyadic.npz")
    G = sp_dyadic_graphs()
    tr, te = G[:16], G[16:]
    for name, algo in (("auto", "auto"), ("fw", "floyd_warshall")):
        sp = gk.ShortestPath(algorithm_type=algo)
        assert np.array_equal(sp.fit_transform(tr), z["K_fit_" + name])
        assert np.array_equal(sp.transform(te), z["K_tr_" + name])
        keys = sorted(sp._enum.items(), key=lambda kv: kv[1])
        assert [[k[0], k[1]] for k, _ in keys] == z["enum_labels_" + name].tolist()
        assert [float(k[2]) for k, _ in keys] == z["enum_dist_" + name].tolist()
    spn = gk.ShortestPath(normalize=True)
    assert np.allclose(spn.fit_transform(tr), z["K_fit_norm"], rtol=1e-5, atol=0)
    assert np.allclose(spn.transform(te), z["K_tr_norm"], rtol=1e-5, atol=0)
    assert np.array_equal(gk.ShortestPath(with_labels=False).fit_transform([[g[0]] for g in tr]), z["K_fit_unlabelled"])
    # targets with integer weights against a fit in units of 1/8: the union counts in the finer unit
    sp = gk.ShortestPath()
    sp.fit(tr)
    ints = [[np.rint(np.asarray(g[0]) * 8), g[1]] for g in te if not isinstance(g[0], dict)]
    eighths = [[np.asarray(g[0]) * 8 / 8.0, g[1]] for g in te if not isinstance(g[0], dict)]
    spo = O.SPOracle()
    spo.fit_transform(tr)
    assert np.array_equal(sp.transform(ints), spo.transform(ints))
    assert np.array_equal(sp.transform(eighths), spo.transform(eighths))


@pytest.mark.parametrize("route", [(), ("sp.no_hist",)], ids=lambda r: "+".join(r) or "default")
def test_sp_general_float_weights_against_reference_goldens(gk, gkopt, route):
    """General positive float edge weights (0.1-multiples, random floats, directed matrices, edge dictionaries).  The
    reference keys its features by its own rounded float path sums, which differ between its floyd_warshall and its dijkstra
    (and between the two directions of a pair): the device reproduces both bit for bit (sp.hip: sp_f64_kernel), ranks the
    distinct distances and goes on as with integer distances.  Golden: the real reference's three different matrices for
    algorithm_type auto / floyd_warshall / dijkstra, its transform, its ``_enum`` keys down to the bits of the distances,
    normalised and unlabelled variants, WL over ShortestPath (tests/golden/sp_float.npz)."""
    import sys
    sys.path.insert(0, GOLDEN)
    from small_sets import sp_float_graphs
    for name in route:
        gkopt(name, 1)
    z = load_golden("sp_float.npz")
    G = sp_float_graphs()
    tr, te = G[:28], G[28:]
    for name, algo in (("auto", "auto"), ("fw", "floyd_warshall"), ("dij", "dijkstra")):
        sp = gk.ShortestPath(algorithm_type=algo)
        assert np.array_equal(sp.fit_transform(tr), z["K_fit_" + name]), name
        assert np.array_equal(sp.transform(te), z["K_tr_" + name]), name
        keys = sorted(sp._enum.items(), key=lambda kv: kv[1])
        assert [[k[0], k[1]] for k, _ in keys] == z["enum_labels_" + name].tolist()
        assert np.array([float(k[2]) for k, _ in keys]).view(np.int64).tolist() == z["enum_dist_bits_" + name].tolist()
    spn = gk.ShortestPath(normalize=True)
    assert np.allclose(spn.fit_transform(tr), z["K_fit_norm"], rtol=1e-12, atol=0, equal_nan=True)
    assert np.allclose(spn.transform(te), z["K_tr_norm"], rtol=1e-12, atol=0, equal_nan=True)
    assert np.array_equal(gk.ShortestPath(with_labels=False).fit_transform([[g[0]] for g in tr]), z["K_fit_unlabelled"])
    wl = gk.WeisfeilerLehman(n_iter=2, base_graph_kernel=gk.ShortestPath)
    assert np.array_equal(wl.fit_transform(tr), z["K_fit_wl_sp"])
    assert np.array_equal(wl.transform(te), z["K_tr_wl_sp"])
    # a fit on integer weights, targets with general floats: the union counts in float64 (the oracle is pinned to the
    # reference on this golden by tests/test_oracle.py)
    ints = [[np.rint(np.asarray(g[0]) * 10), g[1]] for g in tr if not isinstance(g[0], dict)]
    sp, spo = gk.ShortestPath(), O.SPOracle()
    sp.fit(ints), spo.fit_transform(ints)
    assert np.array_equal(sp.transform(te), spo.transform(te))


def test_sp_general_float_weights_above_143_vertices_and_core_framework(gk):
    """Round 4: a graph whose float64 distance matrix does not fit LDS works on it in HBM (sp.hip: sp_f64_big_kernel) -- the
    reference's three different matrices on graphs of 150-200 vertices, the float distances of the feature keys bit for bit;
    CoreFramework over ShortestPath on general float weights (its subgraphs: dijkstra under "auto").  Goldens from the real
    reference (tests/golden/sp_float_big.npz)."""
    from golden.small_sets import sp_float_big_graphs, sp_float_graphs
    z = load_golden("sp_float_big.npz")
    G = sp_float_big_graphs()
    tr, te = G[:6], G[6:]
    for name, kw in (("auto", {}), ("fw", dict(algorithm_type="floyd_warshall")), ("dij", dict(algorithm_type="dijkstra"))):
        sp = gk.ShortestPath(**kw)
        assert np.array_equal(sp.fit_transform(tr), z["K_fit_" + name]), name
        assert np.array_equal(sp.transform(te), z["K_tr_" + name]), name
        keys = sorted(sp._enum.items(), key=lambda kv: kv[1])
        assert np.array([float(k[2]) for k, _ in keys]).view(np.int64).tolist() == z["enum_dist_bits_" + name].tolist()
    S = [g for i, g in enumerate(sp_float_graphs()) if i < 4 or i % 7]
    ctr, cte = S[:24], S[24:]
    for name, kw in (("auto", {}), ("fw", dict(algorithm_type="floyd_warshall"))):
        cf = gk.CoreFramework(base_graph_kernel=(gk.ShortestPath, kw))
        assert np.array_equal(cf.fit_transform(ctr), z["K_core_fit_" + name]), name
        assert np.array_equal(cf.transform(cte), z["K_core_tr_" + name]), name
    with np.errstate(divide="ignore", invalid="ignore"):
        assert np.allclose(gk.CoreFramework(normalize=True).fit_transform(ctr), z["K_core_fit_norm"], rtol=1e-12, atol=0, equal_nan=True)


def test_errors_match_reference(gk):
    with pytest.raises(TypeError):
        gk.WeisfeilerLehman().fit_transform(5)
    with pytest.raises(ValueError):
        with pytest.warns(UserWarning):
            gk.WeisfeilerLehman().fit_transform([[]])
    with pytest.raises(TypeError):
        gk.WeisfeilerLehman(n_iter=0).fit_transform([[{0: [1], 1: [0]}, {0: 1, 1: 2}]])
    with pytest.raises(ValueError):
        gk.ShortestPath(algorithm_type="bfs").fit([[{0: [1], 1: [0]}, {0: 1, 1: 2}]])
    from sklearn.exceptions import NotFittedError
    with pytest.raises(NotFittedError):
        gk.WeisfeilerLehman().transform([[{0: [1], 1: [0]}, {0: 1, 1: 2}]])


def test_sharded_path_single_rank_matches_plain_path(gk):
    """grakel_amd/dist.py end to end on one GPU: a 1-rank RCCL group exercises the all-gather,
    the device-side CSR rebuild (gk_batch_from_shards) and gk_gram_rows."""
    import torch
    import torch.distributed as dist
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.dist import ShardedWL
    from grakel_amd.engine import get_engine
    X = er_dataset(300, 30, 0.1, 4, 5)
    K = gk.WeisfeilerLehman(n_iter=3).fit_transform(X)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        gb, _ = wl_batch_from_input(X)
        sw = ShardedWL(get_engine(), n_iter=3)
        for _ in range(3):                           # repeated steps reuse the gather buffer and the side stream
            Kr, info = sw.step(gb, to_host=True)
            assert info["rows"] == (0, 300) and np.array_equal(Kr, K)
        sw.close()
        # a row block (what rank r of R computes) equals the same rows of the full matrix
        eng = get_engine()
        db = eng.upload(gb)
        eng.wl_relabel(db, 3)
        feat = eng.features(db, 4)
        assert np.array_equal(eng.gram(feat, 0, rows=(37, 211)), K[37:211])
        Kn = gk.WeisfeilerLehman(n_iter=3, normalize=True).fit_transform(X)
        assert np.allclose(eng.gram(feat, 2, rows=(100, 300)), Kn[100:300], rtol=REL_TOL, atol=0)
    finally:
        if created:
            dist.destroy_process_group()


def test_c_abi_collectives_with_a_world_of_one(gk):
    """include/gk_hip.h "multi-GPU" (csrc/comm.hip): gk_comm_unique_id / gk_comm_init / gk_batch_allgather /
    gk_gram_sharded through RCCL itself -- a communicator of one rank (this box has one GPU; RCCL wants one GPU per
    rank).  The message layout and the device-side rebuild are the ones the two-process tests below drive over gloo."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    from grakel_amd import _lib
    eng = get_engine()
    X = er_dataset(257, 25, 0.12, 4, 8)
    K = gk.WeisfeilerLehman(n_iter=3).fit_transform(X)
    Kn = gk.WeisfeilerLehman(n_iter=3, normalize=True).fit_transform(X)
    gb, _ = wl_batch_from_input(X)
    uid = eng.comm_unique_id()
    assert len(uid) == 128
    comm = eng.comm_init(0, 1, uid)
    try:
        for _ in range(2):
            db, bounds = eng.batch_allgather(comm, gb)
            assert bounds.tolist() == [0, 257] and (db.n_graphs, db.n_nodes, db.n_edges) == (gb.n_graphs, gb.n_nodes, gb.n_edges)
            eng.wl_relabel(db, 3)
            feat = eng.features(db, 4)
            assert np.array_equal(eng.gram_sharded(comm, feat, bounds, 0), K)
            assert np.allclose(eng.gram_sharded(comm, feat, bounds, 2), Kn, rtol=REL_TOL, atol=0)
            # the rows of another split of the same job (what rank 1 of 3 would own)
            assert np.array_equal(eng.gram(feat, 0, rows=(86, 172)), K[86:172])
            feat.close()
            db.close()
        with pytest.raises(_lib.GkError):                         # a shard that does not start at 0
            bad = type("B", (), dict(n_graphs=gb.n_graphs, n_nodes=gb.n_nodes, n_edges=gb.n_edges, graph_ptr=gb.graph_ptr + 1,
                                     row_ptr=gb.row_ptr, col_idx=gb.col_idx, node_label=gb.node_label, n_labels=gb.n_labels))
            eng.batch_allgather(comm, bad)
        with pytest.raises(_lib.GkError):
            eng.comm_init(2, 2, uid)                                  # rank outside [0, n_ranks)
    finally:
        comm.close()


def _two_rank_worker(rank, world, port, out_dir):
    """One of two processes sharing cuda:0 (gloo moves the shard messages; RCCL needs one GPU per rank)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.dist import ShardedWL, shard_bounds
    from grakel_amd.engine import get_engine
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full, _ = wl_batch_from_input(er_dataset(301, 30, 0.1, 4, 5))      # 301 graphs: ragged over 2 ranks
        b = shard_bounds(full.n_graphs, world)
        local = full.slice_graphs(b[rank], b[rank + 1])                     # this rank only holds its shard
        sw = ShardedWL(get_engine(0), n_iter=3, symmetric=True)            # the symmetric plan (not the default)
        for _ in range(2):                                                  # the second step reuses the exchange
            K, info = sw.step(local, to_host=True)
        assert info["rows"] == (b[rank], b[rank + 1]) and info["n_graphs"] == 301
        np.save(os.path.join(out_dir, "K_%d.npy" % rank), K)
        np.save(os.path.join(out_dir, "flops_%d.npy" % rank), np.array([info["gram"][0], info["n_cols"]]))
        # the DEFAULT plan: every rank multiplies and stores its full row block (no exchange); same rows
        assert ShardedWL(get_engine(0), n_iter=3).symmetric is False
        sw.symmetric = False
        K_full, info_full = sw.step(local, to_host=True)
        assert np.array_equal(K_full, K) and info_full["gram"][0] > 1.8 * info["gram"][0]
        # the operand-row exchange north_star names (exchange="phi"): every rank assembles the rows of ITS graphs only, the
        # row shards (151 and 150 rows: ragged) are all-gathered -- same rows of the same matrix, normalised too
        sp = ShardedWL(get_engine(0), n_iter=3, exchange="phi")
        sp._stream, sw._stream = sw._stream, None
        K_phi, info_phi = sp.step(local, to_host=True)
        assert np.array_equal(K_phi, K) and sp.phi_bytes > 0 and info_phi["n_cols"] == info_full["n_cols"]
        K_phi2, _ = sp.step(local, to_host=True)
        assert np.array_equal(K_phi2, K)
        sw._stream, sp._stream = sp._stream, None
        swn = ShardedWL(get_engine(0), n_iter=3, normalize=True, exchange="phi")
        swn._stream, sw._stream = sw._stream, None
        Kn, _ = swn.step(local, to_host=True)
        np.save(os.path.join(out_dir, "Kn_%d.npy" % rank), Kn)
        swn.close()
    finally:
        dist.destroy_process_group()


def test_sharded_path_two_processes_on_one_gpu(gk, tmp_path):
    """The multi-GPU step end to end with TWO ranks (shard -> all-gather -> gk_batch_from_shards ->
    relabel -> features -> gk_gram_rows of the rank's rows): both processes use cuda:0 and the gloo
    backend, so everything except RCCL itself is what runs on an 8-GPU node."""
    import torch.multiprocessing as mp
    K = gk.WeisfeilerLehman(n_iter=3).fit_transform(er_dataset(301, 30, 0.1, 4, 5))
    port = 29000 + (os.getpid() % 2000)
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    blocks = [np.load(os.path.join(str(tmp_path), "K_%d.npy" % r)) for r in range(2)]
    assert blocks[0].shape == (151, 301) and blocks[1].shape == (150, 301)
    assert np.array_equal(np.vstack(blocks), K)
    # symmetric sharding: the two ranks TOGETHER multiply what one GPU multiplies alone (entries on/above the
    # diagonal x dense columns), each about half of it
    fl = [np.load(os.path.join(str(tmp_path), "flops_%d.npy" % r)) for r in range(2)]
    one_gpu = 2.0 * (301 * 302 / 2) * fl[0][1]
    assert fl[0][0] + fl[1][0] == one_gpu
    assert abs(fl[0][0] - fl[1][0]) < 0.02 * one_gpu
    Kn = np.vstack([np.load(os.path.join(str(tmp_path), "Kn_%d.npy" % r)) for r in range(2)])
    d = np.sqrt(np.diagonal(K))
    assert np.allclose(Kn, K / np.outer(d, d), rtol=REL_TOL, atol=0)


def test_bench_two_rank_code_path_on_one_gpu(gk):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per process), here
    with both ranks on cuda:0 over gloo (GK_BENCH_BACKEND test hook): the script's sharded branch, its
    max-over-ranks timing and its JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GK_BENCH_BACKEND="gloo", GK_BENCH_DEVICE="0")
    port = 29100 + (os.getpid() % 800)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--graphs", "600"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["cpu_baseline"] is None and d["roofline"]["traffic"] is None
    from grakel_amd import GraphBatch
    from grakel_amd.engine import get_engine
    eng = get_engine()
    db = eng.upload(GraphBatch(*er_dataset_csr(600, 100, 0.05, 5, 0), 5))
    assert d["config"]["label_counts"] == eng.wl_relabel(db, 5)


# ------------------------------------------------------------------------------------------
# edge cases (ragged / degenerate inputs), each against the CPU oracle
# ------------------------------------------------------------------------------------------
def _edge_sets():
    g_single = [{0: []}, {0: 'x'}]                                   # one vertex, no edge
    g_loop = [{(0, 0): 1, (0, 1): 1, (1, 0): 1}, {0: 'a', 1: 'b'}]    # self loop
    g_directed = [[(0, 1), (1, 2), (2, 0), (0, 3)], {0: 'a', 1: 'a', 2: 'b', 3: 'a'}]
    g_big_ids = [{10**9: [-5], -5: [10**9]}, {10**9: -7, -5: 2**40}]
    g_str = [{'u': {'v': 1.0, 'w': 1.0}, 'v': {'u': 1.0}, 'w': {'u': 1.0}}, {'u': 'N', 'v': 'C', 'w': 'C'}]
    g_dup = [np.array([[0, 1, 1], [1, 0, 0], [1, 0, 0]]), {0: 'N', 1: 'C', 2: 'C'}]
    g_float_lab = [np.array([[0, 1], [1, 0]]), {0: 0.5, 1: 1.5}]
    g_tuple_lab = [np.array([[0, 1], [1, 0]]), {0: (1, 2), 1: (1, 3)}]
    return dict(single=[g_single], loops=[g_loop, g_single, g_loop], directed=[g_directed, g_loop],
                big_ids=[g_big_ids, g_big_ids], isomorphic=[g_str, g_dup, g_str],
                float_labels=[g_float_lab, g_float_lab], tuple_labels=[g_tuple_lab, g_float_lab[:1] + [{0: (1, 2), 1: (9, 9)}]])


@pytest.mark.parametrize("name", sorted(_edge_sets()))
def test_edge_case_inputs_match_oracle(gk, name):
    X = _edge_sets()[name]
    for h in (1, 2, 4):
        assert np.array_equal(gk.WeisfeilerLehman(n_iter=h).fit_transform(X),
                              O.WLOracle(n_iter=h).fit_transform(X)), (name, h)
    Kn = gk.WeisfeilerLehman(n_iter=2, normalize=True).fit_transform(X)
    assert np.allclose(Kn, O.WLOracle(n_iter=2, normalize=True).fit_transform(X), rtol=REL_TOL, atol=0)
    assert np.array_equal(gk.VertexHistogram().fit_transform(X), O.VHOracle().fit_transform(X))
    wl, wo = gk.WeisfeilerLehman(n_iter=3).fit(X[:1]), O.WLOracle(n_iter=3)
    wo.fit_transform(X[:1])
    assert np.array_equal(wl.transform(X), wo.transform(X))


def test_transform_with_only_unseen_labels_and_zero_diagonals(gk):
    tr = [[{0: [1], 1: [0]}, {0: 'a', 1: 'b'}], [{0: [1, 2], 1: [0], 2: [0]}, {0: 'a', 1: 'a', 2: 'b'}]]
    te = [[{0: [1], 1: [0]}, {0: 'zz', 1: 'yy'}], [{0: [1], 1: [0]}, {0: 'a', 1: 'yy'}]]
    for norm in (False, True):
        wl, wo = gk.WeisfeilerLehman(n_iter=2, normalize=norm), O.WLOracle(n_iter=2, normalize=norm)
        wl.fit(tr), wo.fit_transform(tr)
        K, Ko = wl.transform(te), wo.transform(te)
        assert np.allclose(K, Ko, rtol=REL_TOL, atol=0) and np.array_equal(K[0], [0, 0])
        vh, vo = gk.VertexHistogram(normalize=norm), O.VHOracle(normalize=norm)
        vh.fit(tr), vo.fit_transform(tr)
        assert np.allclose(vh.transform(te), vo.transform(te), rtol=REL_TOL, atol=0)
    # a graph without any labelled vertex: VertexHistogram leaves 0/0 = NaN and warns
    # (kernel.py:199-234); WeisfeilerLehman refuses it like the reference (graph.py:737-738)
    Xz = [["unused", {}], ["unused", {0: 'a'}]]
    with pytest.warns(RuntimeWarning):
        Kz = gk.VertexHistogram(normalize=True).fit_transform(Xz)
    with np.errstate(all='ignore'):
        Kzo = O.VHOracle(normalize=True).fit_transform(Xz)
    assert np.array_equal(np.isnan(Kz), np.isnan(Kzo))
    assert np.array_equal(np.isnan(Kz), [[True, True], [True, False]]) and Kz[1, 1] == 1.0
    with pytest.raises(ValueError):
        gk.WeisfeilerLehman(n_iter=1).fit_transform([[{0: [1], 1: [0]}, {}]])


def test_shortest_path_edge_cases_match_oracle(gk):
    A1 = np.array([[0, 2, 0, 0], [2, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 0]])      # weighted + isolated vertex
    A2 = np.array([[0, 1, 0], [0, 0, 1], [1, 0, 0]])                            # directed cycle
    A3 = np.zeros((2, 2), dtype=int)                                            # no edges at all
    X = [[A1, {0: 'a', 1: 'b', 2: 'a', 3: 'c'}], [A2, {0: 'a', 1: 'a', 2: 'b'}], [A3, {0: 'a', 1: 'b'}]]
    for kw in (dict(), dict(with_labels=False), dict(algorithm_type="floyd_warshall")):
        assert np.array_equal(gk.ShortestPath(**kw).fit_transform(X), O.SPOracle(**kw).fit_transform(X)), kw
    sp, so = gk.ShortestPath(), O.SPOracle()
    sp.fit(X[:2]), so.fit_transform(X[:2])
    assert np.array_equal(sp.transform(X), so.transform(X))
    # dictionary input with weights, symbols as vertex names (Dijkstra route of the reference)
    D = [{'p': {'q': 3}, 'q': {'p': 3, 'r': 1}, 'r': {'q': 1}}, {'p': 0, 'q': 1, 'r': 0}]
    assert np.array_equal(gk.ShortestPath().fit_transform([D, D]), O.SPOracle().fit_transform([D, D]))
    with np.errstate(all='ignore'):
        Kn, Kno = gk.ShortestPath(normalize=True).fit_transform(X), O.SPOracle(normalize=True).fit_transform(X)
    assert np.allclose(Kn, Kno, rtol=REL_TOL, atol=0, equal_nan=True)


def test_reference_identical_inv_labels(gk, mutag_graphs):
    """the host post-pass reproduces the reference's label dictionaries
    (credential strings and ids) exactly; tests/golden/doc_goldens.json holds the real
    reference's dictionary for H2O/H3O."""
    import json
    H2O = [{'a': ['b', 'c'], 'b': ['a'], 'c': ['a']}, {'a': 'O', 'b': 'H', 'c': 'H'}]
    H3O = [{'a': ['b', 'c', 'd'], 'b': ['a'], 'c': ['a'], 'd': ['a']},
           {'a': 'O', 'b': 'H', 'c': 'H', 'd': 'H'}]
    with open(os.path.join(os.path.dirname(__file__), "golden", "doc_goldens.json")) as f:
        doc = json.load(f)
    wl = gk.WeisfeilerLehman(n_iter=5).fit([H2O, H3O])
    assert {str(k): v for k, v in wl.inv_labels().items()} == doc["wl5_inv_labels"]
    G, z = mutag_graphs
    for X, h in ((G, 4), (random_labelled_graphs(40, 3, 14, 0.3, 3, 11, fmt="dict"), 3)):
        wo = O.WLOracle(n_iter=h)
        wo.fit_transform(X, keep_levels=True)
        wl = gk.WeisfeilerLehman(n_iter=h)
        wl.fit_transform(X)
        assert wl.inv_labels() == wo.inv_labels
        for lvl in range(h + 1):       # and the per-node ids are the reference's ids, not just a bijection
            want = np.array([l for d in wo.levels[lvl] for l in d.values()])
            assert np.array_equal(wl._reference_labels[lvl], want)


def test_graph_kernel_wrapper_and_extras(gk, mutag_graphs):
    """The dispatcher and inputs with edge labels / extra tuple elements (how CoreFramework and
    GraphKernel hand graphs to the kernels, weisfeiler_lehman.py:157-171) give the same matrices."""
    G, z = mutag_graphs
    K = gk.GraphKernel(kernel=[{"name": "WL", "n_iter": 5}, "VH"]).fit_transform(G)
    assert np.array_equal(K, z["K_wl5"])
    assert np.array_equal(gk.GraphKernel(kernel="SP").fit_transform(G[:60]), z["K_sp"][:60, :60])
    Gx = [[g[0], g[1], {e: 1 for e in g[0]}, "extra", 7] for g in G[:50]]      # edge labels + extras
    assert np.array_equal(gk.WeisfeilerLehman(n_iter=5).fit_transform(Gx), z["K_wl5"][:50, :50])
    gkk = gk.GraphKernel(kernel="WL", normalize=True).fit(G[:120])
    wl = gk.WeisfeilerLehman(normalize=True).fit(G[:120])
    assert np.array_equal(gkk.transform(G[120:]), wl.transform(G[120:]))


def test_tu_loader_batch_gives_the_reference_gram(gk, mutag_graphs, tmp_path):
    from grakel_amd.datasets import read_tu
    from test_host import _write_mutag_tu
    G, z = mutag_graphs
    batch, _ = read_tu(_write_mutag_tu(tmp_path, z), "MUTAG")
    assert np.array_equal(gk.WeisfeilerLehman(n_iter=5).fit_transform(batch), z["K_wl5"])
    assert np.array_equal(gk.VertexHistogram().fit_transform(batch), z["K_vh"])


def test_edge_histogram_against_reference_goldens(gk, mutag_graphs):
    """EdgeHistogram = the VertexHistogram path over edge labels."""
    G, z = mutag_graphs
    eh = gk.EdgeHistogram()
    assert np.array_equal(eh.fit_transform(G[:120]), z["K_eh"])
    assert np.array_equal(eh.transform(G[120:]), z["K_eh_tr"])
    assert np.allclose(gk.EdgeHistogram(normalize=True).fit_transform(G), z["K_eh_norm"], rtol=REL_TOL, atol=0)
    assert np.array_equal(gk.GraphKernel(kernel="EH").fit_transform(G[:120]), z["K_eh"])
    with pytest.raises(TypeError):
        gk.EdgeHistogram().fit_transform([[g[0], g[1]] for g in G[:3]])      # needs edge labels


@pytest.mark.parametrize("world", [2, 3, 8])
def test_batch_from_shards_rebuilds_the_global_batch(gk, world):
    """What every rank of a `world`-rank job does after the all-gather, on one GPU: the messages
    are packed per shard exactly as ShardExchange packs them (ragged shards, incl. graphs without
    edges), laid out back to back as ncclAllGather leaves them, and gk_batch_from_shards must give
    the same Gram rows as the plain path on the whole batch."""
    import torch
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.dist import shard_bounds
    from grakel_amd.engine import get_engine
    X = er_dataset(203, 25, 0.08, 4, 17) + [[{0: [], 1: []}, {0: 1, 1: 2}]] * 2
    gb, _ = wl_batch_from_input(X)
    K = gk.WeisfeilerLehman(n_iter=3).fit_transform(gb)
    b = shard_bounds(gb.n_graphs, world)
    shards = [gb.slice_graphs(b[r], b[r + 1]) for r in range(world)]
    sizes = np.array([[s.n_graphs, s.n_nodes, s.n_edges] for s in shards], np.int64)
    mg, mv, me = (int(x) for x in sizes.max(axis=0))
    msgs = []
    for s in shards:
        m = np.zeros(mg + 2 * mv + me, np.int32)
        m[:s.n_graphs] = np.diff(s.graph_ptr)
        m[mg:mg + s.n_nodes] = np.diff(s.row_ptr)
        m[mg + mv:mg + mv + s.n_nodes] = s.node_label
        m[mg + 2 * mv:mg + 2 * mv + s.n_edges] = s.col_idx          # local node ids
        msgs.append(m)
    flat = torch.from_numpy(np.concatenate(msgs)).cuda()
    torch.cuda.synchronize()
    eng = get_engine()
    db = eng.batch_from_shards(sizes, mg, mv, me, flat.data_ptr(), gb.n_labels)
    assert (db.n_graphs, db.n_nodes, db.n_edges) == (gb.n_graphs, gb.n_nodes, gb.n_edges)
    eng.wl_relabel(db, 3)
    feat = eng.features(db, 4)
    for r in range(world):
        assert np.array_equal(eng.gram(feat, 0, rows=(b[r], b[r + 1])), K[b[r]:b[r + 1]])


# ------------------------------------------------------------------------------------------
# WL optimal assignment: histogram intersection over the WL hierarchy
# ------------------------------------------------------------------------------------------
def test_wloa_against_reference_goldens(gk, mutag_graphs):
    G, z = mutag_graphs
    oa = gk.WeisfeilerLehmanOptimalAssignment(n_iter=4)
    assert np.array_equal(oa.fit_transform(G[:120]), z["K_oa4"])
    assert np.array_equal(oa.diagonal(), np.diagonal(z["K_oa4"]))
    assert np.array_equal(oa.transform(G[120:]), z["K_oa4_tr"])
    oan = gk.WeisfeilerLehmanOptimalAssignment(n_iter=2, normalize=True)
    assert np.allclose(oan.fit_transform(G[:120]), z["K_oa2_norm"], rtol=REL_TOL, atol=0)
    assert np.allclose(oan.transform(G[120:]), z["K_oa2_norm_tr"], rtol=REL_TOL, atol=0)
    assert np.array_equal(gk.GraphKernel(kernel={"name": "WL-OA", "n_iter": 4}).fit_transform(G[:120]), z["K_oa4"])


@pytest.mark.parametrize("name", ["dict_u", "adj_u", "adj_d", "tuples_d", "dense_big"])
def test_wloa_small_sets_against_reference(gk, name):
    """Every input format incl. the dict set whose isolated ``{v: []}`` vertices WL-OA drops."""
    z = load_golden("small_sets.npz")
    tr, te = split(random_labelled_graphs(**dict(SMALL_SETS)[name]))
    oa = gk.WeisfeilerLehmanOptimalAssignment(n_iter=3)
    assert np.array_equal(oa.fit_transform(tr), z[name + "/oa3_fit"])
    assert np.array_equal(oa.transform(te), z[name + "/oa3_tr"])


@pytest.mark.parametrize("low_df", [2, 32, 1000000])
def test_wloa_er_set_against_oracle_all_column_classes(gk, low_df, gkopt):
    """600 ER graphs: dense (unary-expanded), rare (pair updates with min) and dead columns all
    occur; the option feat.low_df moves the dense/rare boundary to both extremes."""
    gkopt("feat.low_df", low_df)
    G = er_dataset(600, 30, 0.12, 3, 5)
    want = O.WLOAOracle(n_iter=3)
    Kw = want.fit_transform(G[:400])
    oa = gk.WeisfeilerLehmanOptimalAssignment(n_iter=3)
    assert np.array_equal(oa.fit_transform(G[:400]), Kw)
    assert np.array_equal(oa.transform(G[400:]), want.transform(G[400:]))
    X_diag, Y_diag = oa.diagonal()
    assert np.array_equal(X_diag, want.x_diag) and np.array_equal(Y_diag, want.y_diag)


def test_wloa_counts_above_int8_and_properties(gk):
    """Graphs with > 127 equally labelled vertices (unary width > 127), and the kernel's own
    invariants at a size the oracle cannot reach: K_ii = (n_iter+1) * |V_i|, K symmetric,
    K_ij <= min(K_ii, K_jj), and the h-level kernel is monotone in h."""
    G = random_labelled_graphs(30, 150, 300, 0.03, 2, 77, fmt="adj")
    want = O.WLOAOracle(n_iter=2).fit_transform(G)
    assert np.array_equal(gk.WeisfeilerLehmanOptimalAssignment(n_iter=2).fit_transform(G), want)
    batch = gk.GraphBatch(*er_dataset_csr(3000, 60, 0.08, 4, 9), 4)
    K2 = gk.WeisfeilerLehmanOptimalAssignment(n_iter=2).fit_transform(batch)
    K3 = gk.WeisfeilerLehmanOptimalAssignment(n_iter=3).fit_transform(batch)
    assert np.array_equal(K2, K2.T) and np.array_equal(np.diagonal(K2), np.full(3000, 3 * 60.0))
    assert np.all(K2 <= np.minimum.outer(np.diagonal(K2), np.diagonal(K2)))
    assert np.all(K3 >= K2) and np.array_equal(np.diagonal(K3), np.full(3000, 4 * 60.0))


def test_wloa_er_configs_against_reference_checksums(gk):
    """ER n200 and BASELINE config 2 (1000 graphs, n=50, h=3; the real reference takes 185 s).
    Dict-of-lists input on purpose: its isolated ``{v: []}`` vertices have no edge-dictionary
    entry and drop out of WL-OA (a packed GraphBatch would keep them)."""
    for tag in ("n200", "config2"):
        z = load_golden("er_%s.npz" % tag)
        N, n, L, seed, h = z["params"].tolist()
        K = gk.WeisfeilerLehmanOptimalAssignment(n_iter=h).fit_transform(
            er_dataset(N, n, float(z["p"][0]), L, seed))
        assert int(K.sum()) == int(z["oa_sum"][0]) and np.array_equal(K[:64, :64], z["oa_block"])
        assert np.array_equal(K.sum(axis=1), z["oa_row_sums"])
        assert np.array_equal(K[z["oa_samp_i"], z["oa_samp_j"]], z["oa_samp_v"])
        assert np.array_equal(K, K.T)


def test_wloa_error_behaviour(gk):
    with pytest.raises(TypeError):
        gk.WeisfeilerLehmanOptimalAssignment(n_iter=0).fit([[{0: [1], 1: [0]}, {0: 'a', 1: 'b'}]])
    with pytest.raises(KeyError):       # an edge-dictionary entry without a label (:177)
        gk.WeisfeilerLehmanOptimalAssignment().fit_transform([[{0: [1], 1: [0, 2]}, {0: 'a', 1: 'b'}]])
    oa = gk.WeisfeilerLehmanOptimalAssignment(n_iter=2).fit([[{0: [1], 1: [0]}, {0: 'a', 1: 'b'}]])
    with pytest.raises(ValueError):
        oa.transform([[{0: [1], 1: [0]}]])                 # transform wants 2 or 3 elements (:327-337)
    with pytest.raises(ValueError):
        oa.transform(None)


# ------------------------------------------------------------------------------------------
# WL framework over the ShortestPath base kernel
# ------------------------------------------------------------------------------------------
def test_wl_with_shortest_path_base_against_reference_goldens(gk, mutag_graphs):
    G, z = mutag_graphs
    wsp = gk.WeisfeilerLehman(n_iter=2, base_graph_kernel=gk.ShortestPath)
    assert np.array_equal(wsp.fit_transform(G[:100]), z["K_wlsp2"])
    assert np.array_equal(wsp.diagonal(), np.diagonal(z["K_wlsp2"]))
    assert np.array_equal(wsp.transform(G[100:140]), z["K_wlsp2_tr"])
    wspn = gk.WeisfeilerLehman(n_iter=1, normalize=True,
                               base_graph_kernel=(gk.ShortestPath, {"with_labels": True}))
    assert np.allclose(wspn.fit_transform(G[:100]), z["K_wlsp1_norm"], rtol=REL_TOL, atol=0)
    assert np.allclose(wspn.transform(G[100:140]), z["K_wlsp1_norm_tr"], rtol=REL_TOL, atol=0)
    K = gk.GraphKernel(kernel=[{"name": "WL", "n_iter": 2}, {"name": "SP"}]).fit_transform(G[:100])
    assert np.array_equal(K, z["K_wlsp2"])
    with pytest.raises(ValueError):
        gk.WeisfeilerLehman(base_graph_kernel=(gk.ShortestPath, {"algorithm_type": "bfs"})).fit(G[:3])
    # round 5: any OTHER kernel class is a host base kernel over the device relabel (here: WL-OA under WL, on MUTAG's
    # tuple-set graphs with global vertex ids): level l's matrix is that kernel on the oracle's level-l labels
    ref = O.WLOracle(n_iter=1)
    ref.fit_transform(G[:12], keep_levels=True)
    want = sum(gk.WeisfeilerLehmanOptimalAssignment(n_iter=2).fit_transform([(g[0], ref.levels[l][j]) for j, g in enumerate(G[:12])])
               for l in range(2))
    woa = gk.WeisfeilerLehman(n_iter=1, base_graph_kernel=(gk.WeisfeilerLehmanOptimalAssignment, {"n_iter": 2}))
    assert np.array_equal(woa.fit_transform(G[:12]), want)


@pytest.mark.parametrize("name", ["dict_u", "adj_u", "adj_d", "tuples_d", "dense_big"])
def test_wl_with_shortest_path_base_small_sets(gk, name):
    z = load_golden("small_sets.npz")
    if name + "/wlsp2_fit" not in z.files:
        pytest.skip("the reference's ShortestPath raises on this set (Dijkstra sink-vertex KeyError)")
    kw = dict(SMALL_SETS)[name]
    tr, te = split(random_labelled_graphs(**kw))
    trs, tes = sp_inputs(kw, tr), sp_inputs(kw, te)
    wsp = gk.WeisfeilerLehman(n_iter=2, base_graph_kernel=gk.ShortestPath)
    assert np.array_equal(wsp.fit_transform(trs), z[name + "/wlsp2_fit"])
    assert np.array_equal(wsp.transform(tes), z[name + "/wlsp2_tr"])


def test_wl_with_shortest_path_base_weighted_and_unlabelled_against_oracle(gk):
    """NCI1-like graphs (distances up to ~25, 37 labels) incl. the with_labels=False base: every
    level then carries the same distance histogram, K = (n_iter + 1) * K_SP."""
    G = nci1_like(80, 5, as_adj=True)
    want = O.WLSPOracle(n_iter=3)
    Kw = want.fit_transform(G[:50])
    wsp = gk.WeisfeilerLehman(n_iter=3, base_graph_kernel=gk.ShortestPath)
    assert np.array_equal(wsp.fit_transform(G[:50]), Kw)
    assert np.array_equal(wsp.transform(G[50:]), want.transform(G[50:]))
    X_diag, Y_diag = wsp.diagonal()
    assert np.array_equal(X_diag, want.x_diag) and np.array_equal(Y_diag, want.y_diag)
    Ku = gk.WeisfeilerLehman(n_iter=2, base_graph_kernel=(gk.ShortestPath, {"with_labels": False})).fit_transform(G[:50])
    assert np.array_equal(Ku, 3 * gk.ShortestPath(with_labels=False).fit_transform(G[:50]))


# ------------------------------------------------------------------------------------------
# Core framework
# ------------------------------------------------------------------------------------------
def test_core_numbers_match_the_oracle(gk):
    from grakel_amd.batch import sp_batch_from_input
    from grakel_amd.engine import get_engine
    G = (random_labelled_graphs(40, 2, 60, 0.15, 3, 21, fmt="adj") +
         random_labelled_graphs(6, 300, 700, 0.02, 2, 22, fmt="adj") +
         [[np.zeros((3, 3)), {0: 0, 1: 0, 2: 1}], [np.ones((9, 9)) - np.eye(9), {i: 0 for i in range(9)}]])
    gb, _ = sp_batch_from_input(G, True)
    eng = get_engine()
    got = eng.core_numbers(eng.upload(gb))
    want = np.concatenate([O.core_numbers(np.asarray(g[0])) for g in G])
    assert np.array_equal(got, want)
    assert got[-9:].tolist() == [8] * 9 and got[-12:-9].tolist() == [0, 0, 0]


def test_core_framework_against_reference_goldens(gk, mutag_graphs):
    G, z = mutag_graphs
    for tag, base in (("sp", None), ("vh", gk.VertexHistogram), ("wl2", (gk.WeisfeilerLehman, {"n_iter": 2}))):
        cf = gk.CoreFramework(base_graph_kernel=base)
        assert np.array_equal(cf.fit_transform(G[:100]), z["K_core_%s" % tag]), tag
        assert np.array_equal(cf.diagonal(), np.diagonal(z["K_core_%s" % tag])), tag
        assert np.array_equal(cf.transform(G[100:140]), z["K_core_%s_tr" % tag]), tag
    cfn = gk.CoreFramework(normalize=True)
    assert np.allclose(cfn.fit_transform(G[:100]), z["K_core_sp_norm"], rtol=REL_TOL, atol=0)
    assert np.allclose(cfn.transform(G[100:140]), z["K_core_sp_norm_tr"], rtol=REL_TOL, atol=0)
    K = gk.GraphKernel(kernel=[{"name": "CORE"}, {"name": "VH"}]).fit_transform(G[:100])
    assert np.array_equal(K, z["K_core_vh"])
    assert np.array_equal(gk.GraphKernel(kernel="core_framework").fit_transform(G[:100]), z["K_core_sp"])
    fitted = gk.CoreFramework(base_graph_kernel=gk.VertexHistogram).fit(G[:100])     # fit, then transform
    assert np.array_equal(fitted.transform(G[100:140]), z["K_core_vh_tr"])
    with pytest.raises(NotImplementedError):
        gk.CoreFramework(base_graph_kernel=gk.EdgeHistogram).fit(G[:3])


@pytest.mark.parametrize("name", ["dict_u", "adj_u", "dense_big"])
def test_core_framework_small_sets(gk, name):
    """dense_big reaches core numbers around 20, so ~20 base-kernel levels; the targets of adj_u /
    dict_u exercise levels the fitted graphs never reach (diagonal-only "dummy" levels)."""
    z = load_golden("small_sets.npz")
    kw = dict(SMALL_SETS)[name]
    tr, te = split(random_labelled_graphs(**kw))
    trs, tes = sp_inputs(kw, tr), sp_inputs(kw, te)
    for tag, base, orc in (("sp", None, O.SPOracle), ("vh", gk.VertexHistogram, O.VHOracle)):
        cf = gk.CoreFramework(base_graph_kernel=base)
        assert np.array_equal(cf.fit_transform(trs), z[name + "/core_%s_fit" % tag]), tag
        assert np.array_equal(cf.transform(tes), z[name + "/core_%s_tr" % tag]), tag
        want = O.CoreOracle(orc)
        want.fit_transform(trs), want.transform(tes)
        X_diag, Y_diag = cf.diagonal()
        assert np.array_equal(X_diag, want.x_diag) and np.array_equal(Y_diag, want.y_diag), tag
    # swapped roles: sparse fit, dense targets -> levels above the fitted maximum
    cf, want = gk.CoreFramework(base_graph_kernel=gk.VertexHistogram), O.CoreOracle(O.VHOracle)
    assert np.array_equal(cf.fit_transform(tes[:3]), want.fit_transform(tes[:3]))
    assert np.array_equal(cf.transform(trs), want.transform(trs))
    assert np.array_equal(cf.diagonal()[1], want.y_diag)


@pytest.mark.parametrize("name", ["adj_u", "dense_big"])
def test_caller_protocols_of_hadamard_code_and_core_framework(gk, name):
    """SURVEY 8b "Callers": the accelerated BASE classes driven exactly as the reference's own frameworks drive them --
    HadamardCode (hadamard_code.py:189-260: per level a list of `(graph, {vertex: tuple-valued label})` elements to
    `fit_transform` / `transform`, `diagonal()` of every level for the normalisation) and CoreFramework
    (core_framework.py:173-219: per core level a list of graph OBJECTS to `fit` / `fit_transform` / `transform`, `diagonal()`
    after both).  The calling sequences are restated in tests/caller_protocols.py (the GPU box has no reference; on the CPU
    box tests/test_host.py checks that the REAL frameworks hand the accelerated classes the same batches); the matrices are
    compared with what the real reference's frameworks computed (tests/golden/callers.npz, small_sets.npz)."""
    from caller_protocols import CoreCaller, HadamardCaller
    z, zs = load_golden("callers.npz"), load_golden("small_sets.npz")
    kw = dict(SMALL_SETS)[name]
    tr, te = split(random_labelled_graphs(**kw))
    hc = HadamardCaller(lambda: gk.VertexHistogram(normalize=False, verbose=False, n_jobs=None), 3)
    assert np.array_equal(hc.fit_transform(tr), z[name + "/hc_vh_fit"])
    assert np.array_equal(hc.transform(te), z[name + "/hc_vh_tr"])
    hcn = HadamardCaller(lambda: gk.VertexHistogram(normalize=False, verbose=False, n_jobs=None), 2, normalize=True)
    assert np.allclose(hcn.fit_transform(tr), z[name + "/hc_vh_norm_fit"], rtol=REL_TOL, atol=0)
    assert np.allclose(hcn.transform(te), z[name + "/hc_vh_norm_tr"], rtol=REL_TOL, atol=0)
    hs = HadamardCaller(lambda: gk.ShortestPath(normalize=False, verbose=False, n_jobs=None), 2)
    assert np.array_equal(hs.fit_transform(tr), z[name + "/hc_sp_fit"])
    assert np.array_equal(hs.transform(te), z[name + "/hc_sp_tr"])
    for tag, make in (("sp", lambda: gk.ShortestPath(normalize=False, verbose=False, n_jobs=None)),
                      ("vh", lambda: gk.VertexHistogram(normalize=False, verbose=False, n_jobs=None))):
        cc = CoreCaller(make)
        assert np.array_equal(cc.fit_transform(tr), zs[name + "/core_%s_fit" % tag]), tag
        assert np.array_equal(cc.transform(te), zs[name + "/core_%s_tr" % tag]), tag
        want = O.CoreOracle(O.SPOracle if tag == "sp" else O.VHOracle)
        want.fit_transform(tr), want.transform(te)
        assert np.array_equal(cc.x_diag, want.x_diag) and np.array_equal(cc.y_diag, want.y_diag), tag


# ------------------------------------------------------------------------------------------
# config 5 (BASELINE.json: 50 000 synthetic n=30 graphs, WL h=5).  The reference cannot run it (six dense
# 50k x 50k float64 matrices, weisfeiler_lehman.py:269-270), so parity is block-wise: a WL kernel value
# only depends on the two graphs, hence K[block, block] of the big job must equal the oracle run on the
# block alone, and K[i, j] the oracle run on the pair.
# ------------------------------------------------------------------------------------------
CONFIG5 = dict(N=50000, n=30, p=0.1, L=5, seed=0, h=5)
CONFIG5_LABEL_COUNTS = [5, 6987, 1106456, 1386518, 1408511, 1408933]     # oracle.WLOracle.label_counts_only


def _config5_parts():
    from grakel_amd import GraphBatch
    c = CONFIG5
    gp, rp, ci, lab = er_dataset_csr(c["N"], c["n"], c["p"], c["L"], c["seed"])
    return GraphBatch(gp, rp, ci, lab, c["L"])


def _graphs_from_batch(gb, idx):
    """grakel input form [edge dict, labels] of the listed graphs of a packed batch."""
    out = []
    for g in idx:
        v0, v1 = int(gb.graph_ptr[g]), int(gb.graph_ptr[g + 1])
        ed = {v - v0: (gb.col_idx[gb.row_ptr[v]:gb.row_ptr[v + 1]] - v0).tolist() for v in range(v0, v1)}
        out.append([ed, {v - v0: int(gb.node_label[v]) for v in range(v0, v1)}])
    return out


def test_config5_full_size_blockwise_against_the_oracle(gk):
    from grakel_amd.engine import get_engine
    c = CONFIG5
    N, h = c["N"], c["h"]
    gb = _config5_parts()
    eng = get_engine()
    db = eng.upload(gb)
    assert eng.wl_relabel(db, h) == CONFIG5_LABEL_COUNTS
    feat = eng.features(db, h + 1)
    eng.gram(feat, 0, to_host=False)                      # the 20 GB float64 matrix stays in HBM
    total, trace, asym = eng.gram_checksum(feat)
    selfk = eng.selfk(feat)
    assert asym == 0.0                                    # K == K.T, checked in place
    assert trace == selfk.sum()                           # diag(K) sums to the exact self similarities
    rs = np.random.RandomState(11)
    # (a) eight disjoint 500-graph diagonal blocks, every entry against the oracle on the block alone
    row_sum_all = 0.0
    for b0 in (0, 3500, 9000, 17500, 24000, 31000, 42500, 49500):
        rows = eng.gram(feat, 0, rows=(b0, b0 + 500))     # [500 x N] row block, recomputed by gk_gram_rows
        Kref = O.WLOracle(n_iter=h).fit_transform(_graphs_from_batch(gb, range(b0, b0 + 500)))
        assert np.array_equal(rows[:, b0:b0 + 500], Kref), "diagonal block at %d" % b0
        assert np.array_equal(np.diagonal(rows[:, b0:b0 + 500]), selfk[b0:b0 + 500])
        # (b) off-diagonal entries of these rows against the pairwise oracle
        for _ in range(260):
            i, j = b0 + int(rs.randint(0, 500)), int(rs.randint(0, N))
            if b0 <= j < b0 + 500:
                continue
            want = O.WLOracle(n_iter=h).fit_transform(_graphs_from_batch(gb, [i, j]))[0, 1]
            assert rows[i - b0, j] == want, (i, j)
        row_sum_all += rows.sum()
    # (c) the row blocks recomputed by gk_gram_rows and the in-place matrix agree on the symmetric counterpart
    eng.gram(feat, 0, to_host=False)
    assert eng.gram_checksum(feat)[0] == total
    cols = eng.gram(feat, 0, rows=(12345, 12346))[0]
    blk = eng.gram(feat, 0, rows=(3500, 4000))
    assert np.array_equal(blk[:, 12345], cols[3500:4000])


def _config5_rank_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grakel_amd.dist import ShardedWL, shard_bounds
    from grakel_amd.engine import get_engine
    torch.cuda.set_device(0)
    full = _config5_parts()
    b = shard_bounds(full.n_graphs, world)
    eng = get_engine(0)
    sw = ShardedWL(eng, n_iter=CONFIG5["h"])               # default plan: plain row blocks
    _, info = sw.step(full.slice_graphs(b[rank], b[rank + 1]), keep=True)
    feat = info["feat"]                                    # holds this rank's [N/2 x N] row block, 10 GB in HBM
    s = eng.gram_checksum(feat)[0]
    lo = info["rows"][0]
    row7 = eng.gram(feat, 0, rows=(lo + 7, lo + 8))[0]
    np.save(os.path.join(outdir, "c5_%d.npy" % rank), np.concatenate([[s, info["gram"][0], info["n_cols"]], row7]))
    feat.close()
    info["batch"].close()
    del info
    sw.close()
    dist.barrier()
    dist.destroy_process_group()


def test_config5_row_sharded_over_two_processes(gk, tmp_path):
    """Config 5 with the Gram rows sharded over two ranks (both on cuda:0, gloo): each rank holds a 10 GB row
    block; the blocks' sums add up to the single-process matrix and a row of each block equals the row the
    single process computes."""
    import torch.multiprocessing as mp
    from grakel_amd.engine import get_engine
    c = CONFIG5
    eng = get_engine()
    db = eng.upload(_config5_parts())
    eng.wl_relabel(db, c["h"])
    feat = eng.features(db, c["h"] + 1)
    eng.gram(feat, 0, to_host=False)
    total = eng.gram_checksum(feat)[0]
    want = {r: eng.gram(feat, 0, rows=(lo + 7, lo + 8))[0] for r, lo in ((0, 0), (1, c["N"] // 2))}
    feat.close()
    db.close()
    port = 29300 + (os.getpid() % 500)
    mp.spawn(_config5_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [np.load(os.path.join(str(tmp_path), "c5_%d.npy" % r)) for r in range(2)]
    assert got[0][0] + got[1][0] == total
    for r in range(2):
        assert np.array_equal(got[r][3:], want[r])
    # plain row blocks: each rank multiplied its whole block (rows x N x dense columns)
    assert got[0][1] + got[1][1] == 2.0 * c["N"] * c["N"] * got[0][2] and abs(got[0][1] - got[1][1]) < 0.01 * got[0][1]


# ------------------------------------------------------------------------------------------
# the dense kernel against the host product of its own operand, every operand form and both kernel forms
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("env", [(), ("gram.no_fp4",), ("gram.no_ws",), ("gram.no_ws", "gram.no_fp4"), ("gram.no_sym",),
                                 ("gram.no_patch",), ("gram.dd",), ("gram.dd", "gram.no_fp4"), ("gram.dd=2",)],
                         ids=lambda e: "+".join(e) or "default")
@pytest.mark.parametrize("N,n", [(40, 20), (300, 20), (1001, 12)])
def test_dense_gram_equals_the_product_of_its_own_operand(gk, gkopt, env, N, n):
    from grakel_amd import GraphBatch
    from grakel_amd.engine import get_engine
    gkopt("feat.low_df", 2)                                # every useful column is dense
    for k in env:
        gkopt(k.split("=")[0], int(k.split("=")[1]) if "=" in k else 1)
    eng = get_engine()
    db = eng.upload(GraphBatch(*er_dataset_csr(N, n, 0.15, 3, 0), 3))
    eng.wl_relabel(db, 2)
    feat = eng.features(db, 3)
    assert feat.n_cols_low == 0 and feat.max_count > 4     # counts 5..127 exist: the secondary int8 region is in use
    phi = eng.debug_phi(feat)
    K = eng.gram(feat, 0)
    R = phi @ phi.T
    np.fill_diagonal(R, eng.selfk(feat))
    assert np.array_equal(K, R)
    s, t, a = eng.gram_checksum(feat)
    assert (s, t, a) == (K.sum(), np.trace(K), 0.0)
    d = eng.selfk(feat)                                      # normalised epilogue and a row block with an odd offset
    Kn = eng.gram(feat, 2, rows=(3, N - 2))
    with np.errstate(divide="ignore", invalid="ignore"):
        want = np.nan_to_num(R[3:N - 2] / np.sqrt(np.outer(d[3:N - 2], d)))
    assert np.allclose(Kn, want, rtol=1e-13, atol=0)


@pytest.mark.parametrize("big_n,want", [(4095, "fp4+i8+f64"), (4096, "i8+f64"), (46340, "i8+f64"), (46341, "f64")])
def test_operand_type_switches_at_the_exactness_bounds(gk, big_n, want):
    """The dense operand type is chosen from the bound levels * max_n^2 on a Gram entry: below 2^24 counts <= 4 travel
    as MX fp4 codes (float32 accumulation exact), below 2^31 everything is int8 (int32 accumulation), above it the
    float64 MFMA path.  One VertexHistogram job on each side of both bounds (4095^2 < 2^24 <= 4096^2,
    46340^2 < 2^31 < 46341^2), with columns of every count class (<= 4, 5..127, > 127), against the oracle."""
    rs = np.random.RandomState(big_n)

    def labels(n, big_graph):
        # label 0 heavy (> 127 per graph), 1..20 medium (5..127), 21..40 light (<= 4, in every graph: dense fp4 columns
        # when fp4 is allowed), >= 41 filler of the big graph (<= 4 each, in no other graph: dead columns)
        lab = np.zeros(n, dtype=np.int64)
        pos = int(n * (0.4 if big_graph else 0.5))
        for l in range(1, 21):
            c = 60 if big_graph else int(rs.randint(5, 9))
            lab[pos:pos + c] = l
            pos += c
        for l in range(21, 41):
            c = 3 if big_graph else int(rs.randint(1, 3))
            lab[pos:pos + c] = l
            pos += c
        rest = n - pos
        assert rest >= 0
        if big_graph:
            lab[pos:] = 41 + np.arange(rest) // 4
        else:
            lab[pos:] = 0
        return dict(enumerate(rs.permutation(lab).tolist()))

    big = [{i: [] for i in range(big_n)}, labels(big_n, True)]
    small = []
    for _ in range(40):
        n = int(rs.randint(400, 700))
        small.append([{i: [] for i in range(n)}, labels(n, False)])
    X = [big] + small
    vh = gk.VertexHistogram()
    K = vh.fit_transform(X)
    assert vh._last_info["dtype"] == want, vh._last_info
    assert vh._last_info["max_count"] > 127
    assert np.array_equal(K, O.VHOracle().fit_transform(X))
    assert np.array_equal(vh.transform(X[:3]), K[:3])


def test_wl_over_edge_histogram_and_deep_hierarchies_against_reference_goldens(gk, mutag_graphs):
    """tests/golden/round3.npz (the real reference on MUTAG): WeisfeilerLehman over the EdgeHistogram base kernel -- the
    edge labels reach every level's base kernel untouched, so the matrix is (n_iter + 1) x EdgeHistogram's -- and
    hierarchies of more than 48 levels, which grakel_amd builds in chunks of 48 (gk_features_build_range) and sums."""
    G, _ = mutag_graphs
    z = load_golden("round3.npz")
    wl = gk.WeisfeilerLehman(n_iter=3, base_graph_kernel=gk.EdgeHistogram)
    assert np.array_equal(wl.fit_transform(G[:100]), z["wleh_fit"])
    assert np.array_equal(wl.transform(G[100:130]), z["wleh_tr"])
    assert np.array_equal(wl.diagonal()[0], np.diagonal(z["wleh_fit"]))
    wln = gk.WeisfeilerLehman(n_iter=2, base_graph_kernel=(gk.EdgeHistogram, {}), normalize=True)
    assert np.allclose(wln.fit_transform(G[:100]), z["wleh_fit_norm"], rtol=REL_TOL, atol=0)
    assert np.allclose(wln.transform(G[100:130]), z["wleh_tr_norm"], rtol=REL_TOL, atol=0)
    deep = gk.WeisfeilerLehman(n_iter=55)
    assert np.array_equal(deep.fit_transform(G[:40]), z["deep_fit"])
    assert np.array_equal(deep.transform(G[40:52]), z["deep_tr"])
    assert np.array_equal(deep.diagonal()[0], np.diagonal(z["deep_fit"]))
    deepn = gk.WeisfeilerLehman(n_iter=50, normalize=True)
    assert np.allclose(deepn.fit_transform(G[:40]), z["deep_fit_norm"], rtol=REL_TOL, atol=0)
    assert np.allclose(deepn.transform(G[40:52]), z["deep_tr_norm"], rtol=REL_TOL, atol=0)
    X = er_dataset(60, 20, 0.15, 3, 8)                       # the oracle on another set; WL-OA takes the same chunks
    assert np.array_equal(gk.WeisfeilerLehman(n_iter=60).fit_transform(X), O.WLOracle(n_iter=60).fit_transform(X))
    assert np.array_equal(gk.WeisfeilerLehmanOptimalAssignment(n_iter=49).fit_transform(X), O.WLOAOracle(n_iter=49).fit_transform(X))


def test_export_and_import_of_the_fitted_state_through_the_c_abi(gk):
    """gk_export_state / gk_import_state: the fitted batch as a blob; a context that imports it computes the same matrix,
    a damaged blob is an argument error (not a fault)."""
    from grakel_amd._lib import GkError
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    X = er_dataset(120, 25, 0.1, 3, 4)
    K = O.WLOracle(n_iter=3).fit_transform(X)
    eng = get_engine()
    gb, _ = wl_batch_from_input(X)
    db = eng.upload(gb)
    blob = eng.export_state(db)
    db.close()
    db2 = eng.import_state(blob)
    assert (db2.n_graphs, db2.n_nodes, db2.n_edges) == (gb.n_graphs, gb.n_nodes, gb.n_edges)
    eng.wl_relabel(db2, 3)
    assert np.array_equal(eng.gram(eng.features(db2, 4)), K)
    bad = blob.copy()
    bad[0] ^= 0xff
    with pytest.raises(GkError, match="not a gk_hip state blob"):
        eng.import_state(bad)
    with pytest.raises(GkError, match="truncated"):
        eng.import_state(blob[:blob.size // 2])
    worse = blob.copy()
    worse[40 + 4 * 5] = 0xff                                  # graph_ptr[5] far out of range
    worse[40 + 4 * 5 + 3] = 0x7f
    with pytest.raises(GkError, match="malformed batch"):
        eng.import_state(worse)


def test_malformed_batches_are_rejected_at_the_c_abi(gk):
    """gk_batch_create validates the CSR on the device: a malformed batch is GK_ERR_ARG, not an out-of-bounds gather."""
    from grakel_amd import GraphBatch
    from grakel_amd._lib import GkError
    from grakel_amd.engine import get_engine
    eng = get_engine()
    gp, rp, ci, lab = er_dataset_csr(50, 12, 0.3, 3, 1)

    def bad(**kw):
        a = dict(gp=gp.copy(), rp=rp.copy(), ci=ci.copy(), lab=lab.copy())
        for k, f in kw.items():
            f(a[k])
        gb = GraphBatch.__new__(GraphBatch)                # bypass the host-side checks: this is the C ABI's job
        gb.graph_ptr, gb.row_ptr, gb.col_idx, gb.node_label = a["gp"], a["rp"], a["ci"], a["lab"]
        gb.n_labels, gb.edge_weight = 3, None
        with pytest.raises(GkError, match="malformed batch"):
            eng.upload(gb)

    def swap_rows(r): r[5], r[6] = r[6] + 1, r[5]
    def far_col(c): c[3] = len(rp) - 2                    # a neighbour in another graph
    def neg_col(c): c[0] = -1
    def big_label(l): l[7] = 3
    def swap_graphs(g): g[3] = g[5]
    bad(rp=swap_rows), bad(ci=far_col), bad(ci=neg_col), bad(lab=big_label), bad(gp=swap_graphs)
    assert eng.upload(GraphBatch(gp, rp, ci, lab, 3)).n_graphs == 50


# ------------------------------------------------------------------------------------------
# fitted state a consumer may read, against the real reference on MUTAG
# (tests/golden/make_golden.py: mutag_state)
# ------------------------------------------------------------------------------------------
def test_fitted_state_equals_the_reference(gk, mutag_graphs):
    import json
    G, _ = mutag_graphs
    z = load_golden("mutag_state.npz")
    G = G[:int(z["n_graphs"])]
    # VertexHistogram: X (graphs x labels, first-seen columns), _labels
    vh = gk.VertexHistogram().fit(G)
    assert sorted([int(k), int(v)] for k, v in vh._labels.items()) == json.loads(str(z["vh_labels"]))
    assert vh.X.shape == z["vh_X"].shape and np.array_equal(vh.X.toarray(), z["vh_X"])
    assert np.array_equal(np.asarray(vh.X), z["vh_X"]) and vh.sparse_ is True
    # WeisfeilerLehman: _inv_labels (filled by reading it), X[i].X, X[i]._labels
    wl = gk.WeisfeilerLehman(n_iter=3).fit(G)
    want = {int(i): {k: int(v) for k, v in d.items()} for i, d in json.loads(str(z["wl_inv_labels"])).items()}
    assert {str(k): v for k, v in wl._inv_labels[0].items()} == {str(k): v for k, v in want[0].items()}
    for i in (1, 2, 3):
        assert wl._inv_labels[i] == want[i]
    assert sorted(wl._inv_labels.keys()) == [0, 1, 2, 3] and wl._nx == len(G)
    for i in range(4):
        assert sorted([int(k), int(v)] for k, v in wl.X[i]._labels.items()) == json.loads(str(z["wl_labels%d" % i]))
        assert np.array_equal(wl.X[i].X.toarray(), z["wl_X%d" % i])
    wl2 = pickle.loads(pickle.dumps(wl))                    # the state survives a pickle round trip
    assert wl2._inv_labels[2] == want[2] and np.array_equal(wl2.X[3].X.toarray(), z["wl_X3"])
    # ShortestPath: _enum ((l_u, l_v, d) -> column, first seen first), X (count dicts), _phi_X, _phi_Y
    sp = gk.ShortestPath()
    K = sp.fit_transform(G)
    assert len(sp._enum) == z["sp_enum"].shape[0]           # known without rebuilding the dictionary
    enum = sorted(sp._enum.items(), key=lambda kv: kv[1])
    assert [[int(k[0]), int(k[1]), int(k[2])] for k, _ in enum] == z["sp_enum"].tolist()
    assert np.array_equal(sp._phi_X, z["sp_phi_X"]) and np.array_equal(sp._phi_X @ sp._phi_X.T, K)
    assert sorted([int(k), int(v)] for k, v in sp.X[0].items()) == json.loads(str(z["sp_X_graph0"]))
    Kt = sp.transform(G[:7])
    assert sp._phi_Y.shape == tuple(z["sp_phi_Y_shape"].tolist()) and len(sp._Y_enum) == 0
    assert np.array_equal(sp._phi_Y @ sp._phi_X.T, Kt)
    sp2 = pickle.loads(pickle.dumps(sp))
    assert np.array_equal(sp2._phi_X[:, :len(enum)], z["sp_phi_X"])


def test_shortest_path_distance_range_is_checked(gk):
    """A chain whose end-to-end distance leaves the int32 distance range must raise, not count as unreachable."""
    from grakel_amd._lib import GkError
    n = 1200
    w = 1000000 - 1                                           # below the host's per-edge cap, (n-1)*w > 2^30
    ed = {(i, i + 1): w for i in range(n - 1)}
    ed.update({(i + 1, i): w for i in range(n - 1)})
    with pytest.raises((GkError, NotImplementedError)):
        gk.ShortestPath().fit_transform([[ed, {i: 0 for i in range(n)}]])


def test_integration_md_ctypes_stub_runs_verbatim(gk, mutag_graphs):
    """INTEGRATION.md section B is the binding a GraKeL maintainer would paste: execute that code block exactly
    as printed (only the library path is resolved) and compare with the reference's MUTAG WL(h=5) matrix."""
    import re
    from grakel_amd import _lib
    from grakel_amd.batch import wl_batch_from_input
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    stub = [b for b in blocks if "def wl_gram(" in b]
    assert len(stub) == 1
    code = stub[0].replace('C.CDLL("libgk_hip.so")', 'C.CDLL(%r)' % _lib.LIB_PATH)
    assert code != stub[0]
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    G, z = mutag_graphs
    gb, _ = wl_batch_from_input(G)
    K, counts = ns["wl_gram"](gb.graph_ptr, gb.row_ptr, gb.col_idx, gb.node_label, gb.n_labels, 5, False)
    assert np.array_equal(K, z["K_wl5"]) and counts == z["wl5_label_counts"].tolist()
    Kn, _ = ns["wl_gram"](gb.graph_ptr, gb.row_ptr, gb.col_idx, gb.node_label, gb.n_labels, 5, True)
    d = np.sqrt(np.diagonal(K))
    assert np.allclose(Kn, K / np.outer(d, d), rtol=REL_TOL, atol=0)
    # ... and its transform stub: the first 150 graphs fitted, the rest looked up in their dictionaries
    fit, tgt = gb.slice_graphs(0, 150), gb.slice_graphs(150, gb.n_graphs)
    Kt = ns["wl_transform"]((fit.graph_ptr, fit.row_ptr, fit.col_idx, fit.node_label),
                            (tgt.graph_ptr, tgt.row_ptr, tgt.col_idx, tgt.node_label), gb.n_labels, 5)
    assert Kt is not None and np.array_equal(Kt, K[150:, :150])


# ------------------------------------------------------------------------------------------
# round 5: stand-ins for the TU datasets the reference PUBLISHES its running times on
# (grakel_amd/synthetic.py PUBLISHED_LIKE; goldens by the real grakel 0.1.11, tests/golden/pub_*.npz)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["nci1", "dd", "reddit", "collab"])
def test_published_like_sets_against_reference_goldens(gk, name):
    """WL-subtree h=5 on the FULL set (packed CSR through the C ABI: label counts, checksums, diagonal, a corner, 20 000
    sampled entries and every row sum of the reference's matrix), normalised, `transform` of 20 graphs against a fit on 200
    through the estimator on Python objects, and ShortestPath on the subsample the reference finished.  These sets leave
    the Erdos-Renyi sweet spot: thousands of vertices per graph (D&D), hubs of degree > 1 000 and > 256 input labels
    (REDDIT), mean degree ~ 60 (COLLAB), most WL labels shared by many graphs (NCI1)."""
    from grakel_amd import GraphBatch, synthetic as S
    from grakel_amd.engine import get_engine
    z = load_golden("pub_%s.npz" % name)
    graphs = S.PUBLISHED_LIKE[name][0]()
    gp, rp, ci, lab, nl = S.as_csr(graphs)
    eng = get_engine()
    db = eng.upload(GraphBatch(gp, rp, ci, lab, nl))
    feat, K = eng.wl_fit_transform(db, 5, to_host=True)
    assert db.label_counts == z["label_counts"].tolist()
    assert K.sum() == float(z["K_sum"][0]) and np.trace(K) == float(z["K_trace"][0]) and K.max() == float(z["K_max"][0])
    assert np.array_equal(np.diagonal(K), z["diag"]) and np.array_equal(K[:64, :64], z["K_block"])
    assert np.array_equal(K[z["samp_i"], z["samp_j"]], z["samp_v"])
    assert np.array_equal(K.sum(axis=1), z["row_sums"]) and np.array_equal(K, K.T)
    feat.close()
    feat, Kn = eng.wl_fit_transform(db, 5, normalize=2, to_host=True)
    d = np.sqrt(np.diagonal(K))
    assert np.abs(Kn - K / np.outer(d, d)).max() <= REL_TOL and np.all(np.diagonal(Kn) == 1.0)
    feat.close()
    db.close()
    del K, Kn
    Gt = S.as_grakel(graphs[-220:])
    est = gk.WeisfeilerLehman(n_iter=5)
    est.fit(Gt[:200])
    assert np.array_equal(est.transform(Gt[200:]), z["tr_block"])
    assert np.array_equal(est.transform(iter(Gt[200:])), z["tr_block"])      # a one-shot iterable: ONE ingestion per call
    if "sp_K" in z.files:
        sub = S.as_grakel([graphs[i] for i in z["sp_index"].tolist()], adjacency=True)
        sp = gk.ShortestPath()
        assert np.array_equal(sp.fit_transform(sub), z["sp_K"])
        assert len(sp._enum) == int(z["sp_n_features"][0])


@pytest.mark.parametrize("name", ["dd", "reddit", "collab"])
def test_published_like_sets_shortest_path_at_full_size(gk, name):
    """Round 6: ShortestPath(with_labels) on the FULL D&D-, REDDIT-BINARY- and COLLAB-like sets (172 M / 643 M / 36 M vertex
    pairs; graphs of up to 5 748 vertices: the counter-row, row-slab and 64 / 32 / 16-column breadth-first-search size
    classes at the sizes the README quotes times for).  The matrix is compared with (1) tests/golden/pub_<set>_sp_full.npz --
    checksums, diagonal, every row sum, a corner and 20 000 sampled entries of the full matrix, written by oracle/sp_fast.py,
    which tests/test_oracle.py pins entry for entry to the real reference -- and (2) pub_<set>_sp_big.npz: the block of the
    set's largest graphs (the 5 748-vertex giant; the ten largest REDDIT-like threads) as grakel 0.1.11 itself computed it
    in 6.5 / ~ 20 minutes.  Feature and pair counts too (= len(ShortestPath._enum), shortest_path.py:468-490)."""
    import bench
    from grakel_amd import GraphBatch, synthetic as S
    from grakel_amd.engine import get_engine
    z = load_golden("pub_%s_sp_full.npz" % name)
    graphs = S.PUBLISHED_LIKE[name][0]()
    gp, rp, ci, lab, nl = S.as_csr(graphs)
    eng = get_engine()
    db = eng.upload(GraphBatch(gp, rp, ci, lab, nl))
    pb = eng.sp_build(db, None, True)
    feat = eng.features(pb, 1)
    K = eng.gram(feat, 0, to_host=True)
    assert pb.label_counts[0] == int(z["n_features"][0]) and pb.n_nodes == int(z["n_pairs"][0])
    rec = bench.check_sp_matrix(K, name)
    assert rec["equals_full_set_fixture"] and (name == "collab" or rec["equals_real_reference_on_largest_graphs"])
    # normalised, through the same job: within REL_TOL of the reference's formula on the exact integers
    Kn = eng.gram(feat, 2, to_host=True)
    d = np.sqrt(np.diagonal(K))
    assert np.abs(Kn - K / np.outer(d, d)).max() <= REL_TOL and np.all(np.diagonal(Kn) == 1.0)
    feat.close(), pb.close(), db.close()
    if name == "collab":
        return
    # the estimator on Python objects: the largest graphs against the small ones, as the real reference ran them
    zb = load_golden("pub_%s_sp_big.npz" % name)
    ix = zb["index"].tolist()
    sub = S.as_grakel([graphs[i] for i in ix], adjacency=True)
    sp = gk.ShortestPath()
    assert np.array_equal(sp.fit_transform(sub), zb["K"]) and len(sp._enum) == int(zb["n_features"][0])
    spn = gk.ShortestPath(normalize=True)
    spn.fit(sub[-6:])
    assert np.allclose(spn.transform(sub[:1]), zb["Kn_tr"], rtol=REL_TOL, atol=0)


@pytest.mark.parametrize("no_huge", [0, 1])
def test_graphs_of_thousands_of_vertices_through_the_graph_major_builder(gk, gkopt, no_huge):
    """Round 6: a graph above 1 024 vertices is counted by a whole workgroup of the graph-major feature builder
    (gm_pairs_huge_kernel; up to 8 192 vertices) instead of sending the job to the label-major builder and the relabel route
    with full sorts (option feat.gm_no_huge = 1: rounds 1-5).  Chains with contacts of 1 100 .. 6 000 vertices next to small
    graphs, few labels so that large classes survive several levels; partitions, matrix, normalised matrix and transform
    against the oracle, and the route the relabel took."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    gkopt("feat.gm_no_huge", no_huge)
    rs = np.random.RandomState(17)
    X = []
    for n in (1100, 40, 2500, 6000, 25, 1025, 3, 1024, 700):
        ed = {i: [] for i in range(n)}
        for v in range(1, n):
            for u in {v - 1, max(0, v - 3) if rs.rand() < 0.5 else v - 1, int(rs.randint(0, v)) if rs.rand() < 0.05 else v - 1}:
                if u != v and u not in ed[v]:
                    ed[v].append(u), ed[u].append(v)
        X.append([ed, dict(enumerate(rs.randint(0, 3, n).tolist()))])
    ref = O.WLOracle(n_iter=3)
    K = ref.fit_transform(X)
    est = gk.WeisfeilerLehman(n_iter=3)
    assert np.array_equal(est.fit_transform(X), K)
    eng = get_engine()
    db = eng.upload(wl_batch_from_input(X)[0])
    assert eng.wl_relabel(db, 3) == ref.label_counts
    assert db.stream_route == (not no_huge)                       # the route without host round trips takes the job now
    db.close()
    Kn = gk.WeisfeilerLehman(n_iter=3, normalize=True).fit_transform(X)
    d = np.sqrt(np.diagonal(K))
    assert np.allclose(Kn, K / np.outer(d, d), rtol=REL_TOL, atol=0)
    assert np.array_equal(est.transform(X[2:5]), K[2:5])
    # the host-driven route (wl.no_stream) with such graphs: the sort-free dictionary, no label-grouped order
    gkopt("wl.no_stream", 1)
    assert np.array_equal(gk.WeisfeilerLehman(n_iter=3).fit_transform(X), K)


def test_transform_of_a_generator_above_the_lookup_threshold(gk):
    """`transform` ingests its input ONCE: a generator whose targets hold more than 1/32 of the fitted nodes (the look-up
    route declines, the joint route takes over) used to be exhausted by the first ingestion."""
    X = er_dataset(60, 20, 0.2, 3, 9)
    ref = O.WLOracle(n_iter=3)
    ref.fit_transform(X[:40])
    Kt = ref.transform(X[40:])
    est = gk.WeisfeilerLehman(n_iter=3)
    est.fit(X[:40])
    assert np.array_equal(est.transform(x for x in X[40:]), Kt)
    est.transform_route = "lookup"
    assert np.array_equal(est.transform(x for x in X[40:]), Kt)


@pytest.mark.parametrize("no_wave", [0, 1, 2, 3])
def test_dense_graphs_and_hubs_through_the_wave_signature_kernels(gk, gkopt, no_wave):
    """Graphs whose vertices have 33..1024 neighbours (near-cliques, ego networks: the COLLAB kind) take the wave-per-node
    signature and verification kernels (round 5), hubs beyond 1024 the workgroup kernel; option wl.no_wave_sig = 1 keeps the
    rounds 1-4 kernels.  Partitions, matrix and transform against the oracle either way."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    gkopt("wl.no_wave_sig", 1 if no_wave == 1 else 0)
    gkopt("wl.no_converge", 1 if no_wave == 2 else 0)          # (2: every level computed even when the partition has converged)
    gkopt("wl.no_frozen_skip", 1 if no_wave == 3 else 0)       # (3: hubs that are alone in their class are sorted at every level all the same)
    rs = np.random.RandomState(3)
    X = random_labelled_graphs(12, 60, 90, 0.7, 3, 17, fmt="dict")                  # degree ~ 50
    X += random_labelled_graphs(3, 150, 160, 0.9, 2, 18, fmt="dict")                # degree ~ 140 (R = 4)
    for hub in (70, 600, 1500):                                                      # stars with a few cross links
        ed = {0: list(range(1, hub + 1))}
        ed.update({i: [0] + ([i % hub + 1] if i % 3 == 0 else []) for i in range(1, hub + 1)})
        for i in range(1, hub + 1):                                                  # make the cross links symmetric
            for j in ed[i][1:]:
                if i not in ed[j]:
                    ed[j].append(i)
        X.append([ed, {i: int(rs.randint(0, 2)) for i in range(hub + 1)}])
    X.append(X[0]), X.append(X[-2])                                                  # isomorphic copies: shared classes everywhere
    wl, K, levels = _oracle_levels(X, 3)
    gb, _ = wl_batch_from_input(X)
    eng = get_engine()
    db = eng.upload(gb)
    counts = eng.wl_relabel(db, 3)
    assert counts == [len(set(l.tolist())) for l in levels]
    for lvl in range(4):
        assert same_partition(eng.wl_labels(db, lvl), levels[lvl])
    est = gk.WeisfeilerLehman(n_iter=3)
    assert np.array_equal(est.fit_transform(X), K)
    assert np.array_equal(est.transform(X[3:9]), K[3:9])


@pytest.mark.parametrize("no_converge", [0, 1])
def test_converged_partition_levels_are_copies(gk, gkopt, no_converge):
    """The host-driven relabel stops computing once two consecutive levels have the same number of labels (classes only
    split: the partition has converged and every later level repeats it; wl.hip: RelabelState::converged) and copies the
    remaining levels.  Dense graphs (degree ~ 50: off the stream route) converge after a level or two; n_iter = 7 leaves five
    levels to copy.  Counts, partitions, matrix and transform against the oracle, with the shortcut and without
    (wl.no_converge)."""
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.engine import get_engine
    gkopt("wl.no_converge", no_converge)
    X = random_labelled_graphs(14, 60, 90, 0.7, 3, 21, fmt="dict")
    X.append(X[0]), X.append(X[5])                                    # isomorphic copies: shared classes at every level
    wl, K, levels = _oracle_levels(X, 7)
    counts_ref = [len(set(l.tolist())) for l in levels]
    assert counts_ref[-1] == counts_ref[-2] == counts_ref[-3]         # (the case the test is about)
    gb, _ = wl_batch_from_input(X)
    eng = get_engine()
    db = eng.upload(gb)
    assert eng.wl_relabel(db, 7) == counts_ref
    for lvl in range(8):
        assert same_partition(eng.wl_labels(db, lvl), levels[lvl])
    est = gk.WeisfeilerLehman(n_iter=7)
    assert np.array_equal(est.fit_transform(X), K)
    assert np.array_equal(est.transform(X[2:9]), K[2:9])
    estn = gk.WeisfeilerLehman(n_iter=7, normalize=True)
    refn = O.WLOracle(n_iter=7, normalize=True)
    assert np.allclose(estn.fit_transform(X), refn.fit_transform(X), rtol=REL_TOL, atol=0)


@pytest.mark.parametrize("gpus", [1, 2])
def test_config6_subsampled_against_the_oracle(gk, gpus):
    """bench.py --workload config6 (200 000 graphs: a 320 GB matrix that no single GPU holds) multiplies every rank's rows
    in sub-blocks that reuse one device buffer.  Here the same code path on a 2 400-graph prefix of the same generator with
    forced 500-row sub-blocks, on one rank and -- `bench.py --gpus 2` launching ITSELF under torch.distributed.run, both
    ranks on cuda:0 over gloo (test hook) -- on two: the sum over all sub-blocks of all ranks must be the oracle's sum of K
    (K[i, j] only depends on graphs i and j, so the prefix's matrix is the corner of the full set's)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GK_BENCH_BACKEND="gloo", GK_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--workload", "config6", "--graphs", "2400",
                          "--block-rows", "500", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    d = json.loads(lines[0])
    K = O.WLOracle(n_iter=5).fit_transform(er_dataset(2400, 30, 0.1, 5, 0))
    assert d["n_gpus"] == gpus and d["checks"]["K_sum_over_sub_blocks"] == float(K.sum())
    assert d["checks"]["sub_blocks_this_rank"] == (5 if gpus == 1 else 3) and d["checks"]["two_passes_agree"]
    if gpus == 2:
        assert d["rccl"]["ranks"] == 2 and len(d["per_rank"]) == 2 and all(r["gram_rows_ms"] > 0 for r in d["per_rank"])


def test_integration_md_section_c_runs_with_one_process_per_gpu(gk, tmp_path):
    """INTEGRATION.md section C as a RUNNING C program (tests/c_abi/multi_gpu_run.c + the section's own code,
    tests/c_abi/multi_gpu_stub.c): one process per visible GPU, every process with its own shard file, the communicator id
    handed over through a file, gk_comm_init / gk_batch_allgather / gk_gram_sharded over RCCL.  On a one-GPU box this is a
    communicator of one rank; on an 8-GPU node it is the first run of the C path with real ranks.  The rows the processes
    write must be the oracle's matrix."""
    import subprocess
    from grakel_amd import GraphBatch, _lib
    from grakel_amd.dist import shard_bounds
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "multi_gpu_run")
    libdir = os.path.join(root, "grakel_amd")
    cc = subprocess.run(["gcc", "-O1", "-DGK_STUB_ID_IS_DISTRIBUTED", "-I", os.path.join(root, "include"),
                         os.path.join(root, "tests", "c_abi", "multi_gpu_run.c"), os.path.join(root, "tests", "c_abi", "multi_gpu_stub.c"),
                         "-o", exe, "-L", libdir, "-l:libgk_hip.so", "-Wl,-rpath," + libdir,
                         "-Wl,--unresolved-symbols=ignore-in-shared-libs"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    n_ranks = max(1, min(_lib.device_count(), 8))
    N, n_iter = 96, 3
    X = er_dataset(N, 24, 0.15, 4, 21)
    full = GraphBatch(*er_dataset_csr(N, 24, 0.15, 4, 21), 4)
    b = shard_bounds(N, n_ranks)
    procs = []
    for r in range(n_ranks):
        sh = full.slice_graphs(b[r], b[r + 1])
        path = str(tmp_path / ("shard%d.bin" % r))
        with open(path, "wb") as f:
            np.array([sh.n_graphs, sh.n_nodes, sh.n_edges, sh.n_labels], np.int64).tofile(f)
            for arr in (sh.graph_ptr, sh.row_ptr, sh.col_idx, sh.node_label):
                np.ascontiguousarray(arr, np.int32).tofile(f)
        np.array([N], np.int64).tofile(path + ".total")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([exe, str(r), str(n_ranks), str(r), str(tmp_path / "comm.id"), path,
                                       str(tmp_path / ("rows%d.bin" % r)), str(n_iter)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    K = O.WLOracle(n_iter=n_iter).fit_transform(X)
    for r in range(n_ranks):
        raw = np.fromfile(str(tmp_path / ("rows%d.bin" % r)), np.uint8)
        lo, hi, total = raw[:24].view(np.int64).tolist()
        assert (lo, hi, total) == (b[r], b[r + 1], N)
        assert np.array_equal(raw[24:].view(np.float64).reshape(hi - lo, N), K[lo:hi])


def test_debug_guard_catches_a_write_behind_a_block(gk, gkopt):
    """Option debug.guard (round 5): every block of the context's allocator gets a red zone on
    either side; a kernel that writes outside its block -- provoked here with gk_block_copy aimed 32 bytes behind the end of
    a Gram matrix -- turns the next gk_synchronize into GK_ERR_STATE instead of a silent corruption.  The whole GPU suite
    runs under it in tests/tools/guard_suite.sh (profiles/r05_guard_suite.txt)."""
    import ctypes
    from grakel_amd import GraphBatch, _lib
    from grakel_amd.engine import get_engine
    eng = get_engine()
    gkopt("debug.guard", 1)
    N = 37
    db = eng.upload(GraphBatch(*er_dataset_csr(N, 12, 0.3, 3, 4), 3))
    eng.wl_relabel(db, 2)
    feat = eng.features(db, 3)
    eng.gram(feat, 0, to_host=False)
    eng.synchronize()                                            # nothing wrong so far
    p, r, c = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
    _lib.check(eng.lib.gk_gram_dev_ptr(feat.handle, ctypes.byref(p), ctypes.byref(r), ctypes.byref(c)))
    assert (r.value, c.value) == (N, N)
    eng.block_copy(p.value, 1, 4, N, p.value + N * N * 8, 4)     # four float64 just behind the matrix
    with pytest.raises(_lib.GkError, match="red zone"):
        eng.synchronize()
    eng.synchronize()                                            # reported once; the context stays usable
    assert np.array_equal(eng.gram(feat, 0), O.WLOracle(n_iter=2).fit_transform(er_dataset(N, 12, 0.3, 3, 4)))
    feat.close(), db.close()


@pytest.mark.parametrize("no_fused", [0, 1])
def test_lookup_transform_match_kernels_fused_and_per_level(gk, gkopt, no_fused):
    """The look-up transform matches target classes to fitted classes with one wave per class (round 5: neighbour lists in
    registers, no scratch); a handful of targets run all levels in ONE single-workgroup launch, larger sets (or option
    transform.no_fused) two launches per level.  Degrees up to the 64-neighbour limit, unseen labels, isolated vertices."""
    gkopt("transform.no_fused", no_fused)
    X = random_labelled_graphs(40, 5, 30, 0.35, 3, 31, fmt="dict")
    hub = {0: list(range(1, 61))}
    hub.update({i: [0] for i in range(1, 61)})
    X.append([hub, {i: i % 3 for i in range(61)}])               # a representative with 60 neighbours
    Y = X[5:9] + [[{0: [1], 1: [0], 2: []}, {0: 0, 1: 7, 2: 1}]] + [X[-1]] + random_labelled_graphs(6, 5, 30, 0.35, 3, 77, fmt="dict")
    ref = O.WLOracle(n_iter=4)
    ref.fit_transform(X)
    Kt = ref.transform(Y)
    est = gk.WeisfeilerLehman(n_iter=4)
    est.transform_route = "lookup"
    est.fit(X)
    assert np.array_equal(est.transform(Y), Kt)
    assert np.array_equal(est.transform(Y[:1]), Kt[:1])
    estn = gk.WeisfeilerLehman(n_iter=4, normalize=True)
    estn.transform_route = "lookup"
    estn.fit(X)
    refn = O.WLOracle(n_iter=4, normalize=True)
    refn.fit_transform(X)
    assert np.abs(estn.transform(Y) - refn.transform(Y)).max() <= REL_TOL


class _LabelledEdgeKernel(object):
    """A base kernel grakel_amd knows nothing about (host Python, the estimator methods the WL framework calls): features
    of a graph = its vertex labels and the unordered label pairs of its adjacency entries, K = dot product of the counts.
    It depends on the relabelled node labels AND on the graph structure, on label IDENTITY across fit and transform."""

    def __init__(self, **kw):
        self.kw = kw

    @staticmethod
    def _features(X):
        out = []
        for x in X:
            g, lab = x[0], x[1]
            c = {}
            for v, l in lab.items():
                c[("v", l)] = c.get(("v", l), 0) + 1
                for u in g.get(v, []):
                    k = ("e", min(l, lab[u]), max(l, lab[u]))
                    c[k] = c.get(k, 0) + 1
            out.append(c)
        return out

    @staticmethod
    def _dot(A, B):
        return np.array([[float(sum(v * b.get(k, 0) for k, v in a.items())) for b in B] for a in A])

    def fit(self, X):
        self.F = self._features(X)
        return self

    def fit_transform(self, X):
        self.fit(X)
        return self._dot(self.F, self.F)

    def transform(self, Y):
        self.G = self._features(Y)
        return self._dot(self.G, self.F)

    def diagonal(self):
        xd = np.array([float(sum(v * v for v in a.values())) for a in self.F])
        if hasattr(self, "G"):
            return xd, np.array([float(sum(v * v for v in a.values())) for a in self.G])
        return xd


def test_wl_over_an_arbitrary_host_base_kernel(gk):
    """WeisfeilerLehman(base_graph_kernel=<any kernel class>) (weisfeiler_lehman.py:77-109; round 5): the relabelling runs on
    the device, every level's relabelled graphs -- reference-identical label ids for everything the fit has seen -- go to a
    host base kernel per level, as the reference hands them over.  Checked against the same base kernel applied to the
    oracle's levels: fit_transform, transform (seen and unseen classes), diagonal, normalised, pickling."""
    X = random_labelled_graphs(18, 4, 14, 0.35, 3, 41, fmt="dict")
    Y = X[2:5] + random_labelled_graphs(5, 4, 14, 0.35, 4, 43, fmt="dict") + [[{0: [1], 1: [0], 2: []}, {0: 9, 1: 0, 2: 1}]]
    h = 3
    ref = O.WLOracle(n_iter=h)
    ref.fit_transform(X, keep_levels=True)
    ref.transform(Y, keep_levels=True)
    bases, Kx, Ky, xd, yd = [], 0, 0, 0, 0
    for l in range(h + 1):
        b = _LabelledEdgeKernel()
        Kx = Kx + b.fit_transform([(x[0], ref.levels[l][j]) for j, x in enumerate(X)])
        Ky = Ky + b.transform([(y[0], ref.y_levels[l][j]) for j, y in enumerate(Y)])
        d = b.diagonal()
        xd, yd = xd + d[0], yd + d[1]
    est = gk.WeisfeilerLehman(n_iter=h, base_graph_kernel=_LabelledEdgeKernel)
    assert np.array_equal(est.fit_transform(X), Kx) and np.array_equal(est.diagonal(), xd)
    assert sorted(est.X) == list(range(h + 1)) and all(isinstance(b, _LabelledEdgeKernel) for b in est.X.values())
    assert np.array_equal(est.transform(Y), Ky)
    d = est.diagonal()
    assert np.array_equal(d[0], xd) and np.array_equal(d[1], yd)
    est2 = gk.WeisfeilerLehman(n_iter=h, base_graph_kernel=(_LabelledEdgeKernel, {"anything": 1}), normalize=True)
    est2.fit(iter(X))
    with np.errstate(divide="ignore", invalid="ignore"):
        want = np.nan_to_num(Ky / np.sqrt(np.outer(yd, xd)))
    assert np.abs(est2.transform(Y) - want).max() <= REL_TOL
    assert np.abs(gk.WeisfeilerLehman(n_iter=h, base_graph_kernel=_LabelledEdgeKernel, normalize=True).fit_transform(X)
                  - Kx / np.sqrt(np.outer(xd, xd))).max() <= REL_TOL
    est3 = pickle.loads(pickle.dumps(est))
    assert np.array_equal(est3.transform(Y), Ky)
    # the same route with one of the ACCELERATED kernels handed in as a foreign class gives the accelerated matrix
    class _VHClone(gk.VertexHistogram):
        pass
    assert np.array_equal(gk.WeisfeilerLehman(n_iter=h, base_graph_kernel=_VHClone).fit_transform(X), ref.fit_transform(X))

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference; see oracle/build_ref.sh):

    bash oracle/build_ref.sh && python tests/golden/make_golden.py [--skip-big]

Every number stored here was produced by grakel 0.1.11's own classes
(WeisfeilerLehman / VertexHistogram / ShortestPath); the fixtures are data, no
reference code is copied.  The GPU box has no /root/reference, so the parity
tests compare against these files (and against oracle/grakel_oracle.py, which
tests/test_oracle.py pins to the same files).
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("GK_REF_BUILD", "/tmp/grakel_oracle")
sys.path.insert(0, REF)

import grakel  # noqa: E402  (the real reference)
from grakel import WeisfeilerLehman, VertexHistogram, ShortestPath, EdgeHistogram  # noqa: E402
from grakel import WeisfeilerLehmanOptimalAssignment, CoreFramework  # noqa: E402
from grakel.datasets.base import read_data  # noqa: E402

from grakel_amd.synthetic import er_dataset, nci1_like, random_labelled_graphs  # noqa: E402
sys.path.insert(0, HERE)
from small_sets import SMALL_SETS, split, sp_inputs, sp_dyadic_graphs, sp_float_graphs, sp_float_big_graphs  # noqa: E402
from small_sets import sp_large_unit_graphs, sp_large_unit_paths  # noqa: E402

warnings.filterwarnings("ignore")


def as_int(K):
    Ki = np.rint(K).astype(np.int64)
    assert np.array_equal(Ki.astype(np.float64), K), "Gram matrix is not integer valued"
    return Ki


def sample_entries(K, m, seed):
    rs = np.random.RandomState(seed)
    i = rs.randint(0, K.shape[0], m)
    j = rs.randint(0, K.shape[1], m)
    return i.astype(np.int32), j.astype(np.int32), K[i, j]


def doc_goldens():
    H2O = [{'a': ['b', 'c'], 'b': ['a'], 'c': ['a']}, {'a': 'O', 'b': 'H', 'c': 'H'}]
    H3O = [{'a': ['b', 'c', 'd'], 'b': ['a'], 'c': ['a'], 'd': ['a']},
           {'a': 'O', 'b': 'H', 'c': 'H', 'd': 'H'}]
    out = {}
    sp = ShortestPath()
    out["sp_fit_h2o"] = sp.fit_transform([H2O]).tolist()
    out["sp_tr_h3o"] = sp.transform([H3O]).tolist()
    spn = ShortestPath(normalize=True)
    spn.fit([H2O])
    out["sp_norm_tr_h3o"] = spn.transform([H3O]).tolist()
    vh = VertexHistogram(normalize=True)
    vh.fit([H2O])
    out["vh_norm_tr_h3o"] = vh.transform([H3O]).tolist()
    wl = WeisfeilerLehman(n_iter=5)
    out["wl5_fit_both"] = wl.fit_transform([H2O, H3O]).tolist()
    out["wl5_inv_labels"] = {str(k): v for k, v in wl._inv_labels.items()}
    wl1 = WeisfeilerLehman(n_iter=5)
    wl1.fit([H2O])
    out["wl5_fit_h2o_tr_h3o"] = wl1.transform([H3O]).tolist()
    # the 4x4 known-answer APSP matrix of grakel/tests/test_graph.py:61-74
    from grakel import Graph
    A = np.array([[0, 1, 0, 3], [1, 0, 0, 2], [2, 3, 0, 1], [1, 0, 0, 0]])
    S, _ = Graph(A, {0: 'a', 1: 'b', 2: 'c', 3: 'd'}).build_shortest_path_matrix("auto")
    out["apsp_4x4"] = [[None if np.isinf(x) else float(x) for x in r] for r in S]
    with open(os.path.join(HERE, "doc_goldens.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("doc goldens", out["sp_fit_h2o"], out["sp_tr_h3o"], out["wl5_fit_both"])


def mutag():
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "grakel", "tests", "data"))
    try:
        G = read_data('MUTAG', with_classes=True).data
    finally:
        os.chdir(cwd)
    # pack the dataset itself (public TU data: arrays only)
    gi, labs, src, dst, elab = [], [], [], [], []
    node_ids = []
    for g, (edges, nl, el) in enumerate(G):
        for v in sorted(nl):
            node_ids.append(v), gi.append(g), labs.append(nl[v])
        for (a, b) in sorted(edges):
            src.append(a), dst.append(b), elab.append(el[(a, b)])
    oa = WeisfeilerLehmanOptimalAssignment(n_iter=4)
    K_oa = as_int(oa.fit_transform(G[:120]))
    K_oa_tr = as_int(oa.transform(G[120:]))
    oan = WeisfeilerLehmanOptimalAssignment(n_iter=2, normalize=True)
    K_oa_norm = oan.fit_transform(G[:120])
    K_oa_norm_tr = oan.transform(G[120:])
    wsp = WeisfeilerLehman(n_iter=2, base_graph_kernel=ShortestPath)
    K_wlsp = as_int(wsp.fit_transform(G[:100]))
    K_wlsp_tr = as_int(wsp.transform(G[100:140]))
    wspn = WeisfeilerLehman(n_iter=1, normalize=True, base_graph_kernel=(ShortestPath, {"with_labels": True}))
    K_wlspn = wspn.fit_transform(G[:100])
    K_wlspn_tr = wspn.transform(G[100:140])
    core = dict()
    for tag, base in (("sp", None), ("vh", VertexHistogram), ("wl2", (WeisfeilerLehman, {"n_iter": 2}))):
        cf = CoreFramework(base_graph_kernel=base)
        core["K_core_%s" % tag] = as_int(cf.fit_transform(G[:100])).astype(np.int64)
        core["K_core_%s_tr" % tag] = as_int(cf.transform(G[100:140])).astype(np.int64)
    cfn = CoreFramework(normalize=True)
    core["K_core_sp_norm"] = cfn.fit_transform(G[:100])
    core["K_core_sp_norm_tr"] = cfn.transform(G[100:140])
    eh = EdgeHistogram()
    K_eh = as_int(eh.fit_transform(G[:120]))
    K_eh_tr = as_int(eh.transform(G[120:]))
    K_eh_norm = EdgeHistogram(normalize=True).fit_transform(G)
    K_vh = as_int(VertexHistogram().fit_transform(G))
    K_wl = as_int(WeisfeilerLehman(n_iter=5).fit_transform(G))
    K_sp = as_int(ShortestPath().fit_transform(G))
    wl = WeisfeilerLehman(n_iter=3)
    wl.fit(G[:120])
    K_wl_tr = as_int(wl.transform(G[120:]))
    sp = ShortestPath()
    sp.fit(G[:120])
    K_sp_tr = as_int(sp.transform(G[120:]))
    wln = WeisfeilerLehman(n_iter=3, normalize=True)
    wln.fit(G[:120])
    K_wl_tr_norm = wln.transform(G[120:])
    wl5 = WeisfeilerLehman(n_iter=5)
    wl5.fit(G)
    np.savez_compressed(
        os.path.join(HERE, "mutag.npz"),
        node_id=np.array(node_ids, np.int32), node_graph=np.array(gi, np.int32),
        node_label=np.array(labs, np.int32), edge_src=np.array(src, np.int32),
        edge_dst=np.array(dst, np.int32), edge_label=np.array(elab, np.int32),
        K_eh=K_eh.astype(np.int32), K_eh_tr=K_eh_tr.astype(np.int32), K_eh_norm=K_eh_norm,
        K_wlsp2=K_wlsp.astype(np.int64), K_wlsp2_tr=K_wlsp_tr.astype(np.int64), K_wlsp1_norm=K_wlspn,
        K_wlsp1_norm_tr=K_wlspn_tr,
        K_oa4=K_oa.astype(np.int32), **core, K_oa4_tr=K_oa_tr.astype(np.int32), K_oa2_norm=K_oa_norm,
        K_oa2_norm_tr=K_oa_norm_tr,
        K_vh=K_vh.astype(np.int32), K_wl5=K_wl.astype(np.int32), K_sp=K_sp.astype(np.int64),
        K_wl3_tr=K_wl_tr.astype(np.int32), K_sp_tr=K_sp_tr.astype(np.int64),
        K_wl3_tr_norm=K_wl_tr_norm,
        wl5_label_counts=np.array([len(wl5._inv_labels[i]) for i in range(6)], np.int64))
    print("MUTAG sums", K_vh.sum(), K_wl.sum(), K_sp.sum(), "traces",
          np.trace(K_vh), np.trace(K_wl), np.trace(K_sp))


def sp_dyadic():
    """ShortestPath on float edge weights that are multiples of a power of two: the
    reference's Gram matrices, and its ``_enum`` keys (float distances) for the fitted-state comparison."""
    G = sp_dyadic_graphs()
    tr, te = G[:16], G[16:]
    out = {}
    for name, kw in (("auto", {}), ("fw", dict(algorithm_type="floyd_warshall"))):
        sp = ShortestPath(normalize=False, **kw)
        out["K_fit_" + name] = as_int(sp.fit_transform(tr))
        out["K_tr_" + name] = as_int(sp.transform(te))
        keys = sorted(sp._enum.items(), key=lambda kv: kv[1])
        out["enum_labels_" + name] = np.array([[k[0], k[1]] for k, _ in keys])
        out["enum_dist_" + name] = np.array([float(k[2]) for k, _ in keys])
    spn = ShortestPath(normalize=True)
    out["K_fit_norm"] = spn.fit_transform(tr)
    out["K_tr_norm"] = spn.transform(te)
    spu = ShortestPath(normalize=False, with_labels=False)
    out["K_fit_unlabelled"] = as_int(spu.fit_transform([[g[0]] for g in tr]))
    np.savez_compressed(os.path.join(HERE, "sp_dyadic.npz"), **out)
    print("sp_dyadic: fit", out["K_fit_auto"].shape, "sum", int(out["K_fit_auto"].sum()), "features", len(out["enum_dist_auto"]),
          "auto == fw:", bool(np.array_equal(out["K_fit_auto"], out["K_fit_fw"])))


def sp_float():
    """ShortestPath on GENERAL float edge weights (round 3): the reference's matrices for its three algorithm settings,
    its ``_enum`` keys as float64 bit patterns, transform, normalisation, unlabelled, and WL over ShortestPath."""
    G = sp_float_graphs()
    tr, te = G[:28], G[28:]
    out = {}
    for name, kw in (("auto", {}), ("fw", dict(algorithm_type="floyd_warshall")), ("dij", dict(algorithm_type="dijkstra"))):
        sp = ShortestPath(normalize=False, **kw)
        out["K_fit_" + name] = as_int(sp.fit_transform(tr))
        out["K_tr_" + name] = as_int(sp.transform(te))
        keys = sorted(sp._enum.items(), key=lambda kv: kv[1])
        out["enum_labels_" + name] = np.array([[k[0], k[1]] for k, _ in keys])
        out["enum_dist_bits_" + name] = np.array([float(k[2]) for k, _ in keys], np.float64).view(np.int64)
    spn = ShortestPath(normalize=True)
    out["K_fit_norm"] = spn.fit_transform(tr)
    out["K_tr_norm"] = spn.transform(te)
    spu = ShortestPath(normalize=False, with_labels=False)
    out["K_fit_unlabelled"] = as_int(spu.fit_transform([[g[0]] for g in tr]))
    wl = WeisfeilerLehman(n_iter=2, base_graph_kernel=ShortestPath)
    out["K_fit_wl_sp"] = as_int(wl.fit_transform(tr))
    out["K_tr_wl_sp"] = as_int(wl.transform(te))
    np.savez_compressed(os.path.join(HERE, "sp_float.npz"), **out)
    print("sp_float: fit", out["K_fit_auto"].shape, "sums auto / fw / dij", int(out["K_fit_auto"].sum()), int(out["K_fit_fw"].sum()),
          int(out["K_fit_dij"].sum()), "features", len(out["enum_labels_auto"]),
          "auto == fw:", bool(np.array_equal(out["K_fit_auto"], out["K_fit_fw"])),
          "fw == dij:", bool(np.array_equal(out["K_fit_fw"], out["K_fit_dij"])))


def sp_float_big():
    """Round 4: general float edge weights on graphs above 143 vertices (all three algorithm settings, fit + transform, the
    float distances of the _enum keys), and CoreFramework over ShortestPath on general float weights (its subgraphs are
    dictionary-format Graph objects: dijkstra under "auto")."""
    G = sp_float_big_graphs()
    tr, te = G[:6], G[6:]
    out = {}
    for name, kw in (("auto", {}), ("fw", dict(algorithm_type="floyd_warshall")), ("dij", dict(algorithm_type="dijkstra"))):
        sp = ShortestPath(normalize=False, **kw)
        out["K_fit_" + name] = as_int(sp.fit_transform(tr))
        out["K_tr_" + name] = as_int(sp.transform(te))
        keys = sorted(sp._enum.items(), key=lambda kv: kv[1])
        out["enum_n_" + name] = np.array([len(keys)])
        out["enum_dist_bits_" + name] = np.array([float(k[2]) for k, _ in keys], np.float64).view(np.int64)
    from grakel import CoreFramework
    S = [g for i, g in enumerate(sp_float_graphs()) if i < 4 or i % 7]      # undirected only: the reference's core_number needs symmetric neighbourhoods
    ctr, cte = S[:24], S[24:]
    for name, kw in (("auto", {}), ("fw", dict(algorithm_type="floyd_warshall"))):
        cf = CoreFramework(base_graph_kernel=(ShortestPath, kw))
        out["K_core_fit_" + name] = as_int(cf.fit_transform(ctr))
        out["K_core_tr_" + name] = as_int(cf.transform(cte))
    cfn = CoreFramework(normalize=True)
    out["K_core_fit_norm"] = cfn.fit_transform(ctr)
    np.savez_compressed(os.path.join(HERE, "sp_float_big.npz"), **out)
    print("sp_float_big: sums auto / fw / dij", int(out["K_fit_auto"].sum()), int(out["K_fit_fw"].sum()), int(out["K_fit_dij"].sum()),
          "features", int(out["enum_n_auto"][0]), "core sums", int(out["K_core_fit_auto"].sum()), int(out["K_core_fit_fw"].sum()))


def mutag_state(n_graphs=60):
    """Fitted state a consumer may read on the first graphs of MUTAG: VertexHistogram ``X`` /
    ``_labels``; WeisfeilerLehman ``_inv_labels`` and per level ``X[i].X`` / ``X[i]._labels``; ShortestPath
    ``_enum`` / ``_phi_X`` / ``X``."""
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "grakel", "tests", "data"))
    try:
        G = read_data('MUTAG', with_classes=True).data[:n_graphs]
    finally:
        os.chdir(cwd)

    def dense(X):
        return np.asarray(X.todense() if hasattr(X, "todense") else X)

    vh = VertexHistogram()
    vh.fit(G)
    out = dict(n_graphs=n_graphs, vh_X=dense(vh.X).astype(np.int32),
               vh_labels=json.dumps(sorted([[int(k), int(v)] for k, v in vh._labels.items()])))
    wl = WeisfeilerLehman(n_iter=3)
    wl.fit(G)
    out["wl_inv_labels"] = json.dumps({str(i): {str(k): int(v) for k, v in d.items()} for i, d in wl._inv_labels.items()})
    for i in range(4):
        out["wl_X%d" % i] = dense(wl.X[i].X).astype(np.int32)
        out["wl_labels%d" % i] = json.dumps(sorted([[int(k), int(v)] for k, v in wl.X[i]._labels.items()]))
    sp = ShortestPath()
    sp.fit_transform(G)
    enum = sorted(sp._enum.items(), key=lambda kv: kv[1])
    out["sp_enum"] = np.array([[int(k[0]), int(k[1]), int(k[2])] for k, _ in enum], np.int64)
    assert [v for _, v in enum] == list(range(len(enum)))
    out["sp_phi_X"] = sp._phi_X.astype(np.int32)
    out["sp_X_graph0"] = json.dumps(sorted([[int(k), int(v)] for k, v in sp.X[0].items()]))
    sp.transform(G[:7])
    out["sp_phi_Y_shape"] = np.array(sp._phi_Y.shape, np.int64)
    np.savez_compressed(os.path.join(HERE, "mutag_state.npz"), **out)
    print("MUTAG fitted state:", {k: (v.shape if hasattr(v, "shape") else len(v)) for k, v in out.items() if k != "n_graphs"})


def small_sets():
    out = {}
    for name, kw in SMALL_SETS:
        G = random_labelled_graphs(**kw)
        tr, te = split(G)
        for h in (1, 3):
            wl = WeisfeilerLehman(n_iter=h)
            out["%s/wl%d_fit" % (name, h)] = as_int(wl.fit_transform(tr))
            out["%s/wl%d_tr" % (name, h)] = as_int(wl.transform(te))
            out["%s/wl%d_counts" % (name, h)] = np.array(
                [len(wl._inv_labels[i]) for i in range(h + 1)], np.int64)
        wln = WeisfeilerLehman(n_iter=2, normalize=True)
        out[name + "/wl2n_fit"] = wln.fit_transform(tr)
        out[name + "/wl2n_tr"] = wln.transform(te)
        if True:                      # dict sets carry isolated {v: []} vertices, which WL-OA drops entirely
            oa = WeisfeilerLehmanOptimalAssignment(n_iter=3)
            out[name + "/oa3_fit"] = as_int(oa.fit_transform(tr))
            out[name + "/oa3_tr"] = as_int(oa.transform(te))
        vh = VertexHistogram()
        out[name + "/vh_fit"] = as_int(vh.fit_transform(tr))
        out[name + "/vh_tr"] = as_int(vh.transform(te))
        vhn = VertexHistogram(normalize=True)
        out[name + "/vhn_fit"] = vhn.fit_transform(tr)
        out[name + "/vhn_tr"] = vhn.transform(te)
        trs, tes = sp_inputs(kw, tr), sp_inputs(kw, te)
        try:
            sp = ShortestPath()
            out[name + "/sp_fit"] = as_int(sp.fit_transform(trs))
            out[name + "/sp_tr"] = as_int(sp.transform(tes))
            spn = ShortestPath(normalize=True)
            out[name + "/spn_fit"] = spn.fit_transform(trs)
            out[name + "/spn_tr"] = spn.transform(tes)
            spu = ShortestPath(with_labels=False)
            out[name + "/spu_fit"] = as_int(spu.fit_transform(trs))
            out[name + "/spu_tr"] = as_int(spu.transform(tes))
            wsp = WeisfeilerLehman(n_iter=2, base_graph_kernel=ShortestPath)
            out[name + "/wlsp2_fit"] = as_int(wsp.fit_transform(trs))
            out[name + "/wlsp2_tr"] = as_int(wsp.transform(tes))
            if not kw["directed"]:        # core_number() needs symmetric neighbour lists
                for tag, base in (("sp", None), ("vh", VertexHistogram)):
                    cf = CoreFramework(base_graph_kernel=base)
                    out[name + "/core_%s_fit" % tag] = as_int(cf.fit_transform(trs))
                    out[name + "/core_%s_tr" % tag] = as_int(cf.transform(tes))
        except KeyError as e:           # tuples sets can hit the Dijkstra sink-vertex bug
            print("  SP skipped for", name, "(reference KeyError %s)" % e)
    np.savez_compressed(os.path.join(HERE, "small_sets.npz"), **out)
    print("small sets:", len(out), "arrays")


def er_config(tag, N, n, p, L, seed, h, nsamp, with_oa=True):
    G = er_dataset(N, n, p, L, seed)
    oa = dict()
    if with_oa:      # WL-OA keeps a dense N x (all labels) float64 histogram: config 3 would need 400 GB
        t0 = time.perf_counter()
        Ko = as_int(WeisfeilerLehmanOptimalAssignment(n_iter=h).fit_transform(G))
        oa_dt = time.perf_counter() - t0
        oi, oj, ov = sample_entries(Ko, nsamp, 321)
        oa = dict(oa_sum=np.array([Ko.sum()], np.int64), oa_row_sums=Ko.sum(axis=1).astype(np.int64),
                  oa_block=Ko[:64, :64].astype(np.int32), oa_samp_i=oi, oa_samp_j=oj,
                  oa_samp_v=ov.astype(np.int32), oa_ref_seconds=np.array([oa_dt]))
        print("ER", tag, "WL-OA ref %.2fs" % oa_dt, "sum", Ko.sum())
    t0 = time.perf_counter()
    wl = WeisfeilerLehman(n_iter=h)
    K = wl.fit_transform(G)
    dt = time.perf_counter() - t0
    Ki = as_int(K)
    i, j, v = sample_entries(Ki, nsamp, 123)
    np.savez_compressed(
        os.path.join(HERE, "er_%s.npz" % tag),
        params=np.array([N, n, L, seed, h], np.int64), p=np.array([p]),
        label_counts=np.array([len(wl._inv_labels[k]) for k in range(h + 1)], np.int64),
        K_sum=np.array([Ki.sum()], np.int64), K_trace=np.array([np.trace(Ki)], np.int64),
        K_max=np.array([Ki.max()], np.int64), diag=np.diagonal(Ki).astype(np.int32),
        K_block=Ki[:64, :64].astype(np.int32), row_sums=Ki.sum(axis=1).astype(np.int64),
        samp_i=i, samp_j=j, samp_v=v.astype(np.int32),
        ref_seconds=np.array([dt]), **oa)
    print("ER", tag, "ref %.2fs" % dt, "sum", Ki.sum(), "trace", np.trace(Ki), "max", Ki.max(),
          "counts", [len(wl._inv_labels[k]) for k in range(h + 1)])


def nci1_sp(N):
    G = nci1_like(N, 0, as_adj=True)
    t0 = time.perf_counter()
    sp = ShortestPath()
    K = sp.fit_transform(G)
    dt = time.perf_counter() - t0
    Ki = as_int(K)
    i, j, v = sample_entries(Ki, 10000, 321)
    np.savez_compressed(
        os.path.join(HERE, "nci1_like_sp_%d.npz" % N),
        K_sum=np.array([Ki.sum()], np.int64), K_max=np.array([Ki.max()], np.int64),
        n_features=np.array([len(sp._enum)], np.int64), diag=np.diagonal(Ki).astype(np.int64),
        K_block=Ki[:64, :64].astype(np.int64), row_sums=Ki.sum(axis=1).astype(np.int64),
        samp_i=i, samp_j=j, samp_v=v.astype(np.int64), ref_seconds=np.array([dt]))
    print("NCI1-like SP N=%d ref %.2fs sum %d max %d nfeat %d" % (N, dt, Ki.sum(), Ki.max(),
                                                                  len(sp._enum)))


def round3():
    """Round-3 fixtures: WeisfeilerLehman over the EdgeHistogram base kernel, and more than 48 WL levels (the level
    chunking of grakel_amd), both on MUTAG."""
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "grakel", "tests", "data"))
    try:
        G = read_data('MUTAG', with_classes=True).data
    finally:
        os.chdir(cwd)
    out = {}
    wl = WeisfeilerLehman(n_iter=3, base_graph_kernel=EdgeHistogram)
    out["wleh_fit"] = as_int(wl.fit_transform(G[:100]))
    out["wleh_tr"] = as_int(wl.transform(G[100:130]))
    wln = WeisfeilerLehman(n_iter=2, base_graph_kernel=(EdgeHistogram, {}), normalize=True)
    out["wleh_fit_norm"] = wln.fit_transform(G[:100])
    out["wleh_tr_norm"] = wln.transform(G[100:130])
    deep = WeisfeilerLehman(n_iter=55)
    out["deep_fit"] = as_int(deep.fit_transform(G[:40]))
    out["deep_tr"] = as_int(deep.transform(G[40:52]))
    deepn = WeisfeilerLehman(n_iter=50, normalize=True)
    out["deep_fit_norm"] = deepn.fit_transform(G[:40])
    out["deep_tr_norm"] = deepn.transform(G[40:52])
    np.savez_compressed(os.path.join(HERE, "round3.npz"), **out)
    print("round3.npz", {k: v.shape for k, v in out.items()})


def sp_large_unit():
    """sp_large_unit.npz: ShortestPath of the real reference on unit-weight graphs above 128 vertices (directed, hubs,
    unreachable pairs, a 300-vertex path) -- what the breadth-first search of round 5 has to reproduce."""
    t0 = time.time()
    G = sp_large_unit_graphs()
    P = sp_large_unit_paths()
    out = {"K": as_int(ShortestPath().fit_transform(G)),
           "K_nolabels": as_int(ShortestPath(with_labels=False).fit_transform(G)),
           "K_paths": as_int(ShortestPath().fit_transform(P))}
    sp = ShortestPath()
    sp.fit(G[:3])
    out["K_tr"] = as_int(sp.transform(G[3:] + P[2:]))
    out["ref_seconds"] = np.array([time.time() - t0])
    np.savez_compressed(os.path.join(HERE, "sp_large_unit.npz"), **out)
    print("sp_large_unit: sums", int(out["K"].sum()), int(out["K_nolabels"].sum()), int(out["K_paths"].sum()), int(out["K_tr"].sum()),
          "in %.1f s" % (time.time() - t0))


def published_like(only=None):
    """Round-5 fixtures: stand-ins for the TU datasets the reference publishes its running times on
    (grakel_amd/synthetic.py: PUBLISHED_LIKE; doc/benchmarks/evaluation.rst:19-73).  WL-subtree h=5 on the FULL sets
    through the real reference, ShortestPath (labels, adjacency input -> its Floyd-Warshall) on a leading subsample
    (the full sets take the reference an hour and more: evaluation.rst:25,69)."""
    from grakel_amd import synthetic as S
    sp_sample = {"nci1": 0, "dd": 40, "reddit": 24, "collab": 64}
    for name, (gen, pub) in S.PUBLISHED_LIKE.items():
        if only and name not in only:
            continue
        graphs = gen()
        G = S.as_grakel(graphs)
        t0 = time.perf_counter()
        wl = WeisfeilerLehman(n_iter=5)
        K = wl.fit_transform(G)
        dt = time.perf_counter() - t0
        Ki = as_int(K)
        del K
        i, j, v = sample_entries(Ki, 20000, 123)
        out = dict(n_graphs=np.array([len(G)], np.int64),
                   label_counts=np.array([len(wl._inv_labels[k]) for k in range(6)], np.int64),
                   K_sum=np.array([Ki.sum()], np.int64), K_trace=np.array([np.trace(Ki)], np.int64),
                   K_max=np.array([Ki.max()], np.int64), diag=np.diagonal(Ki).astype(np.int64),
                   K_block=Ki[:64, :64].astype(np.int64), row_sums=Ki.sum(axis=1).astype(np.int64),
                   samp_i=i, samp_j=j, samp_v=v.astype(np.int64), ref_seconds=np.array([dt]))
        print("published-like", name, "WL h=5 ref %.1fs" % dt, "sum", Ki.sum(), "trace", np.trace(Ki), "max", Ki.max(),
              "counts", out["label_counts"].tolist(), flush=True)
        # a transform block: the last 20 graphs against a fit on the 200 before them
        wl2 = WeisfeilerLehman(n_iter=5)
        wl2.fit(G[-220:-20])
        out["tr_block"] = as_int(wl2.transform(G[-20:]))
        del Ki
        m = sp_sample[name]
        if m:
            sub = [g for g in graphs[:4 * m] if g[0] <= 700][:m]
            out["sp_index"] = np.array([k for k, g in enumerate(graphs[:4 * m]) if g[0] <= 700][:m], np.int64)
            Ga = S.as_grakel(sub, adjacency=True)
            t0 = time.perf_counter()
            sp = ShortestPath()
            Ks = sp.fit_transform(Ga)
            sdt = time.perf_counter() - t0
            out["sp_K"] = as_int(Ks)
            out["sp_n_features"] = np.array([len(sp._enum)], np.int64)
            out["sp_ref_seconds"] = np.array([sdt])
            print("   SP on", len(sub), "graphs ref %.1fs" % sdt, "sum", out["sp_K"].sum(), "features", len(sp._enum), flush=True)
        np.savez_compressed(os.path.join(HERE, "pub_%s.npz" % name), **out)


def callers():
    """callers.npz (round 6): the frameworks SURVEY 8b names as CALLERS of the path, run by the real reference on small
    sets -- HadamardCode (default base VertexHistogram, hadamard_code.py:19,189; and over ShortestPath), so that a test can
    drive the accelerated base classes through the same per-level calls and compare matrices."""
    from grakel import HadamardCode
    out = dict()
    for name in ("adj_u", "dense_big"):
        kw = dict(SMALL_SETS)[name]
        G = random_labelled_graphs(**kw)
        tr, te = split(G)
        hc = HadamardCode(n_iter=3)
        out[name + "/hc_vh_fit"] = as_int(hc.fit_transform(tr))
        out[name + "/hc_vh_tr"] = as_int(hc.transform(te))
        hcn = HadamardCode(n_iter=2, normalize=True)
        out[name + "/hc_vh_norm_fit"] = hcn.fit_transform(tr)
        out[name + "/hc_vh_norm_tr"] = hcn.transform(te)
        hs = HadamardCode(n_iter=2, base_graph_kernel=ShortestPath)
        out[name + "/hc_sp_fit"] = as_int(hs.fit_transform(tr))
        out[name + "/hc_sp_tr"] = as_int(hs.transform(te))
        print("callers", name, "HadamardCode sums", out[name + "/hc_vh_fit"].sum(), out[name + "/hc_vh_tr"].sum(),
              out[name + "/hc_sp_fit"].sum(), out[name + "/hc_sp_tr"].sum())
    np.savez_compressed(os.path.join(HERE, "callers.npz"), **out)


def sp_big_selection(name, graphs):
    """Which graphs of a published-like set the REAL reference runs ShortestPath on for pub_<set>_sp_big.npz: the
    largest ones (D&D-like: the 5 748-vertex giant and the runner-up; REDDIT-like: the ten largest threads) next to a
    handful of small ones -- sizes the round-5 goldens (at most 700 vertices) never reached."""
    sizes = np.array([g[0] for g in graphs])
    order = np.argsort(-sizes, kind="stable")
    big = order[:2] if name == "dd" else order[:10]
    small = [k for k in range(len(graphs)) if sizes[k] <= 300 and k not in set(big.tolist())][:6]
    return np.array(sorted(big.tolist()) + small, np.int64)


def published_like_sp_big(only=None):
    """Round-6 fixtures pub_<set>_sp_big.npz: ShortestPath of the REAL reference (adjacency input -> its Floyd-Warshall,
    graph.py:1767-1794: n^2 numpy row operations per graph -- minutes per graph of thousands of vertices; pair walk
    shortest_path.py:468-490) on the largest graphs of the D&D-like and REDDIT-like sets against a few small ones."""
    from grakel_amd import synthetic as S
    for name in ("dd", "reddit"):
        if only and name not in only:
            continue
        graphs = S.PUBLISHED_LIKE[name][0]()
        idx = sp_big_selection(name, graphs)
        sub = [graphs[k] for k in idx.tolist()]
        print("published-like", name, "ShortestPath through the real reference on graphs", idx.tolist(),
              "of", [g[0] for g in sub], "vertices", flush=True)
        Ga = S.as_grakel(sub, adjacency=True)
        t0 = time.perf_counter()
        sp = ShortestPath()
        Ks = sp.fit_transform(Ga)
        dt = time.perf_counter() - t0
        out = dict(index=idx, sizes=np.array([g[0] for g in sub], np.int64), K=as_int(Ks),
                   n_features=np.array([len(sp._enum)], np.int64), ref_seconds=np.array([dt]))
        spn = ShortestPath(normalize=True)
        spn.fit(Ga[-6:])
        out["Kn_tr"] = spn.transform(Ga[:1])            # the largest-index big graph against the small ones, normalised
        np.savez_compressed(os.path.join(HERE, "pub_%s_sp_big.npz" % name), **out)
        print("   ref %.1fs" % dt, "sum", out["K"].sum(), "features", len(sp._enum), flush=True)


def published_like_sp_full(only=None):
    """Round-6 fixtures pub_<set>_sp_full.npz: ShortestPath on the FULL published-like sets.  The real reference needs
    hours for these (evaluation.rst:25,69), so the numbers come from oracle/sp_fast.py -- the vectorised restatement that
    tests/test_oracle.py pins to the real reference's matrices (pub_<set>.npz[sp_K], pub_<set>_sp_big.npz) and to the
    literal oracle.  Stored: checksums, diagonal, row sums, a corner, 20 000 sampled entries, feature / pair counts."""
    from grakel_amd import synthetic as S
    from oracle import sp_fast
    for name in ("dd", "reddit", "collab"):
        if only and name not in only:
            continue
        graphs = S.PUBLISHED_LIKE[name][0]()
        t0 = time.perf_counter()
        K, nf, pairs = sp_fast.sp_unit_gram(graphs, progress=200)
        dt = time.perf_counter() - t0
        i, j, v = sample_entries(K, 20000, 321)
        out = dict(n_graphs=np.array([len(graphs)], np.int64), n_features=np.array([nf], np.int64),
                   n_pairs=np.array([pairs], np.int64), K_sum=np.array([K.sum()], np.int64),
                   K_trace=np.array([np.trace(K)], np.int64), K_max=np.array([K.max()], np.int64),
                   diag=np.diagonal(K).copy(), row_sums=K.sum(axis=1), K_block=K[:64, :64].copy(),
                   samp_i=i, samp_j=j, samp_v=v.astype(np.int64), oracle_seconds=np.array([dt]))
        big = os.path.join(HERE, "pub_%s_sp_big.npz" % name)
        if os.path.exists(big):                               # the real reference's block of the same matrix
            zb = np.load(big)
            ix = zb["index"]
            assert np.array_equal(K[np.ix_(ix, ix)], zb["K"]), "sp_fast differs from the real reference on " + name
            print("   equal to the real reference on the %d x %d block of its largest graphs" % (len(ix), len(ix)))
        np.savez_compressed(os.path.join(HERE, "pub_%s_sp_full.npz" % name), **out)
        print("published-like", name, "ShortestPath full set (oracle/sp_fast.py, %.0f s): sum" % dt, K.sum(), "trace",
              np.trace(K), "max", K.max(), "features", nf, "pairs", pairs, flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-big", action="store_true", help="skip config 3 (~100 s) and NCI1-4110")
    ap.add_argument("--only-state", action="store_true", help="only the fitted-state fixture (mutag_state.npz)")
    ap.add_argument("--only-dyadic", action="store_true", help="only the float-weight ShortestPath fixture (sp_dyadic.npz)")
    ap.add_argument("--only-published", nargs="*", default=None, metavar="SET",
                    help="only the round-5 published-dataset stand-ins (pub_<set>.npz); no names = all four")
    ap.add_argument("--only-round3", action="store_true", help="only round3.npz (WL over EdgeHistogram, more than 48 levels)")
    ap.add_argument("--only-float", action="store_true", help="only sp_float.npz (ShortestPath on general float edge weights)")
    ap.add_argument("--only-float-big", action="store_true", help="only sp_float_big.npz (general float weights above 143 vertices, CoreFramework)")
    ap.add_argument("--only-large-unit", action="store_true", help="only sp_large_unit.npz (unit weights above 128 vertices: directed, hubs, a long path)")
    ap.add_argument("--only-sp-big", nargs="*", default=None, metavar="SET",
                    help="only pub_<set>_sp_big.npz (round 6: the real reference's ShortestPath on the largest D&D-/REDDIT-like graphs)")
    ap.add_argument("--only-sp-full", nargs="*", default=None, metavar="SET",
                    help="only pub_<set>_sp_full.npz (round 6: full-set ShortestPath checksums from oracle/sp_fast.py)")
    ap.add_argument("--only-callers", action="store_true", help="only callers.npz (round 6: HadamardCode driving the base kernels)")
    a = ap.parse_args()
    print("reference grakel", grakel.__version__, "from", os.path.dirname(grakel.__file__))
    if a.only_callers:
        callers()
        sys.exit(0)
    if a.only_sp_big is not None:
        published_like_sp_big(a.only_sp_big or None)
        sys.exit(0)
    if a.only_sp_full is not None:
        published_like_sp_full(a.only_sp_full or None)
        sys.exit(0)
    if a.only_published is not None:
        published_like(a.only_published or None)
        sys.exit(0)
    if a.only_round3:
        round3()
        sys.exit(0)
    if a.only_float:
        sp_float()
        sys.exit(0)
    if a.only_float_big:
        sp_float_big()
        sys.exit(0)
    if a.only_large_unit:
        sp_large_unit()
        sys.exit(0)
    sp_dyadic()
    if a.only_dyadic:
        sys.exit(0)
    mutag_state()
    if a.only_state:
        sys.exit(0)
    doc_goldens()
    mutag()
    small_sets()
    er_config("n200", 200, 30, 0.1, 4, 3, 4, 4000)
    er_config("config2", 1000, 50, 0.1, 5, 0, 3, 20000)
    nci1_sp(300)
    if not a.skip_big:
        nci1_sp(4110)
        er_config("config3", 10000, 100, 0.05, 5, 0, 5, 20000, with_oa=False)
    round3()
    sp_float()
    sp_float_big()
    sp_large_unit()

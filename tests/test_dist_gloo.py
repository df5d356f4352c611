"""Multi-process CPU test (gloo, world_size 2) of the N>1 path's host/collective logic:
shard the graphs, all-gather the packed CSR shards, rebuild the global batch on every rank,
and partition the Gram rows.  The device kernels themselves are covered by the -m gpu tests;
here each rank checks with the CPU oracle that its row block of the oracle Gram, computed
from the batch it rebuilt after the all-gather, is the right slice of the global matrix."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch.distributed as dist
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.dist import all_gather_batch, shard_bounds, tensors_to_batch
    from grakel_amd.synthetic import random_labelled_graphs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = random_labelled_graphs(23, 3, 12, 0.35, 3, 77)          # ragged, not divisible by 2
        full, _ = wl_batch_from_input(X)
        b = shard_bounds(full.n_graphs, world)
        assert b[0] == 0 and b[-1] == full.n_graphs and all(b[i] < b[i + 1] for i in range(world))
        local = full.slice_graphs(b[rank], b[rank + 1])               # this rank only holds its shard
        gp, rp, ci, lab, n_labels, bounds = all_gather_batch(local, None, None)
        assert bounds == b
        g = tensors_to_batch(gp, rp, ci, lab, n_labels)
        ok = (np.array_equal(g.graph_ptr, full.graph_ptr) and np.array_equal(g.row_ptr, full.row_ptr)
              and np.array_equal(g.col_idx, full.col_idx) and np.array_equal(g.node_label, full.node_label))
        # the fused form the GPU path uses: ONE buffer holding the ranks' messages back to back
        # ([graph sizes | degrees | labels | col_idx], each padded to the largest shard) -- rebuilt here
        # with numpy exactly as gk_batch_from_shards does on the device
        from grakel_amd.dist import ShardExchange
        ex = ShardExchange(local)
        flat = ex.gather_flat().numpy()
        stride = ex.mg + 2 * ex.mv + ex.me
        # the C ABI's own form of this rank's message (csrc/comm.hip: gk_shard_message, what gk_batch_allgather sends over
        # RCCL) is word for word the message this exchange sent -- a host function, no device
        from ctypes import c_void_p
        from grakel_amd import _lib
        cmsg = np.full(stride, -1, dtype=np.int32)

        def ptr(a):
            return np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(c_void_p)
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (local.graph_ptr, local.row_ptr, local.col_idx, local.node_label)]
        _lib.check(_lib.load().gk_shard_message(local.n_graphs, local.n_nodes, local.n_edges, *[a.ctypes.data_as(c_void_p) for a in arrs],
                                                ex.mg, ex.mv, ex.me, cmsg.ctypes.data_as(c_void_p)))
        ok = ok and np.array_equal(cmsg, ex.msg.numpy()) and np.array_equal(cmsg, flat[rank * stride:(rank + 1) * stride])
        gs, dg, lb, cc, v0 = [], [], [], [], 0
        for r in range(world):
            ng, nv, ne = (int(x) for x in ex.all_sizes[r, :3])
            m = flat[r * stride:(r + 1) * stride]
            gs.append(m[:ng]), dg.append(m[ex.mg:ex.mg + nv]), lb.append(m[ex.mg + ex.mv:ex.mg + ex.mv + nv])
            cc.append(m[ex.mg + 2 * ex.mv:ex.mg + 2 * ex.mv + ne] + v0)
            v0 += nv
        ok = ok and (np.array_equal(np.concatenate([[0], np.cumsum(np.concatenate(gs))]), full.graph_ptr)
                     and np.array_equal(np.concatenate([[0], np.cumsum(np.concatenate(dg))]), full.row_ptr)
                     and np.array_equal(np.concatenate(lb), full.node_label)
                     and np.array_equal(np.concatenate(cc), full.col_idx))
        np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([int(ok), b[rank], b[rank + 1]]))
    finally:
        dist.destroy_process_group()


def _worker_own_ingestion(rank, world, port, out_dir):
    """Each rank ingests ONLY its own graphs (shard-local level-0 ids, different label sets per rank)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.dist import all_gather_batch, shard_bounds, tensors_to_batch
    from grakel_amd.synthetic import random_labelled_graphs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = random_labelled_graphs(16, 3, 9, 0.4, 6, 5)
        for i, x in enumerate(X):                        # label values that only occur in one half
            x[1] = {v: "shared" if l % 3 == 0 else ("a%d" % l if i < 8 else "b%d" % (l % 2)) for v, l in x[1].items()}
        full, full_map = wl_batch_from_input(X)
        b = shard_bounds(len(X), world)
        local, local_map = wl_batch_from_input(X[b[rank]:b[rank + 1]])
        assert len(local_map) < len(full_map)            # the shard does not see every label
        gp, rp, ci, lab, n_labels, _ = all_gather_batch(local, None, None, label_map=local_map)
        g = tensors_to_batch(gp, rp, ci, lab, n_labels)
        ok = (n_labels == len(full_map) and np.array_equal(g.node_label, full.node_label)
              and np.array_equal(g.col_idx, full.col_idx) and np.array_equal(g.graph_ptr, full.graph_ptr))
        # without the map the ids collide: the same call must NOT reproduce the global labels
        _, _, _, lab_bad, _, _ = all_gather_batch(local, None, None)
        ok = ok and not np.array_equal(lab_bad.numpy(), full.node_label)
        np.save(os.path.join(out_dir, "own_%d.npy" % rank), np.array([int(ok)]))
    finally:
        dist.destroy_process_group()


def _worker_weights(rank, world, port, out_dir):
    """ShardedSP's exchange: edge weights travel with the shards (one rank without weights sends ones)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from grakel_amd.batch import GraphBatch, sp_batch_from_input
    from grakel_amd.dist import ShardExchange, gram_plan, shard_bounds
    from grakel_amd.synthetic import random_labelled_graphs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G = random_labelled_graphs(17, 3, 10, 0.4, 3, 9, fmt="adj")
        rs = np.random.RandomState(1)
        for g in G[:9]:                                   # only the first shard's graphs carry weights 1..4
            W = np.triu(rs.randint(1, 5, g[0].shape), 1)
            g[0] = g[0] * (W + W.T)
        full, _ = sp_batch_from_input(G, True)
        b = shard_bounds(len(G), world)
        assert b[1] == 9
        local = full.slice_graphs(b[rank], b[rank + 1])
        if rank == 1:                                      # a shard of unit-weight graphs has no weight array at all
            local = GraphBatch(local.graph_ptr, local.row_ptr, local.col_idx, local.node_label, local.n_labels, None)
        ex = ShardExchange(local)
        w = ex.gather_weights()
        ok = w is not None and np.array_equal(w, full.edge_weight)
        # no rank has weights -> no weight exchange at all
        plain = GraphBatch(local.graph_ptr, local.row_ptr, local.col_idx, local.node_label, local.n_labels, None)
        full1, _ = sp_batch_from_input([[np.minimum(g[0], 1), g[1]] for g in G], True)
        l1 = full1.slice_graphs(b[rank], b[rank + 1])
        ok = ok and ShardExchange(GraphBatch(l1.graph_ptr, l1.row_ptr, l1.col_idx, l1.node_label, l1.n_labels, None)).gather_weights() is None
        del plain
        # general float weights on ONE rank: every rank sends float64 weights (its exact integer weights as floats) and
        # the per-graph dictionary flags; the gathered arrays are those of the whole batch ingested at once
        Gf = [[g[0] * 1.0, g[1]] for g in G]
        for g in Gf[9:]:
            g[0] = g[0] * 0.1
        Gf[12] = [{i: {j: float(Gf[12][0][i, j]) for j in range(Gf[12][0].shape[0]) if Gf[12][0][i, j] > 0}
                   for i in range(Gf[12][0].shape[0])}, Gf[12][1]]
        fullf, _ = sp_batch_from_input(Gf, True)
        assert fullf.float_weight is not None and fullf.from_dict[12] == 1
        lf, _ = sp_batch_from_input(Gf[b[rank]:b[rank + 1]], True, fitted_labels=None)
        lf = GraphBatch(lf.graph_ptr, lf.row_ptr, lf.col_idx, fullf.slice_graphs(b[rank], b[rank + 1]).node_label, fullf.n_labels,
                        lf.edge_weight, lf.weight_step, lf.float_weight, lf.from_dict)
        assert (lf.float_weight is None) == (rank == 0)              # rank 0's shard has integer weights only
        exf = ShardExchange(lf)
        fw, fd = exf.gather_float_weights()
        ok = ok and exf.gather_weights() is None and np.array_equal(fw, fullf.float_weight) and np.array_equal(fd, fullf.from_dict)
        # shards ingested ON THEIR OWN, with different weight units: rank 0's weights are multiples of 0.5, rank 1's are
        # integers -- each shard's quantisation picks its own step (0.5 and 1.0), the exchange must re-express both in
        # the finer one, or a distance of 0.5 and a distance of 1.0 would be the same integer
        Gs = [[np.minimum(g[0], 1) * (0.5 if i < 9 else 1.0), g[1]] for i, g in enumerate(G)]
        fulls, _ = sp_batch_from_input(Gs, True)
        ls, _ = sp_batch_from_input(Gs[b[rank]:b[rank + 1]], True)
        assert ls.weight_step == (0.5 if rank == 0 else 1.0) and fulls.weight_step == 0.5
        ls = GraphBatch(ls.graph_ptr, ls.row_ptr, ls.col_idx, fulls.slice_graphs(b[rank], b[rank + 1]).node_label, fulls.n_labels,
                        ls.edge_weight, ls.weight_step)
        exs = ShardExchange(ls)
        ws = exs.gather_weights()
        ok = ok and exs.weight_step == 0.5 and np.array_equal(ws, fulls.edge_weight)
        np.save(os.path.join(out_dir, "w_%d.npy" % rank), np.array([int(ok)]))
    finally:
        dist.destroy_process_group()


def test_edge_weights_travel_with_the_shards_world2(tmp_path):
    import torch.multiprocessing as mp
    port = 33000 + (os.getpid() % 2000)
    mp.spawn(_worker_weights, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(os.path.join(str(tmp_path), "w_%d.npy" % r)).tolist() == [1]


def test_gram_plan_model_prefers_plain_row_blocks_while_the_kernel_is_store_bound():
    """The bytes model behind the default plan: with the measured store rate of the Gram kernel the
    symmetric plan loses at every world size; it would only win if a rank produced entries slower than xGMI
    delivers them."""
    from grakel_amd.dist import gram_plan
    for N in (10000, 50000):
        for R in (2, 4, 8):
            m = gram_plan(N, R)
            assert m["choice"] == "plain" and m["plain"]["xgmi_recv_bytes"] == 0
            assert abs(m["plain"]["hbm_store_bytes"] - 8.0 * N * N / R) < 1
            assert m["symmetric"]["xgmi_recv_bytes"] >= 0.25 * m["plain"]["hbm_store_bytes"]
    assert gram_plan(50000, 8, store_bps=20e9)["choice"] == "symmetric"       # a (hypothetical) 20 GB/s producer
    assert gram_plan(10000, 1)["choice"] == "plain"


def test_independently_ingested_shards_get_global_label_ids_world2(tmp_path):
    import torch.multiprocessing as mp
    port = 31000 + (os.getpid() % 2000)
    mp.spawn(_worker_own_ingestion, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.load(os.path.join(str(tmp_path), "own_%d.npy" % r)).tolist() == [1]


def test_symmetric_plan_covers_every_entry_exactly_once():
    """Blocks a rank multiplies + mirrored blocks it receives tile its row block exactly; every unordered pair of
    row blocks is multiplied once; the multiplied entries are balanced (1/R of the upper triangle each)."""
    from grakel_amd.dist import shard_bounds, symmetric_plan
    for R in range(1, 10):
        for N in (R, 37, 100, 257):
            if N < R:
                continue
            b = shard_bounds(N, R)
            owner_count = np.zeros((N, N), np.int32)          # how often entry (i, j) is produced for row i's owner
            multiplied = np.zeros((N, N), np.int32)
            work = []
            for r in range(R):
                compute, recv = symmetric_plan(b, r)
                w = 0
                for (r0, r1, c0, c1, peer) in compute:
                    assert b[r] <= r0 <= r1 <= b[r + 1]      # rows inside the rank's own block
                    owner_count[r0:r1, c0:c1] += 1
                    if peer < 0:                              # diagonal block: tiles on/above the diagonal only
                        multiplied[r0:r1, c0:c1] += np.triu(np.ones((r1 - r0, c1 - c0), np.int32))
                    else:
                        multiplied[r0:r1, c0:c1] += 1
                    w += (r1 - r0) * (r1 - r0 + 1) // 2 if peer < 0 else (r1 - r0) * (c1 - c0)
                    if peer >= 0:
                        assert (r, r0, r1, c0, c1) in symmetric_plan(b, peer)[1]   # the peer expects exactly this block
                for (peer, r0, r1, c0, c1) in recv:           # stored transposed: rows c0:c1 (mine), columns r0:r1
                    assert b[r] <= c0 <= c1 <= b[r + 1]
                    owner_count[c0:c1, r0:r1] += 1
                work.append(w)
            assert np.all(owner_count == 1)
            sym = multiplied + multiplied.T
            np.fill_diagonal(sym, np.diagonal(multiplied))
            assert np.all(sym == 1)                           # each unordered pair of graphs multiplied once
            if N % (2 * R) == 0:
                assert max(work) - min(work) <= N // R        # balanced up to a block edge


def test_shard_bounds():
    from grakel_amd.dist import shard_bounds
    assert shard_bounds(10, 4) == [0, 3, 6, 8, 10]
    assert shard_bounds(8, 8) == list(range(9))
    assert shard_bounds(10000, 8)[-1] == 10000


def test_all_gather_rebuilds_the_global_batch_world2(tmp_path):
    import torch.multiprocessing as mp
    port = 29000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    rows = []
    for r in range(2):
        ok, lo, hi = np.load(os.path.join(str(tmp_path), "ok_%d.npy" % r)).tolist()
        assert ok == 1
        rows.append((lo, hi))
    assert rows[0][0] == 0 and rows[0][1] == rows[1][0] and rows[1][1] == 23   # rows partition K


def _rows_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from grakel_amd.dist import all_gather_rows, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = []
        for n, width in ((301, 96), (300, 128), (7, 16), (2, 16)):          # ragged, even, tiny shards
            b = shard_bounds(n, world)
            whole = (torch.arange(n * width, dtype=torch.int64).reshape(n, width) % 251).to(torch.uint8)
            got = all_gather_rows(whole[b[rank]:b[rank + 1]].clone(), b)
            out.append(bool(torch.equal(got, whole)) and tuple(got.shape) == (n, width))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_operand_row_all_gather_world2():
    """SURVEY 8e / north_star: "RCCL all-gather of per-graph feature vectors" -- grakel_amd.dist.all_gather_rows (the
    exchange="phi" payload of ShardedWL) over gloo with two ranks: row shards of unequal height are padded for the
    collective and come back as the whole operand, byte for byte, on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_rows_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert got[0] == [True] * 4 and got[1] == [True] * 4

"""Experiment: two processes, ONE GPU, the C-ABI collectives (RCCL normally refuses two ranks on one device)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multiprocessing as mp


def worker(rank, world, path):
    from grakel_amd.engine import get_engine
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.dist import shard_bounds
    from grakel_amd.synthetic import er_dataset
    eng = get_engine()
    if rank == 0:
        uid = eng.comm_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(path + ".tmp", path)
    else:
        while not os.path.exists(path):
            time.sleep(0.05)
        uid = open(path, "rb").read()
    try:
        comm = eng.comm_init(rank, world, uid)
    except Exception as e:
        print("rank", rank, "comm_init failed:", e, flush=True)
        return
    full, _ = wl_batch_from_input(er_dataset(301, 30, 0.1, 4, 5))
    b = shard_bounds(full.n_graphs, world)
    db, bounds = eng.batch_allgather(comm, full.slice_graphs(b[rank], b[rank + 1]))
    eng.wl_relabel(db, 3)
    feat = eng.features(db, 4)
    K = eng.gram_sharded(comm, feat, bounds, 0)
    print("rank", rank, "rows", bounds.tolist(), K.shape, K.sum(), flush=True)


if __name__ == "__main__":
    path = "/tmp/gk_uid_%d" % os.getpid()
    ps = [mp.get_context("spawn").Process(target=worker, args=(r, 2, path)) for r in range(2)]
    [p.start() for p in ps]
    for p in ps:
        p.join(120)
        if p.is_alive():
            p.terminate()
            print("timeout: terminated", p.pid)

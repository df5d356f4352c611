"""Cost of the multi-GPU code path at world_size 1 (RCCL group of one rank) against the plain path:
python tools/bench_sharded_1rank.py   -- config 3, HIP-event time per step."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29581")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from grakel_amd import GraphBatch                      # noqa: E402
from grakel_amd.dist import ShardedWL                  # noqa: E402
from grakel_amd.engine import get_engine               # noqa: E402
from grakel_amd.synthetic import er_dataset_csr        # noqa: E402

eng = get_engine()
gb = GraphBatch(*er_dataset_csr(10000, 100, 0.05, 5, 0), 5)
sw = ShardedWL(eng, n_iter=5)
import time
for it in range(5):
    sw.step(gb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(20):
    sw.step(gb)
torch.cuda.synchronize()
t_sh = (time.perf_counter() - t0) / 20 * 1e3
db = eng.upload(gb)
for it in range(5):
    eng.wl_relabel(db, 5); f = eng.features(db, 6); eng.gram(f, 0, to_host=False); f.close()
eng.synchronize()
t0 = time.perf_counter()
for it in range(20):
    eng.wl_relabel(db, 5); f = eng.features(db, 6); eng.gram(f, 0, to_host=False); f.close()
eng.synchronize()
t_pl = (time.perf_counter() - t0) / 20 * 1e3
print("sharded path (1 rank) %.3f ms/step, plain path %.3f ms/step" % (t_sh, t_pl))
dist.destroy_process_group()

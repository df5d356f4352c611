import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1] if len(sys.argv) > 1 else "lib_first"
import numpy as np
if order == "lib_first":
    from grakel_amd.engine import get_engine
    eng = get_engine()
import torch
print("torch", torch.__version__, "cuda avail", torch.cuda.is_available(), "count", torch.cuda.device_count())
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
try:
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda"); dist.all_reduce(t); print("allreduce ok", t.tolist())
    from grakel_amd.engine import get_engine
    from grakel_amd.batch import wl_batch_from_input
    from grakel_amd.dist import ShardedWL
    from grakel_amd.synthetic import er_dataset
    import grakel_amd as gk
    X = er_dataset(300, 30, 0.1, 4, 5)
    K = gk.WeisfeilerLehman(n_iter=3).fit_transform(X)
    gb, _ = wl_batch_from_input(X)
    Kr, info = ShardedWL(get_engine(), n_iter=3).step(gb, to_host=True)
    print("sharded == plain:", np.array_equal(Kr, K), info["rows"])
    dist.destroy_process_group()
except Exception as e:
    import traceback; traceback.print_exc()

"""Multi-GPU path: one process per GPU, graphs sharded across ranks, Gram rows sharded too.

The path shards naturally with ONE exchange step: rank r owns the graphs
[lo_r, hi_r) -- it ingests/packs only those and holds only their CSR shard in HBM -- but WL
labels are a *global* dictionary, so before relabelling every rank needs every graph's
level-0 features (labels + adjacency).  The level-0 ids must mean the same label on every rank:
a shard cut out of a globally ingested batch (``GraphBatch.slice_graphs``) already has global ids; a
shard ingested on its own (``wl_batch_from_input`` on the rank's graphs) has shard-local ids and MUST be
passed together with its ``label_map`` -- the ranks then exchange their distinct label values once and
remap to the ids of the sorted union, exactly the reference's ``sorted(distinct_values)`` numbering
(weisfeiler_lehman.py:199-210).  The exchange is therefore a single RCCL
``all_gather`` of the packed shards over xGMI (config 3: ~25 MB in total, i.e. ~3 MB per
rank; a direct all-gather at ~77 GB/s per link and direction takes tens of microseconds), after which

    relabel + label-count features : replicated on every rank (config 5: 1.06 ms of a 7.7 ms step)
    Gram                           : rank r computes and stores its row block K[lo_r:hi_r, :] -- PLAIN ROW BLOCKS,
                                     no data-path collective after the all-gather.

Why not the symmetric plan (multiply half, ship the mirrored blocks): the Gram kernel is bound by the float64
STORE of K (3 TB/s measured, 17 % MFMA-busy), so the multiply-adds a rank would save are free, while the blocks
it would receive instead arrive over xGMI at a small fraction of the rate at which it can produce them locally --
``gram_plan`` puts numbers on both plans (config 5, 8 ranks: 1.1 ms plain against ~5.7 ms symmetric).  The
symmetric plan stays available (``ShardedWL(symmetric=True)``) for a future MFMA-bound operand.

``ShardedSP`` is the same scheme for the ShortestPath kernel ("identical scheme; the ``_enum``
dictionary is global", shortest_path.py:370-499): all-gather of the CSR shards (+ edge weights), all-pairs
distances, pair dictionary and features replicated, Gram rows sharded.

``torch.distributed`` is plumbing only: backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests of the
shard/gather/rebuild logic.
"""
import numpy as np

from .batch import GraphBatch, MAX_EDGE_WEIGHT


def shard_bounds(n_graphs, world_size):
    """Contiguous, balanced graph ranges; rank r owns [b[r], b[r+1])."""
    base, rem = divmod(n_graphs, world_size)
    b = [0]
    for r in range(world_size):
        b.append(b[-1] + base + (1 if r < rem else 0))
    return b


# measured on one MI355X: float64 store rate of the Gram kernel in ROW-BLOCK mode (a rank's rows x all columns: no tile is
# shared with its mirror image, so the operand stream per output byte is twice the symmetric job's).  Round 6, strip-walk tile
# order: 3.3 TB/s on config 6's 25 000 x 200 000 sub-blocks, 4.1 on 12 500 x 100 000, 4.6 on 6 250 x 50 000
# (profiles/r06_strip_sweep.txt; round 5: 2.2-2.3 / 3.6 / 3.9) -- the model uses the lowest of them,
# device-to-device transposed placement (read + write, profiles/r03_*), and the xGMI link rate per direction
# (MI355X_MICROARCH.md: 7 links x 153.6 GB/s bidirectional, point to point -- a rank talking to p peers uses p links)
GRAM_STORE_BPS = 3.3e12
PLACE_BPS = 1.5e12
XGMI_LINK_BPS = 76.8e9


def gram_plan(n_graphs, world, store_bps=GRAM_STORE_BPS, link_bps=XGMI_LINK_BPS, place_bps=PLACE_BPS):
    """Bytes and predicted Gram-phase time per rank of the two sharding plans (a model from measured rates, not a
    measurement): ``plain`` = every rank multiplies and stores its whole row block; ``symmetric`` = every rank
    multiplies 1/R of the upper triangle (``symmetric_plan``), ships the mirrored blocks point to point and stores
    what it receives transposed.  Returns {"plain": {...}, "symmetric": {...}, "choice": "plain" | "symmetric"}."""
    N, R = float(n_graphs), int(world)
    block = 8.0 * (N / R) * N                                   # float64 row block of one rank
    plain = dict(hbm_store_bytes=block, xgmi_recv_bytes=0.0, seconds=block / store_bps)
    if R == 1:
        return dict(plain=plain, symmetric=dict(plain), choice="plain")
    own = block * 0.5 * (1.0 + 1.0 / R)                          # diagonal block + the blocks towards half of the peers
    recv = block - own
    peers = max(1, (R - 1 + 1) // 2)                             # senders a rank receives from, one xGMI link each
    sym = dict(hbm_store_bytes=own + recv, xgmi_recv_bytes=recv, peers=peers,
               seconds=own / store_bps + recv / (peers * link_bps) + 2.0 * recv / place_bps)
    return dict(plain=plain, symmetric=sym, choice="symmetric" if sym["seconds"] < plain["seconds"] else "plain")


def symmetric_plan(bounds, rank):
    """Which blocks of its row block rank ``rank`` multiplies itself ("symmetry can halve work").

    ``bounds``: row-block bounds of the R ranks.  Returns (compute, recv):
      compute = [(row_lo, row_hi, col_lo, col_hi, peer)]  blocks rank ``rank`` computes; ``peer`` is the rank that
                owns the mirrored block and receives a copy (-1: the diagonal block, mirrored in place);
      recv    = [(peer, row_lo, row_hi, col_lo, col_hi)]  blocks computed by ``peer`` -- rows inside ITS row block,
                columns inside this rank's -- that arrive here and are stored transposed.
    Rank r takes the diagonal block, the full blocks towards the next ceil(R/2)-1 ranks (cyclically) and, for
    even R, half of the block towards the rank R/2 away (the lower rank the first half of the columns, the
    higher rank the second half of its rows): every unordered pair of blocks is multiplied exactly once and
    every rank multiplies (N/R)^2 * R/2 entries -- 1/R of the upper triangle."""
    R = len(bounds) - 1

    def jobs(r):
        lo, hi = bounds[r], bounds[r + 1]
        out = [(lo, hi, lo, hi, -1)]
        for d in range(1, (R + 1) // 2):
            q = (r + d) % R
            out.append((lo, hi, bounds[q], bounds[q + 1], q))
        if R % 2 == 0 and R > 1:
            q = (r + R // 2) % R
            a, b = min(r, q), max(r, q)
            half = bounds[b] + (bounds[b + 1] - bounds[b]) // 2
            if r == a:        # all rows of a  x  first half of b's columns
                out.append((lo, hi, bounds[b], half, b))
            else:             # second half of b's rows  x  all columns of a
                out.append((half, hi, bounds[a], bounds[a + 1], a))
        return [j for j in out if j[1] > j[0] and j[3] > j[2]]

    compute = jobs(rank)
    recv = [(p, j[0], j[1], j[2], j[3]) for p in range(R) if p != rank for j in jobs(p) if j[4] == rank]
    return compute, recv


def reconcile_label_ids(local, label_map, group=None):
    """Shard-local level-0 ids -> ids of the sorted union of every rank's label values.

    ``label_map`` is the ``{label value: local id}`` dictionary ``wl_batch_from_input`` returned for this
    rank's shard.  Returns (node_label int32 with global ids, number of global labels).  One
    ``all_gather_object`` of the distinct values (a few hundred python objects at most)."""
    import torch.distributed as dist
    ws = dist.get_world_size(group)
    mine = sorted(label_map, key=lambda v: label_map[v])           # local id order == sorted order
    everyone = [None] * ws
    dist.all_gather_object(everyone, mine, group=group)
    union = sorted(set(v for part in everyone for v in part))
    gid = {v: i for i, v in enumerate(union)}
    lut = np.zeros(max(len(mine), 1), dtype=np.int32)
    for v, i in label_map.items():
        lut[i] = gid[v]
    return lut[local.node_label], len(union)


def _pad_to(t, n, torch):
    if t.shape[0] == n:
        return t
    out = torch.zeros(n, dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


class ShardExchange(object):
    """All-gather of the packed CSR shards.  The shard sizes are exchanged once (first call);
    every later call is ONE fused all_gather of [graph sizes | degrees | labels | col_idx] plus
    the pointer rebuild (two cumsums) on the device."""

    def __init__(self, local, group=None, device=None, label_map=None):
        import torch
        import torch.distributed as dist
        self.group, self.ws = group, dist.get_world_size(group)
        if label_map is not None:          # the shard was ingested on its own: make the level-0 ids global
            ids, n_labels = reconcile_label_ids(local, label_map, group)
            local = GraphBatch(local.graph_ptr, local.row_ptr, local.col_idx, ids, n_labels, local.edge_weight,
                               getattr(local, "weight_step", 1.0), getattr(local, "float_weight", None),
                               getattr(local, "from_dict", None))
        self.dev = torch.device("cpu") if device is None else device
        dev = self.dev

        def T(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

        sizes = torch.tensor([local.n_graphs, local.n_nodes, local.n_edges, local.n_labels],
                             dtype=torch.int64, device=dev)
        all_sizes = [torch.zeros_like(sizes) for _ in range(self.ws)]
        dist.all_gather(all_sizes, sizes, group=group)
        self.all_sizes = torch.stack(all_sizes).cpu().numpy()
        a = self.all_sizes
        self.mg, self.mv, self.me = int(a[:, 0].max()), int(a[:, 1].max()), int(a[:, 2].max())
        self.n_labels = int(a[:, 3].max())
        mg, mv, me = self.mg, self.mv, self.me
        # this rank's message lives on the device (the shard is resident in HBM)
        self.msg = torch.cat([_pad_to(T(np.diff(local.graph_ptr).astype(np.int32)), mg, torch),
                              _pad_to(T(np.diff(local.row_ptr).astype(np.int32)), mv, torch),
                              _pad_to(T(local.node_label), mv, torch),
                              _pad_to(T(local.col_idx), me, torch)])
        self.bounds = [0]
        for r in range(self.ws):
            self.bounds.append(self.bounds[-1] + int(a[r, 0]))
        self.zero = torch.zeros(1, dtype=torch.int32, device=dev)
        self.flat = None
        self._into_tensor = True
        # edge weights (ShortestPath): present on any rank -> every rank sends a (unit-filled) weight segment
        # general float weights on ANY rank (GraphBatch.float_weight): every rank then sends float64 weights -- its exact
        # integer / power-of-two-multiple weights as floats -- and, per graph, whether the element came as a dictionary
        fw_local = getattr(local, "float_weight", None)
        has_w = torch.tensor([2 if fw_local is not None else (1 if local.edge_weight is not None else 0)],
                             dtype=torch.int64, device=dev)
        dist.all_reduce(has_w, op=dist.ReduceOp.MAX, group=group)
        self.weights, self.float_weights, self.from_dict = None, None, None
        mode = int(has_w.item())
        if mode == 2:
            if fw_local is None:
                fw_local = (np.ones(local.n_edges, np.float64) if local.edge_weight is None
                            else local.edge_weight.astype(np.float64) * float(getattr(local, "weight_step", 1.0)))
            fd = getattr(local, "from_dict", None)
            self.float_weights = _pad_to(T(np.asarray(fw_local, np.float64)), me, torch)
            self.from_dict = _pad_to(T(np.asarray(fd if fd is not None else np.zeros(local.n_graphs), np.int32)), mg, torch)
        elif mode == 1:
            # every shard was quantised on its own (batch.quantise_weights): one weight unit for the whole job is the
            # finest of the ranks' steps (all are powers of two), and every rank re-expresses its integers in it --
            # otherwise a distance of 0.5 on one rank and of 1.0 on another would both travel as the integer 1
            step_local = float(getattr(local, "weight_step", 1.0)) if local.edge_weight is not None else 1.0
            st = torch.tensor([step_local], dtype=torch.float64, device=dev)
            dist.all_reduce(st, op=dist.ReduceOp.MIN, group=group)
            self.weight_step = float(st.item())
            w = (np.asarray(local.edge_weight, dtype=np.int64) if local.edge_weight is not None
                 else np.ones(local.n_edges, dtype=np.int64))
            # the ratio of two powers of two, as a Python int: the check below cannot overflow, and it runs BEFORE any
            # int64 multiply so that every rank reaches the collective (every rank raises, or none does)
            ratio = int(round(step_local / self.weight_step))
            wmax = int(w.max()) if w.size else 0
            too_big = torch.tensor([1 if ratio * wmax >= MAX_EDGE_WEIGHT else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(too_big, op=dist.ReduceOp.MAX, group=group)
            if int(too_big.item()):
                raise NotImplementedError('edge weights of the shards need more than 20 bits at their common '
                                          'power-of-two step')
            w = w * ratio
            self.weights = _pad_to(T(w.astype(np.int32)), me, torch)

    def gather_weights(self):
        """Global int32 edge-weight array (host) in the order of the gathered col_idx, or None (unit weights)."""
        import torch
        import torch.distributed as dist
        if self.weights is None:
            return None
        parts = [torch.empty_like(self.weights) for _ in range(self.ws)]
        dist.all_gather(parts, self.weights, group=self.group)
        return np.concatenate([parts[r][:int(self.all_sizes[r, 2])].cpu().numpy() for r in range(self.ws)]).astype(np.int32)

    def gather_float_weights(self):
        """(float64 edge weights, uint8 dictionary flags) of the global batch in the order of the gathered col_idx /
        graphs, or (None, None): general float weights on some rank (GraphBatch.float_weight)."""
        import torch
        import torch.distributed as dist
        if self.float_weights is None:
            return None, None
        pw = [torch.empty_like(self.float_weights) for _ in range(self.ws)]
        pf = [torch.empty_like(self.from_dict) for _ in range(self.ws)]
        dist.all_gather(pw, self.float_weights, group=self.group)
        dist.all_gather(pf, self.from_dict, group=self.group)
        w = np.concatenate([pw[r][:int(self.all_sizes[r, 2])].cpu().numpy() for r in range(self.ws)]).astype(np.float64)
        f = np.concatenate([pf[r][:int(self.all_sizes[r, 0])].cpu().numpy() for r in range(self.ws)]).astype(np.uint8)
        return w, f

    def gather_flat(self):
        """ONE collective into a preallocated buffer: the ws messages back to back (what
        ``gk_batch_from_shards`` consumes on the device)."""
        import torch
        import torch.distributed as dist
        if self.flat is None:
            self.flat = torch.empty(self.ws * self.msg.shape[0], dtype=self.msg.dtype, device=self.dev)
        if self._into_tensor:
            try:
                dist.all_gather_into_tensor(self.flat, self.msg, group=self.group)
                return self.flat
            except (RuntimeError, NotImplementedError):      # a backend without the fused form (some gloo builds)
                self._into_tensor = False
        dist.all_gather(list(self.flat.view(self.ws, -1).unbind(0)), self.msg, group=self.group)
        return self.flat

    def gather(self):
        import torch
        import torch.distributed as dist
        gathered = [torch.empty_like(self.msg) for _ in range(self.ws)]
        dist.all_gather(gathered, self.msg, group=self.group)
        mg, mv = self.mg, self.mv
        gs, dg, lb, ci = [], [], [], []
        node_off = 0
        for r in range(self.ws):
            ng, nv, ne = (int(x) for x in self.all_sizes[r, :3])
            m = gathered[r]
            gs.append(m[:ng])
            dg.append(m[mg:mg + nv])
            lb.append(m[mg + mv:mg + mv + nv])
            ci.append(m[mg + 2 * mv:mg + 2 * mv + ne] + node_off)      # local -> global node ids
            node_off += nv
        graph_ptr = torch.cat([self.zero, torch.cumsum(torch.cat(gs), 0).to(torch.int32)])
        row_ptr = torch.cat([self.zero, torch.cumsum(torch.cat(dg), 0).to(torch.int32)])
        col_idx = torch.cat(ci).to(torch.int32)
        labels = torch.cat(lb).to(torch.int32)
        return graph_ptr.contiguous(), row_ptr.contiguous(), col_idx.contiguous(), labels.contiguous()


def all_gather_batch(local, group=None, device=None, label_map=None):
    """All-gather CSR shards -> the global batch as int32 torch tensors on ``device``.

    ``local`` is this rank's ``GraphBatch`` (local node numbering); ``label_map`` as in ``ShardExchange``
    (None: the level-0 ids are already global).  Returns
    (graph_ptr, row_ptr, col_idx, node_label, n_labels, shard_graph_bounds) where the four
    arrays describe ALL graphs in rank order with global node numbering.
    """
    ex = ShardExchange(local, group, device, label_map)
    gp, rp, ci, lab = ex.gather()
    return gp, rp, ci, lab, ex.n_labels, ex.bounds


def tensors_to_batch(graph_ptr, row_ptr, col_idx, labels, n_labels):
    return GraphBatch(graph_ptr.cpu().numpy(), row_ptr.cpu().numpy(), col_idx.cpu().numpy(),
                      labels.cpu().numpy(), n_labels)


def all_gather_rows(own, bounds, group=None):
    """All-gather of ROW shards of unequal height: ``own`` is this rank's [n_own x row_bytes] uint8 tensor (rows
    bounds[rank] .. bounds[rank + 1] of the whole), the result is the [bounds[-1] x row_bytes] tensor of all ranks' rows.
    The collective wants equal shards (``all_gather_into_tensor``; RCCL on GPUs, gloo on the CPU box): every shard is padded
    to the tallest one and the padding dropped afterwards.  This is the exchange ``north_star`` names -- the per-graph feature
    vectors (operand rows of Phi) travel, not the graphs."""
    import torch
    import torch.distributed as dist
    ws = dist.get_world_size(group)
    heights = [bounds[r + 1] - bounds[r] for r in range(ws)]
    tall, width = max(heights), own.shape[1]
    send = own if own.shape[0] == tall else torch.cat([own, own.new_zeros((tall - own.shape[0], width))])
    recv = own.new_empty((ws * tall, width))
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if all(h == tall for h in heights):
        return recv
    return torch.cat([recv[r * tall:r * tall + heights[r]] for r in range(ws)])


class ShardedWL(object):
    """WL-subtree Gram with graphs and Gram rows sharded over the ranks of ``group``."""

    def __init__(self, engine, n_iter=5, normalize=False, group=None, symmetric=False, exchange="csr"):
        import torch.distributed as dist
        self.engine, self.n_iter, self.normalize, self.group = engine, n_iter, normalize, group
        # What crosses xGMI besides the packed CSR shards (which every rank needs for the global label dictionary either way):
        # "csr" (default): nothing -- every rank assembles the whole operand Phi_s itself (replicated);
        # "phi": every rank assembles the operand rows of ITS graphs only and the row shards are all-gathered -- the
        #        exchange north_star names.  Same matrix; an A/B switch for the first multi-GPU node this runs on.
        if exchange not in ("csr", "phi"):
            raise ValueError('exchange must be "csr" or "phi"')
        self.exchange = exchange
        # False (default): plain row blocks, every rank multiplies and stores its whole block, no exchange.
        # True: the symmetric plan (half the multiply-adds, mirrored blocks over xGMI) -- slower as long as the
        # Gram kernel is store-bound (``gram_plan``)
        self.symmetric = symmetric
        self.ws = dist.get_world_size(group)
        self._exchange, self._local = None, None
        self._stream = None

    def _symmetric_rows(self, eng, feat, bounds, rank, N, dev):
        """This rank's row block [n_local x N] (a torch tensor: device memory for the point-to-point exchange):
        multiply the blocks of ``symmetric_plan``, ship the off-diagonal ones to the owners of the mirrored
        blocks, store the received ones transposed."""
        import torch
        import torch.distributed as dist
        lo, hi = bounds[rank], bounds[rank + 1]
        K = torch.empty((hi - lo, N), dtype=torch.float64, device=dev)
        base = K.data_ptr()
        compute, recv = symmetric_plan(bounds, rank)
        eng.gram_reset_stats(feat)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]      # block products | exchange + placement
        ev[0].record()
        # gloo (the CPU-side test backend) has no device-to-device send / recv: stage through the host there
        on_host = dist.get_backend(self.group) == "gloo"
        ops, keepalive = [], []
        for (r0, r1, c0, c1, peer) in compute:
            eng.gram_block(feat, (r0, r1), (c0, c1), base + ((r0 - lo) * N + c0) * 8, N)
            if peer >= 0:        # contiguous copy of the block for the wire
                buf = torch.empty((r1 - r0, c1 - c0), dtype=torch.float64, device=dev)
                eng.block_copy(base + ((r0 - lo) * N + c0) * 8, r1 - r0, c1 - c0, N, buf.data_ptr(), c1 - c0)
                if on_host:
                    torch.cuda.current_stream(dev).synchronize()
                    buf = buf.cpu()
                ops.append(dist.P2POp(dist.isend, buf, peer, group=self.group))
                keepalive.append(buf)
        ev[1].record()
        landed = []
        for (peer, r0, r1, c0, c1) in recv:          # rows [r0,r1) of the peer, columns [c0,c1) of this rank
            buf = torch.empty((r1 - r0, c1 - c0), dtype=torch.float64, device="cpu" if on_host else dev)
            ops.append(dist.P2POp(dist.irecv, buf, peer, group=self.group))
            landed.append((buf, r0, r1, c0, c1))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for (buf, r0, r1, c0, c1) in landed:         # K[c0:c1, r0:r1] = buf^T
            if on_host:
                buf = buf.to(dev)
                keepalive.append(buf)
            eng.block_copy(buf.data_ptr(), r1 - r0, c1 - c0, c1 - c0, base + ((c0 - lo) * N + r0) * 8, N, transpose=True)
        ev[2].record()
        self.last_events = ev        # elapsed_time(ev[0], ev[1]) = block products + packing, (ev[1], ev[2]) = exchange + placement
        return K

    def _gather_operand_rows(self, eng, feat, bounds, rank, dev):
        """exchange="phi": this rank's operand rows out of the library's buffer, all-gathered, the whole operand back in
        (left operand and, when the job has split columns, right operand).  Staged through torch tensors: the collective
        runs on torch's memory, two device copies of the operand (25-70 MB) are the price of keeping the library's buffers
        its own.  ``self.phi_bytes`` = what one rank sent."""
        import torch
        import torch.distributed as dist
        left, right, row_bytes, n_rows, own = eng.operand_rows(feat)
        assert own == (bounds[rank], bounds[rank + 1]) and n_rows == bounds[-1], (own, bounds)
        on_host = dist.get_backend(self.group) == "gloo"       # the CPU-side test backend has no device collectives
        self.phi_bytes = 0
        for ptr in (left, right):
            if not ptr:
                continue
            mine = torch.empty((own[1] - own[0], row_bytes), dtype=torch.uint8, device=dev)
            eng.memcpy_dev(mine.data_ptr(), ptr + own[0] * row_bytes, mine.numel())
            if on_host:
                torch.cuda.current_stream(dev).synchronize()
                whole = all_gather_rows(mine.cpu(), bounds, self.group).to(dev)
            else:
                whole = all_gather_rows(mine, bounds, self.group)
            eng.memcpy_dev(ptr, whole.data_ptr(), n_rows * row_bytes)
            self.phi_bytes += mine.numel()
            self._keepalive = (mine, whole)              # until the copies on the shared stream have run

    def _shared_stream(self, dev):
        """One torch side stream carries both the collective and the library's kernels, so the
        all-gather and its consumers are stream-ordered without a host synchronisation.  It has to be
        a real stream: torch's default stream has the handle 0, which ``gk_set_stream`` reads as
        "use the context's own stream" -- and that stream is not ordered against the collective."""
        import torch
        if self._stream is None:
            self._stream = torch.cuda.Stream(dev)
            self.engine.set_stream(self._stream.cuda_stream)
        return self._stream

    def close(self):
        """Give the engine its own stream back (the engine is process-wide)."""
        if self._stream is not None:
            self.engine.set_stream(0)
            self._stream = None

    def step(self, local_batch, to_host=False, label_map=None, keep=False, block_rows=0, on_block=None):
        """One fit_transform: returns (row block [n_local x N] or None, info dict).  ``label_map``: see
        ``ShardExchange`` (needed when the shard was ingested on its own).  ``keep``: leave the features (and
        with them the rank's row block in HBM) alive as info["feat"] / info["batch"]; the caller closes them.
        ``block_rows`` > 0: the rank's rows are multiplied in sub-blocks of at most that many rows which REUSE one device
        buffer (a row block that does not fit HBM: 200 000 graphs are 320 GB of float64); ``on_block(feat, (lo, hi))`` is
        called after each sub-block is queued (checksums, a copy to a pinned buffer) -- nothing is returned then.
        ``self.phase_events`` = four events on the shared stream: start | exchange + rebuild of the global batch | relabel +
        features (replicated) | the rank's Gram rows."""
        import torch
        import torch.distributed as dist
        rank = dist.get_rank(self.group)
        dev = torch.device("cuda", self.engine.device)
        if self._local is not local_batch:           # shard sizes are exchanged once per local shard
            self._exchange, self._local = ShardExchange(local_batch, self.group, dev, label_map), local_batch
        s = self._shared_stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))   # the shard message was built on the caller's stream
        ex = self._exchange
        eng = self.engine
        pe = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        with torch.cuda.stream(s):
            pe[0].record()
            flat = ex.gather_flat()
            n_labels, bounds = ex.n_labels, ex.bounds
            # the global CSR is rebuilt from the gathered messages by the library (two scans + one copy kernel)
            db = eng.batch_from_shards(ex.all_sizes[:, :3], ex.mg, ex.mv, ex.me, flat.data_ptr(), n_labels)
            pe[1].record()
            counts = eng.wl_relabel(db, self.n_iter)
            rows = (bounds[rank], bounds[rank + 1])
            if self.exchange == "phi" and self.ws > 1:
                with eng.options(**{"feat.rows_lo": rows[0], "feat.rows_hi": rows[1]}):
                    feat = eng.features(db, self.n_iter + 1)
                self._gather_operand_rows(eng, feat, bounds, rank, dev)
            else:
                feat = eng.features(db, self.n_iter + 1)
            pe[2].record()
            N = db.n_graphs
            if block_rows and block_rows > 0:
                K, info, gms, gfl = None, dict(), 0.0, 0.0
                for lo in range(rows[0], rows[1], int(block_rows)):
                    sub = (lo, min(lo + int(block_rows), rows[1]))
                    eng.gram(feat, 2 if self.normalize else 0, rows=sub, to_host=False)
                    if on_block is not None:
                        on_block(feat, sub)
                    fl, ms = eng.gram_stats(feat)         # (reads the block's events: the sub-blocks run back to back anyway)
                    gms, gfl = gms + ms, gfl + fl
                info["gram_blocks"] = -(-(rows[1] - rows[0]) // int(block_rows))
                info["gram_sum"] = (gfl, gms)
            elif self.symmetric and self.ws > 1:
                Kdev = self._symmetric_rows(eng, feat, bounds, rank, N, dev)
                if self.normalize:
                    eng.gram_normalize_rows(feat, rows, Kdev.data_ptr(), 2)
                K = Kdev.cpu().numpy() if to_host else None
                info = dict(K_dev=Kdev)
            else:
                K = eng.gram(feat, 2 if self.normalize else 0, rows=rows, to_host=to_host)
                info = dict()
            pe[3].record()
            self.phase_events = pe
            info.update(label_counts=counts, n_cols=feat.n_cols, n_cols_low=feat.n_cols_low, rows=rows,
                        n_graphs=N, gram=info.pop("gram_sum", None) or eng.gram_stats(feat), dtype=feat.dtype, operand=feat.operand)
            if keep:
                info["feat"], info["batch"] = feat, db
            else:
                feat.close()
                db.close()
        torch.cuda.current_stream(dev).wait_stream(s)
        return K, info


class ShardedSP(object):
    """ShortestPath Gram with graphs and Gram rows sharded over the ranks of ``group`` ("SP: identical
    scheme"): the (l_u, l_v, d) feature dictionary is global (shortest_path.py:412-499: ``_enum`` grows over ALL
    graphs), so after the all-gather of the CSR shards (+ edge weights when any rank has them) every rank runs the
    all-pairs distances, the pair dictionary and the feature builder on the global batch and then multiplies and
    stores only its own row block.  Shards must carry global level-0 ids (or their ``label_map``, see ShardExchange)."""

    def __init__(self, engine, normalize=False, with_labels=True, group=None, algorithm_type="auto"):
        import torch.distributed as dist
        if algorithm_type not in ("auto", "floyd_warshall", "dijkstra"):
            raise ValueError('Unsupported "algorithm_type"')
        self.engine, self.normalize, self.with_labels, self.group = engine, normalize, with_labels, group
        self.algorithm_type = algorithm_type          # only matters for general float edge weights (shortest_path.py)
        self.ws = dist.get_world_size(group)
        self._exchange, self._local, self._weights, self._stream = None, None, None, None

    _shared_stream = ShardedWL._shared_stream
    close = ShardedWL.close

    def step(self, local_batch, to_host=False, label_map=None, keep=False):
        """One fit_transform: returns (row block [n_local x N] or None, info dict)."""
        import torch
        import torch.distributed as dist
        rank = dist.get_rank(self.group)
        dev = torch.device("cuda", self.engine.device)
        if self._local is not local_batch:
            self._exchange, self._local = ShardExchange(local_batch, self.group, dev, label_map), local_batch
            self._weights = self._exchange.gather_weights()        # host array: gk_sp_build validates and uploads it
            self._fweights, fd = self._exchange.gather_float_weights()
            self._algo = None
            if self._fweights is not None:
                self._algo = fd if self.algorithm_type == "auto" else np.full(
                    fd.shape[0], 1 if self.algorithm_type == "dijkstra" else 0, np.uint8)
        s = self._shared_stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        ex, eng = self._exchange, self.engine
        with torch.cuda.stream(s):
            flat = ex.gather_flat()
            db = eng.batch_from_shards(ex.all_sizes[:, :3], ex.mg, ex.mv, ex.me, flat.data_ptr(), ex.n_labels)
            pb = eng.sp_build(db, self._weights, self.with_labels, float_weights=self._fweights, graph_algo=self._algo)
            feat = eng.features(pb, 1)
            rows = (ex.bounds[rank], ex.bounds[rank + 1])
            K = eng.gram(feat, 1 if self.normalize else 0, rows=rows, to_host=to_host)
            info = dict(rows=rows, n_graphs=db.n_graphs, n_pairs=pb.n_nodes, n_keys=pb.label_counts[0],
                        n_cols=feat.n_cols, n_cols_low=feat.n_cols_low, gram=eng.gram_stats(feat), operand=feat.operand)
            if keep:
                info["feat"], info["batch"], info["pairs"] = feat, db, pb
            else:
                feat.close()
                pb.close()
                db.close()
        torch.cuda.current_stream(dev).wait_stream(s)
        return K, info

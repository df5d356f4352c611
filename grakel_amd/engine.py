"""Thin Python wrapper over the C ABI: device context, device batches, features, Gram.

Host logic only; all compute happens in libgk_hip.so.  One ``Engine`` per device/process
(multi-GPU = one process per GPU, see ``grakel_amd.dist``).
"""
import ctypes
from ctypes import byref, c_double, c_int, c_int64, c_void_p

import numpy as np

from . import _lib
from ._lib import check


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class _PinnedBlock(object):
    """Owner of one pinned host block; goes back to the pool when the last ndarray view dies."""

    def __init__(self, pool, ptr, cap):
        self.pool, self.ptr, self.cap = pool, ptr, cap

    def __del__(self):
        try:
            self.pool._release(self.ptr, self.cap)
        except Exception:
            pass


class PinnedPool(object):
    """Gram outputs live in pinned host memory: the 8 N^2-byte device -> host copy then runs at the PCIe
    rate (57 GB/s measured) instead of the pageable rate (12-18 GB/s).  Pinning is slow (~0.14 ms per MB),
    so released blocks are kept and handed out again: the first ``fit_transform`` of a given size pays for
    the allocation, the following ones do not.  Pinned memory cannot be swapped, so the pool is bounded:
    one block is never larger than ``MAX_BLOCK`` (bigger outputs are ordinary pageable arrays), the blocks
    retained for reuse never add up to more than ``MAX_RETAINED`` (least recently released go first), and
    ``Engine.close()`` frees them all.  ``GK_PINNED_OUTPUT=0`` turns the pool off."""

    GRANULE = 2 << 20
    MIN_BYTES = 1 << 20
    KEEP = 2                       # blocks retained per size
    MAX_BLOCK = 24 << 30           # 50 000 x 50 000 float64 = 20 GB still gets a pinned block
    MAX_RETAINED = 32 << 30

    def __init__(self, lib):
        import collections
        import threading
        self.lib = lib
        self.free = collections.OrderedDict()        # (cap, ptr) in release order: oldest first
        self.retained = 0
        # _release runs from __del__, i.e. on any thread and -- when a cyclic-GC pass starts inside one of the critical
        # sections below -- possibly on the thread that already holds the lock.  So __del__ never waits: it parks the
        # block in `pending` (deque.append is atomic) and files it only if the lock is free right now; whoever holds
        # the lock files the parked blocks before it leaves.
        self.pending = collections.deque()
        self.lock = threading.Lock()
        self.closed = False

    def _file_pending_locked(self):
        """(lock held) parked blocks -> the free list, or a list of pointers to hand back to the driver."""
        drop = []
        while True:
            try:
                ptr, cap = self.pending.popleft()
            except IndexError:
                break
            same = 0
            for k in self.free:
                if k[0] == cap:
                    same += 1
            if self.closed or same >= self.KEEP or cap > self.MAX_RETAINED:
                drop.append(ptr)
                continue
            self.free[(cap, ptr)] = True
            self.retained += cap
            while self.retained > self.MAX_RETAINED and self.free:
                (c, q), _ = self.free.popitem(last=False)
                self.retained -= c
                drop.append(q)
        return drop

    def empty(self, shape):
        import os
        nbytes = 8
        for d in shape:
            nbytes *= int(d)
        if nbytes < self.MIN_BYTES or nbytes > self.MAX_BLOCK or os.environ.get("GK_PINNED_OUTPUT", "1") == "0":
            return np.empty(shape, dtype=np.float64)
        cap = -(-nbytes // self.GRANULE) * self.GRANULE
        ptr = None
        with self.lock:
            drop = self._file_pending_locked()
            for key in self.free:
                if key[0] == cap:
                    ptr = key[1]
                    del self.free[key]
                    self.retained -= cap
                    break
        for q in drop:
            self.lib.gk_host_free(c_void_p(q))
        if ptr is None:
            p = c_void_p()
            if self.lib.gk_host_alloc(ctypes.c_uint64(cap), byref(p)) != 0 or not p.value:
                self.trim(0)                                          # give the retained blocks back and try once more
                if self.lib.gk_host_alloc(ctypes.c_uint64(cap), byref(p)) != 0 or not p.value:
                    return np.empty(shape, dtype=np.float64)          # no pinned memory left: pageable output
            ptr = p.value
        buf = (ctypes.c_char * nbytes).from_address(ptr)
        buf._gk_block = _PinnedBlock(self, ptr, cap)               # lives as long as any view of the array
        return np.frombuffer(buf, dtype=np.float64).reshape(shape)

    def _release(self, ptr, cap):
        self.pending.append((ptr, cap))
        if not self.lock.acquire(False):             # held (maybe by this very thread): the holder files the block
            return
        try:
            drop = self._file_pending_locked()
        finally:
            self.lock.release()
        for q in drop:
            self.lib.gk_host_free(c_void_p(q))

    def trim(self, keep_bytes=0):
        """Free retained blocks, oldest first, until at most ``keep_bytes`` stay pinned."""
        with self.lock:
            drop = self._file_pending_locked()
            while self.retained > keep_bytes and self.free:
                (c, q), _ = self.free.popitem(last=False)
                self.retained -= c
                drop.append(q)
        for q in drop:
            self.lib.gk_host_free(c_void_p(q))

    def close(self):
        self.closed = True           # blocks still referenced by live arrays are freed when those die
        self.trim(0)


class DeviceBatch(object):
    def __init__(self, engine, handle, n_graphs, n_nodes, n_edges):
        self.engine, self.handle = engine, handle
        self.n_graphs, self.n_nodes, self.n_edges = n_graphs, n_nodes, n_edges
        self.label_counts = None

    def close(self):
        if self.handle is not None and self.engine.handle is not None:
            self.engine.lib.gk_batch_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm(object):
    """A communicator of the C ABI (``gk_comm``): one process per GPU, RCCL underneath."""

    def __init__(self, engine, handle, rank, n_ranks):
        self.engine, self.handle, self.rank, self.n_ranks = engine, handle, rank, n_ranks

    def close(self):
        if self.handle is not None and self.engine.handle is not None:
            self.engine.lib.gk_comm_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FittedWL(object):
    """Fitted WL dictionaries on the device (csrc/wl_transform.hip): what ``transform`` looks target signatures up in."""

    def __init__(self, engine, handle, batch):
        self.engine, self.handle, self.batch = engine, handle, batch

    def close(self):
        if self.handle is not None and self.engine.handle is not None:
            self.engine.lib.gk_wl_fitted_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceFeatures(object):
    def __init__(self, engine, handle, batch, n_fit):
        self.engine, self.handle, self.batch, self.n_fit = engine, handle, batch, n_fit
        nc, nl, nnz, mc, dt = c_int64(), c_int64(), c_int64(), c_int64(), c_int()
        check(engine.lib.gk_features_info(handle, byref(nc), byref(nl), byref(nnz), byref(mc), byref(dt)))
        self.n_cols, self.n_cols_low, self.nnz = nc.value, nl.value, nnz.value
        self.max_count, self.dtype = mc.value, dt.value
        fp4, k1, k8, nw = c_int(), c_int(), c_int(), c_int64()
        check(engine.lib.gk_features_operand(handle, byref(fp4), byref(k1), byref(k8), byref(nw)))
        # arithmetic type of the dense product: "fp4+i8" (MX fp4 codes for counts <= 4, int8 for 5..127), "i8", "f64"
        self.operand = "f64" if dt.value else ("fp4+i8" if fp4.value else "i8")
        if nw.value and not dt.value:
            self.operand += "+f64"
        self.symmetric = n_fit == batch.n_graphs
        self.n_rows = batch.n_graphs if self.symmetric else batch.n_graphs - n_fit
        self.n_out_cols = n_fit

    def close(self):
        if self.handle is not None and self.engine.handle is not None:
            self.engine.lib.gk_features_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


MAX_LEVELS_PER_JOB = 48          # FEAT_MAX_LEVELS of csrc/features.h


class ChunkedFeatures(object):
    """Features of a hierarchy deeper than one job holds (48 levels): one ``DeviceFeatures`` per chunk of levels.  K is a
    sum over levels (weisfeiler_lehman.py:269-270), so the chunks' matrices and self-similarity vectors add up;
    ``Engine.gram`` / ``Engine.selfk`` do that, normalisation happens on the sum."""

    def __init__(self, parts):
        self.parts = parts
        p0 = parts[0]
        self.engine, self.batch, self.n_fit = p0.engine, p0.batch, p0.n_fit
        self.symmetric, self.n_rows, self.n_out_cols = p0.symmetric, p0.n_rows, p0.n_out_cols
        self.n_cols = sum(p.n_cols for p in parts)
        self.n_cols_low = sum(p.n_cols_low for p in parts)
        self.nnz = sum(p.nnz for p in parts)
        self.max_count = max(p.max_count for p in parts)
        self.dtype = max(p.dtype for p in parts)
        self.operand = "|".join(sorted(set(p.operand for p in parts)))
        self.handle = None

    def close(self):
        for p in self.parts:
            p.close()


class Engine(object):
    """A libgk_hip context bound to one GPU."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        if _lib.device_count() <= 0:
            raise _lib.GkError("no MI355X / HIP device visible: grakel_amd has no CPU fallback")
        h = c_void_p()
        check(self.lib.gk_create(int(device), byref(h)))
        self.handle, self.device = h, int(device)
        self.pinned = PinnedPool(self.lib)

    def close(self):
        if self.handle is not None:
            self.pinned.close()
            self.lib.gk_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing -------------------------------------------------------------------------
    def set_option(self, name, value):
        """Route / capacity options of the context (``gk_set_option``, include/gk_hip.h): results never change."""
        check(self.lib.gk_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name):
        v = c_int64()
        check(self.lib.gk_get_option(self.handle, name.encode(), byref(v)))
        return v.value

    def options(self, **kw):
        """Context manager: ``with eng.options(**{"wl.no_tiny": 1}): ...`` -- restores the old values."""
        eng = self

        class _Scope(object):
            def __enter__(self_):
                self_.old = {k: eng.get_option(k) for k in kw}
                for k, v in kw.items():
                    eng.set_option(k, v)
                return eng

            def __exit__(self_, *exc):
                for k, v in self_.old.items():
                    eng.set_option(k, v)
                return False

        return _Scope()

    def set_stream(self, stream_ptr):
        check(self.lib.gk_set_stream(self.handle, c_void_p(stream_ptr) if stream_ptr else None))

    def synchronize(self):
        check(self.lib.gk_synchronize(self.handle))

    def timer_start(self):
        check(self.lib.gk_timer_start(self.handle))

    def timer_stop_ms(self):
        ms = c_double()
        check(self.lib.gk_timer_stop_ms(self.handle, byref(ms)))
        return ms.value

    def profile(self, enable):
        check(self.lib.gk_profile_enable(self.handle, 1 if enable else 0))
        check(self.lib.gk_profile_reset(self.handle))

    def profile_get(self, name):
        ms, n = c_double(), c_int64()
        check(self.lib.gk_profile_get(self.handle, name.encode(), byref(ms), byref(n)))
        return ms.value, n.value

    # -- batches --------------------------------------------------------------------------
    def upload(self, gb):
        h = c_void_p()
        check(self.lib.gk_batch_create(self.handle, gb.n_graphs, gb.n_nodes, gb.n_edges,
                                       _ptr(gb.graph_ptr), _ptr(gb.row_ptr), _ptr(gb.col_idx),
                                       _ptr(gb.node_label), gb.n_labels, 0, byref(h)))
        return DeviceBatch(self, h, gb.n_graphs, gb.n_nodes, gb.n_edges)

    def concat(self, a, b, n_labels):
        """Union batch on the device (graphs of ``a`` first); ``a`` and ``b`` stay valid."""
        h = c_void_p()
        check(self.lib.gk_batch_concat(self.handle, a.handle, b.handle, int(n_labels), byref(h)))
        return DeviceBatch(self, h, a.n_graphs + b.n_graphs, a.n_nodes + b.n_nodes, a.n_edges + b.n_edges)

    def upload_from_device(self, n_graphs, n_nodes, n_edges, graph_ptr, row_ptr, col_idx, node_label,
                           n_labels):
        """Arrays are raw device pointers (ints), e.g. torch tensors' ``data_ptr()``."""
        h = c_void_p()
        check(self.lib.gk_batch_create(self.handle, n_graphs, n_nodes, n_edges, c_void_p(graph_ptr),
                                       c_void_p(row_ptr), c_void_p(col_idx), c_void_p(node_label),
                                       n_labels, 1, byref(h)))
        return DeviceBatch(self, h, n_graphs, n_nodes, n_edges)

    def batch_from_shards(self, shard_sizes, mg, mv, me, gathered_ptr, n_labels):
        """shard_sizes: int64 [n_ranks, 3] (graphs, nodes, edges); gathered_ptr: device address of the
        all-gathered int32 messages (grakel_amd.dist.ShardExchange layout)."""
        sizes = np.ascontiguousarray(shard_sizes, dtype=np.int64)
        h = c_void_p()
        check(self.lib.gk_batch_from_shards(self.handle, int(sizes.shape[0]), sizes.ctypes.data_as(ctypes.POINTER(c_int64)),
                                            int(mg), int(mv), int(me), c_void_p(int(gathered_ptr)), int(n_labels), byref(h)))
        return DeviceBatch(self, h, int(sizes[:, 0].sum()), int(sizes[:, 1].sum()), int(sizes[:, 2].sum()))

    # -- multi-GPU through the C ABI (csrc/comm.hip; grakel_amd.dist is the torch.distributed form of the same scheme) ----
    def comm_unique_id(self):
        """128 bytes rank 0 creates and hands to the other processes (``ncclGetUniqueId``)."""
        buf = ctypes.create_string_buffer(128)
        check(self.lib.gk_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, n_ranks, unique_id):
        """Collective: every rank calls it with rank 0's id.  Returns a ``Comm``."""
        h = c_void_p()
        check(self.lib.gk_comm_init(self.handle, int(rank), int(n_ranks), ctypes.c_char_p(bytes(unique_id)), byref(h)))
        return Comm(self, h, int(rank), int(n_ranks))

    def batch_allgather(self, comm, gb):
        """Collective: this rank's ``GraphBatch`` shard (local numbering, global level-0 label ids) -> (the batch of all
        graphs in rank order on this rank's device, graph bounds of the ranks)."""
        h = c_void_p()
        bounds = np.zeros(comm.n_ranks + 1, dtype=np.int64)
        check(self.lib.gk_batch_allgather(self.handle, comm.handle, gb.n_graphs, gb.n_nodes, gb.n_edges,
                                          _ptr(np.ascontiguousarray(gb.graph_ptr, dtype=np.int32)), _ptr(gb.row_ptr),
                                          _ptr(gb.col_idx), _ptr(gb.node_label), int(gb.n_labels), byref(h), _ptr(bounds)))
        ng, nv, ne = c_int64(), c_int64(), c_int64()
        check(self.lib.gk_batch_info(h, byref(ng), byref(nv), byref(ne)))
        return DeviceBatch(self, h, ng.value, nv.value, ne.value), bounds

    def gram_sharded(self, comm, feat, bounds, normalize=0):
        """The rows of K that belong to this rank's graphs, float64 [n_own x n_cols] on the host."""
        bounds = np.ascontiguousarray(bounds, dtype=np.int64)
        lo, hi = int(bounds[comm.rank]), int(bounds[comm.rank + 1])
        n_cols = feat.n_fit if feat.n_fit else feat.batch.n_graphs
        out = np.empty((hi - lo, n_cols), dtype=np.float64)
        rl, rh = c_int64(), c_int64()
        check(self.lib.gk_gram_sharded(self.handle, comm.handle, feat.handle, _ptr(bounds), int(normalize), _ptr(out),
                                       byref(rl), byref(rh)))
        assert (rl.value, rh.value) == (lo, hi)
        return out

    def export_state(self, db):
        """The fitted batch as one self-describing blob (``gk_export_state``): what a consumer of the C ABI persists."""
        need = ctypes.c_uint64()
        check(self.lib.gk_export_state(self.handle, db.handle, None, 0, byref(need)))
        buf = np.empty(need.value, dtype=np.uint8)
        check(self.lib.gk_export_state(self.handle, db.handle, _ptr(buf), need.value, byref(need)))
        return buf

    def import_state(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        h = c_void_p()
        check(self.lib.gk_import_state(self.handle, _ptr(blob), int(blob.size), byref(h)))
        ng, nv, ne = c_int64(), c_int64(), c_int64()
        check(self.lib.gk_batch_info(h, byref(ng), byref(nv), byref(ne)))
        return DeviceBatch(self, h, ng.value, nv.value, ne.value)

    # -- WL ---------------------------------------------------------------------------------
    def wl_relabel(self, db, n_iter, hash_bits=0):
        counts = (c_int64 * (n_iter + 1))()
        rounds = c_int(0)
        check(self.lib.gk_wl_relabel(self.handle, db.handle, int(n_iter), int(hash_bits), counts,
                                     byref(rounds)))
        db.label_counts = [int(c) for c in counts]
        db.refine_rounds = rounds.value
        route = c_int(0)
        check(self.lib.gk_wl_route(db.handle, byref(route)))
        db.stream_route = bool(route.value)           # relabelled without host round trips (csrc/wl_stream.hip)
        return db.label_counts

    GK_ERR_STATE, GK_ERR_UNSUPPORTED = -3, -4

    def wl_fitted(self, db, n_iter):
        """Fitted dictionaries of a batch relabelled with ``n_iter`` (``gk_wl_fitted_create``); None when the job needs the
        joint transform route (GK_ERR_UNSUPPORTED)."""
        h = c_void_p()
        rc = self.lib.gk_wl_fitted_create(self.handle, db.handle, int(n_iter), byref(h))
        if rc == self.GK_ERR_UNSUPPORTED:
            return None
        check(rc)
        return FittedWL(self, h, db)

    def wl_fitted_selfk(self, wf):
        out = np.empty(wf.batch.n_graphs, dtype=np.float64)
        check(self.lib.gk_wl_fitted_selfk(self.handle, wf.handle, _ptr(out)))
        return out

    def wl_transform(self, wf, tb, normalize=0):
        """(K [n_targets x n_fitted], target diagonal) by look-up; "stale": rebuild the fitted state; None: unsupported job."""
        K = self.pinned.empty((tb.n_graphs, wf.batch.n_graphs))       # pinned when large: the copy runs at the PCIe rate
        yd = np.empty(tb.n_graphs, dtype=np.float64)
        rc = self.lib.gk_wl_transform(self.handle, wf.handle, tb.handle, int(normalize), _ptr(K), _ptr(yd))
        if rc == self.GK_ERR_STATE:
            return "stale"                       # the fitted batch was relabelled since the state was built
        if rc == self.GK_ERR_UNSUPPORTED:
            return None
        check(rc)
        return K, yd

    def wl_fit_transform(self, db, n_iter, kind=0, normalize=0, to_host=True, hash_bits=0):
        """relabel + features (all graphs fitted) + Gram in one library call (``gk_wl_fit_transform``): the relabel is queued
        without a host round trip when the job allows.  Returns (DeviceFeatures, K or None); ``db.label_counts`` is set."""
        counts = (c_int64 * (n_iter + 1))()
        rounds = c_int(0)
        h = c_void_p()
        out = self.pinned.empty((db.n_graphs, db.n_graphs)) if to_host else None
        check(self.lib.gk_wl_fit_transform(self.handle, db.handle, int(n_iter), int(hash_bits), int(kind), int(normalize),
                                           counts, byref(rounds), byref(h), _ptr(out)))
        db.label_counts = [int(c) for c in counts]
        db.refine_rounds = rounds.value
        route = c_int(0)
        check(self.lib.gk_wl_route(db.handle, byref(route)))
        db.stream_route = bool(route.value)
        return DeviceFeatures(self, h, db, db.n_graphs), out

    def wl_labels(self, db, level):
        out = np.empty(db.n_nodes, dtype=np.int32)
        check(self.lib.gk_wl_get_labels(self.handle, db.handle, int(level), _ptr(out)))
        return out

    def wl_debug_signature(self, db, level, seed):
        h = np.empty(db.n_nodes, dtype=np.uint64)
        s = np.empty(db.n_edges, dtype=np.int32)
        check(self.lib.gk_wl_debug_signature(self.handle, db.handle, int(level),
                                             ctypes.c_uint64(seed), _ptr(h), _ptr(s)))
        return h, s

    # -- features / Gram -----------------------------------------------------------------------
    def features(self, db, n_levels, n_fit=None, kind=0):
        """kind 0: dot product of label counts; 1: histogram intersection (min-sum).  More than 48 levels: a
        ``ChunkedFeatures`` (one job per 48 levels; ``gram`` / ``selfk`` add the chunks up)."""
        n_fit = db.n_graphs if n_fit is None else int(n_fit)
        n_levels = int(n_levels)
        if n_levels <= MAX_LEVELS_PER_JOB:
            h = c_void_p()
            check(self.lib.gk_features_build_ex(self.handle, db.handle, n_levels, n_fit, int(kind), byref(h)))
            return DeviceFeatures(self, h, db, n_fit)
        parts = []
        for lo in range(0, n_levels, MAX_LEVELS_PER_JOB):
            h = c_void_p()
            check(self.lib.gk_features_build_range(self.handle, db.handle, lo, min(lo + MAX_LEVELS_PER_JOB, n_levels), n_fit,
                                                   int(kind), byref(h)))
            parts.append(DeviceFeatures(self, h, db, n_fit))
        return ChunkedFeatures(parts)

    def operand_rows(self, feat):
        """gk_features_operand_rows: (left operand ptr, right operand ptr or 0, bytes per row, rows, (own_lo, own_hi)) -- device
        addresses as ints; the multi-GPU operand-row exchange (grakel_amd/dist.py) fills in the rows of the other ranks."""
        a, b = c_void_p(), c_void_p()
        rb, nr, lo, hi = c_int64(), c_int64(), c_int64(), c_int64()
        check(self.lib.gk_features_operand_rows(feat.handle, byref(a), byref(b), byref(rb), byref(nr), byref(lo), byref(hi)))
        return int(a.value or 0), int(b.value or 0), rb.value, nr.value, (lo.value, hi.value)

    def memcpy_dev(self, dst_ptr, src_ptr, n_bytes):
        check(self.lib.gk_memcpy_dev(self.handle, c_void_p(int(dst_ptr)), c_void_p(int(src_ptr)), int(n_bytes)))

    def selfk(self, feat):
        if isinstance(feat, ChunkedFeatures):
            return sum(self.selfk(p) for p in feat.parts)
        out = np.empty(feat.batch.n_graphs, dtype=np.float64)
        check(self.lib.gk_features_selfk(self.handle, feat.handle, _ptr(out)))
        return out

    def debug_phi(self, feat):
        out = np.empty((feat.batch.n_graphs, feat.n_cols), dtype=np.float64)
        check(self.lib.gk_features_debug_phi(self.handle, feat.handle, _ptr(out)))
        return out

    def debug_phi_right(self, feat):
        """(right operand, parts): differs from debug_phi only in split columns (counts of 128..381 as int8 digits)"""
        out = np.empty((feat.batch.n_graphs, feat.n_cols), dtype=np.float64)
        parts = ctypes.c_int(0)
        check(self.lib.gk_features_debug_phi_right(self.handle, feat.handle, _ptr(out), ctypes.byref(parts)))
        return out, parts.value

    def gram(self, feat, normalize=0, rows=None, to_host=True):
        if isinstance(feat, ChunkedFeatures):
            if not to_host:
                raise _lib.GkError("a hierarchy of more than 48 levels is summed on the host: to_host=False is not available")
            lo, hi = (0, feat.n_rows) if rows is None else rows
            K = None
            for p in feat.parts:                      # integer-valued float64 matrices: the sum is exact
                Kp = self.gram(p, 0, rows=rows)
                K = Kp if K is None else np.add(K, Kp, out=K)
            if normalize:
                d = self.selfk(feat)
                dr = d[lo:hi] if feat.symmetric else d[feat.n_fit + lo:feat.n_fit + hi]
                with np.errstate(divide="ignore", invalid="ignore"):
                    K = np.divide(K, np.sqrt(np.outer(dr, d[:feat.n_out_cols])), out=K)
                if normalize == 2:
                    K = np.nan_to_num(K, copy=False)
            return K
        lo, hi = (0, feat.n_rows) if rows is None else rows
        out = self.pinned.empty((hi - lo, feat.n_out_cols)) if to_host else None
        check(self.lib.gk_gram_rows(self.handle, feat.handle, lo, hi, int(normalize), _ptr(out)))
        return out

    # -- block-wise Gram (multi-GPU path): caller-owned device memory, addresses as ints -------------------
    def gram_block(self, feat, rows, cols, out_ptr, ld):
        check(self.lib.gk_gram_block(self.handle, feat.handle, int(rows[0]), int(rows[1]), int(cols[0]), int(cols[1]),
                                     c_void_p(int(out_ptr)), int(ld)))

    def gram_reset_stats(self, feat):
        check(self.lib.gk_gram_reset_stats(feat.handle))

    def block_copy(self, src_ptr, rows, cols, ld_src, dst_ptr, ld_dst, transpose=False):
        check(self.lib.gk_block_copy(self.handle, c_void_p(int(src_ptr)), int(rows), int(cols), int(ld_src),
                                     c_void_p(int(dst_ptr)), int(ld_dst), 1 if transpose else 0))

    def gram_normalize_rows(self, feat, rows, k_ptr, mode):
        check(self.lib.gk_gram_normalize_rows(self.handle, feat.handle, int(rows[0]), int(rows[1]), c_void_p(int(k_ptr)), int(mode)))

    def gram_checksum(self, feat):
        """(sum, trace, max |K - K^T|) of the matrix the last ``gram`` call left on the device."""
        a, b, c = c_double(), c_double(), c_double()
        check(self.lib.gk_gram_checksum(self.handle, feat.handle, byref(a), byref(b), byref(c)))
        return a.value, b.value, c.value

    def host_copy_stats(self):
        """(cpu dict, copy dict) -- gk_host_copy_stats: the host-thread budget and the stages of the last host copy."""
        cpu, cp = (c_int * 4)(), (c_double * 8)()
        check(self.lib.gk_host_copy_stats(self.handle, cpu, cp))
        form = {0: "plain float64 copy", 1: "upper-triangle blocks", 2: "rectangular narrow"}.get(int(cp[0]), "?")
        return (dict(online=cpu[0], affinity=cpu[1], cgroup_quota_cpus=cpu[2], thread_budget=cpu[3]),
                dict(form=form, widening_threads=int(cp[1]), pcie_bytes=int(cp[2]), ms_until_last_chunk_landed=cp[3],
                     ms_copy_out=cp[4], widen_busy_ms_mean=cp[5], widen_busy_ms_max=cp[6], chunks=int(cp[7])))

    def gram_stats(self, feat):
        fl, ms = c_double(), c_double()
        check(self.lib.gk_gram_last_stats(feat.handle, byref(fl), byref(ms)))
        return fl.value, ms.value

    # -- shortest paths -------------------------------------------------------------------------
    def sp_build(self, db, edge_weight, with_labels, n_levels=1, float_weights=None, graph_algo=None):
        """Pair batch of the ShortestPath features; n_levels > 1 keys level l by the WL labels of
        level l (``wl_relabel(db, n_levels - 1)`` first): the WL framework over the SP base kernel.
        ``float_weights`` (float64[n_edges]) + ``graph_algo`` (uint8[n_graphs]: 0 floyd_warshall, 1 dijkstra): arbitrary
        positive float weights, distances bit-identical to the reference's (gk_sp_build_f64)."""
        h = c_void_p()
        npairs, nkeys = c_int64(), (c_int64 * int(n_levels))()
        if float_weights is not None:
            fw = np.ascontiguousarray(float_weights, np.float64)
            ga = np.ascontiguousarray(graph_algo, np.uint8)
            check(self.lib.gk_sp_build_f64(self.handle, db.handle, _ptr(fw), _ptr(ga), 1 if with_labels else 0,
                                           int(n_levels), byref(h), byref(npairs), nkeys))
        else:
            check(self.lib.gk_sp_build_levels(self.handle, db.handle, _ptr(edge_weight), 1 if with_labels else 0,
                                              int(n_levels), byref(h), byref(npairs), nkeys))
        pb = DeviceBatch(self, h, db.n_graphs, npairs.value, 0)
        pb.label_counts = [int(k) for k in nkeys]
        return pb

    def core_numbers(self, db):
        out = np.empty(db.n_nodes, dtype=np.int32)
        check(self.lib.gk_core_numbers(self.handle, db.handle, _ptr(out)))
        return out

    def sp_debug_apsp_f64(self, db, float_weights, graph_algo, graph, n):
        out = np.empty((n, n), dtype=np.float64)
        fw = np.ascontiguousarray(float_weights, np.float64)
        ga = np.ascontiguousarray(graph_algo, np.uint8)
        check(self.lib.gk_sp_debug_apsp_f64(self.handle, db.handle, _ptr(fw), _ptr(ga), int(graph), _ptr(out)))
        return out

    def sp_debug_apsp(self, db, edge_weight, graph, n):
        out = np.empty((n, n), dtype=np.int32)
        check(self.lib.gk_sp_debug_apsp(self.handle, db.handle, _ptr(edge_weight), int(graph), _ptr(out)))
        return out


_engines = {}


def get_engine(device=None):
    """Process-wide engine per device (created on first use; raises if no GPU)."""
    import os
    if device is None:
        device = int(os.environ.get("GK_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = _lib.device_count()
        if n > 0:
            device %= n
    e = _engines.get(device)
    if e is None or e.handle is None:
        e = Engine(device)
        _engines[device] = e
    return e

#!/usr/bin/env python3
"""Where the host-to-host wall of config 3 goes: upload (H2D of the packed CSR) | device step | copy-out, each bracketed by
synchronisations; the copy-out over its forms (triangle / rectangular / plain) and host thread counts.
usage: python tools/dev/h2h_breakdown.py [config3|config5|nci1|...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: F401,E402
from grakel_amd.engine import get_engine  # noqa: E402

wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "config3")
eng = get_engine()
N, h = wl.N, wl.h


def med(f, n=7):
    t = []
    for _ in range(n):
        eng.synchronize()
        t0 = time.perf_counter()
        f()
        eng.synchronize()
        t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t))


db = eng.upload(wl.batch)
print("upload (H2D of %.1f MB, pageable numpy arrays): %.3f ms" % (
    (wl.batch.row_ptr.nbytes + wl.batch.col_idx.nbytes + wl.batch.node_label.nbytes + wl.batch.graph_ptr.nbytes) / 1e6,
    med(lambda: eng.upload(wl.batch).close())))
feat, _ = eng.wl_fit_transform(db, h, to_host=False)
print("device step: %.3f ms" % med(lambda: eng.wl_fit_transform(db, h, to_host=False)[0].close()))
for norm in (0, 2):
    for name, opts in (("triangle", {}), ("rectangular", {"gram.no_tri": 1}), ("plain float64", {"gram.no_compact": 1})):
        for thr in ((0, 8, 16, 32, 64) if name != "plain float64" else (0,)):
            o = dict(opts)
            o["gram.copy_threads"] = thr
            with eng.options(**o):
                eng.gram(feat, norm)
                ms = med(lambda: eng.gram(feat, norm))
            print("gram + copy-out normalize=%d %-14s threads=%-3d %.3f ms" % (norm, name, thr, ms))
gk = med(lambda: eng.gram(feat, 0, to_host=False))
print("gram alone (device): %.3f ms" % gk)
full = med(lambda: (lambda d: (eng.wl_fit_transform(d, h, to_host=True)[0].close(), d.close()))(eng.upload(wl.batch)))
print("host to host, one call sequence: %.3f ms" % full)

# ---- what the bus does alone: 105 MB (the triangle's uint16 blocks) device -> pinned host, in one piece and in pieces
import torch  # noqa: E402
src = torch.empty(105 << 20, dtype=torch.uint8, device="cuda")
dst = torch.empty(105 << 20, dtype=torch.uint8).pin_memory()
for piece_mb in (105, 32, 16, 8, 4):
    piece = piece_mb << 20
    def run():
        for o in range(0, 105 << 20, piece):
            dst[o:o + piece].copy_(src[o:o + piece], non_blocking=True)
        torch.cuda.synchronize()
    run()
    t = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        t.append((time.perf_counter() - t0) * 1e3)
    print("D2H of 105 MB into pinned memory in %3d MB pieces: %.3f ms (%.1f GB/s)" % (piece_mb, min(t), 105 * 1.048576 / min(t)))

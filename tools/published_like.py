#!/usr/bin/env python3
"""One stand-in for a published TU dataset (grakel_amd/synthetic.py PUBLISHED_LIKE) through the device path:

    python tools/published_like.py SET wl|sp|wl_e2e [steps]

wl      WL-subtree h=5 device step (packed CSR in HBM -> float64 K in HBM): ms, phases, relabel route, operand, the Gram
        kernel against both roofs, the matrix against the reference's checksums (tests/golden/pub_SET.npz)
sp      ShortestPath(with_labels) fit_transform on the FULL set from the packed CSR in HBM: ms, phases, all-pairs rate; the
        matrix asserted against the full-set fixture and the real reference's block of the largest graphs (round 6)
wl_e2e  packed CSR on the host -> float64 K on the host, unnormalised and normalised, entry-wise against the golden

Prints one JSON line.  (Each mode is its own process so that tools/profile_published.sh can put a timeout and a
rocprofv3 kernel trace around it.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def sp_summary(eng, wl, steps):
    """bench.published_sp: timed, and ASSERTED against tests/golden/pub_<set>_sp_full.npz / pub_<set>_sp_big.npz (round 6)"""
    return bench.published_sp(eng, wl, steps)


def main():
    name, mode = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    import torch  # noqa: F401
    from grakel_amd.engine import get_engine
    wl = bench.Workload(name)
    eng = get_engine()
    for o in os.environ.get("GK_TOOL_OPTS", "").split():
        eng.set_option(o.split("=")[0], int(o.split("=")[1]))
    if mode == "wl":
        out = bench.device_step_summary(eng, wl, steps)
    elif mode == "sp":
        out = sp_summary(eng, wl, steps)
    else:
        out, Ku = bench.host_to_host(eng, wl, steps, 2)
        if wl.golden is not None:
            err, n = bench.check_host_matrix(Ku, wl.golden)
            out["gram_max_abs_err"], out["entries_checked"] = err, n
    print(json.dumps(out))


if __name__ == "__main__":
    main()

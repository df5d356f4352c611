#!/usr/bin/env python3
"""Kernel timeline WITH GAPS of one steady-state step of a rocprofv3 --kernel-trace csv:
    python tools/dev/trace_gaps.py kernel_trace.csv MARKER [which=-2]
The step runs from the launch after the `which`-th occurrence of a kernel whose name contains MARKER to the next one."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
lo, hi = idx[which - 1] + 1, idx[which] + 1
t0, prev, busy = int(rows[lo]["Start_Timestamp"]), None, 0.0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += (e - s) / 1e3
    print("%8.1f %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0,
                                           r["Kernel_Name"].split("(")[0].replace("void ", "")[:70]))
    prev = e
print("# %d kernels, busy %.1f us of %.1f us" % (hi - lo, busy, (prev - t0) / 1e3))

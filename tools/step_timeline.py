#!/usr/bin/env python3
"""Kernel-by-kernel timeline of the LAST bench step of a rocprofv3 --kernel-trace --output-format csv run.

usage: python tools/step_timeline.py gpurun_out/prof_dir      (start us, duration us, kernel)
"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "gram_ws_kernel<" in r["Kernel_Name"] or "gram_tile_kernel" in r["Kernel_Name"] or "gram_i8" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
busy = 0.0
for r in rows[lo:hi]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    print("%8.1f %7.1f %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d,
                              r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]))
span = (int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3
print("# %d kernels, busy %.1f us of %.1f us" % (hi - lo, busy, span))

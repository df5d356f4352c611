#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats) of a results .db as CSV.

usage: python tools/prof_summary.py gpurun_out/prof_r1 > profiles/r01_bench_kernel_stats.csv
"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
dbs = glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)
assert dbs, "no rocprofv3 results db under " + root
cur = sqlite3.connect(dbs[0]).cursor()
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in cur.execute(
        "select name,total_calls,total_duration,average,percentage from top_kernels"):
    short = name.split("(")[0].replace("void ", "")
    print("%s,%d,%.3f,%.3f,%.2f" % (short, calls, total, avg, pct))

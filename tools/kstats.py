#!/usr/bin/env python3
"""Compact view of a rocprofv3 --kernel-trace --stats kernel_stats.csv: calls, average us, total us per kernel.
usage: python tools/kstats.py <dir-or-csv> [top]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("%-62s %6s %10s %10s" % ("kernel", "calls", "avg_us", "total_us"))
for r in rows[:top]:
    name = r["Name"].split("(")[0].replace("void ", "")
    print("%-62s %6d %10.1f %10.1f" % (name[:62], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))

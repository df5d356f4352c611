#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc ... --output-format csv runs.

usage: python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE [gpurun_out/pmc_WRITE_SIZE ...] > profiles/x.csv
Sums every counter over the dispatch's rows (rocprofv3 emits one row per counter instance/XCD),
then averages over the launches of a kernel.  FETCH_SIZE / WRITE_SIZE are in KiB.
"""
import collections
import csv
import glob
import os
import sys

print("kernel,counter,launches,avg_value")
for root in sys.argv[1:]:
    for path in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            per_dispatch[(name, row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
        agg = collections.defaultdict(list)
        for (name, ctr, _), v in per_dispatch.items():
            agg[(name, ctr)].append(v)
        for (name, ctr), vals in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            print('"%s",%s,%d,%.3f' % (name, ctr, len(vals), sum(vals) / len(vals)))

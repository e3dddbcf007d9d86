#!/usr/bin/env python3
"""tools/slim_counters.py <dir>: rewrite every rocprofv3 *counter_collection.csv under <dir> with only what tools/summarize_profiles.py reads —
the rows of mppi:: kernels, columns Kernel_Name (up to its argument list), Counter_Name, Counter_Value — and drop the per-dispatch
*kernel_trace.csv of the counter passes. gpurun copies at most 64 MiB back from a box; five workloads' raw passes exceed that."""
import csv
import glob
import os
import sys

for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows = []
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "mppi::" in name:
            rows.append({"Kernel_Name": name.split("(")[0], "Counter_Name": row["Counter_Name"], "Counter_Value": row["Counter_Value"]})
    with open(f, "w", newline="") as out:
        w = csv.DictWriter(out, ["Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        w.writerows(rows)
for f in glob.glob(os.path.join(sys.argv[1], "pmc_*", "**", "*kernel_trace.csv"), recursive=True):
    os.remove(f)

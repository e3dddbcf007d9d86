#!/usr/bin/env python3
"""Summarise gpurun_out/profiles_<tag>/ (tools/collect_profiles.sh) into the committed evidence:
   profiles/<tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of bench.py
   profiles/<tag>_pmc_summary.json     per-kernel averages of the PMC counters (separate passes)
   profiles/traffic_latest.json        HBM bytes per launch of the dominant kernel, read by bench.py
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads (x2 correction); WRITE_SIZE is exact for dword stores."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))

summary = collections.defaultdict(dict)
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if "mppi::" not in name:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, cs in acc.items():
        for c, v in cs.items():
            summary[name][c] = sum(v) / len(v)
            summary[name]["launches_" + c] = len(v)

dominant = next((k for k in summary if "k_rollout" in k), None)
out = {"tag": tag, "kernels": summary}
if dominant and "FETCH_SIZE" in summary[dominant] and "WRITE_SIZE" in summary[dominant]:
    fetch_kib, write_kib = summary[dominant]["FETCH_SIZE"], summary[dominant]["WRITE_SIZE"]
    hbm = (2.0 * fetch_kib + write_kib) * 1024.0
    traffic = {"kernel": dominant, "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
               "fetch_correction": "x2 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B)",
               "hbm_bytes_per_launch": hbm, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tag " + tag}
    out["traffic"] = traffic
    if "k_rollout_pc" in dominant:  # bench.py's default workload reads this file
        json.dump(traffic, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
b = os.path.join(src, "bench_under_profiler.json")
if os.path.exists(b) and os.path.getsize(b):
    out["bench_line_under_profiler"] = json.loads(open(b).read())
json.dump(out, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out.get("traffic", {}), indent=1))
print("kernels:", list(summary))

#!/usr/bin/env python3
"""tools/kernel_duration_hist.py <rocprofv3 *kernel_trace.csv> <kernel name substring> [bin_ns]: the distribution of one kernel's dispatch
durations in a trace (is an average made of one mode or two?), in launch order by quarters as well."""
import collections
import csv
import sys

f, pat = sys.argv[1], sys.argv[2]
binw = int(sys.argv[3]) if len(sys.argv) > 3 else 250
d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
d.sort()
dur = [x[1] for x in d]
n = len(dur)
print("%d launches of %s: mean %.0f ns, median %d, min %d, p95 %d" % (n, pat, sum(dur) / n, sorted(dur)[n // 2], min(dur), sorted(dur)[int(.95 * n)]))
for q in range(4):
    part = dur[q * n // 4:(q + 1) * n // 4]
    print("  quarter %d of the run: mean %.0f ns" % (q + 1, sum(part) / len(part)))
h = collections.Counter(v // binw * binw for v in dur)
for b in sorted(h):
    if h[b] * 200 >= n:
        print("  %6d ns  %5d  %s" % (b, h[b], "#" * (h[b] * 120 // n)))
# two modes? how long does the run stay in one (launch-to-launch flips: placement; long stretches: clocks / power management)
med = sorted(dur)[n // 2]
lo = sorted(dur)[n // 10]
thr = lo * 1.05
mode = [v > thr for v in dur]
runs, cur = [], 1
for a, b in zip(mode, mode[1:]):
    if a == b:
        cur += 1
    else:
        runs.append(cur); cur = 1
runs.append(cur)
print("  threshold %.0f ns (1.05 x the 10th percentile): %.1f %% of the launches above it; %d stretches, mean length %.1f launches, longest %d" % (
    thr, 100.0 * sum(mode) / n, len(runs), sum(runs) / len(runs), max(runs)))
# and the neighbour in the trace: does the OTHER kernel of the step slow down in the same stretches?

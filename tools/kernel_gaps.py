"""Gaps between consecutive kernels of the bench step from a rocprofv3 --kernel-trace CSV: for every kernel class the mean
duration, and the mean idle time between the end of the previous kernel and its start (same stream, steady state)."""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list)
gap = collections.defaultdict(list)
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:60]
    dur[name].append((e - s) / 1e3)
    if prev_end is not None and 0 <= s - prev_end < 50_000:
        gap[name].append((s - prev_end) / 1e3)
    prev_end = e
for name in sorted(dur, key=lambda n: -sum(dur[n]))[:14]:
    d, g = dur[name], gap[name] or [0.0]
    print("%-60s n=%5d  dur %7.2f us  gap before %5.2f us" % (name, len(d), sum(d) / len(d), sum(g) / len(g)))

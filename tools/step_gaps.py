"""Idle gaps > 5 us between consecutive kernels in a rocprofv3 --kernel-trace CSV (steady state: the last third of the trace):
what ran before and after each, how long the GPU idled."""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.3, 0.6)
rows = rows[int(len(rows) * lo):int(len(rows) * hi)]
gaps = collections.defaultdict(list)
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None:
        g = (s - int(prev["End_Timestamp"])) / 1e3
        if g > 5:
            gaps[(prev["Kernel_Name"][:50], r["Kernel_Name"][:50])].append(g)
    prev = r
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
print("window: %d kernels, span %.0f us, busy %.0f us (%.1f %%)" % (len(rows), span, busy, 100 * busy / span))
for (a, b), g in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%4d x %7.1f us  after %-50s before %s" % (len(g), sum(g) / len(g), a, b))

"""Timeline of ONE steady-state request (image encode at batch 1 + top-10) from a rocprofv3 --kernel-trace CSV of
tools/single_request.py: every kernel of the request in launch order with its duration and the idle time before it."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a request starts at the patch-embedding kernel; take the one in the middle of the trace that is followed by a query
starts = [i for i, r in enumerate(rows) if "patch" in r["Kernel_Name"].lower() or "im2col" in r["Kernel_Name"].lower()]
if not starts:
    starts = [0]
pick = starts[min(len(starts) - 1, int(sys.argv[2]) if len(sys.argv) > 2 else 150)]
nxt = [s for s in starts if s > pick]
end = nxt[0] if nxt else len(rows)
t0 = int(rows[pick]["Start_Timestamp"])
prev = None
busy = 0.0
for r in rows[pick:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += (e - s) / 1e3
    print("%8.1f us  +%6.2f gap  %7.2f us  %s  grid %s wg %s" % ((s - t0) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, (e - s) / 1e3,
                                                    r["Kernel_Name"][:70], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
    prev = e
print("request: %d kernels, span %.1f us, busy %.1f us" % (end - pick, (prev - t0) / 1e3, busy))
if nxt:
    print("next request starts %.1f us after this one's last kernel ended" % ((int(rows[nxt[0]]["Start_Timestamp"]) - prev) / 1e3))

#!/usr/bin/env python3
"""Timeline of ONE steady-state request out of a rocprofv3 --kernel-trace of tools/single_request.py: per kernel its duration and
the gap behind its predecessor. usage: request_timeline.py <kernel_trace.csv> [launches per request, default: detected]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a request starts at im2col
starts = [i for i, n in enumerate(names) if "im2col" in n]
mid = starts[len(starts) // 3]
nxt = starts[len(starts) // 3 + 1]
seg = rows[mid:nxt]
tot_d = tot_g = 0.0
prev = None
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", n))[:64]
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{short(r['Kernel_Name']):64s} {(e - s) / 1e3:6.2f} us  gap {gap:6.2f}")
    tot_d += (e - s) / 1e3
    tot_g += gap
    prev = e
print(f"{len(seg)} launches, kernels {tot_d:.1f} us + gaps {tot_g:.1f} us = {tot_d + tot_g:.1f} us "
      f"(first start to last end {(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3:.1f})")

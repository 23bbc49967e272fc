#!/usr/bin/env python3
"""Q = 1 top-10 over a 100k x 512 f16 index (the query half of one request): ms per query under scan / merge options."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from mmiss_amd.index import FlatIndex
rows = torch.randn(100000, 512, device="cuda")
index = FlatIndex(512, dtype="f16", device=0)
index.add(rows, np.arange(100000, dtype=np.int64))
q = torch.randn(1, 512, device="cuda")
def run(n=300):
    for _ in range(20): index.query(q, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): index.query(q, 10)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ref = index.query(q, 10)
for slabs in (1024, 512, 256, 128):
    for two in (64, 4096):
        _lib.set_option("scan_max_slabs", slabs); _lib.set_option("merge_two_level_min", two)
        ms = min(run(), run())
        out = index.query(q, 10)
        same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
        _lib.prof_filter(None, 1); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(5): index.query(q, 10)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        k = {p["kernel"]: (p["launches"] // 5, round(1e3 * p["ms"] / p["launches"], 1)) for p in _lib.prof_read()}
        print(f"scan_max_slabs={slabs:5d} merge_two_level_min={two:5d}: {ms:.4f} ms/query same={same} {k}")

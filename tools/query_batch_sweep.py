#!/usr/bin/env python3
"""Cosine top-10 over an N x 512 f16 index against the query batch size: looks for cliffs where the path changes
(scan kernel up to 16 queries, 128-row score GEMM up to 128, strip-persistent 256x256 score GEMM above)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd.index import FlatIndex  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
D = 512
idx = FlatIndex(D, "f16", device=0, capacity=N)
for c in range(0, N, 1_000_000):
    n = min(1_000_000, N - c)
    rows = torch.nn.functional.normalize(torch.randn(n, D, device="cuda"), dim=1)
    idx.add(rows, np.arange(c, c + n, dtype=np.int64))
del rows
for Q in (1, 4, 8, 16, 17, 24, 32, 48, 64, 96, 128, 129, 192, 256, 512, 1024):
    q = torch.nn.functional.normalize(torch.randn(Q, D, device="cuda"), dim=1)
    for _ in range(2):
        idx.query(q, 10)
    torch.cuda.synchronize()
    n_it = 5
    t0 = time.perf_counter()
    for _ in range(n_it):
        idx.query(q, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_it
    print({"Q": Q, "ms": round(dt * 1e3, 3), "ms_per_query": round(dt * 1e3 / Q, 4), "gpairs_per_s": round(Q * N / dt / 1e9, 1)}, flush=True)

#!/usr/bin/env python3
"""The second half of the metric under rocprofv3: cosine top-10 over 10M x 512 f16 rows at Q = 1 (streaming scan) and
Q = 1024 (score GEMM with threshold-filtered selection), 5 passes each after one warm-up."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd.index import FlatIndex  # noqa: E402

N, D = 10_000_000, 512
PASSES = 6   # queries per Q leg (one warm-up + 5); tools/traffic_from_pmc.py records it so per-batch traffic is PMC total / PASSES
idx = FlatIndex(D, "f16", capacity=N)
g = torch.Generator(device="cuda").manual_seed(4)
for r0 in range(0, N, 1_000_000):
    idx.add(torch.randn(1_000_000, D, device="cuda", generator=g), np.arange(r0, r0 + 1_000_000, dtype=np.int64))
for Q in (1, 1024):
    q = torch.randn(Q, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    for _ in range(PASSES):
        idx.query(q, 10)
    torch.cuda.synchronize()
print(idx.guard_stats())

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mmiss_amd
from mmiss_amd import _lib
from mmiss_amd.index import FlatIndex
for N in (100000, 1000000):
    rows = torch.nn.functional.normalize(torch.randn(N, 512, device="cuda"), dim=1)
    idx = FlatIndex(512, "f16", device=0, capacity=N); idx.add(rows, np.arange(N, dtype=np.int64)); del rows
    for Q in (256, 512):
        q = torch.nn.functional.normalize(torch.randn(Q, 512, device="cuda"), dim=1)
        for minq in (129, 100000):
            _lib.set_option("score_big_min_q", minq)
            for _ in range(3): idx.query(q, 10)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): idx.query(q, 10)
            torch.cuda.synchronize()
            print({"N": N, "Q": Q, "kernel": "256x256 strip" if minq == 129 else "128-row", "us": round((time.perf_counter() - t0) / 20 * 1e6, 1)}, flush=True)

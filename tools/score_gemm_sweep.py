#!/usr/bin/env python3
"""Q = 1024 score GEMM (strip-persistent 256x256 kernel) vs index size: does the rate depend on where the index rows
come from (L2 / Infinity Cache / HBM)?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402
from mmiss_amd.index import FlatIndex  # noqa: E402

D, Q = 512, 1024
q = torch.nn.functional.normalize(torch.randn(Q, D, device="cuda"), dim=1)
for N in (16384, 65536, 131072, 262144, 1048576, 4194304):
    rows = torch.nn.functional.normalize(torch.randn(N, D, device="cuda"), dim=1)
    idx = FlatIndex(D, "f16", device=0, capacity=N)
    idx.add(rows, np.arange(N, dtype=np.int64))
    del rows
    for strip in (0, 1):
        _lib.set_option("score_strip", strip)
        for _ in range(3):
            idx.query(q, 10)
        torch.cuda.synchronize()
        _lib.prof_filter(None, 1)
        _lib.prof_enable(True)
        _lib.prof_reset()
        for _ in range(10):
            idx.query(q, 10)
        torch.cuda.synchronize()
        r = {k["kernel"]: k for k in _lib.prof_read()}
        _lib.prof_enable(False)
        g = r["score_gemm_f16"]
        us = g["ms"] / g["launches"] * 1e3
        print({"N": N, "MB": N * D * 2 >> 20, "strip": "auto" if strip == 0 else strip, "score_gemm_us": round(us, 1),
               "tflops": round(2.0 * Q * N * D / (us * 1e-6) / 1e12, 1)}, flush=True)
    _lib.set_option("score_strip", 0)
    del idx

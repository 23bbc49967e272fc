#!/usr/bin/env python3
"""Why does the widen pass's threshold GEMM (sweep_gemm_f16: gemm256s_kernel<f16, FILTER = 2>) take twice the time of the first
pass's score GEMM on the same operands (78 vs 39 us at Q = 256, 100k x 512 f16: bench line, round 6)? Per-kernel table of one
query batch on (a) a clustered index (every query widened, ~60 rows per query pass the threshold) and (b) a random index with
the guard forced (every query widened, ~10 rows pass)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402
from mmiss_amd.index import FlatIndex  # noqa: E402

N, D, Q = 100_000, 512, 256
g = torch.Generator(device="cuda").manual_seed(1)
centre = torch.randn(1, D, device="cuda", generator=g)
for name, rows, force in (("clustered (cos ~0.99)", centre + 0.1 * torch.randn(N, D, device="cuda", generator=g), 0),
                          ("random, guard forced", torch.randn(N, D, device="cuda", generator=g), 1)):
    idx = FlatIndex(D, "f16", device=0, capacity=N)
    idx.add(rows, np.arange(N, dtype=np.int64))
    q = rows[:Q].contiguous()
    _lib.set_option("guard_force", force)
    for _ in range(3):
        idx.query(q, 10)
    torch.cuda.synchronize()
    gs0 = idx.guard_stats()
    _lib.prof_filter(None, 1)
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(10):
        idx.query(q, 10)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    gs1 = idx.guard_stats()
    _lib.set_option("guard_force", 0)
    print(name, {k: (gs1[k] - gs0[k]) / 10 for k in gs1})
    for k in sorted(_lib.prof_read(), key=lambda k: -k["ms"]):
        print(f"   {k['kernel']:28s} {k['launches'] // 10:2d} x {k['ms'] / k['launches'] * 1e3:7.1f} us", flush=True)
    idx.close()

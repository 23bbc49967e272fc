#!/usr/bin/env python3
"""Same-box A/B of library options on ONE query shape (default: the bench step's 256 queries x 100k x 512 f16 rows, k = 10).
Usage: query_option_ab.py base score_big_min_q=257 score_strip=2 ...   (env Q, N to change the shape)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from mmiss_amd.index import FlatIndex

Q, N, D = int(os.environ.get("Q", "256")), int(os.environ.get("N", "100000")), 512
sets = [(a, {} if a == "base" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")}) for a in (sys.argv[1:] or ["base"])]
keys = sorted({k for _, d in sets for k in d})
defaults = {"score_big_min_q": 129, "score_strip": 0, "score_filter": 1}
g = torch.Generator(device="cuda").manual_seed(1)
ix = FlatIndex(D, "f16", capacity=N)
ix.add(torch.randn(N, D, device="cuda", generator=g), np.arange(N, dtype=np.int64))
q = torch.randn(Q, D, device="cuda", generator=g)
ref = None
res = {n: [] for n, _ in sets}
for rep in range(5):
    for name, d in sets:
        for k in keys:
            _lib.set_option(k, d.get(k, defaults.get(k, 0)))
        for _ in range(3):
            out = ix.query(q, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = ix.query(q, 10)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 20 * 1e6)
        if ref is None:
            ref = out
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), name
for name, d in sets:
    for k in keys:
        _lib.set_option(k, d.get(k, defaults.get(k, 0)))
    _lib.prof_filter(None, 1); _lib.prof_enable(True); _lib.prof_reset()
    for _ in range(10):
        ix.query(q, 10)
    torch.cuda.synchronize()
    ks = " ".join(f"{p['kernel']}={p['ms'] / p['launches'] * 1e3:.1f}" for p in _lib.prof_read())
    _lib.prof_enable(False)
    w = sorted(res[name])
    print(f"{name:40s} median {w[len(w)//2]:7.1f} us per query call (Q={Q}, N={N})  {ks}", flush=True)

#!/usr/bin/env python3
"""A/B of the Q<=16 scan kernel knobs on a 10M x 512 f16 index (run on the MI355X box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from mmiss_amd.index import FlatIndex

N, D = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 512
idx = FlatIndex(D, "f16", capacity=N)
g = torch.Generator(device="cuda").manual_seed(0)
for r0 in range(0, N, 1_000_000):
    n = min(1_000_000, N - r0)
    idx.add(torch.randn(n, D, device="cuda", generator=g), np.arange(r0, r0 + n, dtype=np.int64))
for Q in (1, 16):
    q = torch.randn(Q, D, device="cuda")
    for group in (8, 16):
        for rounds in (1, 2, 4):
            for max_slabs in (1024, 4096):
                _lib.set_option("scan_group", group); _lib.set_option("scan_rounds", rounds); _lib.set_option("scan_max_slabs", max_slabs)
                for _ in range(3): idx.query(q, 10)
                torch.cuda.synchronize()
                _lib.prof_reset(); _lib.prof_enable(True)
                for _ in range(10): idx.query(q, 10)
                torch.cuda.synchronize(); _lib.prof_enable(False)
                p = {x["kernel"]: x for x in _lib.prof_read()}
                sc = p["scan_topk_f16"]; ms = sc["ms"] / sc["launches"]
                other = sum(x["ms"] / x["launches"] for k, x in p.items() if k != "scan_topk_f16")
                print(f"Q={Q} group={group} rounds={rounds} max_slabs={max_slabs}: scan {ms*1e3:.0f} us = {N*D*2/ms/1e9:.2f} TB/s, other {other*1e3:.0f} us", flush=True)

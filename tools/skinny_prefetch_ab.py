#!/usr/bin/env python3
"""VERDICT r5 next #8, ONE experiment: in the one-request path every skinny GEMM touches the weight lines of the NEXT skinny
launch (GemmEpi::pf, option skinny_prefetch) so that launch's first round trip is an L2 hit. ViT-B/32 image encode at batch 1 +
top-10 over 100k x 512 f16, device-resident, alternating off / on / off / on (300 requests each); the embedding must not
change by a bit (the prefetched values are never used)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict  # noqa: E402
from mmiss_amd.index import FlatIndex  # noqa: E402

n = 300
enc = ClipEncoder(VIT_B32, device=0, max_batch_image=8, max_batch_text=8)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
rows = torch.nn.functional.normalize(torch.randn(100000, 512, device="cuda"), dim=1)
index = FlatIndex(512, dtype="f16", device=0)
index.add(rows, np.arange(100000, dtype=np.int64))
px = torch.randn(1, 3, 224, 224, device="cuda")
emb = torch.empty(1, 512, device="cuda")


def run(with_query):
    for _ in range(20):
        enc.encode_image(px, out=emb)
        if with_query:
            index.query(emb, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        enc.encode_image(px, out=emb)
        if with_query:
            index.query(emb, 10)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ref = None
for rep in range(3):
    for on in (0, 1):
        _lib.set_option("skinny_prefetch", on)
        e, r = run(False), run(True)
        if ref is None:
            ref = emb.clone()
        same = bool(torch.equal(emb, ref))
        print("skinny_prefetch = %d: encode %.4f ms, encode + top-10 %.4f ms, embedding bits unchanged: %s" % (on, e, r, same), flush=True)
_lib.set_option("skinny_prefetch", 0)

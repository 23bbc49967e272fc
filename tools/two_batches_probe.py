#!/usr/bin/env python3
"""Two (or more) FULL batches in flight: one encoder handle + index per HIP stream, each encoding its own 256 images and
querying its own embeddings, launched from one thread per stream (ctypes drops the GIL inside the library). Does the tail
of one batch's kernel (its last, partly filled round of tiles + its store burst) fill with the other batch's workgroups?
Reported: images/s over all streams, against the same work run by one stream."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mmiss_amd  # noqa
from mmiss_amd.encoder import ClipEncoder, VIT_B32, random_state_dict
from mmiss_amd.index import FlatIndex

B, N, D = 256, 100_000, 512
W = random_state_dict(VIT_B32, 0)
STEPS = int(os.environ.get("STEPS", "24"))
with_query = os.environ.get("QUERY", "1") == "1"

shared = os.environ.get("SHARED_INDEX", "1") == "1"   # one index handle for all lanes (its calls serialise)
_ix = []

def make(i):
    e = ClipEncoder(VIT_B32, max_batch_image=B, max_batch_text=8)
    e.load_state_dict(W)
    g = torch.Generator(device="cuda").manual_seed(7 + i)
    if shared and _ix:
        ix = _ix[0]
    else:
        ix = FlatIndex(D, "f16", capacity=N)
        ix.add(torch.randn(N, D, device="cuda", generator=g), np.arange(N, dtype=np.int64))
        _ix.append(ix)
    return e, ix, torch.cuda.Stream(), torch.randn(B, 3, 224, 224, device="cuda", generator=g), torch.empty(B, D, device="cuda")

def worker(ctx, n, barrier):
    e, ix, s, px, out = ctx
    torch.cuda.set_device(0)
    barrier.wait()
    with torch.cuda.stream(s):
        for _ in range(n):
            e.encode_image(px, out=out)
            if with_query:
                ix.query(out, 10)
    s.synchronize()

for nstream in (1, 2, 3, 1, 2):
    ctxs = [make(i) for i in range(nstream)]
    for warm in (True, False):
        n = 4 if warm else STEPS // nstream
        bar = threading.Barrier(nstream + 1)
        th = [threading.Thread(target=worker, args=(c, n, bar)) for c in ctxs]
        for t in th:
            t.start()
        torch.cuda.synchronize()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"streams={nstream} query={with_query} shared_index={shared}: {n * nstream} steps in {dt*1e3:.1f} ms -> {dt/(n*nstream)*1e3:.3f} ms/step, "
          f"{n * nstream * B / dt:.0f} img/s", flush=True)
    for c in ctxs:
        c[0].close()
    for ix in _ix:
        ix.close()
    _ix.clear()
    del ctxs

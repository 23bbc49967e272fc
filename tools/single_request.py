#!/usr/bin/env python3
"""One request at a time (the reference's regime): ViT-B/32 image encode at batch 1 + top-10 over a 100k x 512 f16
index, device-resident. Prints ms per request; run under `rocprofv3 --kernel-trace --stats` for the kernel table."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict  # noqa: E402
from mmiss_amd.index import FlatIndex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
enc = ClipEncoder(VIT_B32, device=0, max_batch_image=8, max_batch_text=8)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
rows = torch.nn.functional.normalize(torch.randn(100000, 512, device="cuda"), dim=1)
index = FlatIndex(512, dtype="f16", device=0)
index.add(rows, np.arange(100000, dtype=np.int64))
px = torch.randn(1, 3, 224, 224, device="cuda")
emb = torch.empty(1, 512, device="cuda")
for _ in range(10):
    enc.encode_image(px, out=emb)
    index.query(emb, 10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    enc.encode_image(px, out=emb)
    index.query(emb, 10)
torch.cuda.synchronize()
print({"ms_per_request": round((time.perf_counter() - t0) / n * 1e3, 4), "requests": n})
t0 = time.perf_counter()
for _ in range(n):
    enc.encode_image(px, out=emb)
torch.cuda.synchronize()
print({"ms_per_encode_only": round((time.perf_counter() - t0) / n * 1e3, 4)})

# ---- the encode captured in a HIP graph (torch.cuda.CUDAGraph on the library's stream = torch's current stream). The query is
# not capturable since round 2: its exactness guard reads one flag back to decide on the host whether to widen.
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            enc.encode_image(px, out=emb)
        torch.cuda.synchronize()
        ref = emb.clone()
        with torch.cuda.graph(g, stream=s):
            enc.encode_image(px, out=emb)
    torch.cuda.synchronize()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    print({"ms_per_encode_graph_replay": round((time.perf_counter() - t0) / n * 1e3, 4), "same_result": bool(torch.equal(emb, ref))})
except Exception as e:  # noqa: BLE001
    print({"graph_capture_failed": repr(e)})

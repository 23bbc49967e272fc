#!/usr/bin/env python3
"""Per-kernel table of the ViT-B/32 text tower at batch 256 x 77 tokens (library HIP-event profiler)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
enc = ClipEncoder(VIT_B32, device=0, max_batch_image=8, max_batch_text=B)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
rng = np.random.default_rng(0)
ids = np.full((B, 77), 49407, dtype=np.int32)
ids[:, 0] = 49406
ids[:, 1:76] = rng.integers(0, 49406, size=(B, 75))
ids_d = torch.from_numpy(ids).cuda()
out = torch.empty(B, 512, device="cuda")
for _ in range(3):
    enc.encode_text(ids_d, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    enc.encode_text(ids_d, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print({"ms_per_batch": round(dt * 1e3, 3), "texts_per_s": round(B / dt, 1), "tflops": round(B * 5.960e9 / dt / 1e12, 1)})
_lib.prof_filter(None, 1)
_lib.prof_enable(True)
_lib.prof_reset()
for _ in range(5):
    enc.encode_text(ids_d, out=out)
torch.cuda.synchronize()
rows = sorted(_lib.prof_read(), key=lambda k: -k["ms"])
_lib.prof_enable(False)
for k in rows:
    n = k["launches"] // 5
    us = k["ms"] / k["launches"] * 1e3
    print(f'{k["kernel"]:28s} x{n:3d}  {us:8.2f} us  {k["flops"] / k["launches"] / (us * 1e-6) / 1e12 if us else 0:7.1f} TF  {k["bytes"] / k["launches"] / (us * 1e-6) / 1e9 if us else 0:8.1f} GB/s')

#!/usr/bin/env python3
"""ViT-B/32 text tower (77 tokens) against batch size; device-resident ids."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict  # noqa: E402

enc = ClipEncoder(VIT_B32, device=0, max_batch_image=8, max_batch_text=256)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
rng = np.random.default_rng(0)
for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256):
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
    print({"batch": B, "rows": B * 77, "ms": round(dt * 1e3, 3), "texts_per_s": round(B / dt, 1)}, flush=True)

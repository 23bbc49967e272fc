#!/usr/bin/env python3
"""ViT-L/14 (the reference checkpoint's vision geometry) image encode against batch size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd.encoder import LONGCLIP_L14, ClipEncoder, random_state_dict  # noqa: E402

enc = ClipEncoder(LONGCLIP_L14, device=0, max_batch_image=128, max_batch_text=8)
enc.load_state_dict(random_state_dict(LONGCLIP_L14, seed=0))
for B in (1, 2, 4, 8, 16, 32, 64, 128):
    x = torch.randn(B, 3, 224, 224, device="cuda")
    out = torch.empty(B, 768, device="cuda")
    for _ in range(3):
        enc.encode_image(x, out=out)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        enc.encode_image(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print({"batch": B, "rows": B * 257, "ms": round(dt * 1e3, 3), "images_per_s": round(B / dt, 1),
           "tflops": round(B * 162.03e9 / dt / 1e12, 1)}, flush=True)

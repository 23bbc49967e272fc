#!/usr/bin/env python3
"""ViT-B/32 image encode throughput and latency against batch size (device-resident pixels), to see where the small-M
(skinny) and tiled GEMM paths hand over."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict  # noqa: E402

enc = ClipEncoder(VIT_B32, device=0, max_batch_image=256, max_batch_text=8)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
if os.environ.get("MMISS_LN_MODE"):
    enc.set_fuse_ln(int(os.environ["MMISS_LN_MODE"]))  # 0 separate LayerNorm kernels, 1 operand-fused, 2 folded
for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256):
    x = torch.randn(B, 3, 224, 224, device="cuda")
    out = torch.empty(B, 512, device="cuda")
    for _ in range(5):
        enc.encode_image(x, out=out)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        enc.encode_image(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print({"batch": B, "rows": B * 50, "ms": round(dt * 1e3, 3), "images_per_s": round(B / dt, 1)}, flush=True)

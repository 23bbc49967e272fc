#!/usr/bin/env python3
"""The reference's actual checkpoint geometry (CLIP_MODEL_ID = LongCLIP-GmP-ViT-L-14, backend/app/utils.py:16-17,41-45):
ViT-L/14 vision tower (257 tokens, d=1024, 24 layers) at batch 128 and the 248-token text tower (d=768, 12 layers) at
batch 64, random-init weights; BASELINE config 5 shapes in bf16."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402
from mmiss_amd.encoder import LONGCLIP_L14, ClipEncoder, random_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
BT = 64
t0 = time.perf_counter()
enc = ClipEncoder(LONGCLIP_L14, device=0, max_batch_image=B, max_batch_text=BT)
enc.load_state_dict(random_state_dict(LONGCLIP_L14, seed=0))
if os.environ.get("PRECISION"):   # "bf16-f32resid" = f32 residual stream at every batch size, "fp8"
    enc.set_precision(os.environ["PRECISION"])
print({"load_s": round(time.perf_counter() - t0, 1)}, flush=True)
x = torch.randn(B, 3, 224, 224, device="cuda")
out = torch.empty(B, 768, device="cuda")
for _ in range(2):
    enc.encode_image(x, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    enc.encode_image(x, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print({"l14_images_per_s": round(B / dt, 1), "ms_per_batch": round(dt * 1e3, 2), "tflops": round(B * 162.03e9 / dt / 1e12, 1)}, flush=True)
ids = np.full((BT, 248), 49407, dtype=np.int32)
ids[:, 0] = 49406
ids[:, 1:247] = np.random.default_rng(0).integers(0, 49406, size=(BT, 246))
ids_d = torch.from_numpy(ids).cuda()
tout = torch.empty(BT, 768, device="cuda")
for _ in range(2):
    enc.encode_text(ids_d, out=tout)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    enc.encode_text(ids_d, out=tout)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print({"longclip_texts_per_s_T248": round(BT / dt, 1), "ms_per_batch": round(dt * 1e3, 2), "tflops": round(BT * 44.39e9 / dt / 1e12, 1)}, flush=True)
# one request at a time
x1 = x[:1].contiguous()
o1 = torch.empty(1, 768, device="cuda")
for _ in range(3):
    enc.encode_image(x1, out=o1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    enc.encode_image(x1, out=o1)
torch.cuda.synchronize()
print({"l14_single_image_ms": round((time.perf_counter() - t0) / 20 * 1e3, 3)}, flush=True)
_lib.prof_filter(None, 1)
_lib.prof_enable(True)
_lib.prof_reset()
for _ in range(2):
    enc.encode_image(x, out=out)
torch.cuda.synchronize()
for k in sorted(_lib.prof_read(), key=lambda k: -k["ms"])[:8]:
    us = k["ms"] / k["launches"] * 1e3
    print(f'{k["kernel"]:26s} x{k["launches"] // 2:3d} {us:9.1f} us {k["flops"] / k["launches"] / (us * 1e-6) / 1e12:7.1f} TF')
_lib.prof_enable(True)
_lib.prof_reset()
for _ in range(5):
    enc.encode_image(x1, out=o1)
torch.cuda.synchronize()
print("--- one image")
for k in sorted(_lib.prof_read(), key=lambda k: -k["ms"])[:10]:
    us = k["ms"] / k["launches"] * 1e3
    print(f'{k["kernel"]:26s} x{k["launches"] // 5:3d} {us:9.1f} us  total {us * (k["launches"] // 5):8.1f}')
_lib.prof_enable(False)

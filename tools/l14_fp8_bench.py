#!/usr/bin/env python3
"""ViT-L/14 geometry, bs 128: images/s in bf16 and fp8, with the per-kernel table of the fp8 encode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from mmiss_amd.encoder import LONGCLIP_L14, ClipEncoder, random_state_dict

enc = ClipEncoder(LONGCLIP_L14, device=0, max_batch_image=128, max_batch_text=8)
enc.load_state_dict(random_state_dict(LONGCLIP_L14, seed=0))
x = torch.randn(128, 3, 224, 224, device="cuda")
o = torch.empty(128, 768, device="cuda")
for prec in ("bf16", "fp8"):
    enc.set_precision(prec)
    for _ in range(3):
        enc.encode_image(x, out=o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        enc.encode_image(x, out=o)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
    print(prec, round(128 / dt, 1), "images/s", round(dt * 1e3, 2), "ms")
_lib.prof_reset(); _lib.prof_enable(True)
enc.encode_image(x, out=o); torch.cuda.synchronize()
_lib.prof_enable(False)
for k in sorted(_lib.prof_read(), key=lambda k: -k["ms"])[:9]:
    print(f'{k["kernel"]:26s} x{k["launches"]:3d} {1e3*k["ms"]/k["launches"]:8.1f} us {k["flops"]/k["ms"]/1e9 if k["flops"] else 0:8.1f} TF')

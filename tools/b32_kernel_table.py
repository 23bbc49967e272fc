"""Per-kernel table (instrumented replay: HIP events around every launch) of one ViT-B/32 bs-256 encode, for an option set:
python tools/b32_kernel_table.py [precision=fp8] [key=value ...]; then the wall time of 20 encodes without events."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict
enc = ClipEncoder(VIT_B32, device=0, max_batch_image=256, max_batch_text=8)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
x = torch.randn(256, 3, 224, 224, device="cuda")
o = torch.empty(256, 512, device="cuda")
prec = "fp8"
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "precision":
        prec = v
    else:
        _lib.set_option(k, int(v))
enc.set_precision(prec)
for _ in range(3):
    enc.encode_image(x, out=o)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    enc.encode_image(x, out=o)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
_lib.prof_filter(None, 1)
_lib.prof_reset()
_lib.prof_enable(True)
for _ in range(5):
    enc.encode_image(x, out=o)
_lib.prof_enable(False)
tot = 0.0
for k in sorted(_lib.prof_read(), key=lambda k: -k["ms"]):
    tot += k["ms"] / 5
    print(f"{k['kernel']:36s} {k['launches'] // 5:3d} x {k['ms'] / k['launches'] * 1e3:7.1f} us = {k['ms'] / 5:6.3f} ms" +
          (f"  {k['flops'] / k['ms'] / 1e9:7.1f} TF" if k["flops"] else ""))
print(f"sum {tot:.3f} ms; encode without events {wall:.3f} ms = {256 / wall:.1f} k images/s  ({' '.join(sys.argv[1:]) or 'defaults'})", flush=True)

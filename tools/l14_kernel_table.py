"""Per-kernel table (instrumented replay: HIP events around every launch) of one ViT-L/14 bs-128 fp8 encode, for an option set:
python tools/l14_kernel_table.py [precision=bf16] [key=value ...]"""
import sys
sys.path.insert(0, "/root/repo")
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from mmiss_amd.encoder import LONGCLIP_L14, ClipEncoder, random_state_dict
enc = ClipEncoder(LONGCLIP_L14, device=0, max_batch_image=128, max_batch_text=8)
enc.load_state_dict(random_state_dict(LONGCLIP_L14, seed=0))
x = torch.randn(128, 3, 224, 224, device="cuda")
o = torch.empty(128, 768, device="cuda")
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
_lib.prof_filter(None, 1)
_lib.prof_reset()
_lib.prof_enable(True)
enc.encode_image(x, out=o)
_lib.prof_enable(False)
tot = 0.0
for k in sorted(_lib.prof_read(), key=lambda k: -k["ms"]):
    tot += k["ms"]
    print(f"{k['kernel']:36s} {k['launches']:3d} x {k['ms'] / k['launches'] * 1e3:7.1f} us = {k['ms']:6.3f} ms" +
          (f"  {k['flops'] / k['ms'] / 1e9:7.1f} TF" if k["flops"] else ""))
print(f"sum {tot:.3f} ms  ({' '.join(sys.argv[1:]) or 'defaults'})")

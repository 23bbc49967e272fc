import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mmiss_amd
from mmiss_amd import _lib
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict
enc = ClipEncoder(VIT_B32, device=0, max_batch_image=256, max_batch_text=8)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
x = torch.randn(256, 3, 224, 224, device="cuda"); out = torch.empty(256, 512, device="cuda")
for mode in (0, 2):
    enc.set_fuse_ln(mode)
    for _ in range(3): enc.encode_image(x, out=out)
    torch.cuda.synchronize()
    _lib.prof_filter(None, 1); _lib.prof_enable(True); _lib.prof_reset()
    for _ in range(10): enc.encode_image(x, out=out)
    torch.cuda.synchronize()
    rows = sorted(_lib.prof_read(), key=lambda k: -k["ms"]); _lib.prof_enable(False)
    print("mode", mode, "sum_us_per_encode", round(sum(k["ms"] for k in rows) / 10 * 1e3, 1))
    for k in rows[:9]:
        print(f'  {k["kernel"]:26s} x{k["launches"] // 10:3d} {k["ms"] / k["launches"] * 1e3:8.2f} us')

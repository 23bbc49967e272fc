"""ViT-B/32 bs-256 encode (the headline's encode half) against the band height of the persistent GEMMs' tile order (option gemm_p256_band)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict
enc = ClipEncoder(VIT_B32, device=0, max_batch_image=256, max_batch_text=8)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
x = torch.randn(256, 3, 224, 224, device="cuda")
o = torch.empty(256, 512, device="cuda")
def t(n=20):
    for _ in range(5): enc.encode_image(x, out=o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): enc.encode_image(x, out=o)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for rnd in range(3):
    for band in (8, 2, 3, 4, 5, 6, 10, 12, 16, 25, 50):
        _lib.set_option("gemm_p256_band", band)
        dt = t()
        print(f"band {band:3d}: {256/dt:9.1f} img/s {dt*1e3:6.3f} ms", flush=True)
_lib.set_option("gemm_p256_band", 8)

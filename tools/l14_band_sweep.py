"""ViT-L/14 bs-128 fp8 encode against the band height of the persistent GEMMs' tile order (option gemm_p256_band: row blocks per band)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from mmiss_amd.encoder import LONGCLIP_L14, ClipEncoder, random_state_dict
enc = ClipEncoder(LONGCLIP_L14, device=0, max_batch_image=128, max_batch_text=8)
enc.load_state_dict(random_state_dict(LONGCLIP_L14, seed=0))
x = torch.randn(128, 3, 224, 224, device="cuda")
o = torch.empty(128, 768, device="cuda")
enc.set_precision("fp8")
def t(n=8):
    for _ in range(3): enc.encode_image(x, out=o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): enc.encode_image(x, out=o)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for rnd in range(3):
    for band in (8, 6, 4, 16):
        _lib.set_option("gemm_p256_band", band)
        dt = t()
        print(f"band {band:3d}: {128/dt:8.1f} img/s {dt*1e3:6.2f} ms", flush=True)
_lib.set_option("gemm_p256_band", 8)

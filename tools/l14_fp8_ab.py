import os, sys, time
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
    for name, opts in (("base", {}), ("attention_long", {"attention_stream": 0}), ("LayerNorm folded (fp8_ln_fold)", {"fp8_ln_fold": 1}), ("ragged off", {"gemm_p256_ragged": 0}), ("p256 fp8 off", {"gemm_p256_fp8": 0})):
        for k, v in opts.items(): _lib.set_option(k, v)
        dt = t()
        for k in opts: _lib.set_option(k, {"ln_mxfp8_wide": 1, "gemm_p256_ragged": 1, "gemm_p256_fp8": 1, "attention_stream": 1, "fp8_ln_fold": 0}[k])
        print(f"{name:14s} {128/dt:8.1f} img/s {dt*1e3:6.2f} ms", flush=True)

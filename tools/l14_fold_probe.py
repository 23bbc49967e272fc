import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import mmiss_amd
from mmiss_amd import _lib
from mmiss_amd.encoder import ClipEncoder, LONGCLIP_L14, random_state_dict
dev = torch.device("cuda", 0)
BL = 128
enc = ClipEncoder(LONGCLIP_L14, device=0, max_batch_image=BL, max_batch_text=8)
enc.load_state_dict(random_state_dict(LONGCLIP_L14, seed=0))
x = torch.randn(BL, 3, 224, 224, device=dev); o = torch.empty(BL, 768, device=dev)
def t(fn, warm=3, it=8):
    for _ in range(warm): fn()
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / it)
    return best
res = {}
for opt in (1, 0):
    _lib.set_option("ln_fold_1024", opt)
    dt = t(lambda: enc.encode_image(x, out=o))
    res[f"bf16_fold{opt}"] = (round(BL / dt, 1), o.clone())
print({k: v[0] for k, v in res.items()}, "1-cos fold vs separate", float((1 - (res["bf16_fold1"][1] * res["bf16_fold0"][1]).sum(1)).max()))
_lib.set_option("ln_fold_1024", 1)
_lib.prof_filter(None, 1); _lib.prof_reset(); _lib.prof_enable(True)
enc.encode_image(x, out=o); torch.cuda.synchronize(); _lib.prof_enable(False)
for k in sorted(_lib.prof_read(), key=lambda p: -p["ms"]): print(f'{k["kernel"]:40s} {k["launches"]:4d} {1e3*k["ms"]/k["launches"]:8.1f} us')
ref = res["bf16_fold0"][1]
enc.set_precision("fp8")
for opt in (1, 0):
    _lib.set_option("fp8_outproj", opt)
    dt = t(lambda: enc.encode_image(x, out=o))
    print(f"fp8 fp8_outproj={opt}: {BL / dt:.1f} img/s, max 1-cos vs bf16 (separate LN) path {float((1 - (o * ref).sum(1)).max()):.3e}")
_lib.set_option("fp8_outproj", 1)
_lib.prof_reset(); _lib.prof_enable(True)
enc.encode_image(x, out=o); torch.cuda.synchronize(); _lib.prof_enable(False)
for k in sorted(_lib.prof_read(), key=lambda p: -p["ms"]): print(f'{k["kernel"]:40s} {k["launches"]:4d} {1e3*k["ms"]/k["launches"]:8.1f} us')

#!/usr/bin/env python3
"""GPU sweep of the GEMM tile order (option gemm_group_m: 0 = n fastest, 1 = m fastest, g>=2 = bands of g m-tiles)
on the ViT-B/32 layer shapes and on the whole bs-256 image encode (run on the MI355X box)."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict  # noqa: E402

lib = _lib.load()
M0 = 256 * 50
shapes = [("qkv", _lib.EPI_BIAS_BF16, 2304, 768, 192), ("out", _lib.EPI_BIAS_RESID_F32, 768, 768, 160),
          ("fc1", _lib.EPI_BIAS_QGELU_BF16, 3072, 768, 192), ("fc2", _lib.EPI_BIAS_RESID_F32, 768, 3072, 160)]
groups = [0, 1, 2, 3, 4, 6, 8, 12, 16, 24]
res = []
for name, epi, N, K, bm in shapes:
    M = (M0 + bm - 1) // bm * bm
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == _lib.EPI_BIAS_RESID_F32 else torch.bfloat16)
    for g in groups:
        _lib.set_option("gemm_group_m", g)
        ms = C.c_float(0)
        _lib.check(lib.mmiss_dbg_gemm_time(0, epi, bm, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), None,
                                           M, N, K, 0, 0, 50, C.byref(ms)))
        tf = 2.0 * M0 * N * K / (ms.value * 1e-3) / 1e12
        res.append({"gemm": name, "bm": bm, "group_m": g, "us": round(ms.value * 1e3, 2), "tflops": round(tf, 1)})
        print(res[-1], flush=True)

enc = ClipEncoder(VIT_B32, device=0)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
x = torch.randn(256, 3, 224, 224, device="cuda")
for g in groups:
    _lib.set_option("gemm_group_m", g)
    for _ in range(3):
        enc.encode_image(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        enc.encode_image(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    res.append({"encode_bs256": True, "group_m": g, "ms": round(dt * 1e3, 3), "img_s": round(256 / dt, 1)})
    print(res[-1], flush=True)
print(json.dumps(res))

#!/usr/bin/env python3
"""GPU micro-benchmark of the bf16 GEMM tile variants on the ViT-B/32 shapes (run on the MI355X box)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M0 = B * 50
shapes = [("qkv", _lib.EPI_BIAS_BF16, 2304, 768), ("out", _lib.EPI_BIAS_RESID_F32, 768, 768),
          ("fc1", _lib.EPI_BIAS_QGELU_BF16, 3072, 768), ("fc2", _lib.EPI_BIAS_RESID_F32, 768, 3072)]
if len(sys.argv) > 2 and sys.argv[2] == "l14":  # ViT-L/14: 257 tokens per image
    M0 = B * 257
    shapes = [("L14 qkv", _lib.EPI_BIAS_BF16, 3072, 1024), ("L14 out", _lib.EPI_BIAS_RESID_F32, 1024, 1024),
              ("L14 fc1", _lib.EPI_BIAS_QGELU_BF16, 4096, 1024), ("L14 fc2", _lib.EPI_BIAS_RESID_F32, 1024, 4096)]
res = []
for name, epi, N, K in shapes:
    for bm in (128, 160, 192, 256, 1128, 1160, 1192, 2128, 2160, 2192):
        if (bm == 256 or bm > 2000) and N % 256:
            continue
        tile = bm % 1000
        M = (M0 + tile - 1) // tile * tile
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == _lib.EPI_BIAS_RESID_F32 else torch.bfloat16)
        ms = C.c_float(0)
        _lib.check(lib.mmiss_dbg_gemm_time(0, epi, bm, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), None,
                                           M, N, K, 0, 0, 50, C.byref(ms)))
        tf = 2.0 * M0 * N * K / (ms.value * 1e-3) / 1e12
        res.append({"gemm": name, "bm": bm, "M": M, "N": N, "K": K, "us": round(ms.value * 1e3, 2), "tflops": round(tf, 1)})
        print(res[-1], flush=True)
print(json.dumps(res))

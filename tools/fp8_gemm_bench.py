#!/usr/bin/env python3
"""fp8 (block-scaled e4m3, gemm_fp8.h) vs bf16 (gemm_bf16.h) GEMM at the ViT-L/14 bs-128 shapes (32896 rows) and the
ViT-B/32 bs-256 shapes (12800 rows): microseconds per launch and TFLOP/s, random operands, same epilogue class."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402

lib = _lib.load()
rows = []
for name, M, N, K, epi8, epi16 in [("L14 qkv", 32896, 3072, 1024, 0, 1), ("L14 fc1", 32896, 4096, 1024, 1, 2),
                                   ("L14 fc2", 32896, 1024, 4096, 2, 3), ("B32 qkv", 12800, 2304, 768, 0, 1),
                                   ("B32 fc1", 12800, 3072, 768, 1, 2), ("B32 fc2", 12800, 768, 3072, 2, 3)]:
    for bm in (128, 160, 192):
        Mp = (M + bm - 1) // bm * bm
        A8 = torch.randint(0, 120, (Mp, K), dtype=torch.uint8, device="cuda")
        As = torch.randint(120, 130, (Mp, 16 * ((K + 511) // 512)), dtype=torch.uint8, device="cuda")
        W8 = torch.randint(0, 120, (N, K), dtype=torch.uint8, device="cuda")
        ws = torch.rand(N, device="cuda") * 1e-3
        bias = torch.randn(N, device="cuda")
        out = torch.zeros((Mp, N), dtype=torch.float32, device="cuda")   # large enough for every epilogue
        osc = torch.zeros((Mp, 16 * ((N + 511) // 512)), dtype=torch.uint8, device="cuda")
        ms = C.c_float(0)
        _lib.check(lib.mmiss_dbg_gemm8_time(0, epi8, bm, A8.data_ptr(), As.data_ptr(), W8.data_ptr(), ws.data_ptr(),
                                            bias.data_ptr(), out.data_ptr(), osc.data_ptr(), Mp, N, K, 20, C.byref(ms)))
        t8 = ms.value
        A16 = torch.randn(Mp, K, device="cuda").to(torch.bfloat16)
        W16 = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        _lib.check(lib.mmiss_dbg_gemm_time(0, epi16, bm, A16.data_ptr(), W16.data_ptr(), out.data_ptr(), bias.data_ptr(), None,
                                           Mp, N, K, 0, 0, 20, C.byref(ms)))
        t16 = ms.value
        fl = 2.0 * M * N * K
        rows.append({"gemm": name, "bm": bm, "fp8_us": round(t8 * 1e3, 1), "fp8_tflops": round(fl / t8 / 1e9, 0),
                     "bf16_us": round(t16 * 1e3, 1), "bf16_tflops": round(fl / t16 / 1e9, 0), "speedup": round(t16 / t8, 2)})
        del A8, As, W8, out, A16, W16
for r in rows:
    print(json.dumps(r))

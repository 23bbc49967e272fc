#!/usr/bin/env python3
"""The block-scaled fp8 GEMMs of BASELINE configs[4] (ViT-L/14, batch 128: 32 896 rows) and of ViT-B/32 in isolation: the
BM x 128 tile kernel (gemm_fp8.h; bm 128 / 160 / 192) against the persistent 256 x 256 kernel (gemm_fp8_p256.h; bm = 256 +
valid rows). us per launch and PFLOP/s on the valid rows; interleaved rounds in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from oracle import fp8_oracle as fo

lib = _lib.load()
SHAPES = [("L/14 QKV", 0, 32896, 3072, 1024), ("L/14 FC1", 1, 32896, 4096, 1024), ("L/14 out-proj", 3, 32896, 1024, 1024),
          ("L/14 FC2", 3, 32896, 1024, 4096), ("B/32 FC2", 3, 12800, 768, 3072),
          # 128 whole row blocks (no ragged block at all): what the tile stream alone costs
          ("QKV 32768", 0, 32768, 3072, 1024), ("FC1 32768", 1, 32768, 4096, 1024), ("out 32768", 3, 32768, 1024, 1024),
          ("FC2 32768", 3, 32768, 1024, 4096)]
g = torch.Generator(device="cuda").manual_seed(1)
for name, epi, mv, N, K in SHAPES:
    Mp = (mv + 255) // 256 * 256 + 192
    A8 = torch.randint(0, 120, (Mp, K), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    W8 = torch.randint(0, 120, (N, K), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    As = torch.full((Mp, fo.scale_row_bytes(K)), 124, dtype=torch.uint8, device="cuda")
    ws = torch.rand(N, device="cuda") * 2.0 ** -8
    bias = torch.randn(N, device="cuda")
    osc = torch.zeros((Mp, fo.scale_row_bytes(N)), dtype=torch.uint8, device="cuda")
    out = torch.zeros((Mp, N), dtype=torch.bfloat16 if epi != 1 else torch.uint8, device="cuda")
    res = {}
    for rnd in range(3):
        for bm in (128, 160, 192, 256 + mv, -(256 + mv)):
            _lib.set_option("gemm_p256_ragged", 0 if bm < 0 else 1)   # (negative: the ragged last row block as a tile, the round-5a form)
            key = bm if 0 < bm < 256 else (256 if bm > 0 else "256t")
            bm = abs(bm)
            M = (mv + 255) // 256 * 256 if bm >= 256 else (mv + bm - 1) // bm * bm
            ms = C.c_float(0)
            _lib.check(lib.mmiss_dbg_gemm8_time(0, epi, bm, A8.data_ptr(), As.data_ptr(), W8.data_ptr(), ws.data_ptr(), bias.data_ptr(),
                                                out.data_ptr(), osc.data_ptr(), M, N, K, 20, C.byref(ms)))
            res.setdefault(key, []).append(ms.value * 1e3)
    fl = 2.0 * mv * N * K
    _lib.set_option("gemm_p256_ragged", 1)
    print(f"{name:14s} M={mv} N={N} K={K}: " + "  ".join(f"bm{b}: {min(v):7.1f} us {fl / min(v) / 1e9:5.2f} PF" for b, v in res.items()), flush=True)

#!/usr/bin/env python3
"""Fixed cost vs per-K-tile cost of the GEMM kernels: time(K) for K = 64 .. 3072 at the ViT-B/32 M."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
lib = _lib.load()
M0 = 12800
for name, epi, N, bm in [("resid N768 bm160", _lib.EPI_BIAS_RESID_F32, 768, 160), ("bias_bf16 N2304 bm192", _lib.EPI_BIAS_BF16, 2304, 192),
                         ("bias_bf16 N2304 bm256", _lib.EPI_BIAS_BF16, 2304, 256), ("qgelu N3072 bm192", _lib.EPI_BIAS_QGELU_BF16, 3072, 192),
                         ("f32 N768 bm160", _lib.EPI_F32, 768, 160),
                         ("RING resid N768 bm160", _lib.EPI_BIAS_RESID_F32, 768, 1160), ("RING bias_bf16 N2304 bm192", _lib.EPI_BIAS_BF16, 2304, 1192),
                         ("RING qgelu N3072 bm192", _lib.EPI_BIAS_QGELU_BF16, 3072, 1192)]:
    M = (M0 + bm % 1000 - 1) // (bm % 1000) * (bm % 1000)
    line = []
    for K in (64, 128, 256, 768, 1536, 3072):
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16 if epi in (_lib.EPI_BIAS_BF16, _lib.EPI_BIAS_QGELU_BF16) else torch.float32)
        ms = C.c_float(0)
        _lib.check(lib.mmiss_dbg_gemm_time(0, epi, bm, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), None, M, N, K, 0, 0, 50, C.byref(ms)))
        line.append(f"K={K}: {ms.value*1e3:.1f}us")
    print(name, " | ".join(line), flush=True)

#!/usr/bin/env python3
"""Per-K-tile cost of the tiled GEMM when the grid is far below the CU count (middle batch sizes): time(K) at M = 800
for N = 768 / 2304 (tiles 128 x 128, split-K and skinny paths off)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.set_option("gemm_skinny", 0)
_lib.set_option("gemm_splitk", 0)
for M in (768, 1536):
    for name, epi, N in (("f32 N768", _lib.EPI_F32, 768), ("f32 N2304", _lib.EPI_F32, 2304), ("resid N768", _lib.EPI_BIAS_RESID_F32, 768)):
        line = []
        for K in (64, 256, 768, 1536, 3072, 6144):
            A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
            bias = torch.randn(N, device="cuda")
            out = torch.zeros(M, N, device="cuda")
            ms = C.c_float(0)
            _lib.check(lib.mmiss_dbg_gemm_time(0, epi, 128, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), None, M, N, K,
                                               0, 0, 100, C.byref(ms)))
            line.append(f"K={K}: {ms.value * 1e3:.1f}")
        print(f"M={M} {name} (us)", " | ".join(line), flush=True)

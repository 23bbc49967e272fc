#!/usr/bin/env python3
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
lib = _lib.load()
M = 12800
for name, epi, N, K, bm in [("qkv", _lib.EPI_BIAS_BF16, 2304, 768, 160), ("out", _lib.EPI_BIAS_RESID_F32, 768, 768, 160),
                            ("fc1", _lib.EPI_BIAS_QGELU_BF16, 3072, 768, 160), ("fc2", _lib.EPI_BIAS_RESID_F32, 768, 3072, 160),
                            ("fc1", _lib.EPI_BIAS_QGELU_BF16, 3072, 768, 128), ("qkv", _lib.EPI_BIAS_BF16, 2304, 768, 128)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == _lib.EPI_BIAS_RESID_F32 else torch.bfloat16)
    ms = (C.c_float * 3)()
    _lib.check(lib.mmiss_dbg_gemm_split_time(0, epi, bm, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), M, N, K, 50, ms))
    print(f"{name} bm{bm}: full {ms[0]*1e3:.1f} us | halves same stream {ms[1]*1e3:.1f} us | halves two streams {ms[2]*1e3:.1f} us", flush=True)

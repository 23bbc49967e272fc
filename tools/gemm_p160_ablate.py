#!/usr/bin/env python3
"""What bounds the two-phase loop of gemm_bf16_p160.h: the FC2 / out-projection shapes with pieces of the K loop compiled out
(EXPERIMENTS build, option gemm_p160_dbg: 1 = no MFMAs, 2 = no LDS-DMA staging, 4 = no fragment reads; results are wrong)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402

lib = _lib.load()
NAMES = {0: "full", 1: "no MFMA", 2: "no staging", 4: "no fragment reads", 3: "no MFMA, no staging", 5: "no MFMA, no reads",
         6: "no staging, no reads", 7: "barriers + waits + epilogue only"}
for name, M, N, K in (("FC2", 12800, 768, 3072), ("out-proj", 12800, 768, 768)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    stats = torch.zeros(M, N // 64, 2, device="cuda")
    line = "%-8s" % name
    for dbg in (0, 1, 2, 4, 3, 5, 6, 7):
        _lib.set_option("gemm_p160_dbg", dbg)
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ms = C.c_float(0)
        _lib.check(lib.mmiss_dbg_gemm_resid16(0, None, 0, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), stats.data_ptr(),
                                              M, N, K, M, 30, C.byref(ms)))
        line += " | %s %.1f us (%.2f us per K-tile)" % (NAMES[dbg], ms.value * 1e3, ms.value * 1e3 / (K // 64))
    _lib.set_option("gemm_p160_dbg", 0)
    print(line, flush=True)

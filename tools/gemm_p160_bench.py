#!/usr/bin/env python3
"""The residual GEMMs of a layer (out-projection, FC2; bf16 residual stream) on the 160 x 256 tile of the staggered loop
(gemm160p_kernel) against the 128-column kernel at its tile heights: HIP-event time of back-to-back launches of one shape."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402

lib = _lib.load()
SHAPES = [("B/32 out-proj", 12800, 768, 768), ("B/32 FC2", 12800, 768, 3072), ("text out-proj", 19840, 512, 512),
          ("text FC2", 19840, 512, 2048), ("L/14 out-proj", 33120, 1024, 1024), ("L/14 FC2", 33120, 1024, 4096),
          ("B/32 bs128 FC2", 6400, 768, 3072)]
for name, M, N, K in SHAPES:
    A = (torch.randn(M, K, device="cuda")).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.01
    stats = torch.zeros(M, N // 64, 2, device="cuda")
    line = "%-14s %6d x %4d x %4d " % (name, M, N, K)
    for variant in (0, 128, 160, 192):
        if variant and M % variant:
            continue
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ms = C.c_float(0)
        rc = lib.mmiss_dbg_gemm_resid16(0, None, variant, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), stats.data_ptr(),
                                        M, N, K, M, 30, C.byref(ms))
        line += "| %4s: %6.1f us %4.0f TF " % (variant or "p160", ms.value * 1e3, 2.0 * M * N * K / ms.value / 1e9) if rc == 0 else "| %4s: n/a " % (variant or "p160")
    print(line, flush=True)

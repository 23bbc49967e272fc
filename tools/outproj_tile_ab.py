#!/usr/bin/env python3
"""the out-projection of ViT-B/32 at batch 256 (12 800 x 768 x 768, bf16 residual stream) on every tile
shape the library has, isolated launches (HIP events around 30 back-to-back launches, three passes):
  p160      160 x 256 tiles, 240 workgroups = one round (gemm160p_kernel, the shipping choice)
  128 / 160 / 192 x 128 tiles of the older loop (gemm16_kernel, residual epilogue): 600 / 480 / 402 workgroups, two per CU
  256 x 256 persistent (gemm256p_kernel, plain bias epilogue — the residual epilogue does not exist on it): 150 tiles = 0.59 round
A 128 x 256 tile does not exist as a kernel: it would be 300 workgroups = 1.17 rounds of a tile that costs ~0.85 of the
160-row one (the K-tile is bound by its staging: 384 rows of operands instead of 416), i.e. ~1.7 x the p160 time."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402

lib = _lib.load()
for name, M, N, K in (("out-projection", 12800, 768, 768), ("FC2", 12800, 768, 3072)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.01
    stats = torch.zeros(M, N // 64, 2, device="cuda")
    out = torch.zeros(M + 256, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rep in range(3):
        for variant in (0, 128, 160, 192):
            if variant and M % variant:
                continue
            ms = C.c_float(0)
            rc = lib.mmiss_dbg_gemm_resid16(0, None, variant, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), stats.data_ptr(),
                                            M, N, K, M, 30, C.byref(ms))
            if rc == 0:
                res.setdefault("p160 (160 x 256)" if variant == 0 else "%d x 128" % variant, []).append(ms.value * 1e3)
        ms = C.c_float(0)
        _lib.check(lib.mmiss_dbg_gemm_p256(0, None, 1, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), None, None, 1e-5,
                                           M, N, K, M, 30, C.byref(ms)))
        res.setdefault("256 x 256 persistent (bias epilogue)", []).append(ms.value * 1e3)
    print("%s %d x %d x %d" % (name, M, N, K))
    for k, v in res.items():
        print("   %-40s %s us   (best %.0f TF)" % (k, " ".join("%.1f" % t for t in v), 2.0 * M * N * K / min(v) / 1e6), flush=True)

"""Where a K-tile of the persistent 256x256 GEMM spends its time: the same launch with parts of the loop removed
(EXPERIMENTS build only: make -C multimodal-image-similarity-search_amd/csrc EXPERIMENTS=1). Results are wrong by
construction; only the times mean anything."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import mmiss_amd  # noqa: F401
from mmiss_amd import _lib

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
K, N = 768, 3072
for M in (256 * 128,):   # 768 / 1536 tiles: 3 / 6 tiles per workgroup exactly
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    cvec = W.float().sum(1).contiguous()
    parts = A.float().view(M, K // 64, 64)
    stats = torch.stack([parts.sum(-1), (parts * parts).sum(-1)], dim=-1).contiguous()
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ms = C.c_float(0)
    for rep in range(2):
        for name, dbg in (("full", 0), ("hot operands", 8), ("no MFMA", 1), ("no staging", 2), ("no fragment reads", 4),
                          ("no MFMA, no reads", 5), ("no staging, no reads", 6), ("only barriers + epilogue", 7),
                          ("only barriers + epilogue without stores", 7 + 32), ("only barriers", 7 + 16), ("empty loop", 7 + 16 + 64),
                          ("full without epilogue", 16), ("full without stores", 32), ("full without barriers", 64),
                          ("full", 0), ("32x32x16 MFMA", 128), ("32x32x16 MFMA, hot operands", 136), ("32x32x16 MFMA, no reads", 132),
                          ("full", 0), ("32x32x16 MFMA", 128)):
            _lib.set_option("gemm_p256_dbg", dbg)
            _lib.check(lib.mmiss_dbg_gemm_p256(0, None, 8, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                               cvec.data_ptr(), stats.data_ptr(), 1e-5, M, N, K, M, 30, C.byref(ms)))
            _lib.set_option("gemm_p256_dbg", 0)
            tiles = (M // 256) * (N // 256) / 256.0
            print("M %6d %-40s %7.1f us  = %.2f us per tile, %.3f us per K-tile" % (M, name, ms.value * 1e3, ms.value * 1e3 / tiles,
                                                                                  ms.value * 1e3 / tiles / 12), flush=True)
        for style in (0, 1):
            _lib.set_option("gemm_p256_style", style)
            for epi in (7, 8):
                _lib.check(lib.mmiss_dbg_gemm_p256(0, None, epi, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                                   cvec.data_ptr(), stats.data_ptr(), 1e-5, M, N, K, M, 30, C.byref(ms)))
                print("M %6d epilogue style %d epi %d: %7.1f us = %.2f us per tile" % (M, style, epi, ms.value * 1e3, ms.value * 1e3 / tiles), flush=True)
        _lib.set_option("gemm_p256_style", 0)

"""Persistent 256x256 GEMM (gemm_bf16_p256.h) against the shipping kernels on the tower shapes, isolated launches.
usage: python tools/gemm_p256_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import mmiss_amd  # noqa: F401
from mmiss_amd import _lib

lib = _lib.load()
SHAPES = [("B/32 QKV", 12800, 2304, 768, 7), ("B/32 FC1", 12800, 3072, 768, 8), ("text QKV", 19712, 1536, 512, 7),
          ("text FC1", 19712, 2048, 512, 8), ("L/14 QKV", 33024, 3072, 1024, 1), ("L/14 FC1", 33024, 4096, 1024, 2)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, epi in SHAPES:
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    cvec = W.float().sum(1).contiguous()
    parts = A.float().view(M, K // 64, 64)
    stats = torch.stack([parts.sum(-1), (parts * parts).sum(-1)], dim=-1).contiguous()
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    ms = C.c_float(0)
    res = {}
    for band in (8, 4, 16, 1):
        _lib.set_option("gemm_p256_band", band)
        _lib.check(lib.mmiss_dbg_gemm_p256(0, None, epi, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                           cvec.data_ptr(), stats.data_ptr(), 1e-5, M, N, K, M, 50, C.byref(ms)))
        res["p256 band %d" % band] = ms.value * 1e3
    old_epi = {7: 1, 8: 2}.get(epi, epi)
    for variant in (256, 192, 160, 128):
        if M % variant:
            continue
        _lib.check(lib.mmiss_dbg_gemm_time(0, old_epi, variant, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                           None, M, N, K, 0, 0, 50, C.byref(ms)))
        res["tile %d (plain epilogue)" % variant] = ms.value * 1e3
    fl = 2.0 * M * N * K
    print(name, M, N, K, " ".join("%s: %.1f us (%.0f TF)" % (k, v, fl / v / 1e6) for k, v in res.items()), flush=True)

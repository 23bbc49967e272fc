import os, sys
sys.path.insert(0, "/root/repo")
import ctypes as C
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from oracle import fp8_oracle as fo
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
for name, epi, mv, N, K in (("QKV", 0, 32896, 3072, 1024), ("FC1", 1, 32896, 4096, 1024), ("out", 3, 32896, 1024, 1024), ("FC2", 3, 32896, 1024, 4096)):
    M = (mv + 255) // 256 * 256
    A8 = torch.randint(0, 120, (M, K), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    W8 = torch.randint(0, 120, (N, K), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    As = torch.full((M, fo.scale_row_bytes(K)), 124, dtype=torch.uint8, device="cuda")
    ws = torch.rand(N, device="cuda") * 2.0 ** -8
    bias = torch.randn(N, device="cuda")
    osc = torch.zeros((M, fo.scale_row_bytes(N)), dtype=torch.uint8, device="cuda")
    out = torch.zeros((M, N), dtype=torch.bfloat16 if epi != 1 else torch.uint8, device="cuda")
    res = {}
    for rnd in range(3):
        for band in (8, 2, 3, 4, 6, 12, 16):
            _lib.set_option("gemm_p256_band", band)
            ms = C.c_float(0)
            _lib.check(lib.mmiss_dbg_gemm8_time(0, epi, 256 + mv, A8.data_ptr(), As.data_ptr(), W8.data_ptr(), ws.data_ptr(), bias.data_ptr(),
                                                out.data_ptr(), osc.data_ptr(), M, N, K, 20, C.byref(ms)))
            res.setdefault(band, []).append(ms.value * 1e3)
    print(name, "  ".join(f"band {b}: {min(v):6.1f}" for b, v in sorted(res.items())), flush=True)

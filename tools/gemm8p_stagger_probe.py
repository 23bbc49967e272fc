#!/usr/bin/env python3
"""EXPERIMENTS build only: the persistent fp8 GEMMs of ViT-L/14 with the odd workgroups started `gemm_p256_stagger` x ~4 us late
(s_sleep). If the epilogues cost more because all 256 workgroups run them in step, the launch grows by less than the delay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MMISS_ALLOW_AB_BUILD", "1")
import ctypes as C
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from oracle import fp8_oracle as fo

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
for name, epi, mv, N, K in (("QKV 32768", 0, 32768, 3072, 1024), ("FC1 32768", 1, 32768, 4096, 1024), ("FC2 32768", 3, 32768, 1024, 4096)):
    M = mv
    A8 = torch.randint(0, 120, (M, K), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    W8 = torch.randint(0, 120, (N, K), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    As = torch.full((M, fo.scale_row_bytes(K)), 124, dtype=torch.uint8, device="cuda")
    ws = torch.rand(N, device="cuda") * 2.0 ** -8
    bias = torch.randn(N, device="cuda")
    osc = torch.zeros((M, fo.scale_row_bytes(N)), dtype=torch.uint8, device="cuda")
    out = torch.zeros((M, N), dtype=torch.bfloat16 if epi != 1 else torch.uint8, device="cuda")
    res = {}
    for rnd in range(3):
        for stg in (0, 1, 2, 3, 4):
            _lib.set_option("gemm_p256_stagger", stg)
            ms = C.c_float(0)
            _lib.check(lib.mmiss_dbg_gemm8_time(0, epi, 256 + mv, A8.data_ptr(), As.data_ptr(), W8.data_ptr(), ws.data_ptr(), bias.data_ptr(),
                                                out.data_ptr(), osc.data_ptr(), M, N, K, 20, C.byref(ms)))
            res.setdefault(stg, []).append(ms.value * 1e3)
    _lib.set_option("gemm_p256_stagger", 0)
    print(f"{name}: " + "  ".join(f"stagger {k}: {min(v):6.1f} us" for k, v in res.items()), flush=True)

#!/usr/bin/env python3
"""The attention kernel alone at the two shapes that matter (ViT-L/14: B=128, T=257, H=16; ViT-B/32: B=256, T=50, H=12),
20 launches each — meant to run under `rocprofv3 --kernel-trace --pmc ...`."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
lib = _lib.load()
for B, T, H in ((128, 257, 16), (256, 50, 12)):
    d = H * 64
    qkv = (torch.randn(B * T, 3 * d, device="cuda") * 0.5).to(torch.bfloat16)
    ctx = torch.empty(B * T, d, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, 0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, 0))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"B={B} T={T} H={H}: {dt*1e6:.1f} us, {4.0*B*H*T*T*64/dt/1e12:.1f} TFLOP/s")

import json
res = {}
for B, T, H, causal in ((256, 50, 12, 0), (256, 77, 8, 1)):
    d = H * 64
    qkv = (torch.randn(B * T, 3 * d, device="cuda") * 0.5).to(torch.bfloat16)
    ctx = torch.empty(B * T, d, device="cuda", dtype=torch.bfloat16)
    ref = None
    for hpb in (1, 2, 3, 4, 6):
        if H % hpb:
            continue
        _lib.set_option("att_hpb", hpb)
        for _ in range(3):
            _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, causal))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, causal))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        if ref is None:
            ref = ctx.clone()
        res[f"T{T}_hpb{hpb}"] = (round(dt * 1e6, 1), bool(torch.equal(ctx, ref)))
    _lib.set_option("att_hpb", 0)
print(json.dumps(res))

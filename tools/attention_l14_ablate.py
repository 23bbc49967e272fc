#!/usr/bin/env python3
"""ViT-L/14 attention (B = 128, T = 257, H = 16) in isolation: whole kernel / without the K-V loads / without the query tiles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
lib = _lib.load()
B, T, H = 128, 257, 16
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").to(torch.bfloat16)
ctx = torch.zeros(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
def t(n=30):
    for _ in range(5): _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, 0))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, 0))
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for dbg, name in ((0, "whole kernel"), (1, "no K/V loads"), (2, "no query tiles (staging only)"), (0, "whole kernel again")):
    _lib.set_option("att_dbg", dbg)
    print(f"{name:32s} {min(t(), t()):7.1f} us", flush=True)
_lib.set_option("att_dbg", 0)

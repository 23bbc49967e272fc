#!/usr/bin/env python3
"""Context for the hand-written GEMMs: what the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS) does
on the same bf16 shapes, WITHOUT the fused epilogues the encoder needs (QuickGELU, f32 residual add, LayerNorm statistics).
Not used by the product."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

shapes = [("B/32 qkv", 12800, 2304, 768), ("B/32 out", 12800, 768, 768), ("B/32 fc1", 12800, 3072, 768),
          ("B/32 fc2", 12800, 768, 3072), ("L/14 qkv", 32896, 3072, 1024), ("L/14 out", 32896, 1024, 1024),
          ("L/14 fc1", 32896, 4096, 1024), ("L/14 fc2", 32896, 1024, 4096), ("score Q1024", 1024, 1048576, 512)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    for label, fn in (("linear", lambda: F.linear(a, w)), ("linear+bias", lambda: F.linear(a, w, b))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print({"gemm": name, "op": label, "us": round(us, 1), "tflops": round(2.0 * M * N * K / (us * 1e-6) / 1e12, 1)}, flush=True)

#!/usr/bin/env python3
"""Weight-streaming GEMMs of one request (M = 50) with HOT weights (the same matrix every launch: L2-resident) against COLD
weights (a ring of matrices larger than the L2 / than the Infinity Cache): what a kernel of the batch-1 encode pays for
meeting its weights for the first time."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402

lib = _lib.load()
M = 50
for name, epi, N, K in (("qkv", _lib.EPI_BIAS_BF16, 2304, 768), ("out", _lib.EPI_BIAS_RESID_F32, 768, 768),
                        ("fc1", _lib.EPI_BIAS_QGELU_BF16, 3072, 768), ("fc2", _lib.EPI_BIAS_RESID_F32, 768, 3072)):
    A = torch.randn(128, K, device="cuda").to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(128, N, device="cuda", dtype=torch.float32 if epi == _lib.EPI_BIAS_RESID_F32 else torch.bfloat16)
    for ring_mb in (0, 64, 400):
        n = max(1, int(ring_mb * 1e6 / (N * K * 2)))
        Ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(n)]
        st = torch.cuda.current_stream().cuda_stream
        def run(iters):
            for i in range(iters):
                _lib.check(lib.mmiss_dbg_gemm(0, st, epi, 128, A.data_ptr(), Ws[i % n].data_ptr(), out.data_ptr(), bias.data_ptr(), None,
                                              M, N, K, 0, 0))
        run(2 * n + 10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 400
        e0.record()
        run(iters)
        e1.record()
        torch.cuda.synchronize()
        print("%-4s N %4d K %4d  weights ring %3d MB (%3d matrices): %.2f us per launch" % (name, N, K, ring_mb, n, e0.elapsed_time(e1) * 1e3 / iters), flush=True)
        del Ws

#!/usr/bin/env python3
"""Per-GEMM cost when TWO independent batches are in flight: each stream launches the same ViT-B/32 GEMM back to back on
its own operands; the aggregate time per GEMM is what a pipeline of two batches pays. 128-column tiles (best height per
shape) against the 256 x 256 tile, whose partly filled last round of tiles can now fill with the other stream's workgroups."""
import ctypes as C
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib

lib = _lib.load()
M0 = 12800
shapes = [("qkv", _lib.EPI_BIAS_BF16, 2304, 768, 192), ("out", _lib.EPI_BIAS_RESID_F32, 768, 768, 160),
          ("fc1", _lib.EPI_BIAS_QGELU_BF16, 3072, 768, 192), ("fc2", _lib.EPI_BIAS_RESID_F32, 768, 3072, 160)]
ITERS = 60

def operands(epi, N, K, tile):
    M = (M0 + tile - 1) // tile * tile
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == _lib.EPI_BIAS_RESID_F32 else torch.bfloat16)
    return M, A, W, bias, out

def run(nstream, epi, N, K, variant):
    tile = variant % 1000
    ops = [operands(epi, N, K, tile) for _ in range(nstream)]
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    def work(i, n, bar):
        M, A, W, bias, out = ops[i]
        bar.wait()
        for _ in range(n):
            _lib.check(lib.mmiss_dbg_gemm(0, C.c_void_p(streams[i].cuda_stream), epi, variant, A.data_ptr(), W.data_ptr(),
                                          out.data_ptr(), bias.data_ptr(), None, M, N, K, 0, 0))
        streams[i].synchronize()
    for n in (5, ITERS):
        bar = threading.Barrier(nstream + 1)
        th = [threading.Thread(target=work, args=(i, n, bar)) for i in range(nstream)]
        for t in th: t.start()
        torch.cuda.synchronize()
        bar.wait()
        t0 = time.perf_counter()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / (ITERS * nstream) * 1e6

for name, epi, N, K, bm in shapes:
    for variant in (bm, 256):
        r = [run(ns, epi, N, K, variant) for ns in (1, 2, 3)]
        tf = [2.0 * M0 * N * K / (u * 1e-6) / 1e12 for u in r]
        print(f"{name:4s} tile {variant:3d}: 1 stream {r[0]:6.1f} us ({tf[0]:5.0f} TF) | 2 streams {r[1]:6.1f} us/GEMM ({tf[1]:5.0f} TF) | "
              f"3 streams {r[2]:6.1f} us/GEMM ({tf[2]:5.0f} TF)", flush=True)

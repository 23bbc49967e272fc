#!/usr/bin/env python3
"""Do a compute-bound GEMM and a memory-bound kernel from two HIP streams overlap on the MI355X?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
lib = _lib.load()
M, N, K = 12800, 3072, 768
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
x = torch.randn(M, 768, device="cuda"); g = torch.randn(768, device="cuda"); b = torch.randn(768, device="cuda"); h = torch.zeros(M, 768, device="cuda", dtype=torch.bfloat16)
qkv = torch.randn(M, 2304, device="cuda").to(torch.bfloat16); ctx = torch.zeros(M, 768, device="cuda", dtype=torch.bfloat16)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def gemm(s, bm=160, M_=M, off=0):
    _lib.check(lib.mmiss_dbg_gemm(0, s.cuda_stream, _lib.EPI_BIAS_QGELU_BF16, bm, A.data_ptr() + off * K * 2, W.data_ptr(), out.data_ptr() + off * N * 2, bias.data_ptr(), None, M_, N, K, 0, 0))
def ln(s):
    _lib.check(lib.mmiss_dbg_layernorm(0, s.cuda_stream, x.data_ptr(), g.data_ptr(), b.data_ptr(), h.data_ptr(), 1, M, 768, 1e-5))
def attn(s):
    _lib.check(lib.mmiss_dbg_attention(0, s.cuda_stream, qkv.data_ptr(), ctx.data_ptr(), 256, 50, 12, 0))
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("gemm alone        %.1f us" % timeit(lambda: gemm(s1)))
print("ln x4 alone       %.1f us" % timeit(lambda: [ln(s2) for _ in range(4)]))
print("attn x3 alone     %.1f us" % timeit(lambda: [attn(s2) for _ in range(3)]))
print("gemm || ln x4     %.1f us" % timeit(lambda: (gemm(s1), [ln(s2) for _ in range(4)])))
print("gemm || attn x3   %.1f us" % timeit(lambda: (gemm(s1), [attn(s2) for _ in range(3)])))
print("gemm ; ln x4 (same stream) %.1f us" % timeit(lambda: (gemm(s1), [ln(s1) for _ in range(4)])))
# two half GEMMs on two streams with different tile heights
print("half gemms 2 streams (bm160 rows 0..6400 | bm128 rest) %.1f us" % timeit(lambda: (gemm(s1, 160, 6400, 0), gemm(s2, 128, 6400, 6400))))
print("half gemms same stream %.1f us" % timeit(lambda: (gemm(s1, 160, 6400, 0), gemm(s1, 128, 6400, 6400))))
print("half gemms 2 streams same bm160 %.1f us" % timeit(lambda: (gemm(s1, 160, 6400, 0), gemm(s2, 160, 6400, 6400))))

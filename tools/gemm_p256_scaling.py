"""Time of the persistent 256x256 GEMM vs tiles per workgroup (1, 2, 3 full rounds of 256 tiles), next to the
one-tile-per-workgroup kernel on the same shapes: slope = steady-state time per tile, intercept = launch + fill + drain."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import mmiss_amd  # noqa: F401
from mmiss_amd import _lib

lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
K = int(os.environ.get("K", 768))
N = 2048
for epi in (1, 2, 7, 8):
    for rounds in (1, 2, 3, 4):
        M = 8192 * rounds
        A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g)
        cvec = W.float().sum(1).contiguous()
        parts = A.float().view(M, K // 64, 64)
        stats = torch.stack([parts.sum(-1), (parts * parts).sum(-1)], dim=-1).contiguous()
        out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        ms = C.c_float(0)
        _lib.check(lib.mmiss_dbg_gemm_p256(0, None, epi, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                           cvec.data_ptr(), stats.data_ptr(), 1e-5, M, N, K, M, 50, C.byref(ms)))
        t_p = ms.value * 1e3
        t_o = float("nan")
        if epi in (1, 2):
            _lib.check(lib.mmiss_dbg_gemm_time(0, epi, 256, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                               None, M, N, K, 0, 0, 50, C.byref(ms)))
            t_o = ms.value * 1e3
        print("epi %d rounds %d: persistent %.1f us, one tile per workgroup %.1f us" % (epi, rounds, t_p, t_o), flush=True)

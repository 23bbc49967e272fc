"""One shape through the persistent 256x256 GEMM (and, with OLD=1, the one-tile-per-workgroup 256x256 kernel), a fixed
number of launches: the program rocprofv3 wraps for counter passes (tools/profile_gemm_pmc.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import mmiss_amd  # noqa: F401
from mmiss_amd import _lib

lib = _lib.load()
M, N, K = (int(os.environ.get(k, d)) for k, d in (("M", 12800), ("N", 3072), ("K", 768)))
epi = int(os.environ.get("EPI", 8))
iters = int(os.environ.get("ITERS", 20))
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda", generator=g)
cvec = W.float().sum(1).contiguous()
parts = A.float().view(M, K // 64, 64)
stats = torch.stack([parts.sum(-1), (parts * parts).sum(-1)], dim=-1).contiguous()
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
ms = C.c_float(0)
if os.environ.get("OLD"):
    _lib.check(lib.mmiss_dbg_gemm_time(0, {7: 1, 8: 2}.get(epi, epi), int(os.environ["OLD"]), A.data_ptr(), W.data_ptr(), out.data_ptr(),
                                       bias.data_ptr(), None, M, N, K, 0, 0, iters, C.byref(ms)))
else:
    _lib.check(lib.mmiss_dbg_gemm_p256(0, None, epi, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                       cvec.data_ptr(), stats.data_ptr(), 1e-5, M, N, K, M, iters, C.byref(ms)))
print("M %d N %d K %d epi %d: %.1f us per launch" % (M, N, K, epi, ms.value * 1e3))

if os.environ.get("CALIB"):   # known byte counts under the same counters: a 256 MiB device-to-device copy (16 B per lane)
    x = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(5):
        y.copy_(x)
    torch.cuda.synchronize()
    print("calibration: 5 copies of %d bytes (read + write each)" % (x.numel() * 4))

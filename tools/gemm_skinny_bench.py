#!/usr/bin/env python3
"""GPU micro-benchmark: weight-streaming kernel (gemm_skinny.h) vs the tiled kernel for small M on the ViT-B/32 layer
shapes; decides the M threshold of the dispatch in launch_gemm."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402

lib = _lib.load()
shapes = [("qkv", _lib.EPI_BIAS_BF16, 2304, 768), ("out", _lib.EPI_BIAS_RESID_F32, 768, 768),
          ("fc1", _lib.EPI_BIAS_QGELU_BF16, 3072, 768), ("fc2", _lib.EPI_BIAS_RESID_F32, 768, 3072),
          ("proj", _lib.EPI_F32, 512, 768)]
l14 = [("L14 qkv", _lib.EPI_BIAS_BF16, 3072, 1024), ("L14 out", _lib.EPI_BIAS_RESID_F32, 1024, 1024),
       ("L14 fc1", _lib.EPI_BIAS_QGELU_BF16, 4096, 1024), ("L14 fc2", _lib.EPI_BIAS_RESID_F32, 1024, 4096),
       ("T248 qkv", _lib.EPI_BIAS_BF16, 2304, 768), ("T248 fc2", _lib.EPI_BIAS_RESID_F32, 768, 3072)]
MS = (1, 16, 32, 50, 64, 77, 100, 128, 154, 200, 256)
if len(sys.argv) > 1 and sys.argv[1] == "l14":
    shapes, MS = l14, (248, 257, 400, 514)
res = []
for name, epi, N, K in shapes:
    for M in MS:
        Mp = (M + 127) // 128 * 128
        A = torch.randn(Mp, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        out = torch.zeros(Mp, N, device="cuda", dtype=torch.float32 if epi in (_lib.EPI_BIAS_RESID_F32, _lib.EPI_F32) else torch.bfloat16)
        row = {"gemm": name, "M": M}
        for label, on, per in (("skinny_us", 1, 0), ("skinny_mt1_us", 1, 1), ("skinny_mt2_us", 1, 2), ("skinny_mt4_us", 1, 4),
                               ("skinny_mt8_us", 1, 8), ("tiled_us", 0, 0)):
            _lib.set_option("gemm_skinny", on)
            _lib.set_option("gemm_skinny_mt", per)
            _lib.set_option("gemm_skinny_max_m", 1024)
            ms = C.c_float(0)
            # the skinny path sees M valid rows; the tiled path needs the padded row count
            _lib.check(lib.mmiss_dbg_gemm_time(0, epi, 128, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(), None,
                                               M if on else Mp, N, K, 0, 0, 200, C.byref(ms)))
            row[label] = round(ms.value * 1e3, 2)
        _lib.set_option("gemm_skinny", 1)
        _lib.set_option("gemm_skinny_mt", 0)
        res.append(row)
        print(row, flush=True)
print(json.dumps(res))

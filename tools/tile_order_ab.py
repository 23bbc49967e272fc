#!/usr/bin/env python3
"""Same-box A/B of the GEMM tile order inside the real bs-256 encode: option gemm_group_m forced to 0 (all GEMMs
n fastest) vs -1 (the per-call-site defaults of run_layers). Alternates the two settings; reports wall ms per encode
(no events) and per-kernel averages (events on every launch)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict  # noqa: E402

settings = [int(a) for a in sys.argv[1:]] or [0, -1]
enc = ClipEncoder(VIT_B32, device=0, max_batch_image=256, max_batch_text=256)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
x = torch.randn(256, 3, 224, 224, device="cuda")
out = {str(s): {"wall_ms": [], "kernels": {}} for s in settings}
for rep in range(6):
    for s in settings:
        _lib.set_option("gemm_group_m", s)
        for _ in range(3):
            enc.encode_image(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            enc.encode_image(x)
        torch.cuda.synchronize()
        out[str(s)]["wall_ms"].append(round((time.perf_counter() - t0) / 30 * 1e3, 4))
for s in settings:
    _lib.set_option("gemm_group_m", s)
    _lib.prof_filter(None, 1)
    _lib.prof_enable(True)
    _lib.prof_reset()
    for _ in range(10):
        enc.encode_image(x)
    torch.cuda.synchronize()
    for k in _lib.prof_read():
        if k["kernel"].startswith("gemm"):
            out[str(s)]["kernels"][k["kernel"]] = round(k["ms"] / k["launches"] * 1e3, 2)
    _lib.prof_enable(False)
for s in settings:
    w = sorted(out[str(s)]["wall_ms"])
    out[str(s)]["wall_ms_median"] = w[len(w) // 2]
print(json.dumps(out, indent=1))

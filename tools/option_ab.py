#!/usr/bin/env python3
"""Same-box A/B of a library option inside the real bs-256 encode (isolated-kernel results do not always transfer: in the
pipeline the operands were just written by the producing kernel). Usage: option_ab.py <option> <value> <value> ...
e.g. `gemm_group_m 0 -1 8` (tile order) or `gemm_wide 0 1` (BM x 256 tiles). Alternates the settings; reports wall ms per
encode (no events) and per-kernel averages (events on every launch)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict  # noqa: E402

OPTION = sys.argv[1] if len(sys.argv) > 1 else "gemm_group_m"
settings = [int(a) for a in sys.argv[2:]] or [0, -1]
enc = ClipEncoder(VIT_B32, device=0, max_batch_image=256, max_batch_text=256)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
x = torch.randn(256, 3, 224, 224, device="cuda")
out = {str(s): {"wall_ms": [], "kernels": {}} for s in settings}
for rep in range(6):
    for s in settings:
        _lib.set_option(OPTION, s)
        for _ in range(3):
            enc.encode_image(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            enc.encode_image(x)
        torch.cuda.synchronize()
        out[str(s)]["wall_ms"].append(round((time.perf_counter() - t0) / 30 * 1e3, 4))
for s in settings:
    _lib.set_option(OPTION, s)
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

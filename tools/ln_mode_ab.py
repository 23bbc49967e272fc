#!/usr/bin/env python3
"""Same-box, alternating A/B of the LayerNorm placement (0 separate kernels, 2 folded into the GEMMs) at batch 256/128."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict  # noqa: E402

enc = ClipEncoder(VIT_B32, device=0, max_batch_image=256, max_batch_text=8)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
for B in (256, 192, 128, 96, 64):
    x = torch.randn(B, 3, 224, 224, device="cuda")
    out = torch.empty(B, 512, device="cuda")
    res = {0: [], 2: []}
    for rep in range(6):
        for mode in (0, 2):
            enc.set_fuse_ln(mode)
            for _ in range(3):
                enc.encode_image(x, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                enc.encode_image(x, out=out)
            torch.cuda.synchronize()
            res[mode].append(round((time.perf_counter() - t0) / 30 * 1e3, 4))
    print({"batch": B, "separate_ms": sorted(res[0])[3], "folded_ms": sorted(res[2])[3], "all": res}, flush=True)

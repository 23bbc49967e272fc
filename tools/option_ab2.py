#!/usr/bin/env python3
"""option_ab.py for option SETS: each argument is a comma-separated list key=value (or "base"), e.g.
`option_ab2.py base gemm_256_fold=2304 gemm_256_fold=2304,gemm_256_fold_mlp=1`. Same protocol: alternating settings on one
box, wall ms per bs-256 encode (median of 6 x 30) and per-kernel averages of the GEMM classes."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict

sets = []
for a in sys.argv[1:] or ["base"]:
    sets.append((a, {} if a == "base" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")}))
keys = sorted({k for _, d in sets for k in d})
B = int(os.environ.get("BATCH", "256"))
enc = ClipEncoder(VIT_B32, device=0, max_batch_image=B, max_batch_text=8)
enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
x = torch.randn(B, 3, 224, 224, device="cuda")
if os.environ.get("TOWER") == "text":   # 77-token prompts through the text tower instead
    import numpy as np
    enc.close()
    enc = ClipEncoder(VIT_B32, device=0, max_batch_image=8, max_batch_text=B)
    enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
    ids = np.full((B, 77), 49407, dtype=np.int32); ids[:, 0] = 49406; ids[:, 1:76] = 1000
    x = torch.from_numpy(ids).cuda()
    enc.encode_image = lambda t: enc.encode_text(t)
defaults = {k: int(os.environ.get("DEFAULT_" + k, "0")) for k in keys}

def apply(d):
    for k in keys:
        _lib.set_option(k, d.get(k, defaults[k]))

out = {name: {"wall_ms": [], "kernels": {}} for name, _ in sets}
for rep in range(6):
    for name, d in sets:
        apply(d)
        for _ in range(3):
            enc.encode_image(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            enc.encode_image(x)
        torch.cuda.synchronize()
        out[name]["wall_ms"].append(round((time.perf_counter() - t0) / 30 * 1e3, 4))
for name, d in sets:
    apply(d)
    _lib.prof_filter(None, 1); _lib.prof_enable(True); _lib.prof_reset()
    for _ in range(10):
        enc.encode_image(x)
    torch.cuda.synchronize()
    for k in _lib.prof_read():
        if k["kernel"].startswith("gemm") or k["kernel"] in ("attention", "layernorm"):
            out[name]["kernels"][k["kernel"]] = round(k["ms"] / k["launches"] * 1e3, 2)
    _lib.prof_enable(False)
for name, _ in sets:
    w = sorted(out[name]["wall_ms"])
    print(f"{name:50s} median {w[len(w)//2]:.4f} ms  min {w[0]:.4f}  " + " ".join(f"{k.replace('gemm_bf16_','')}={v}" for k, v in out[name]["kernels"].items()), flush=True)

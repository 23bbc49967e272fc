#!/usr/bin/env python3
"""Probe: does running two half batches concurrently on two HIP streams (two encoder handles, 128 images each) beat one
batch of 256 on one stream? If kernels of different character (GEMM main loops vs bandwidth-bound epilogues, attention)
overlap across streams the pair should finish sooner. Prints images/s for 1 x 256, 2 x 128 concurrent, 2 x 128 staggered."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd.encoder import VIT_B32, ClipEncoder, random_state_dict  # noqa: E402

W = random_state_dict(VIT_B32, seed=0)
big = ClipEncoder(VIT_B32, device=0, max_batch_image=256, max_batch_text=8)
big.load_state_dict(W)
halves = []
for _ in range(2):
    e = ClipEncoder(VIT_B32, device=0, max_batch_image=128, max_batch_text=8)
    e.load_state_dict(W)
    halves.append(e)
x = torch.randn(256, 3, 224, 224, device="cuda")
xs = [x[:128].contiguous(), x[128:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
out_big = torch.empty(256, 512, device="cuda")
outs = [torch.empty(128, 512, device="cuda") for _ in range(2)]


def run_big(n):
    for _ in range(n):
        big.encode_image(x, out=out_big)


def run_pair(n):
    for _ in range(n):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                halves[i].encode_image(xs[i], out=outs[i])


def timed(fn, n=30):
    fn(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


res = {}
for rep in range(3):
    res.setdefault("one_x_256_ms", []).append(round(timed(run_big) * 1e3, 4))
    res.setdefault("two_x_128_concurrent_ms", []).append(round(timed(run_pair) * 1e3, 4))
res["images_per_s_one"] = round(256 / (min(res["one_x_256_ms"]) * 1e-3), 1)
res["images_per_s_pair"] = round(256 / (min(res["two_x_128_concurrent_ms"]) * 1e-3), 1)
print(json.dumps(res))

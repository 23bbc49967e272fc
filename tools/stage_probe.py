#!/usr/bin/env python3
"""Host buffers -> embeddings: the pinned staging ring of csrc/host_stager.h against plain hipMemcpyAsync from pageable memory,
over host thread counts and block sizes. f32 pixels (4 x 256 images per call) and raw 640 x 480 RGB8 uploads (4 x 256)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from mmiss_amd.encoder import ClipEncoder, VIT_B32, random_state_dict

print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
B = 256
px4 = np.random.default_rng(0).standard_normal((4 * B, 3, 224, 224), dtype=np.float32)
raw = np.random.default_rng(1).integers(0, 256, (4 * B, 480, 640, 3), dtype=np.uint8).reshape(-1)
offs = np.arange(4 * B, dtype=np.int64) * (480 * 640 * 3)
hs, ws = np.full(4 * B, 480, np.int32), np.full(4 * B, 640, np.int32)

def run(tag):
    enc = ClipEncoder(VIT_B32, device=0, max_batch_image=B, max_batch_text=8)
    enc.load_state_dict(random_state_dict(VIT_B32, seed=0))
    out = {}
    for name, fn, nbytes in (("f32 pixels", lambda: enc.encode_image(px4), px4.nbytes),
                             ("rgb8 640x480", lambda: enc.encode_image_rgb_packed(raw, offs, hs, ws), raw.nbytes)):
        fn()
        t = []
        for _ in range(3):
            t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
        dt = min(t)
        out[name] = f"{4 * B / dt / 1e3:6.1f} k img/s {nbytes / dt / 1e9:5.1f} GB/s"
    enc.close()
    print(f"{tag:34s}", out, flush=True)

_lib.set_option("pinned_stage", 0)
run("pageable hipMemcpyAsync")
_lib.set_option("pinned_stage", 1)
for thr in (4, 8, 12, 16):
    for mb in (8, 16, 64):
        _lib.set_option("stage_threads", thr)
        _lib.set_option("stage_block_mb", mb)
        run(f"pinned ring threads={thr} block={mb}MB")

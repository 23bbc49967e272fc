#!/usr/bin/env python3
"""Does running sub-batches on several HIP streams overlap the GEMM epilogue bursts with other sub-batches' MFMA loops?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd.encoder import ClipEncoder, VIT_B32, random_state_dict

W = random_state_dict(VIT_B32, 0)
for nsplit in (1, 2, 4):
    B = 256 // nsplit
    encs = []
    for i in range(nsplit):
        e = ClipEncoder(VIT_B32, max_batch_image=B, max_batch_text=8)
        e.load_state_dict(W)
        encs.append(e)
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    px = [torch.randn(B, 3, 224, 224, device="cuda") for _ in range(nsplit)]
    outs = [torch.empty(B, 512, device="cuda") for _ in range(nsplit)]
    def step():
        for e, s, p, o in zip(encs, streams, px, outs):
            with torch.cuda.stream(s):
                e.encode_image(p, out=o)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"splits={nsplit} sub-batch={B}: {dt*1e3:.3f} ms per 256 images -> {256/dt:.0f} img/s", flush=True)
    del encs

"""One bs-256 batch as TWO half batches on two HIP streams (two host threads, one encoder handle each), against the same
batch on one stream: does the other half's work fill the partly empty tile rounds of the wide GEMMs? Encode only (the query
stage of the step is 0.11 ms on either form). Prints ms per 256 images."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa: F401
from mmiss_amd.encoder import ClipEncoder, VIT_B32, random_state_dict
from mmiss_amd.pipeline import BatchLanes

dev = torch.device("cuda", 0)
B, D = 256, VIT_B32.proj_dim
W = random_state_dict(VIT_B32, seed=0)
gen = torch.Generator(device=dev).manual_seed(1234)
px = [torch.randn(B, 3, 224, 224, device=dev, generator=gen) for _ in range(4)]
STEPS = 40


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


enc = ClipEncoder(VIT_B32, device=0, max_batch_image=B, max_batch_text=8)
enc.load_state_dict(W)
emb = torch.empty(B, D, device=dev)
i = [0]


def one():
    enc.encode_image(px[i[0] % 4], out=emb)
    i[0] += 1


for rep in range(2):
    print("one stream, bs 256:            %.3f ms per 256 images" % timed(one, STEPS), flush=True)

for nl in (2, 4):
    h = B // nl
    encs = [ClipEncoder(VIT_B32, device=0, max_batch_image=h, max_batch_text=8) for _ in range(nl)]
    for e in encs:
        e.load_state_dict(W)
    embs = [torch.empty(h, D, device=dev) for _ in range(nl)]
    ref = torch.empty(B, D, device=dev)
    enc.encode_image(px[0], out=ref)

    def lane_step(lane, it):
        encs[lane].encode_image(px[it % 4][lane * h:(lane + 1) * h], out=embs[lane])

    with BatchLanes(nl, lane_step, device=dev) as bl:
        def batch():
            # every lane gets its part of the SAME batch; the batch is complete when all lanes are (drain)
            for lane in range(nl):
                bl._queues[lane].put((bl._submitted, i[0] % 4))
                bl._submitted += 1
            i[0] += 1

        def run(n):
            for _ in range(5):
                batch()
            bl.drain()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                batch()
            bl.drain()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        for rep in range(2):
            print("%d streams, bs %3d each:        %.3f ms per 256 images (batches back to back)" % (nl, h, run(STEPS)), flush=True)
        # one batch at a time: drain between batches
        def run_sync(n):
            t0 = time.perf_counter()
            for _ in range(n):
                batch()
                bl.drain()
            return (time.perf_counter() - t0) / n * 1e3
        run_sync(3)
        print("%d streams, bs %3d each:        %.3f ms per 256 images (drained after every batch)" % (nl, h, run_sync(STEPS)), flush=True)
        i[0] = 0
        batch()
        bl.drain()
        got = torch.cat(embs)
        print("   max |diff| vs the one-stream embeddings: %.3e" % (got - ref).abs().max().item(), flush=True)
    for e in encs:
        e.close()

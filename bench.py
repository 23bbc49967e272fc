#!/usr/bin/env python3
"""bench.py — throughput of the embed-and-retrieve hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., or
     plainly as above — then this file starts the N ranks itself as fresh child processes and relays rank 0's line)

Workload (BASELINE.json configs[1] as written): CLIP ViT-B/32 image encode at batch 256 on
synthetic, already CLIP-normalised 224x224 pixels resident in HBM, random-init weights (seed 0), each embedding then
queried top-10 (cosine) against the 100k x 512 flat index OF THESE IMAGES' EMBEDDINGS (built before the timed region;
every query is a row of the index and must rank itself first). One step = one batch of 256 images through
patchify -> 12 layers -> pool -> project -> L2-normalise -> index query; `value` = images/s over all ranks.

N > 1 (weak scaling, one process per GPU, RCCL): every rank encodes its own 256 images (no collective),
holds its own 100k-row shard (its own images' embeddings, labels are global), all-gathers the [256,512] embeddings so every rank
searches all N*256 queries in its shard, exchanges the per-shard top-10 in ONE packed all-gather and merges (X1).

Second half of BASELINE.json's metric, reported in the same JSON line under "retrieval": cosine top-10
over a 10M x 512 fp16 index (row-sharded over the ranks) at Q=1 (HBM-bound scan) and Q=1024.

The JSON line also carries `roofline` (dominant kernel class by device time: HIP events recorded by libmmiss
on the stream the kernel runs on INSIDE the timed region, on every 7th launch of that class only — bracketing
all ~100 launches of a step costs ~20 % of it; the full per-kernel table comes from an instrumented replay
of the same K steps right after the timed region) and `cpu_baseline` (a PyTorch-CPU fp32
restatement of the reference's CPU path at batch 1 and 32, one thread and all cores of the box's share, on a bounded
sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0           # spec; ~6.3 TB/s achievable (same guide)

KERNEL_SYMBOL = {  # libmmiss kernel class -> symbol as rocprofv3 --kernel-trace prints it
    "gemm_bf16_f32": "gemm16_kernel<__bf16,BM,0>", "gemm_bf16_bias": "gemm16_kernel<__bf16,BM,1>",
    "gemm_bf16_bias_qgelu": "gemm16_kernel<__bf16,BM,2>", "gemm_bf16_bias_resid": "gemm16_kernel<__bf16,BM,3>",
    "gemm_bf16_bias_resid16": "gemm16_kernel<__bf16,BM,9>", "gemm_bf16_bias_resid16_p160": "gemm160p_kernel<9,0,K/64> (out-projection K = 768: <9,0,12>; FC2 K = 3072: <9,0,48>)",
    "gemm_bf16_patch_p160": "gemm160p_kernel<4,0,0>",
    "gemm_bf16_lnfold_bias": "gemm256_kernel<__bf16,7> (>= 85 % tile fill) / gemm16_kernel<__bf16,BM,7>", "gemm_bf16_lnfold_qgelu": "gemm16_kernel<__bf16,BM,8>",
    "gemm_bf16_lnfold_bias_p256": "gemm256p_kernel<7,K/256,0,0>", "gemm_bf16_lnfold_qgelu_p256": "gemm256p_kernel<8,K/256,0,0>",
    "gemm_bf16_bias_p256": "gemm256p_kernel<1,0,0,0>", "gemm_bf16_bias_qgelu_p256": "gemm256p_kernel<2,0,0,0>",
    "gemm_bf16_patch": "gemm16_kernel<__bf16,BM,4>", "score_gemm_f16": "gemm16_kernel<_Float16,128,5>",
    "attention": "attention_heads_kernel<2,false,4> (short sequences at large batch) / attention_kernel<NKP,causal> / attention_long_kernel<NKP,causal,mx> (> 128 keys) / attention_stream_kernel<mx> (257 keys, >= 256 (item, head) pairs: round 5)",
    "layernorm": "layernorm_kernel<true>", "im2col": "im2col_kernel<false>",
    "scan_topk_f16": "scan_topk_kernel<_Float16,...>", "scan_topk_f32": "scan_topk_kernel<float,...>",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--index-rows", type=int, default=100_000, help="rows per rank of the step's index")
    ap.add_argument("--retrieval-rows", type=int, default=10_000_000, help="total rows of the 10M x 512 f16 scan benchmark (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-text", action="store_true", help="skip the informational text-tower timing")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the instrumented replay (no roofline object)")
    ap.add_argument("--query-stream", choices=("own", "same"), default="same",
                    help="own: the query stage of step i on a second HIP stream beside the encode of step i+1; same: one stream")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: start the N ranks as FRESH child
    processes (python -m torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) and relay rank 0's JSON line and
    the exit code. This parent never imports torch and never touches the GPU (a process that has initialised HIP must
    not be replaced or forked into ranks)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # dmabuf IPC. The pool's images export this already (GPU box and build container alike): the host driver supports only
    # dmabuf IPC, without it cross-process sharing of device memory (RCCL's intra-node transport) fails in
    # hipIpcGetMemHandle. setdefault() only covers a launch from a scrubbed environment; nothing here could verify it on
    # hardware (no multi-GPU box is reachable from this repository's tooling).
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, cwd=ROOT)
    return proc.returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world  # under torch.distributed.run the launcher's world size is the truth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    local_rank = local_rank % torch.cuda.device_count()  # (the gloo dry run puts several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("MMISS_DIST_BACKEND", "nccl")  # "gloo" = dry run of the N>1 control flow on a 1-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def all_gather(out, inp):
        """all_gather_into_tensor on device tensors (RCCL); the gloo dry run stages through host memory."""
        if backend == "nccl":
            dist.all_gather_into_tensor(out, inp)
        else:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu())
            out.copy_(o)

    def all_reduce_max(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    import mmiss_amd  # noqa: F401
    from mmiss_amd import _lib
    from mmiss_amd.encoder import ClipEncoder, VIT_B32, random_state_dict
    from mmiss_amd.index import FlatIndex, merge_topk
    from mmiss_amd.sharded import exchange_topk

    B, D, K_TOP = args.batch, VIT_B32.proj_dim, 10
    W = random_state_dict(VIT_B32, seed=0)
    enc = ClipEncoder(VIT_B32, device=local_rank, max_batch_image=B, max_batch_text=B)
    enc.load_state_dict(W)

    # configs[1] AS WRITTEN (reference flow backend/app/main.py:735-765): the step's index IS the encoder's
    # embeddings of the config's images, and every embedding is queried against it (itself must rank first). Per rank: index_rows
    # seeded N(0,1) images (seed 4321 + rank, consecutive draws of 256) are encoded and added with global labels
    # rank * index_rows + i before the timed region (1.1 s); the first NROT batches stay resident in HBM (154 MB each) and step i
    # encodes batch i % NROT, so every query of the timed region is a row of the index. Random-weight embeddings sit at pairwise
    # cosine ~0.99 — closer than the first pass's error bound — so the exactness guard WIDENS every query (one threshold pass,
    # DESIGN.md section 4): that cost is inside `value`. The same step on an index of random rows (rounds 1-5's headline, where
    # the guard proves ~every query from the first pass) is the secondary leg `random_index`.
    NS = args.index_rows
    NROT = max(1, min(args.steps, 8))
    index = FlatIndex(D, "f16", device=local_rank, capacity=NS)
    g1 = torch.Generator(device=dev).manual_seed(4321 + rank)
    pixel_batches, tmp_e = [], torch.empty(B, D, device=dev)
    t_ing = time.perf_counter()
    for b0 in range(0, NS, B):
        n = min(B, NS - b0)
        px1 = torch.randn(B, 3, 224, 224, device=dev, generator=g1)
        enc.encode_image(px1, out=tmp_e)
        index.add(tmp_e[:n], np.arange(rank * NS + b0, rank * NS + b0 + n, dtype=np.int64))
        if len(pixel_batches) < NROT and (n == B or not pixel_batches):
            pixel_batches.append(px1)
    torch.cuda.synchronize()
    ingest_index_s = time.perf_counter() - t_ing
    del tmp_e, px1
    NROT = len(pixel_batches)
    batches_in_index = NS >= NROT * B
    pixels = pixel_batches[0]
    step_no = [0]
    # the random-row index of the secondary legs (one request at a time, host-buffer steps, `random_index`)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    rindex = FlatIndex(D, "f16", device=local_rank, capacity=NS)
    rows = torch.randn(NS, D, device=dev, generator=gen)
    rindex.add(rows, np.arange(rank * NS, (rank + 1) * NS, dtype=np.int64))
    del rows
    emb = torch.empty(B, D, device=dev)
    emb_all = torch.empty(world * B, D, device=dev) if world > 1 else emb

    # One step = encode batch i -> (N > 1: all-gather the queries) -> top-10 of every query -> (N > 1: exchange + merge).
    # The steps are software-pipelined one deep, as a serving loop would run them: step i+1's encode is QUEUED before the host
    # waits for step i's query (mmiss_index_query_begin / _end: the exactness guard needs one host decision per query batch),
    # so the GPU does not idle for the ~40 us the host needs to wake up, return through Python and launch again. Every timed
    # loop ends with drain(): all K encodes AND all K result sets are complete inside the timed region.
    pending = [None]
    # --query-stream own: the query stage (first pass, and the widen pass query_end decides on) runs on its OWN HIP stream, ordered
    # behind the encode that produced its embeddings by an event, so the host's wait for a step's results no longer waits for the
    # next step's encode and the two overlap on the chip. The embeddings ping-pong between two buffers (an encode may only
    # overwrite a buffer whose first pass is complete: event `read_done`).
    # Measured (round 6, four alternating runs on one box): 2.94 ms per step against 3.00-3.06 on one stream (+2-4 %), while the
    # dominant GEMM's launches slow down under the threshold GEMM beside them (roofline.frac 0.31-0.32 against 0.33): the one-stream
    # form stays the default and the headline; the leg `query_on_own_stream` times this form for the same K steps.
    qs = [torch.cuda.Stream(device=dev) if args.query_stream == "own" else None]
    embs = [emb, torch.empty_like(emb)]
    emb_alls = [emb_all, torch.empty_like(emb_all)] if world > 1 else embs
    read_done = [None, None]
    cur_emb = [emb]

    def finish(h):
        return finish_results(h.result())

    def finish_results(res):
        lab, dst, cnt = res
        if world > 1:
            lab_all, dst_all = exchange_topk(lab, dst, world, all_gather=all_gather)  # X1: ONE packed all-gather
            return merge_topk(dst_all, lab_all)
        return lab, dst, cnt

    def step():
        qstream = qs[0]
        sl = (step_no[0] & 1) if qstream else 0
        e, ea = embs[sl], emb_alls[sl]
        if qstream and read_done[sl] is not None:
            torch.cuda.current_stream().wait_event(read_done[sl])
        enc.encode_image(pixel_batches[step_no[0] % NROT], out=e)
        step_no[0] += 1
        cur_emb[0] = e
        if world > 1:
            all_gather(ea, e)                            # queries: every rank searches all N*256 embeddings in its shard
        prev, pending[0] = pending[0], None
        if prev is not None and qstream is None:
            # the PREVIOUS step's results and this step's first pass back to back (FlatIndex.query_next: no Python between the two)
            res, pending[0] = index.query_next(prev, ea, K_TOP)
            return finish_results(res)
        out = finish(prev) if prev is not None else None   # the PREVIOUS step's results, while this step's encode runs
        if qstream:
            ready = torch.cuda.Event()
            ready.record()                               # this step's embeddings are complete (main stream)
            with torch.cuda.stream(qstream):
                qstream.wait_event(ready)
                pending[0] = index.query_begin(ea, K_TOP)
                read_done[sl] = torch.cuda.Event()
                read_done[sl].record()
        else:
            pending[0] = index.query_begin(ea, K_TOP)
        return out

    def drain():
        prev, pending[0] = pending[0], None
        return finish(prev) if prev is not None else None

    def step_unpipelined():
        enc.encode_image(pixel_batches[step_no[0] % NROT], out=emb)
        step_no[0] += 1
        if world > 1:
            all_gather(emb_all, emb)
            lab, dst, _ = index.query(emb_all, K_TOP)
            lab_all, dst_all = exchange_topk(lab, dst, world, all_gather=all_gather)
            return merge_topk(dst_all, lab_all)
        return index.query(emb, K_TOP)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # warm-up; its last step is fully instrumented to find the dominant kernel class
    dominant = None
    for i in range(args.warmup):
        last = (i == args.warmup - 1) and not args.no_kernel_events
        if last:
            fence()
            _lib.prof_reset()
            _lib.prof_enable(True)
        step()
        if last:
            drain()
            fence()
            _lib.prof_enable(False)
            wp = _lib.prof_read()
            dominant = max(wp, key=lambda p: p["ms"])["kernel"] if wp else None
    drain()
    fence()
    # timed region: EXACTLY K steps. Bracketing EVERY launch with HIP events costs ~20 % of a 3.7 ms step (100
    # launches x 2 event packets; A/B in profiles/), so inside the timed region only the dominant kernel class is
    # bracketed, and only every 7th launch of it (7 is coprime to the launch pattern, so all its shapes are sampled).
    _lib.prof_reset()
    if dominant:
        _lib.prof_filter(dominant, 7)
        _lib.prof_enable(True)
    gs_before = index.guard_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    last_out = drain()
    fence()
    elapsed = time.perf_counter() - t0
    # outside the timed region: the last pipelined result against the synchronous query of the same embeddings
    gs_timed = index.guard_stats()
    pipelined_equals_sync = None
    if world == 1 and last_out is not None:
        lab_s, dst_s, _c = index.query(cur_emb[0], K_TOP)
        pipelined_equals_sync = bool(torch.equal(last_out[0], lab_s) and torch.equal(last_out[1], dst_s))
    # every query of the last step is a row of the index: it must come back first, at distance ~0 (all ranks' queries, global labels)
    self_first, max_self_dist = None, None
    if last_out is not None and batches_in_index:
        last_b = (step_no[0] - 1) % NROT
        want = torch.cat([r * NS + last_b * B + torch.arange(B) for r in range(world)])
        self_first = bool((torch.as_tensor(last_out[0])[:, 0].cpu() == want).all())
        max_self_dist = float(torch.as_tensor(last_out[1])[:, 0].max().item())
    _lib.prof_enable(False)
    timed_prof = _lib.prof_read() if dominant else []
    _lib.prof_filter(None, 1)
    # the same K steps again, every kernel bracketed by HIP events on the stream it runs on -> per-kernel durations
    prof = []
    events_ms_per_step = None
    if not args.no_kernel_events:
        _lib.prof_reset()
        _lib.prof_enable(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        fence()
        events_ms_per_step = (time.perf_counter() - t1) * 1e3 / args.steps
        _lib.prof_enable(False)
        prof = _lib.prof_read()
    # the same K steps with the synchronous query (each step waits for its own results before the next encode is queued)
    fence()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        step_unpipelined()
    fence()
    unpipelined = time.perf_counter() - t2
    if world > 1:
        elapsed = all_reduce_max(elapsed)
        unpipelined = all_reduce_max(unpipelined)
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * B * args.steps / elapsed

    # ---------------------------------------------------------------- N > 1: what a step is made of, stage by stage
    distributed = None
    if world > 1:
        def timed_stage(fn, reps=10):
            fence()
            t = time.perf_counter()
            for _ in range(reps):
                r = fn()
            torch.cuda.synchronize()
            return r, all_reduce_max((time.perf_counter() - t) * 1e3 / reps)

        _, ms_enc = timed_stage(lambda: enc.encode_image(pixels, out=emb))
        _, ms_gq = timed_stage(lambda: all_gather(emb_all, emb))
        (lab_l, dst_l, _c), ms_q = timed_stage(lambda: index.query(emb_all, K_TOP))
        (lab_a, dst_a), ms_x = timed_stage(lambda: exchange_topk(lab_l, dst_l, world, all_gather=all_gather))
        _, ms_m = timed_stage(lambda: merge_topk(dst_a, lab_a))
        dc = torch.tensor([torch.cuda.device_count()], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dcs = [torch.zeros_like(dc) for _ in range(world)]
        dist.all_gather(dcs, dc)
        distributed = {
            "world_size": dist.get_world_size(), "backend": backend + (" (RCCL)" if backend == "nccl" else " (dry run: collectives staged through host memory)"),
            "device_count_seen_by_each_rank": [int(x.item()) for x in dcs],
            "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None),
            "local_rank_to_device": {"rank": rank, "local_rank": local_rank, "device": str(dev)},
            "queries_per_rank_and_step": world * B,
            "query_gather_bytes_per_rank": B * D * 4, "topk_exchange_bytes_per_rank": world * B * K_TOP * 12,
            "stage_ms_max_over_ranks": {"encode": round(ms_enc, 3), "all_gather_queries": round(ms_gq, 3),
                                        "local_query": round(ms_q, 3), "all_gather_topk": round(ms_x, 3), "merge": round(ms_m, 3)},
            "stage_note": "each stage timed alone (10 repetitions between barriers), not inside the timed region",
            "weak_scaling_note": "every query must visit every shard: each rank searches ALL N x 256 queries of the step in its "
                                 "own 100k-row shard, so the local_query stage grows N-fold per rank while the encode stage "
                                 "does not. Read the efficiency of value(N) / (N x value(1)) against (encode + query(256)) / "
                                 "(encode + query(N x 256) + exchange + merge), not against 1.0",
        }

    # ---------------------------------------------------------------- roofline of the dominant kernel class
    roofline = None
    kernels = []
    if prof:
        tot_ms = sum(p["ms"] for p in prof)
        for p in sorted(prof, key=lambda p: -p["ms"]):
            kernels.append({"kernel": p["kernel"], "launches": p["launches"], "avg_us": round(1e3 * p["ms"] / p["launches"], 2),
                            "share": round(p["ms"] / tot_ms, 4),
                            "tflops": round(p["flops"] / p["ms"] / 1e9, 1) if p["flops"] else None,
                            "gbs": round(p["bytes"] / p["ms"] / 1e6, 1)})
        # the roofline numbers come from the launches sampled INSIDE the timed region; the replay gives the table
        top = timed_prof[0] if timed_prof else max(prof, key=lambda p: p["ms"])
        n_l = top["launches"]
        t_s = top["ms"] / n_l * 1e-3                      # mean launch duration, HIP events on the kernel's own stream
        flops_l, bytes_l = top["flops"] / n_l, top["bytes"] / n_l   # ALGORITHMIC work per launch (operands + outputs once)
        tflops, gbs = flops_l / t_s / 1e12, bytes_l / t_s / 1e9
        # which roof binds this shape: the one that needs more time at its peak
        t_mfma, t_hbm = flops_l / (MFMA_BF16_PEAK_TFLOPS * 1e12), bytes_l / (HBM_PEAK_GBS * 1e9)
        bound = "mfma" if t_mfma >= t_hbm else "hbm"
        # HBM traffic per launch of that kernel class: PMC counters cannot be read from inside this process; when present the
        # value is the rocprofv3 FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE measurement of the SAME command taken in
        # separate --pmc passes and kept in profiles/ — i.e. NOT measured in this run
        traffic, traffic_src = None, None
        for fn in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", fn)) as f:
                    tj = json.load(f)
                    t = (tj.get(top["kernel"]) or tj.get(top["kernel"].split("_k")[0]) or {}).get("traffic_bytes")
                if t is not None:
                    traffic, traffic_src = t, f"profiles/{fn}: separate rocprofv3 --pmc passes of this command, not this run"
                    break
            except Exception:
                pass
        roofline = {"bound": bound, "kernel": top["kernel"], "symbol": KERNEL_SYMBOL.get(top["kernel"].split("_k")[0], top["kernel"]),
                    "achieved": round(tflops if bound == "mfma" else gbs, 1),
                    "peak": MFMA_BF16_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS,
                    "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                    "frac": round((tflops / MFMA_BF16_PEAK_TFLOPS) if bound == "mfma" else (gbs / HBM_PEAK_GBS), 4),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "mfma": {"achieved_tflops": round(tflops, 1), "frac": round(tflops / MFMA_BF16_PEAK_TFLOPS, 4)},
                    "hbm": {"achieved_gbs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)},
                    "arithmetic_intensity_flop_per_byte": round(flops_l / bytes_l, 1),
                    "ridge_flop_per_byte": round(MFMA_BF16_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS, 1),
                    "sampled_launches_in_timed_region": n_l, "avg_launch_us": round(t_s * 1e6, 2),
                    "flops_per_launch": flops_l, "bytes_per_launch": bytes_l}

    # ---------------------------------------------------------------- 10M x 512 f16 scan (second half of the metric)
    retrieval = None
    if args.retrieval_rows > 0:
        retrieval = bench_retrieval(args, torch, dist, np, dev, rank, world, local_rank, FlatIndex, merge_topk, _lib,
                                    all_gather, all_reduce_max, exchange_topk)

    # ---------------------------------------------------------------- text tower (BASELINE configs[2] inputs), informational
    text = None
    if not args.no_text:
        rng = np.random.Generator(np.random.Philox(2 + rank))
        ids = np.full((B, 77), 49407, dtype=np.int32)
        ids[:, 0] = 49406
        eos = rng.integers(2, 77, size=B)
        body = rng.integers(0, 49406, size=(B, 77), dtype=np.int32)
        for r in range(B):
            ids[r, 1:eos[r]] = body[r, 1:eos[r]]
        ids_d = torch.from_numpy(ids).to(dev)
        temb = torch.empty(B, D, device=dev)
        for _ in range(3):
            enc.encode_text(ids_d, out=temb)
        fence()
        t0 = time.perf_counter()
        for _ in range(10):
            enc.encode_text(ids_d, out=temb)
        fence()
        tdt = (time.perf_counter() - t0) / 10
        text = {"texts_per_s": round(world * B / tdt, 1), "ms_per_batch": round(tdt * 1e3, 3), "batch": B, "tokens": 77,
                "tflops": round(B * 5.960e9 / tdt / 1e12, 1), "flops_per_text": 5.960e9}

    # ---------------------------------------------------------------- the reference's own regime: one request at a time
    latency = None
    if rank == 0 and world == 1 and not args.no_text:  # single-GPU diagnostics; at N > 1 rank 0 must not lag the others
        one_px = pixels[:1].contiguous()
        one_emb = torch.empty(1, D, device=dev)

        def one_image_query():
            enc.encode_image(one_px, out=one_emb)
            return rindex.query(one_emb, K_TOP)

        for _ in range(5):
            one_image_query()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            one_image_query()
        torch.cuda.synchronize()
        img_ms = (time.perf_counter() - t0) / 50 * 1e3
        host_px = one_px.cpu().numpy()
        t0 = time.perf_counter()
        for _ in range(20):
            e1 = enc.encode_image(host_px)            # host pixels in, host embedding out (PCIe both ways + sync)
            rindex.query(e1, K_TOP)
        host_ms = (time.perf_counter() - t0) / 20 * 1e3
        one_ids = ids_d[:1, :16].contiguous().clone()   # a short prompt: BOS + 14 tokens + EOS (padding trimmed)
        one_ids[0, 15] = 49407
        one_ids[0, 1:15] = torch.clamp(one_ids[0, 1:15], max=49405)

        def one_text_query():
            enc.encode_text(one_ids, out=one_emb)
            return rindex.query(one_emb, K_TOP)

        for _ in range(5):
            one_text_query()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            one_text_query()
        torch.cuda.synchronize()
        txt_ms = (time.perf_counter() - t0) / 50 * 1e3
        _lib.prof_filter(None, 1)
        _lib.prof_enable(True)
        _lib.prof_reset()
        for _ in range(5):
            one_image_query()
        torch.cuda.synchronize()
        one_kernels = sorted(([k["kernel"], k["launches"] // 5, round(k["ms"] / 5 * 1e3, 1)] for k in _lib.prof_read()),
                             key=lambda r: -r[2])
        _lib.prof_enable(False)
        latency = {"image_encode_plus_top10_ms_device_resident": round(img_ms, 3),
                   "image_encode_plus_top10_ms_host_buffers": round(host_ms, 3),
                   "text16_encode_plus_top10_ms_device_resident": round(txt_ms, 3), "batch": 1,
                   "kernels_per_request": one_kernels, "kernels_per_request_columns": ["kernel", "launches", "us"],
                   "note": "the reference runs batch 1 on the CPU (backend/app/utils.py:76-77)"}

    # ---------------------------------------------------------------- ingest: raw RGB uploads -> embeddings
    ingest = None
    if rank == 0 and world == 1 and not args.no_text:  # single-GPU diagnostics; at N > 1 rank 0 must not lag the others
        IH, IW = 480, 640
        raw = torch.randint(0, 256, (B, IH, IW, 3), dtype=torch.uint8, device=dev)
        offs = np.arange(B, dtype=np.int64) * (IH * IW * 3)
        hs, ws = np.full(B, IH, np.int32), np.full(B, IW, np.int32)
        iemb = torch.empty(B, D, device=dev)
        for _ in range(3):
            enc.encode_image_rgb_packed(raw, offs, hs, ws, out=iemb)
        torch.cuda.synchronize()  # rank 0 only: no barrier here
        t0 = time.perf_counter()
        for _ in range(10):
            enc.encode_image_rgb_packed(raw, offs, hs, ws, out=iemb)
        torch.cuda.synchronize()  # rank 0 only: no barrier here
        idt = (time.perf_counter() - t0) / 10
        _lib.prof_filter(None, 1)
        _lib.prof_enable(True)
        _lib.prof_reset()
        for _ in range(5):
            enc.encode_image_rgb_packed(raw, offs, hs, ws, out=iemb)
        torch.cuda.synchronize()  # rank 0 only: no barrier here
        rz = {k["kernel"]: k["ms"] / k["launches"] for k in _lib.prof_read() if k["kernel"].startswith("resize")}
        _lib.prof_enable(False)
        # host buffers: 4 chunks of B images, so that the library's copy/compute pipeline has something to overlap
        NH = 4 * B
        raw_host = np.tile(raw.cpu().numpy().reshape(-1), 4)
        offs_h = np.arange(NH, dtype=np.int64) * (IH * IW * 3)
        hs_h, ws_h = np.full(NH, IH, np.int32), np.full(NH, IW, np.int32)
        enc.encode_image_rgb_packed(raw_host, offs_h, hs_h, ws_h)
        t0 = time.perf_counter()
        for _ in range(2):
            enc.encode_image_rgb_packed(raw_host, offs_h, hs_h, ws_h)
        hdt = (time.perf_counter() - t0) / 2 / 4
        rz_ms = rz.get("resize_crop", 0.0)
        ingest = {"images_per_s_device_resident": round(B / idt, 1), "images_per_s_host_buffers": round(B / hdt, 1),
                  "host_buffers": f"{NH} images per call in pageable host memory, chunks of {B}: the next chunk is copied while the "
                                  "current one is computed",
                  "batch": B, "source": f"{IW}x{IH} RGB8", "resize_crop_kernel_ms": round(rz_ms, 4),
                  "resize_coeffs_kernel_ms": round(rz.get("resize_coeffs", 0.0), 4),
                  "resize_crop_gbs": round((B * IH * IW * 3 + B * 224 * 224 * 3) / (rz_ms * 1e-3) / 1e9, 1) if rz_ms else None,
                  "note": "resize(shortest edge 224, bicubic, Pillow-exact) + centre crop + rescale + normalise + ViT-B/32"}

    # ---------------------------------------------------------------- the step with its pixels handed over as HOST buffers
    # The contract's `value` starts with inputs resident in HBM. The boundary also takes host buffers (the reference's
    # processor leaves float32 pixel_values on the CPU, backend/app/utils.py:76); this is the PCIe-inclusive rate of the same
    # step: 154 MB of float32 pixels per batch cross the bus (pageable numpy memory, copied in chunks behind the compute).
    pcie = None
    if rank == 0 and world == 1 and not args.no_text:
        px_host = pixel_batches[0].cpu().numpy()

        def hstep(px):
            return rindex.query(enc.encode_image(px), K_TOP)   # host in, host out at both calls

        def timed(px, n):
            for _ in range(2):
                hstep(px)
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(n):
                hstep(px)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0_) / n

        pdt1 = timed(px_host, 5)          # ONE batch per call: the copy cannot hide behind anything (round 4's figure)
        # four batches per call (the boundary takes any B): the library moves batch k + 1 across PCIe while batch k is computed;
        # the link itself gives 40-45 GB/s on the pool's boxes (tools/stage_probe.py), with or without the pinned ring
        px4 = np.ascontiguousarray(np.concatenate([pixel_batches[i % len(pixel_batches)].cpu().numpy() for i in range(4)]))
        pdt = timed(px4, 3) / 4
        _lib.set_option("pinned_stage", 1)   # (option: the library's own ring of pinned blocks + parallel host copies)
        try:
            pdt_ring = timed(px4, 2) / 4
        finally:
            _lib.set_option("pinned_stage", 0)
        del px4
        pcie = {"images_per_s": round(B / pdt, 1), "ms_per_step": round(pdt * 1e3, 3),
                "bytes_per_step_over_pcie": int(px_host.nbytes), "gbs_over_pcie": round(px_host.nbytes / pdt / 1e9, 1),
                "one_batch_per_call": {"images_per_s": round(B / pdt1, 1), "ms_per_step": round(pdt1 * 1e3, 3),
                                       "gbs_over_pcie": round(px_host.nbytes / pdt1 / 1e9, 1)},
                "with_pinned_ring_option": {"images_per_s": round(B / pdt_ring, 1), "gbs_over_pcie": round(px_host.nbytes / pdt_ring / 1e9, 1)},
                "note": "the timed step with float32 pixel_values in pageable host memory, four batches of 256 per call (encode, "
                        "then query; batch k + 1 crosses PCIe while batch k is computed); `one_batch_per_call`: nothing to "
                        "overlap; NOT `value`, which is measured with the inputs resident in HBM"}

    # ---------------------------------------------------------------- the same step with TWO batches in flight (N = 1 only)
    lanes = None
    if rank == 0 and world == 1 and not args.no_text:
        from mmiss_amd.pipeline import BatchLanes

        enc2 = ClipEncoder(VIT_B32, device=local_rank, max_batch_image=B, max_batch_text=8)
        enc2.load_state_dict(W)
        lane_enc = [enc, enc2]
        lane_emb = [emb, torch.empty(B, D, device=dev)]

        def lane_step(lane, _):
            lane_enc[lane].encode_image(pixels, out=lane_emb[lane])
            index.query(lane_emb[lane], K_TOP)       # ONE shared index handle: its calls serialise, the encodes do not

        with BatchLanes(2, lane_step, device=dev) as bl:
            bl.map([None] * 6)
            fence()
            t0 = time.perf_counter()
            bl.map([None] * args.steps)
            fence()
            ldt = (time.perf_counter() - t0) / args.steps
        enc2.close()
        lanes = {"lanes": 2, "images_per_s": round(B / ldt, 1), "ms_per_step": round(ldt * 1e3, 3),
                 "vs_one_batch_at_a_time": round(ms_per_step / (ldt * 1e3), 3),
                 "note": "the timed step above, K steps, issued from two host threads on two HIP streams (mmiss_amd.pipeline."
                         "BatchLanes: one encoder handle per lane, one shared index handle): the other batch's workgroups fill "
                         "the partly empty last round of tiles and the store burst of every GEMM. NOT the headline value: kernel "
                         "durations overlap in this mode, so the roofline object is measured one batch at a time"}

    # ---------------------------------------------------------------- the same step with the query stage on its own stream (N = 1 only)
    own_stream_leg = None
    if rank == 0 and world == 1 and not args.no_text and qs[0] is None:
        qs[0] = torch.cuda.Stream(device=dev)
        read_done[0] = read_done[1] = None
        try:
            for _ in range(3):
                step()
            drain()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            o_out = drain()
            fence()
            odt = (time.perf_counter() - t0) / args.steps
            lab_o, dst_o, _c = index.query(cur_emb[0], K_TOP)
            o_same = bool(torch.equal(o_out[0], lab_o) and torch.equal(o_out[1], dst_o))
        finally:
            qs[0] = None
        own_stream_leg = {"images_per_s": round(B / odt, 1), "ms_per_step": round(odt * 1e3, 3), "vs_headline_step": round(odt * 1e3 / ms_per_step, 3),
                          "last_result_equals_synchronous_query": o_same,
                          "note": "the headline's step with the query stage of step i (first pass + widen pass) on a second HIP stream beside the "
                                  "encode of step i+1, ordered behind its own encode by an event (bench.py --query-stream own makes it the timed "
                                  "step). NOT the headline: kernels of the two stages overlap, the dominant GEMM's launches run ~5 % longer"}

    # ---------------------------------------------------------------- the same step on an index of RANDOM rows (N = 1 only)
    # Rounds 1-5 timed this as the headline: unit-normalised N(0,1) rows, where the exactness guard proves ~every query from the
    # first pass (no widen pass). Same pipelined step, same encoder, same batches; only the index differs. Also here: what the
    # widen pass costs the query stage of the headline's own index (guard on / off, outside any timed step).
    random_leg = None
    if rank == 0 and world == 1 and not args.no_text:
        spend = [None]

        def rstep(i):
            enc.encode_image(pixel_batches[i % NROT], out=emb)
            prev, spend[0] = spend[0], None
            out = prev.result() if prev is not None else None
            spend[0] = rindex.query_begin(emb, K_TOP)
            return out

        def rdrain():
            prev, spend[0] = spend[0], None
            return prev.result() if prev is not None else None

        for i in range(3):
            rstep(i)
        rdrain()
        fence()
        gs0 = rindex.guard_stats()
        t0 = time.perf_counter()
        for i in range(args.steps):
            rstep(i)
        rdrain()
        fence()
        rdt = (time.perf_counter() - t0) / args.steps
        gs1 = rindex.guard_stats()

        def q_ms(ix, reps=10):
            ix.query(emb, K_TOP)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                ix.query(emb, K_TOP)
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps * 1e3

        enc.encode_image(pixels, out=emb)
        q_full, q_rand = q_ms(index), q_ms(rindex)
        _lib.set_option("exact_guard", 0)
        try:
            q_first = q_ms(index)
        finally:
            _lib.set_option("exact_guard", 1)
        random_leg = {"index": f"{NS} x {D} f16 unit-normalised N(0,1) rows (seed 1234): the index rounds 1-5 quoted `value` on",
                      "images_per_s": round(B / rdt, 1), "ms_per_step": round(rdt * 1e3, 3),
                      "vs_headline_step": round(rdt * 1e3 / ms_per_step, 3),
                      "exactness_per_step": {k_: round((gs1[k_] - gs0[k_]) / args.steps, 2) for k_ in
                                             ("queries", "widened", "rounds", "pages", "exhaustive", "swept_rows")},
                      "query_stage_ms": {"headline_index_first_pass_plus_widen": round(q_full, 4),
                                         "headline_index_first_pass_only_guard_off": round(q_first, 4),
                                         "random_index": round(q_rand, 4), "widen_ratio": round(q_full / q_first, 3)},
                      "note": "NOT the headline value: no query of this index needs the widen pass"}

    # ---------------------------------------------------------------- the same step with the fp8 GEMMs (N = 1 only, opt-in path)
    b32_fp8 = None
    if rank == 0 and world == 1 and not args.no_text:
        enc.encode_image(pixels, out=emb)
        ref16 = emb.clone()
        enc.set_precision("fp8")
        try:
            for _ in range(3):
                step()
            drain()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            drain()
            fence()
            fdt = (time.perf_counter() - t0) / args.steps
            # its own kernel table and roofline: the same K steps again with every launch bracketed by HIP events (as the bf16
            # table above: an instrumented replay, outside the timed loop)
            fprof = []
            if not args.no_kernel_events:
                _lib.prof_filter(None, 1)
                _lib.prof_reset()
                _lib.prof_enable(True)
                for _ in range(args.steps):
                    step()
                drain()
                fence()
                _lib.prof_enable(False)
                fprof = _lib.prof_read()
            enc.encode_image(pixels, out=emb)     # (the steps rotate through the batches: the comparison re-encodes batch 0)
            cos8 = float((1.0 - (emb * ref16).sum(dim=1)).max().item())
        finally:
            enc.set_precision("bf16")
        MFMA_FP8_PEAK_TFLOPS = 5000.0     # dense, MI355X_MICROARCH.md (AMD's 10 PF headline includes 2:1 sparsity)
        fkern, froof = [], None
        if fprof:
            ftot = sum(p_["ms"] for p_ in fprof)
            for p_ in sorted(fprof, key=lambda p_: -p_["ms"]):
                fkern.append({"kernel": p_["kernel"], "launches": p_["launches"], "avg_us": round(1e3 * p_["ms"] / p_["launches"], 2),
                              "share": round(p_["ms"] / ftot, 4),
                              "tflops": round(p_["flops"] / p_["ms"] / 1e9, 1) if p_["flops"] else None,
                              "gbs": round(p_["bytes"] / p_["ms"] / 1e6, 1)})
            f8 = [p_ for p_ in fprof if p_["kernel"].startswith("gemm_fp8")]
            if f8:
                top8 = max(f8, key=lambda p_: p_["ms"])
                t_s8 = top8["ms"] / top8["launches"] * 1e-3
                tf8 = top8["flops"] / top8["launches"] / t_s8 / 1e12
                allf = sum(p_["flops"] for p_ in f8) / sum(p_["ms"] for p_ in f8) / 1e9
                froof = {"bound": "mfma", "kernel": top8["kernel"], "achieved": round(tf8, 1), "peak": MFMA_FP8_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(tf8 / MFMA_FP8_PEAK_TFLOPS, 4), "traffic": None,
                         "avg_launch_us": round(t_s8 * 1e6, 2), "flops_per_launch": top8["flops"] / top8["launches"],
                         "all_fp8_gemms_tflops": round(allf, 1), "all_fp8_gemms_frac": round(allf / MFMA_FP8_PEAK_TFLOPS, 4),
                         "source": "instrumented replay of this leg's K steps (HIP events on the kernel's own stream, every launch); "
                                   "algorithmic flops 2 M N K on the valid rows"}
        b32_fp8 = {"images_per_s": round(B / fdt, 1), "ms_per_step": round(fdt * 1e3, 3),
                   "vs_headline_step": round(fdt * 1e3 / ms_per_step, 3),
                   "max_1_minus_cos_vs_bf16_path": cos8,
                   "roofline": froof, "kernels": fkern,
                   "note": "set_precision('fp8'), the SAME step as `value` (own index, widen pass): QKV, FC1 (-> MXFP8), FC2 and the "
                           "out-projection (A = the attention's MXFP8 output) of 11 layers + QKV of the 12th on the persistent block-scaled "
                           "GEMM (gemm256p8_kernel, v_mfma_scale_f32_16x16x128_f8f6f4: MXFP8 activations, e4m3 weights with per-channel "
                           "scales; round 6: K = 768 = three K-tile pairs per tile); patch embedding, the pruned last layer's three GEMMs, "
                           "LayerNorm statistics and the head stay bf16 / f32. Opt-in: e4m3's 3 mantissa bits put this tower ~6e-4 from the "
                           "bf16 path (inside 1e-3, asserted in tests/test_headline_gpu.py; the text tower is outside and stays bf16), "
                           "DESIGN.md 3b. NOT the headline value"}

    # ---------------------------------------------------------------- the reference's own checkpoint geometry (N = 1 only)
    l14 = None
    l14_check = None
    if rank == 0 and world == 1 and not args.no_text:
        from mmiss_amd.encoder import LONGCLIP_L14

        del raw, iemb
        BL, BT = 128, 64
        enc_l = ClipEncoder(LONGCLIP_L14, device=local_rank, max_batch_image=BL, max_batch_text=BT)
        W_l14 = random_state_dict(LONGCLIP_L14, seed=0)
        enc_l.load_state_dict(W_l14)
        xl = torch.randn(BL, 3, 224, 224, device=dev)
        ol = torch.empty(BL, 768, device=dev)

        l14_passes = []

        def timed(fn, warm, iters):
            """Mean time per call over `iters` calls, the better of two passes (one pass of eight 17 ms calls is short enough
            for a single host-side stall to cost 15 %: seen once in this round's runs)."""
            for _ in range(warm):
                fn()
            best = None
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / iters
                l14_passes.append(round(dt * 1e3, 3))
                best = dt if best is None or dt < best else best
            return best

        dt_img = timed(lambda: enc_l.encode_image(xl, out=ol), 3, 8)
        idl = np.full((BT, 248), 49407, dtype=np.int32)
        idl[:, 0] = 49406
        idl[:, 1:247] = np.random.Generator(np.random.Philox(9)).integers(0, 49406, size=(BT, 246))
        idl_d = torch.from_numpy(idl).to(dev)
        otl = torch.empty(BT, 768, device=dev)
        dt_txt = timed(lambda: enc_l.encode_text(idl_d, out=otl), 3, 8)
        x1, o1 = xl[:1].contiguous(), torch.empty(1, 768, device=dev)
        dt_one = timed(lambda: enc_l.encode_image(x1, out=o1), 3, 20)
        ref_img, ref_txt = ol.clone(), otl.clone()
        # BASELINE configs[4]: the same towers with the QKV / FC1 / FC2 projections on the block-scaled fp8 matrix cores
        enc_l.set_precision("fp8")
        dt_img8 = timed(lambda: enc_l.encode_image(xl, out=ol), 3, 8)
        dt_txt8 = timed(lambda: enc_l.encode_text(idl_d, out=otl), 3, 8)
        cos_img8 = float((1 - (ol * ref_img).sum(1)).max())
        cos_txt8 = float((1 - (otl * ref_txt).sum(1)).max())
        # two images and two prompts of these batches go to the CPU leg below, which holds them against the fp32 restatement
        l14_check = {"W": W_l14, "px": xl[:2].cpu().numpy(), "ids": idl[:2].copy(),
                     "img_bf16": ref_img[:2].cpu().numpy(), "img_fp8": ol[:2].cpu().numpy(),
                     "txt_bf16": ref_txt[:2].cpu().numpy(), "txt_fp8_setting": otl[:2].cpu().numpy()}
        _lib.prof_filter(None, 1)
        _lib.prof_reset()
        _lib.prof_enable(True)
        enc_l.encode_image(xl, out=ol)
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        k8 = {k["kernel"]: [k["launches"], round(1e3 * k["ms"] / k["launches"], 1),
                            round(k["flops"] / k["ms"] / 1e9, 1) if k["flops"] else None] for k in _lib.prof_read()}
        l14 = {"model": "LongCLIP-L/14 geometry (reference CLIP_MODEL_ID, backend/app/utils.py:16-17): ViT-L/14 vision, "
                        "248-token text tower, proj 768; random-init bf16",
               "images_per_s_bs128": round(BL / dt_img, 1), "image_tflops": round(BL * 162.03e9 / dt_img / 1e12, 1),
               "texts_per_s_bs64_T248": round(BT / dt_txt, 1), "text_tflops": round(BT * 44.39e9 / dt_txt / 1e12, 1),
               "single_image_encode_ms": round(dt_one * 1e3, 3),
               "timing": {"best_of": 2, "ms_per_call_every_pass": l14_passes,
                          "order": "image bf16, text bf16, one image, image fp8, text fp8-setting (two passes each)"},
               "fp8": {"images_per_s_bs128": round(BL / dt_img8, 1), "image_tflops": round(BL * 162.03e9 / dt_img8 / 1e12, 1),
                       "texts_per_s_bs64_T248": round(BT / dt_txt8, 1), "text_tflops": round(BT * 44.39e9 / dt_txt8 / 1e12, 1),
                       "max_1_minus_cos_vs_bf16_path": {"image": cos_img8, "text": cos_txt8},
                       "max_1_minus_cos_vs_fp32_oracle": None,   # filled in by the CPU leg (2 images / 2 prompts of these batches)
                       "towers": "set_precision('fp8') = vision tower on the fp8 GEMMs, text tower on the bf16 kernels (its fp8 "
                                 "form is outside the 1e-3 tolerance: include/mmiss.h)",
                       "kernels_image_bs128": k8, "kernels_columns": ["launches", "avg_us", "tflops"],
                       "note": "QKV / FC1 / FC2 on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, MX block scales on activations, "
                               "per-channel scales on weights); out-proj (bf16 operands), attention, LayerNorm statistics and head "
                               "unchanged; residual stream bf16 as in the bf16 setting at this batch size; fp8 MFMA peak 5 PF dense"}}
        del enc_l, xl, ol

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(W, D, args.index_rows, K_TOP)
        if l14 is not None:   # the L/14 legs against the fp32 restatement, in the one leg that may call it
            chk = l14_oracle_check(l14_check)
            l14["max_1_minus_cos_vs_fp32_oracle"] = chk["bf16"]
            l14["fp8"]["max_1_minus_cos_vs_fp32_oracle"] = chk["fp8"]

    if rank == 0:
        out = {
            "metric": "images/s CLIP-B/32 encode @ bs256; Mvec/s cosine top-10 over 10M x 512",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "ViT-B/32 image encode bs=256 per GPU + cosine top-10 of every embedding vs the "
                                   f"{args.index_rows}x512 f16 flat index of THESE images' embeddings per GPU (BASELINE configs[1] as "
                                   "written, SURVEY 8(d) config 2: every query is a row of the index and must rank itself first; the "
                                   "exactness guard widens every query)",
                       "global_batch": world * B, "image": "3x224x224 f32 resident in HBM", "weights": "random-init seed 0",
                       "index": f"{args.index_rows} seeded N(0,1) images per rank (seed 4321 + rank) encoded and added before the timed "
                                f"region in {ingest_index_s:.2f} s incl. image generation; labels rank * {args.index_rows} + i",
                       "images": f"the first {NROT} batches of {B} of those images stay resident in HBM, step i encodes batch i % {NROT} "
                                 "(the full ingest-then-query-back flow under the oracle is "
                                 "tests/test_headline_gpu.py::test_config1_100k_distinct_images_ingested_then_queried_back)",
                       "every_query_of_the_last_step_finds_itself_first": self_first, "max_self_distance_last_step": max_self_dist,
                       "index_dtype": "f16", "k": K_TOP, "parallelism": f"dp{world}",
                       "flops_per_image": 8.818e9, "flops_per_image_executed": 8.298e9,
                       "pruning": "last layer: out-proj + MLP on the pooled (CLS) rows only",
                       "layernorm": "folded into the QKV / FC1 GEMMs (automatic from 6000 rows per call): the residual GEMM's "
                                    "epilogue also writes the row statistics of the new residual rows",
                       "residual_stream": "bf16 in that mode (read-modify-write of the bf16 rows, f32 accumulate + add, one "
                                          "rounding per add; 1 - cos vs the fp32 oracle 5e-5, tests/test_headline_gpu.py); "
                                          "set_precision('bf16-f32resid') keeps it f32 (5e-6, 4-5 % slower)",
                       "step_pipelining": "one deep: step i+1's encode is queued before the host waits for step i's query results "
                                          "(FlatIndex.query_begin / result()); all K result sets are complete inside the timed region",
                       "query_stream": ("own HIP stream: the query stage of step i (first pass, widen pass) runs beside the encode of step i+1, "
                                        "ordered behind its own encode by an event" if args.query_stream == "own" else "the encoder's stream"),
                       "ms_per_step_unpipelined": round(unpipelined * 1e3 / args.steps, 3),
                       "last_pipelined_result_equals_synchronous_query": pipelined_equals_sync,
                       "kernel_events_in_timed_region": "dominant kernel, every 7th launch",
                       "ms_per_step_with_kernel_events": None if events_ms_per_step is None else round(events_ms_per_step, 3)},
            "encode_tflops": round(value * 8.298e9 / 1e12 / world, 1),
            "exactness": dict({k_: gs_timed[k_] - gs_before[k_] for k_ in ("queries", "widened", "rounds", "pages", "exhaustive", "swept_rows")},
                              note="the timed region's queries on the step's index / of them not provable from the first pass and "
                                   "widened by a threshold pass (mmiss_index_guard_stats, after - before)"),
            "roofline": roofline, "kernels": kernels, "retrieval": retrieval, "text": text, "single_request": latency, "ingest": ingest, "pcie_inclusive": pcie,
            "random_index": random_leg, "query_on_own_stream": own_stream_leg, "two_batches_in_flight": lanes, "fp8_gemms": b32_fp8, "l14": l14,
            "cpu_baseline": cpu if world == 1 else {"see": "the N = 1 line of the same commit: the CPU baseline is timed on rank 0 "
                                                           "at N = 1 only (it needs the host cores the other ranks' launch threads use)"},
            "distributed": distributed,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def bench_retrieval(args, torch, dist, np, dev, rank, world, local_rank, FlatIndex, merge_topk, _lib, all_gather,
                    all_reduce_max, exchange_topk):
    """cosine top-10 over N x 512 f16 rows sharded over the ranks; Q=1 (HBM-bound) and Q=1024."""
    N, D, K_TOP = args.retrieval_rows, 512, 10
    per = N // world
    idx = FlatIndex(D, "f16", device=local_rank, capacity=per)
    gen = torch.Generator(device=dev).manual_seed(4 + rank)
    chunk = 1_000_000
    for r0 in range(0, per, chunk):
        n = min(chunk, per - r0)
        idx.add(torch.randn(n, D, device=dev, generator=gen), np.arange(rank * per + r0, rank * per + r0 + n, dtype=np.int64))
    res = {"rows": N, "rows_per_gpu": per, "dim": D, "dtype": "f16", "k": K_TOP}
    q_all = torch.randn(1024, D, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    first = {}   # query 0's (labels, distance bits) from every leg: scan (Q = 1, 16) and score GEMM (Q = 1024) must agree
    for Q, iters in ((1, 20), (16, 10), (1024, 10)):   # (a steady-state loop behind warm-up calls; round 4 timed Q = 1024 three times)
        q = q_all[:Q].contiguous()

        def run():
            lab, dst, _ = idx.query(q, K_TOP)
            if world > 1:
                lab_all, dst_all = exchange_topk(lab, dst, world, all_gather=all_gather)
                return merge_topk(dst_all, lab_all)
            return lab, dst

        r0_ = run()
        first[Q] = (r0_[0][0].cpu().numpy().copy(), r0_[1][0].cpu().numpy().view(np.uint32).copy())
        run()
        run()   # three warm-up calls in all
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        _lib.prof_reset()
        _lib.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = (time.perf_counter() - t0) / iters
        _lib.prof_enable(False)
        prof = {p["kernel"]: p for p in _lib.prof_read()}
        if world > 1:
            dt = all_reduce_max(dt)
        scan = prof.get("scan_topk_f16")
        entry = {"ms_per_batch": round(dt * 1e3, 3), "mvec_per_s": round(N / dt / 1e6, 1),
                 "gpairs_per_s": round(Q * N / dt / 1e9, 2),
                 "kernel_ms": {k: round(p["ms"] / iters, 4) for k, p in prof.items()}}
        sg = prof.get("score_gemm_f16")
        if sg:
            sg_ms = sg["ms"] / sg["launches"]
            tfl = sg["flops"] / sg["ms"] / 1e9   # the filtered pass covers 15/16 of the rows; its own flops
            entry["score_gemm"] = {"avg_ms": round(sg_ms, 4), "tflops": round(tfl, 1), "mfma_frac": round(tfl / MFMA_BF16_PEAK_TFLOPS, 4)}
            # fabric bytes per launch of the strip score GEMM from the round's PMC passes over tools/retrieval_profile.py (the
            # same 10M x 512 f16 index and Q = 1024; counters cannot be read from inside this process), against the index bytes
            # one pass must read
            import glob
            tfiles = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r??_traffic_retrieval.json")))
            if world == 1 and N == 10_000_000 and Q == 1024 and tfiles:
                try:
                    tj = json.load(open(tfiles[-1])).get("score_gemm_f16_strip") or {}
                    nb = tj.get("batches_if_over_retrieval_profile")   # written by tools/traffic_from_pmc.py from tools/retrieval_profile.py's PASSES
                    if tj.get("traffic_bytes_all_launches") and nb:
                        per_batch = tj["traffic_bytes_all_launches"] / float(nb)   # a batch = the sample + the filtered pass
                        entry["score_gemm"]["traffic_bytes_per_batch"] = int(per_batch)
                        entry["score_gemm"]["traffic_vs_index_bytes"] = round(per_batch / (N * D * 2.0), 3)
                        entry["score_gemm"]["traffic_source"] = "profiles/%s (separate rocprofv3 --pmc passes over tools/retrieval_profile.py, %d batches, not this run)" % (os.path.basename(tfiles[-1]), nb)
                except Exception:
                    pass
        if scan:
            scan_ms = scan["ms"] / scan["launches"]
            gbs = per * D * 2 / scan_ms / 1e6
            tfl = 2.0 * Q * per * D / scan_ms / 1e9
            entry["scan_kernel"] = {"avg_ms": round(scan_ms, 4), "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                                    "tflops": round(tfl, 1), "mfma_frac": round(tfl / MFMA_BF16_PEAK_TFLOPS, 4)}
            other = sum(p["ms"] / iters for k, p in prof.items() if k != "scan_topk_f16")
            entry["other_kernels_ms"] = round(other, 4)
        res[f"Q{Q}"] = entry
    res["headline_mvec_per_s"] = res["Q1"]["mvec_per_s"]
    res["exactness"] = idx.guard_stats()
    same = all(np.array_equal(first[1][0], first[Q][0]) and np.array_equal(first[1][1], first[Q][1]) for Q in (16, 1024))
    res["check"] = {"first_query_ids_and_distance_bits_identical_across_Q1_Q16_Q1024": bool(same),
                    "note": "outside the timed loops: the three legs share their first query; Q = 1 / 16 take the streaming scan, "
                            "Q = 1024 the score GEMM + threshold-filtered selection; the same rows at 10M x 512 and 6.25M x 768 "
                            "against the C oracle: tests/test_headline_gpu.py::test_full_size_index_against_the_oracle"}
    idx.close()
    # the widen pass at full size: a CLUSTERED 10M-row index (one common direction + 10 % noise: pairwise
    # cosine 0.99, like the embeddings of one encoder) queried with 1024 of its own rows — the exactness guard cannot prove
    # any query from the first pass, every one takes the threshold pass
    if world == 1 and N >= 1_000_000:
        idc = FlatIndex(D, "f16", device=local_rank, capacity=N)
        genc = torch.Generator(device=dev).manual_seed(77)
        centre = torch.randn(1, D, device=dev, generator=genc)
        centre = centre / centre.norm()
        qc = None
        for r0 in range(0, N, chunk):
            n = min(chunk, N - r0)
            rows_c = centre + 0.1 * torch.randn(n, D, device=dev, generator=genc) / D ** 0.5
            if qc is None:
                qc = rows_c[:1024].clone()
            idc.add(rows_c, np.arange(r0, r0 + n, dtype=np.int64))
        del rows_c
        lab_c, dst_c, _ = idc.query(qc, K_TOP)
        torch.cuda.synchronize()
        g0 = idc.guard_stats()
        t0 = time.perf_counter()
        for _ in range(3):
            idc.query(qc, K_TOP)
        torch.cuda.synchronize()
        dtc = (time.perf_counter() - t0) / 3
        g1 = idc.guard_stats()
        _lib.set_option("exact_guard", 0)
        try:
            idc.query(qc, K_TOP)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                idc.query(qc, K_TOP)
            torch.cuda.synchronize()
            dtf = (time.perf_counter() - t0) / 3
        finally:
            _lib.set_option("exact_guard", 1)
        res["clustered_Q1024"] = {
            "rows": N, "ms_per_batch": round(dtc * 1e3, 3), "first_pass_only_ms": round(dtf * 1e3, 3),
            "ratio_to_first_pass": round(dtc / dtf, 3),
            "per_batch": {k_: (g1[k_] - g0[k_]) // 3 for k_ in ("queries", "widened", "rounds", "pages", "exhaustive", "swept_rows")},
            "self_match_first": bool((lab_c[:, 0].cpu() == torch.arange(1024)).all()),
            "note": "rows = unit(centre + 0.1 noise): every query fails the guard's proof and is widened by ONE threshold pass "
                    "(score GEMM over the whole index again, rows with approx >= c_k - eps appended, exact re-rank); "
                    "first_pass_only_ms = the same batch with the guard switched off (results unproven)"}
        idc.close()
    # BASELINE configs[4] index shape: one of the 8 row shards of the 50M x 768 f16 index (6.25M rows, 9.6 GB), N = 1 only
    if world == 1 and N >= 10_000_000:
        N8, D8 = 6_250_000, 768
        idx8 = FlatIndex(D8, "f16", device=local_rank, capacity=N8)
        for r0 in range(0, N8, 1_250_000):
            idx8.add(torch.randn(1_250_000, D8, device=dev, generator=gen), np.arange(r0, r0 + 1_250_000, dtype=np.int64))
        shard = {"rows": N8, "dim": D8, "dtype": "f16", "note": "one GPU's shard of BASELINE configs[4] (50M x 768 over 8 GPUs)"}
        q8_all = torch.randn(128, D8, device=dev, generator=torch.Generator(device=dev).manual_seed(6))
        first8 = {}
        for Q, iters in ((1, 10), (128, 5)):
            q = q8_all[:Q].contiguous()
            l8_, d8_, _ = idx8.query(q, K_TOP)
            first8[Q] = (l8_[0].cpu().numpy().copy(), d8_[0].cpu().numpy().view(np.uint32).copy())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                idx8.query(q, K_TOP)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters
            shard[f"Q{Q}"] = {"ms_per_batch": round(dt * 1e3, 3), "mvec_per_s": round(N8 / dt / 1e6, 1),
                              "hbm_gbs_equiv": round(N8 * D8 * 2 / dt / 1e9, 1)}
        shard["first_query_ids_and_distance_bits_identical_across_Q1_Q128"] = bool(
            np.array_equal(first8[1][0], first8[128][0]) and np.array_equal(first8[1][1], first8[128][1]))
        idx8.close()
        res["shard_50Mx768"] = shard
        # the same 10M x 512 rows stored as fp8 (MMISS_F8: e4m3 codes of 128 x, half the bytes of f16 per row), N = 1 only,
        # informational: exact with respect to its stored rows, which are 2^-4-coarse (include/mmiss.h)
        idxf = FlatIndex(D, "f8", device=local_rank, capacity=N)
        gen8 = torch.Generator(device=dev).manual_seed(4 + rank)
        for r0 in range(0, N, chunk):
            n = min(chunk, N - r0)
            idxf.add(torch.randn(n, D, device=dev, generator=gen8), np.arange(r0, r0 + n, dtype=np.int64))
        f8 = {"rows": N, "dim": D, "dtype": "f8 (e4m3 x 2^7 + one f32 inverse norm per row: cosine distances, round 5)", "k": K_TOP}
        for Q, iters in ((1, 20), (16, 10), (1024, 10)):
            q = q_all[:Q].contiguous()
            for _ in range(3):
                idxf.query(q, K_TOP)
            torch.cuda.synchronize()
            _lib.prof_reset()
            _lib.prof_enable(True)
            t0 = time.perf_counter()
            for _ in range(iters):
                idxf.query(q, K_TOP)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters
            _lib.prof_enable(False)
            pr8 = {p["kernel"]: p for p in _lib.prof_read()}
            sc = pr8.get("scan_topk_f8")
            f8[f"Q{Q}"] = {"ms_per_batch": round(dt * 1e3, 3), "mvec_per_s": round(N / dt / 1e6, 1)}
            sg8 = pr8.get("score_gemm_f8")
            if sg8:   # Q = 1024: the strip score GEMM on fp8 rows (codes widened to f16 in the operand load)
                tfl8 = sg8["flops"] / sg8["ms"] / 1e9
                f8[f"Q{Q}"]["score_gemm"] = {"avg_ms": round(sg8["ms"] / sg8["launches"], 4), "tflops": round(tfl8, 1),
                                             "mfma_frac": round(tfl8 / MFMA_BF16_PEAK_TFLOPS, 4)}
                f8[f"Q{Q}"]["kernel_ms"] = {k_: round(p_["ms"] / iters, 4) for k_, p_ in pr8.items()}
            if sc:
                sms = sc["ms"] / sc["launches"]
                f8[f"Q{Q}"]["scan_kernel"] = {"avg_ms": round(sms, 4), "hbm_gbs": round(N * (D + 4) / sms / 1e6, 1),
                                              "hbm_frac": round(N * (D + 4) / sms / 1e6 / HBM_PEAK_GBS, 4)}
        # what the index returns is a cosine distance: a row queried with the vector it represents comes back first at ~0
        own = torch.arange(0, N, N // 16, device=dev)[:16]
        lo_, do_, _ = idxf.query(torch.from_numpy(idxf.get(own.cpu().numpy())).to(dev), 1)
        f8["self_query"] = {"own_row_first": bool((lo_[:, 0] == own).all()), "max_abs_distance": float(do_.abs().max())}
        f8["exactness"] = idxf.guard_stats()
        idxf.close()
        res["f8_rows"] = f8
    return res


def l14_oracle_check(chk):
    """Part of the CPU leg: two images and two prompts of the L/14 legs through the PyTorch-CPU fp32 restatement
    (oracle/clip_oracle_torch.py) -> max 1 - cos of the GPU embeddings (bf16 setting, fp8 setting) against it."""
    import numpy as np
    import torch

    from oracle import clip_oracle as co
    from oracle import clip_oracle_torch as ct

    Wt = ct.to_torch(chk["W"])
    old = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref_i = ct.embed_images(chk["px"], Wt, co.LONGCLIP_L14).numpy()
    ref_t = ct.embed_texts(chk["ids"], Wt, co.LONGCLIP_L14).numpy()
    torch.set_num_threads(old)

    def gap(a, b):
        return float((1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))).max())

    return {"bf16": {"image": gap(chk["img_bf16"], ref_i), "text": gap(chk["txt_bf16"], ref_t)},
            "fp8": {"image": gap(chk["img_fp8"], ref_i), "text": gap(chk["txt_fp8_setting"], ref_t)}}


def cpu_baseline(W, D, index_rows, k):
    """The reference's CPU path restated in the framework it runs on — PyTorch-CPU fp32 (oracle/clip_oracle_torch.py,
    pinned to the numpy oracle and through it to transformers.CLIPModel) + exact brute-force cosine top-k (torch CPU)
    — on a bounded sample of the same workload, as SURVEY.md 8(d) asks: batch 1 (what the reference actually does,
    backend/app/utils.py:76-77) and batch 32, one thread and the box's CPU share. `value` = the batched all-core figure
    (the strongest CPU number), the others are listed beside it."""
    import platform

    import numpy as np
    import torch

    from oracle import clip_oracle as co
    from oracle import clip_oracle_torch as ct

    share = min(16, os.cpu_count() or 1)  # a one-GPU box gives this job a 16-CPU share
    Wt = ct.to_torch(W)
    rng = np.random.Generator(np.random.Philox(99))
    px = rng.standard_normal((32, 3, 224, 224), dtype=np.float32)
    old_threads = torch.get_num_threads()
    enc = {}
    emb = None

    passes = {}

    def rate(name, bs, n_img, threads, n_pass):
        """best of n_pass passes over the same n_img images, every pass listed (one 32-image pass swung 72 <-> 149 images/s
        between boxes of the same CPU model in round 3: a single pass measures the box's other tenants as much as the CPU)"""
        torch.set_num_threads(threads)
        ct.embed_images(px[:bs], Wt, co.VIT_B32)  # warm-up
        best, out = 0.0, None
        passes[name] = []
        for _ in range(n_pass):
            t0 = time.perf_counter()
            out = [ct.embed_images(px[i:i + bs], Wt, co.VIT_B32) for i in range(0, n_img, bs)]
            r = n_img / (time.perf_counter() - t0)
            passes[name].append(round(r, 2))
            best = max(best, r)
        enc[name] = best
        return torch.cat(out).numpy()

    rate("bs1_1thread", 1, 3, 1, 2)
    rate(f"bs1_{share}threads", 1, 8, share, 3)
    rate("bs32_1thread", 32, 32, 1, 1)
    emb = rate(f"bs32_{share}threads", 32, 32, share, 3)
    torch.set_num_threads(old_threads)
    # retrieval on the CPU: exact brute force (what chromadb does below 100 rows and approximates above), unit rows in fp32,
    # one BLAS GEMM + top-k for the 32 queries on all threads of the share
    corpus = torch.from_numpy(rng.standard_normal((index_rows, D), dtype=np.float32))
    corpus = corpus / corpus.norm(dim=1, keepdim=True)
    qt = torch.from_numpy(emb)
    torch.set_num_threads(share)
    torch.topk(qt @ corpus.T, k, dim=1)  # warm-up
    t_q = None
    for _ in range(3):   # best of 3 passes of 3
        t0 = time.perf_counter()
        for _ in range(3):
            d_, i_ = torch.topk(1.0 - qt @ corpus.T, k, dim=1, largest=False)
        t_ = (time.perf_counter() - t0) / 3 / qt.shape[0]
        t_q = t_ if t_q is None or t_ < t_q else t_q
    torch.set_num_threads(old_threads)
    best = enc[f"bs32_{share}threads"]
    cpu_model = platform.processor() or ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), cpu_model)
    except Exception:
        pass
    return {"value": round(1.0 / (1.0 / best + t_q), 2), "unit": "images/s", "cores": int(share), "kind": "port",
            "sample": f"PyTorch-CPU fp32 restatement of the reference's path (best of 3 passes, all listed): 32 images at bs=32 on {share} threads "
                      f"({1e3 / best:.1f} ms/img) + exact brute-force cosine top-{k} of those 32 embeddings vs {index_rows}x{D} "
                      f"fp32 unit rows, one GEMM + topk on {share} threads ({t_q * 1e3:.2f} ms/query); host: {cpu_model}, "
                      f"{os.cpu_count()} logical CPUs",
            "encode_only_images_per_s": {k_: round(v, 2) for k_, v in enc.items()},
            "encode_only_images_per_s_every_pass": passes,
            "reference_regime": "bs1 (backend/app/utils.py:76-77 encodes one image per request)",
            "query_ms": round(t_q * 1e3, 3)}


if __name__ == "__main__":
    main()

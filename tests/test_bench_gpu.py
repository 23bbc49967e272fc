"""bench.py's N > 1 control flow (barriers, rank-0-only sections, sharded retrieval + all-gather merge) rehearsed with
two ranks on the one GPU of the test box over gloo (MMISS_DIST_BACKEND=gloo); the driver runs the same file over RCCL
on 2/4/8 GPUs. Guards against a collective inside a rank-0-only branch, which deadlocks or kills the job."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_two_ranks_gloo_dry_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MMISS_DIST_BACKEND="gloo")
    # exactly the driver's command form: `python3 bench.py --gpus N ...` — bench.py starts the ranks itself
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--retrieval-rows", "200000"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["unit"] == "images/s" and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 512
    assert out["retrieval"]["rows"] == 200000 and out["retrieval"]["rows_per_gpu"] == 100000
    assert out["roofline"]["frac"] > 0 and "N = 1" in out["cpu_baseline"]["see"]  # the CPU baseline runs at N = 1 only
    d = out["distributed"]     # a SCALE line must explain itself: who ran, what was exchanged, where the step's time went
    assert d["world_size"] == 2 and len(d["device_count_seen_by_each_rank"]) == 2 and d["queries_per_rank_and_step"] == 512
    assert d["topk_exchange_bytes_per_rank"] == 512 * 10 * 12 and d["query_gather_bytes_per_rank"] == 256 * 512 * 4
    st = d["stage_ms_max_over_ranks"]
    assert set(st) == {"encode", "all_gather_queries", "local_query", "all_gather_topk", "merge"} and all(v > 0 for v in st.values())
    assert "N-fold" in d["weak_scaling_note"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("query_stream", ["same", "own"])
def test_bench_one_gpu_pipelined_steps_deliver_the_synchronous_results(query_stream):
    """N = 1, the driver's default form with a short run: the timed steps are pipelined one deep (query_begin / query_end);
    the line says what it did, times the unpipelined form beside it, and its last pipelined result set is the synchronous
    query's, id for id and bit for bit."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # (--query-stream own: the query stage on a second HIP stream beside the next encode, embeddings ping-ponging between two
    # buffers behind events — the same results, id for id and bit for bit)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--retrieval-rows", "0",
           "--no-cpu-baseline", "--no-text", "--query-stream", query_stream]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    cfg = out["config"]
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0
    assert "query_begin" in cfg["step_pipelining"] and cfg["ms_per_step_unpipelined"] > 0
    assert ("own HIP stream" in cfg["query_stream"]) == (query_stream == "own")
    assert cfg["last_pipelined_result_equals_synchronous_query"] is True
    # round 6: `value` is timed on configs[1] AS WRITTEN — the index holds the embeddings of the step's own images, every query is
    # a row of it (and must come back first), and the exactness guard widens them (random-weight embeddings: pairwise cosine 0.99)
    assert "THESE images" in cfg["workload"] and cfg["every_query_of_the_last_step_finds_itself_first"] is True
    assert cfg["max_self_distance_last_step"] < 1e-4   # (f16 rows are stored as rounded, not re-normalised: 3.6e-5 here)
    assert out["exactness"]["queries"] == 3 * 256 and out["exactness"]["widened"] > 0
    assert out["roofline"]["kernel"].startswith("gemm_bf16_") and 0 < out["roofline"]["frac"] < 1

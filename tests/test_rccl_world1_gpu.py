"""RCCL itself on the test box's one GPU: a one-rank "nccl" process group and the collectives of the N > 1 path on DEVICE
tensors (tools/rccl_world1_rehearsal.py) — the gloo rehearsals stage through host memory and never load librccl. One rank
exchanging with itself measures nothing; it proves the library initialises here and the device-tensor path runs, and that
ShardedIndex over that group returns the unsharded index's ids and distance bits."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_one_rank_rccl_group_runs_the_sharded_query_and_the_bench_collectives():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    import socket

    with socket.socket() as sk:   # a free rendezvous port (the tool's default is fixed)
        sk.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(sk.getsockname()[1])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1_rehearsal.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "RCCL world-1 rehearsal ok: backend nccl" in r.stdout, r.stdout[-2000:]

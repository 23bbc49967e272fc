"""The N > 1 path on CPU: two processes over gloo, each holding one shard (oracle-backed stand-in for the GPU
index), one all-gather of per-shard top-k, merge -> identical to the unsharded result on every rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import mmiss_amd  # noqa: F401
    from mmiss_amd.sharded import ShardedIndex
    from fakes import OracleIndex

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, D, k = 1000, 128, 10
    rng = np.random.Generator(np.random.Philox(7))
    corpus = rng.standard_normal((N, D), dtype=np.float32)
    corpus[500] = corpus[3]  # an exact duplicate living in the other shard: the tie must break by label
    labels = np.arange(N, dtype=np.int64)
    sh = ShardedIndex(OracleIndex(D, "f16"))
    kept = sh.add_global(corpus, labels, N)
    assert kept == N // world and sh.count() == N
    q = rng.standard_normal((5, D), dtype=np.float32)
    q[0] = corpus[3]
    res_b = sh.query(q if rank == 0 else np.zeros_like(q), k, src=0)            # broadcast from rank 0
    mine = q[rank * 2: rank * 2 + 2]
    res_g = sh.query(mine, k, src=None)                                          # all-gather of per-rank blocks
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), bl=res_b[0], bd=res_b[1], bc=res_b[2], gl=res_g[0], gd=res_g[1])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_query_equals_unsharded(tmp_path):
    import torch.multiprocessing as mp
    from oracle import retrieval_oracle as ro

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    N, D, k = 1000, 128, 10
    rng = np.random.Generator(np.random.Philox(7))
    corpus = rng.standard_normal((N, D), dtype=np.float32)
    corpus[500] = corpus[3]
    q = rng.standard_normal((5, D), dtype=np.float32)
    q[0] = corpus[3]
    full = ro.query(q, ro.normalize_rows(corpus, "f16"), np.arange(N, dtype=np.int64), k)
    assert list(full[0][0][:2]) == [3, 500]
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    for r in (r0, r1):
        np.testing.assert_array_equal(r["bl"], full[0])
        np.testing.assert_array_equal(r["bd"].view(np.uint32), full[1].view(np.uint32))
        np.testing.assert_array_equal(r["bc"], full[2])
        np.testing.assert_array_equal(r["gl"], full[0][:4])
        np.testing.assert_array_equal(r["gd"].view(np.uint32), full[1][:4].view(np.uint32))


def _worker_ragged(rank, world, port, out_dir):
    """N = 1003 rows over `world` ranks (not divisible: the last shard is short), k = 1000 (more than any shard holds: every
    shard answers with fewer than k rows and -1 padding), and a SPARSE placement (only labels < 300 exist although the
    layout is sized for 1003: the upper ranks hold EMPTY shards)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import mmiss_amd  # noqa: F401
    from mmiss_amd.sharded import ShardedIndex
    from fakes import OracleIndex

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, D = 1003, 64
    rng = np.random.Generator(np.random.Philox(11))
    corpus = rng.standard_normal((N, D), dtype=np.float32)
    corpus[900] = corpus[2]
    labels = np.arange(N, dtype=np.int64) * 3 + 1          # global labels need not be 0..N-1
    q = rng.standard_normal((world, D), dtype=np.float32)  # one query row per rank for the all-gather form
    q[0] = corpus[2]
    out = {}
    full = ShardedIndex(OracleIndex(D, "f32"))
    kept = full.add_global(corpus, labels, 3 * N + 1)
    per = -(-(3 * N + 1) // world)
    assert kept == int(((labels // per).clip(max=world - 1) == rank).sum()) and full.count() == N
    for k in (10, 1000):
        l, d, c = full.query(q if rank == 0 else np.zeros_like(q), k, src=0)
        out[f"full_l{k}"], out[f"full_d{k}"], out[f"full_c{k}"] = l, d, c
    lg, dg, cg = full.query(q[rank:rank + 1], 10, src=None)
    out["gath_l"], out["gath_d"] = lg, dg
    sparse = ShardedIndex(OracleIndex(D, "f32"))
    few = labels < 300                                       # ranks whose label range starts above 300 stay empty
    kept_s = sparse.add_global(corpus[few], labels[few], 3 * N + 1)
    out["kept_sparse"] = np.array([kept_s])
    assert sparse.count() == int(few.sum())
    l, d, c = sparse.query(q if rank == 0 else np.zeros_like(q), 1000, src=0)
    out["sparse_l"], out["sparse_d"], out["sparse_c"] = l, d, c
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_ragged_and_empty_shards_k1000(tmp_path, world):
    import torch.multiprocessing as mp
    from oracle import retrieval_oracle as ro

    port = _free_port()
    mp.spawn(_worker_ragged, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    N, D = 1003, 64
    rng = np.random.Generator(np.random.Philox(11))
    corpus = rng.standard_normal((N, D), dtype=np.float32)
    corpus[900] = corpus[2]
    labels = np.arange(N, dtype=np.int64) * 3 + 1
    q = rng.standard_normal((world, D), dtype=np.float32)
    q[0] = corpus[2]
    stored = ro.normalize_rows(corpus, "f32")
    few = labels < 300
    refs = {k: ro.query(q, stored, labels, k) for k in (10, 1000)}
    ref_sparse = ro.query(q, stored[few], labels[few], 1000)
    assert list(refs[10][0][0][:2]) == [7, 2701] and refs[1000][2][0] == 1000 and ref_sparse[2][0] == int(few.sum())
    kept = []
    for r in range(world):
        z = np.load(tmp_path / f"r{r}.npz")
        kept.append(int(z["kept_sparse"][0]))
        for k in (10, 1000):
            np.testing.assert_array_equal(z[f"full_l{k}"], refs[k][0])
            np.testing.assert_array_equal(z[f"full_d{k}"].view(np.uint32), refs[k][1].view(np.uint32))
            np.testing.assert_array_equal(z[f"full_c{k}"], refs[k][2])
        np.testing.assert_array_equal(z["gath_l"], refs[10][0])
        np.testing.assert_array_equal(z["gath_d"].view(np.uint32), refs[10][1].view(np.uint32))
        np.testing.assert_array_equal(z["sparse_l"], ref_sparse[0])
        np.testing.assert_array_equal(z["sparse_d"].view(np.uint32), ref_sparse[1].view(np.uint32))
        np.testing.assert_array_equal(z["sparse_c"], ref_sparse[2])
    assert sum(kept) == int(few.sum()) and kept.count(0) >= world // 2, kept   # most shards really were empty


def test_host_merge_orders_by_distance_then_label():
    import mmiss_amd  # noqa: F401
    from mmiss_amd.sharded import merge_topk_host

    d = np.array([[[0.1, 0.3, np.inf]], [[0.1, 0.2, 0.3]]], np.float32)   # [S=2, Q=1, k=3]
    l = np.array([[[7, 9, -1]], [[2, 5, 4]]], np.int64)
    ol, od, oc = merge_topk_host(d, l)
    assert list(ol[0]) == [2, 7, 5] and oc[0] == 3
    np.testing.assert_allclose(od[0], [0.1, 0.1, 0.2])

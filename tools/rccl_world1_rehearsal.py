"""RCCL on the one GPU of a test box: a process group of ONE rank with backend "nccl" (= RCCL), and through it the
collectives the N > 1 path issues — all_gather_into_tensor on device tensors (the packed top-k exchange and the query
gather), all_reduce(MAX) of the step time, barrier — plus ShardedIndex.query on that group against the unsharded index.
Not a scaling measurement (one rank exchanges with itself); it shows that the library initialises on the pool's boxes and
that the device-tensor code path (no host staging, unlike the gloo rehearsals) runs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import mmiss_amd  # noqa: F401,E402
from mmiss_amd.index import FlatIndex, merge_topk  # noqa: E402
from mmiss_amd.sharded import ShardedIndex, exchange_topk  # noqa: E402

N, D, Q, K = 50_000, 512, 64, 10
g = torch.Generator(device=dev).manual_seed(3)
rows = torch.randn(N, D, device=dev, generator=g)
q = torch.randn(Q, D, device=dev, generator=g)
labels = np.arange(N, dtype=np.int64)
plain = FlatIndex(D, "f16", device=0)
plain.add(rows, labels)


def h(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


lab0, dst0, _ = plain.query(q, K)

sh = ShardedIndex(FlatIndex(D, "f16", device=0))
sh.add_global(rows, labels, N)
lab1, dst1, cnt1 = sh.query(q, K, src=0)
assert np.array_equal(h(lab1), h(lab0)) and np.array_equal(h(dst1).view(np.uint32), h(dst0).view(np.uint32))

# bench.py's own calls on device tensors
lab_d, dst_d = torch.as_tensor(lab0, device=dev), torch.as_tensor(dst0, device=dev)
lab_all, dst_all = exchange_topk(lab_d, dst_d, 1)
ml, md = merge_topk(dst_all, lab_all)[:2]
assert np.array_equal(h(ml), h(lab0)) and np.array_equal(h(md).view(np.uint32), h(dst0).view(np.uint32))
out = torch.empty(1 * Q, D, device=dev)
dist.all_gather_into_tensor(out, q)
assert torch.equal(out, q)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert float(t.item()) == 1.25
print("RCCL world-1 rehearsal ok: backend", dist.get_backend(), "| nccl version", torch.cuda.nccl.version(), flush=True)
dist.destroy_process_group()

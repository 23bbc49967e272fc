"""ShardedIndex — the flat cosine index row-sharded over the ranks of a torch.distributed group
(one process per GPU; backend "nccl" = RCCL over xGMI on the MI355X node, "gloo" in the CPU tests).

No reference analogue: the reference is a single process. Design:
  * rank r owns the rows whose label l satisfies shard_of(l) == r (contiguous blocks of labels by default),
  * a query batch is made identical on every rank (broadcast from a source rank, or all-gather of per-rank
    query blocks), every rank scans only ITS shard -> local top-k with GLOBAL labels and canonical distances,
  * ONE exchange step: all-gather of [Q,k] (distance f32, label i64) = 12*Q*k bytes per rank — latency-bound,
    a single all-gather (direct peer writes over the 7 xGMI links), never a ring of S-1 steps,
  * every rank merges S*k -> k per query by (distance asc, label asc) (mmiss_merge_topk on the GPU).
Because the per-row distance is computed in a canonical order, it does not depend on the shard layout, and the
merged result is bit-identical to the unsharded index.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def merge_topk_host(dist: np.ndarray, labels: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Host-side S*k -> k merge for CPU tensors (gloo tests / tiny exchanges): same order as the device kernel,
    (distance asc, label asc), label -1 = empty slot."""
    S, Q, k = dist.shape
    d = np.transpose(dist, (1, 0, 2)).reshape(Q, S * k)
    l = np.transpose(labels, (1, 0, 2)).reshape(Q, S * k)
    d = np.where(l >= 0, d, np.inf).astype(np.float32)
    lk = np.where(l >= 0, l, np.iinfo(np.int64).max)
    order = np.lexsort((lk, d), axis=1)[:, :k]
    out_d = np.take_along_axis(d, order, axis=1)
    out_l = np.take_along_axis(l, order, axis=1)
    valid = np.take_along_axis(lk, order, axis=1) != np.iinfo(np.int64).max
    out_l = np.where(valid, out_l, -1)
    out_d = np.where(valid, out_d, np.inf).astype(np.float32)
    return out_l, out_d, valid.sum(axis=1).astype(np.int32)


def exchange_topk(lab, dst, world: int, group=None, all_gather=None):
    """X1, the ONE exchange step: every rank contributes its shard's [Q,k] (label i64, distance f32) and receives all
    S of them -> (labels [S,Q,k] i64, distances [S,Q,k] f32). Both arrays travel in ONE all-gather of 12*Q*k bytes per
    rank (labels then distances packed into one byte buffer): the exchange is latency-bound, so one
    collective instead of two halves its cost. `all_gather(out, inp)` may replace torch's (bench.py's gloo rehearsal
    stages device tensors through host memory)."""
    import torch
    import torch.distributed as dist

    Q, k = int(lab.shape[0]), int(lab.shape[1])
    nl, nd = Q * k * 8, Q * k * 4
    send = torch.empty(nl + nd, dtype=torch.uint8, device=lab.device)
    send[:nl].view(torch.int64).copy_(lab.reshape(-1))
    send[nl:].view(torch.float32).copy_(dst.reshape(-1))
    recv = torch.empty(world * (nl + nd), dtype=torch.uint8, device=lab.device)
    if all_gather is not None:
        all_gather(recv, send)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, nl + nd)
    lab_all = recv[:, :nl].contiguous().view(torch.int64).view(world, Q, k)
    dst_all = recv[:, nl:].contiguous().view(torch.float32).view(world, Q, k)
    return lab_all, dst_all


class ShardedIndex:
    """`local` is this rank's shard: any object with add(vecs, labels) / query(q, k) / count() — a FlatIndex on
    the GPU box; the CPU tests plug in an oracle-backed stand-in to exercise the collective path."""

    def __init__(self, local, group=None):
        import torch.distributed as dist

        self.local = local
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    # ------------------------------------------------------------------ placement
    def owner_of(self, labels: np.ndarray, total: int) -> np.ndarray:
        """Contiguous row blocks: rank r holds labels [r*ceil(total/S), (r+1)*ceil(total/S))."""
        per = -(-int(total) // self.world)
        return np.minimum(np.asarray(labels, dtype=np.int64) // per, self.world - 1)

    def add_global(self, vecs, labels: np.ndarray, total: int) -> int:
        """Every rank is handed the same (vecs, labels); each keeps only the rows it owns. Returns rows kept."""
        labels = np.asarray(labels, dtype=np.int64)
        mine = np.nonzero(self.owner_of(labels, total) == self.rank)[0]
        if mine.size:
            sel = vecs[mine] if not hasattr(vecs, "index_select") else vecs[mine.tolist()]
            self.local.add(sel, labels[mine])
        return int(mine.size)

    def count(self) -> int:
        import torch
        import torch.distributed as dist

        n = int(self.local.count())
        if self.world == 1:
            return n
        t = torch.tensor([n], dtype=torch.int64, device=self._device())
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def _device(self):
        import torch
        import torch.distributed as dist

        return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    # ------------------------------------------------------------------ query
    def query(self, queries, k: int, src: Optional[int] = 0):
        """queries: [Q,D] (numpy or torch). src = rank whose queries are used (broadcast); src=None means every
        rank passes its own [Qr,D] block and the blocks are all-gathered (equal Qr on all ranks).
        -> (labels [Q,k], distances [Q,k], counts [Q]) identical on every rank."""
        import torch
        import torch.distributed as dist

        if self.world == 1:
            return self.local.query(queries, k)
        dev = self._device()
        q = torch.as_tensor(np.asarray(queries) if not hasattr(queries, "device") else queries).to(dev, torch.float32).contiguous()
        if src is None:
            gathered = torch.empty((self.world * q.shape[0], q.shape[1]), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(gathered, q, group=self.group)
            q = gathered
        else:
            dist.broadcast(q, src=src, group=self.group)
        lab, dst, _ = self.local.query(q if dev.type == "cuda" else q.numpy(), k)
        lab_all, dst_all = exchange_topk(torch.as_tensor(lab).to(dev), torch.as_tensor(dst).to(dev), self.world, self.group)
        if dev.type == "cuda":
            from .index import merge_topk

            return merge_topk(dst_all, lab_all)
        return merge_topk_host(dst_all.numpy(), lab_all.numpy())

"""FlatIndex — one shard of the in-HBM flat cosine index (libmmiss `mmiss_index_*`).

Numeric half of the chromadb Collection the reference creates with {"hnsw:space": "cosine"}
(backend/app/utils.py:127-130): rows are L2-normalised when added; a query returns the k rows of
smallest cosine distance 1 - cos, ascending, ties by label ascending.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional, Tuple

import numpy as np

from . import _lib

F32, F16, F8 = _lib.MMISS_F32, _lib.MMISS_F16, _lib.MMISS_F8
_DTYPES = {"f32": F32, "float32": F32, "f16": F16, "float16": F16, "f8": F8, "fp8": F8, "e4m3": F8, F32: F32, F16: F16, F8: F8}


def _is_torch(x) -> bool:
    return hasattr(x, "data_ptr")


class FlatIndex:
    def __init__(self, dim: int, dtype="f32", device: int = 0, capacity: int = 0):
        self.dim = int(dim)
        self.dtype = _DTYPES[dtype]
        self.device = int(device)
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.mmiss_index_create(self.dim, self.dtype, self.device, int(capacity), C.byref(h)))
        self._h = h
        # re-entrant: a PendingQuery collected by the cyclic GC on a thread that is inside one of these calls must not
        # deadlock that thread on itself
        self._call_lock = threading.RLock()
        # generation of the handle's ONE open-query slot: bumped by every query_begin, result and abort. A PendingQuery only
        # ends / aborts the C-side query while its own generation is still the current one, so a stale handle that outlives
        # its query (kept alive by a traceback or a reference cycle) cannot abort a newer query of someone else.
        self._query_gen = 0

    # ------------------------------------------------------------------ helpers
    def _vecs(self, v):
        if _is_torch(v):
            import torch

            v = v.to(torch.float32).contiguous()
        else:
            v = np.ascontiguousarray(v, dtype=np.float32)
        if v.ndim == 1:
            v = v.reshape(1, -1)
        if v.ndim != 2 or v.shape[1] != self.dim:
            raise ValueError(f"expected [n,{self.dim}] vectors, got {tuple(v.shape)}")
        return v

    def _sync_stream(self, x):
        if _is_torch(x) and x.is_cuda:
            _lib.check(self._lib.mmiss_index_set_stream(self._h, _lib.current_stream_ptr(x.device), 0))
        else:
            _lib.check(self._lib.mmiss_index_set_stream(self._h, None, 1))

    # ------------------------------------------------------------------ mutation
    def add(self, vecs, labels) -> None:
        v = self._vecs(vecs)
        lab = np.ascontiguousarray(labels, dtype=np.int64).reshape(-1)
        if lab.shape[0] != v.shape[0]:
            raise ValueError("labels and vectors differ in length")
        with self._call_lock:  # stream hand-over + call are one unit per handle (threads: pipeline.BatchLanes)
            self._sync_stream(v)
            _lib.check(self._lib.mmiss_index_add(self._h, _lib.ptr(v), lab.ctypes.data, int(v.shape[0])))

    def update(self, labels, vecs) -> None:
        v = self._vecs(vecs)
        lab = np.ascontiguousarray(labels, dtype=np.int64).reshape(-1)
        with self._call_lock:  # stream hand-over + call are one unit per handle (threads: pipeline.BatchLanes)
            self._sync_stream(v)
            _lib.check(self._lib.mmiss_index_update(self._h, lab.ctypes.data, _lib.ptr(v), int(v.shape[0])))

    def remove(self, labels) -> int:
        lab = np.ascontiguousarray(labels, dtype=np.int64).reshape(-1)
        n = C.c_int64(0)
        _lib.check(self._lib.mmiss_index_remove(self._h, lab.ctypes.data, int(lab.shape[0]), C.byref(n)))
        return n.value

    def clear(self) -> None:
        _lib.check(self._lib.mmiss_index_clear(self._h))

    # ------------------------------------------------------------------ read
    def count(self) -> int:
        n = C.c_int64(0)
        _lib.check(self._lib.mmiss_index_count(self._h, C.byref(n)))
        return n.value

    def __len__(self) -> int:
        return self.count()

    def labels(self) -> np.ndarray:
        n = self.count()
        out = np.empty(n, dtype=np.int64)
        _lib.check(self._lib.mmiss_index_labels(self._h, out.ctypes.data, n))
        return out

    def get(self, labels) -> np.ndarray:
        lab = np.ascontiguousarray(labels, dtype=np.int64).reshape(-1)
        out = np.empty((lab.shape[0], self.dim), dtype=np.float32)
        if lab.shape[0]:
            _lib.check(self._lib.mmiss_index_get(self._h, lab.ctypes.data, int(lab.shape[0]), out.ctypes.data))
        return out

    def query(self, queries, k: int) -> Tuple["np.ndarray", "np.ndarray", "np.ndarray"]:
        """-> (labels int64 [Q,k], distances float32 [Q,k], counts int32 [Q]); numpy in -> numpy out,
        CUDA tensor in -> CUDA tensors out (no host round trip)."""
        return self._query(queries, k, split=False)

    def query_begin(self, queries, k: int) -> "PendingQuery":
        """Queue the query's first pass and return at once (mmiss_index_query_begin); `.result()` of the returned handle waits
        for it, widens what the exactness guard could not prove, and returns what query() returns. Between the two the caller
        may queue other GPU work on the same stream (the next batch's encode) — but no other call on this index."""
        return self._query(queries, k, split=True)

    def query_next(self, pending: "PendingQuery", queries, k: int):
        """`pending.result()` and `query_begin(queries, k)` back to back -> (results of `pending`, the new PendingQuery). Everything
        Python has to do for the NEW query (layout checks, output buffers) happens BEFORE the pending one is ended, so between the
        last operation of its widen pass and the first kernel of the new first pass lie two C calls and nothing else. In a serving
        loop that keeps one query batch open per step (bench.py) the GPU idled ~87 us per step there (rocprofv3 kernel trace of
        round 6: the gap between the widen pass's count readback and prep_queries_kernel)."""
        q, outs = self._prepare(queries, k)
        prev = pending.result()
        with self._call_lock:
            self._sync_stream(q)
            _lib.check(self._lib.mmiss_index_query_begin(self._h, _lib.ptr(q), int(q.shape[0]), int(k), _lib.ptr(outs[0]), _lib.ptr(outs[1]),
                                                         _lib.ptr(outs[2])))
            self._query_gen += 1
            return prev, PendingQuery(self, q, outs, self._query_gen)

    def _prepare(self, queries, k: int):
        q = self._vecs(queries)
        Q = int(q.shape[0])
        if _is_torch(q) and q.is_cuda:
            import torch

            lab = torch.empty((Q, k), dtype=torch.int64, device=q.device)
            dist = torch.empty((Q, k), dtype=torch.float32, device=q.device)
            cnt = torch.empty((Q,), dtype=torch.int32, device=q.device)
        else:
            if _is_torch(q):
                q = q.numpy()
            lab = np.empty((Q, k), dtype=np.int64)
            dist = np.empty((Q, k), dtype=np.float32)
            cnt = np.empty((Q,), dtype=np.int32)
        return q, (lab, dist, cnt)

    def _query(self, queries, k: int, split: bool):
        q, (lab, dist, cnt) = self._prepare(queries, k)
        Q = int(q.shape[0])
        with self._call_lock:  # stream hand-over + call are one unit per handle (threads: pipeline.BatchLanes)
            self._sync_stream(q)
            fn = self._lib.mmiss_index_query_begin if split else self._lib.mmiss_index_query
            _lib.check(fn(self._h, _lib.ptr(q), Q, int(k), _lib.ptr(lab), _lib.ptr(dist), _lib.ptr(cnt)))
            if split:
                self._query_gen += 1
                return PendingQuery(self, q, (lab, dist, cnt), self._query_gen)
        return (lab, dist, cnt)

    def guard_stats(self) -> dict:
        """Exactness accounting (mmiss_index_guard_stats_ex): queries served, queries whose first pass could not be proven
        exact and were widened, widen (threshold) passes run, extra passes over the index they cost, the queries that ended in
        the exhaustive canonical pass (a tie plateau of more than 8192 rows), and the rows the threshold passes re-ranked."""
        out = (C.c_int64 * 8)()
        _lib.check(self._lib.mmiss_index_guard_stats_ex(self._h, out))
        return {"queries": out[0], "widened": out[1], "rounds": out[2], "pages": out[3], "exhaustive": out[4],
                "swept_rows": out[5]}

    # ------------------------------------------------------------------ persistence
    def save(self, path: str) -> None:
        _lib.check(self._lib.mmiss_index_save(self._h, str(path).encode()))

    def load(self, path: str) -> None:
        _lib.check(self._lib.mmiss_index_load(self._h, str(path).encode()))

    def abort_query(self) -> None:
        """Drop a query opened with query_begin() whose handle was lost (mmiss_index_query_abort); no-op otherwise."""
        with self._call_lock:
            self._query_gen += 1   # whatever handle held the open query is stale from here on
            _lib.check(self._lib.mmiss_index_query_abort(self._h))

    def close(self):
        if getattr(self, "_h", None):
            with self._call_lock:
                self._query_gen += 1
                self._lib.mmiss_index_query_abort(self._h)   # (an abandoned PendingQuery must not outlive its buffers)
                self._lib.mmiss_index_destroy(self._h)
                self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PendingQuery:
    """A query between FlatIndex.query_begin and its result(): holds the query rows and the output buffers alive. It owns the
    index's open-query slot only while its generation is the index's current one (FlatIndex._query_gen)."""

    def __init__(self, index: "FlatIndex", q, outs, gen: int):
        self._index, self._q, self._outs, self._gen, self._done = index, q, outs, gen, False

    def _current(self) -> bool:
        idx = self._index
        return getattr(idx, "_h", None) is not None and idx._query_gen == self._gen

    def result(self):
        if not self._done:
            idx = self._index
            with idx._call_lock:
                if not self._current():
                    self._done = True
                    self._q = None
                    raise RuntimeError("this query was aborted (FlatIndex.abort_query / close, or a later query_begin)")
                self._done = True   # (whatever _end returns, the C side has closed the query)
                idx._query_gen += 1
                _lib.check(idx._lib.mmiss_index_query_end(idx._h))
            self._q = None
        return self._outs

    def abort(self) -> None:
        """Give the query up (mmiss_index_query_abort): waits for the queued first pass, delivers nothing, frees the index
        for other calls. Also what happens when the handle is dropped or leaves a `with` block without result(). A handle
        whose query is no longer the index's open one (already aborted through the index, or superseded) does nothing."""
        if not self._done:
            self._done = True
            idx = self._index
            with idx._call_lock:
                if self._current():
                    idx._query_gen += 1
                    idx._lib.mmiss_index_query_abort(idx._h)
            self._q = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.abort()
        return False

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass


def blend(img, txt, weight_image: float, device: int = 0):
    """normalize(w * normalize(img) + (1 - w) * normalize(txt)) row-wise — backend/app/main.py:852-860."""
    lib = _lib.load()
    if _is_torch(img) and img.is_cuda:
        import torch

        img = img.to(torch.float32).contiguous()
        txt = txt.to(torch.float32).contiguous()
        out = torch.empty_like(img)
        stream = _lib.current_stream_ptr(img.device)
        device = img.device.index if img.device.index is not None else device
    else:
        img = np.ascontiguousarray(img, dtype=np.float32)
        txt = np.ascontiguousarray(txt, dtype=np.float32)
        out = np.empty_like(img)
        stream = None
    one_d = img.ndim == 1
    Q = 1 if one_d else int(img.shape[0])
    D = int(img.shape[-1])
    if tuple(img.shape) != tuple(txt.shape):
        raise ValueError("image and text embeddings differ in shape")
    _lib.check(lib.mmiss_blend(device, stream, _lib.ptr(img), _lib.ptr(txt), float(weight_image), Q, D, _lib.ptr(out)))
    return out


def merge_topk(dist, labels, device: int = 0):
    """dist [S,Q,k] f32, labels [S,Q,k] i64 (per-shard results) -> (labels [Q,k], dist [Q,k], count [Q])."""
    lib = _lib.load()
    S, Q, k = (int(x) for x in dist.shape)
    if _is_torch(dist) and dist.is_cuda:
        import torch

        dist = dist.contiguous()
        labels = labels.contiguous()
        od = torch.empty((Q, k), dtype=torch.float32, device=dist.device)
        ol = torch.empty((Q, k), dtype=torch.int64, device=dist.device)
        oc = torch.empty((Q,), dtype=torch.int32, device=dist.device)
        stream = _lib.current_stream_ptr(dist.device)
        device = dist.device.index if dist.device.index is not None else device
    else:
        dist = np.ascontiguousarray(dist, dtype=np.float32)
        labels = np.ascontiguousarray(labels, dtype=np.int64)
        od = np.empty((Q, k), dtype=np.float32)
        ol = np.empty((Q, k), dtype=np.int64)
        oc = np.empty((Q,), dtype=np.int32)
        stream = None
    _lib.check(lib.mmiss_merge_topk(device, stream, _lib.ptr(dist), _lib.ptr(labels), S, Q, k, _lib.ptr(od), _lib.ptr(ol), _lib.ptr(oc)))
    return ol, od, oc

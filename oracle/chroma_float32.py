"""float32 restatements of the two places where the reference's `collection.query` computes a cosine distance.

TEST INFRASTRUCTURE ONLY (tests/ import it; the product never does).

The reference calls chromadb (`/root/reference/backend/requirements.txt:10`, `chromadb>=0.4.13`; call sites
`backend/app/utils.py:127-130` create the collection with {"hnsw:space": "cosine"}, `backend/app/main.py:735-740` add,
`:761-765` query). chromadb and its C++ dependency chroma-hnswlib are NOT in this container and not vendored in the
reference, so what follows is restated from their published sources as the formulas stand there, and is itself UNPINNED —
nothing here was run against the real packages. It exists to measure one thing: how far a float32 evaluation in the
dependency's order lies from the canonical float64 evaluation of `retrieval_oracle.py`, i.e. what a user switching over can
see change (tests/test_chroma_float32_cpu.py: distances within 1e-6, ids identical away from ties closer than that).

1. Rows not yet in the HNSW graph (fewer than `hnsw:batch_size` = 100 pending rows) are searched by brute force with
   chromadb.utils.distance_functions.cosine:
       1.0 - np.dot(x, y) / ((np.linalg.norm(x) * np.linalg.norm(y)) + 1e-30)         on float32 vectors
2. chroma-hnswlib, cosine space = inner-product space on normalised vectors:
       add / query:  norm = sum(x_i^2) in float32;  x_i *= 1.0f / (sqrtf(norm) + 1e-30f)
       distance:     1.0f - sum(a_i * b_i)          float32, accumulated in index order (the scalar InnerProduct; its SIMD
                                                    forms keep 4 / 8 / 16 partial sums and add them at the end)
   evaluated here for EVERY row (the graph search approximates exactly this ranking).
"""
from __future__ import annotations

import numpy as np


def brute_force_cosine(q: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """(1) for every (query, row): float32 [Q, N]."""
    q = np.asarray(q, np.float32)
    rows = np.asarray(rows, np.float32)
    out = np.empty((q.shape[0], rows.shape[0]), np.float32)
    rn = np.linalg.norm(rows, axis=1).astype(np.float32)
    for i in range(q.shape[0]):
        qn = np.float32(np.linalg.norm(q[i]))
        dots = (rows @ q[i]).astype(np.float32)
        out[i] = np.float32(1.0) - dots / ((qn * rn) + np.float32(1e-30))
    return out


def _seq_sum_f32(v: np.ndarray, lanes: int) -> np.ndarray:
    """float32 sum over the last axis with `lanes` interleaved partial sums (1 = strictly sequential), every add rounded."""
    D = v.shape[-1]
    acc = np.zeros(v.shape[:-1] + (lanes,), np.float32)
    for i in range(0, D, lanes):
        acc = (acc + v[..., i:i + lanes]).astype(np.float32)
    tot = np.zeros(v.shape[:-1], np.float32)
    for lane in range(lanes):
        tot = (tot + acc[..., lane]).astype(np.float32)
    return tot


def hnswlib_normalize(x: np.ndarray, lanes: int = 1) -> np.ndarray:
    x = np.asarray(x, np.float32)
    norm = _seq_sum_f32((x * x).astype(np.float32), lanes)
    inv = (np.float32(1.0) / (np.sqrt(norm).astype(np.float32) + np.float32(1e-30))).astype(np.float32)
    return (x * inv[..., None]).astype(np.float32)


def hnswlib_cosine(q: np.ndarray, rows: np.ndarray, lanes: int = 1) -> np.ndarray:
    """(2) for every (query, row): float32 [Q, N]; lanes = 1 (scalar) or 16 (AVX-512 form)."""
    qn, rn = hnswlib_normalize(q, lanes), hnswlib_normalize(rows, lanes)
    out = np.empty((qn.shape[0], rn.shape[0]), np.float32)
    for i in range(qn.shape[0]):
        out[i] = np.float32(1.0) - _seq_sum_f32((rn * qn[i]).astype(np.float32), lanes)
    return out


def topk(dist: np.ndarray, k: int):
    """(ids, distances) of the k smallest, ties by row index (the insertion order chroma returns equal distances in)."""
    order = np.argsort(dist, axis=1, kind="stable")[:, :k]
    return order.astype(np.int64), np.take_along_axis(dist, order, axis=1)

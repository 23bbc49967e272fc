"""CPU restatement (numpy) of the cosine retrieval on the reference's query path.

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package.

What it restates (paths relative to /root/reference):
  collection created with {"hnsw:space": "cosine"}      backend/app/utils.py:127-130
  collection.add(embeddings=...)                         backend/app/main.py:735-740   -> normalize_rows
  collection.query(query_embeddings, n_results, ...)     backend/app/main.py:761-765   -> query
  similarity = 1 - distance / 2                          backend/app/main.py:782        -> similarity_from_distance
  search_multimodal blend                                backend/app/main.py:852-860   -> blend

PARITY UNPINNED at the chromadb boundary: the arithmetic of `collection.query` lives in chromadb
(requirements.txt:10 `chromadb>=0.4.13`, CHANGELOG says 0.6.0+) and its C++ dependency
chroma-hnswlib; neither is vendored under /root/reference nor installed in this container, and the
reference holds no test or golden vector for it. What is restated is the published definition of the
cosine space — stored vectors L2-normalised, distance = 1 - <q^, c^> — evaluated EXACTLY (brute
force), which is also what chromadb does below its `hnsw:batch_size` (100 rows) and what its HNSW
graph approximates above it.

Canonical arithmetic (shared bit for bit with the HIP rerank kernel, csrc/retrieval_kernels.h):
  canon_sum(v[0..D))   : 64 partial sums p[l] = v[l] + v[l+64] + v[l+128] + ... accumulated sequentially in
                         float64, then a fixed tree p[l] += p[l+s] for s = 32,16,8,4,2,1; result p[0].
  norm(x)              = sqrt(canon_sum(x*x))               (float64)
  normalize(x)         = float32(float64(x) / norm(x)), then the storage rounding (float16: RNE)
  distance(q, c)       = float32(1.0 - canon_sum(float64(q^) * float64(c_stored)))
                         fp8 rows (F8Rows): float32(1.0 - canon_sum(float64(q^) * float64(values)) * float64(inv))
  order                = (distance ascending, label ascending)
Products of float32 values are exact in float64, so the only roundings are the additions, in the fixed
order above.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def canon_sum(v: np.ndarray) -> np.ndarray:
    """v: float64 [..., D] with D % 64 == 0 -> float64 [...] in the canonical order."""
    D = v.shape[-1]
    assert D % 64 == 0, "canonical reduction needs D % 64 == 0"
    blocks = v.reshape(v.shape[:-1] + (D // 64, 64))
    p = np.zeros(v.shape[:-1] + (64,), dtype=np.float64)
    for i in range(D // 64):
        p = p + blocks[..., i, :]
    s = 32
    while s >= 1:
        p = p[..., :s] + p[..., s:2 * s]
        s //= 2
    return p[..., 0]


def canon_norm(x: np.ndarray) -> np.ndarray:
    x64 = np.asarray(x, dtype=np.float32).astype(np.float64)
    return np.sqrt(canon_sum(x64 * x64))


class F8Rows(np.ndarray):
    """Stored form of MMISS_F8 rows (include/mmiss.h): float32 [N, D] VALUES the e4m3 codes stand for (decode(code) / 128) plus
    `.inv`, float32 [N]: inv[r] = float32(1 / canon_norm(values[r])). The row the index represents is values[r] * inv[r] (unit
    norm up to inv's own rounding), a distance is float32(1 - canon_sum(q^ * values[r]) * float64(inv[r])). Indexing with a
    slice / index array / mask on the first axis keeps `.inv` in step; anything else yields a plain ndarray."""

    def __new__(cls, values: np.ndarray, inv: np.ndarray):
        obj = np.ascontiguousarray(values, dtype=np.float32).view(cls)
        obj.inv = np.ascontiguousarray(inv, dtype=np.float32)
        assert obj.ndim == 2 and obj.inv.shape == (obj.shape[0],)
        return obj

    def __array_finalize__(self, obj):
        self.inv = None   # (views made by numpy itself carry no inverse norms; __getitem__ below re-attaches them)

    def __getitem__(self, key):
        out = np.asarray(self)[key]
        rows = key[0] if isinstance(key, tuple) else key
        whole_rows = not isinstance(key, tuple) or all(isinstance(k, slice) and k == slice(None) for k in key[1:])
        if self.inv is not None and whole_rows and out.ndim == 2 and not isinstance(rows, (int, np.integer)):
            return F8Rows(out, self.inv[rows])
        return out

    def represented(self) -> np.ndarray:
        """float32 [N, D]: values * inv, one float32 multiply per element — what mmiss_index_get returns for fp8 rows."""
        return (np.asarray(self) * self.inv[:, None]).astype(np.float32)


def concat_rows(parts):
    """np.concatenate for stored rows of any dtype (F8Rows keep their inverse norms)."""
    if all(isinstance(p, F8Rows) for p in parts):
        return F8Rows(np.concatenate([np.asarray(p) for p in parts]), np.concatenate([p.inv for p in parts]))
    return np.concatenate(parts)


def normalize_rows(x: np.ndarray, dtype: str = "f32") -> np.ndarray:
    """Stored form of added vectors: float32(x / norm) then storage rounding."""
    x = np.asarray(x, dtype=np.float32)
    if x.ndim == 1:
        x = x[None]
    y = (x.astype(np.float64) / canon_norm(x)[..., None]).astype(np.float32)
    if dtype in ("f16", "float16", 1):
        return y.astype(np.float16)
    if dtype in ("f8", "fp8", "e4m3", 2):
        # MMISS_F8 (include/mmiss.h): e4m3 code of 128 * y (round to nearest even), stored value = decode(code) / 128 — returned as
        # the float32 values the codes stand for (exactly representable), so that distances() / query() work on them unchanged
        from oracle import fp8_oracle

        codes = fp8_oracle.e4m3_encode(y * np.float32(128.0))
        vals = (fp8_oracle.e4m3_decode(codes).astype(np.float32) * np.float32(1.0 / 128.0)).astype(np.float32)
        # one float per row: the inverse of the canonical norm of the values (csrc/retrieval_kernels.h f8_row_inv_kernel)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = (1.0 / canon_norm(vals)).astype(np.float32)
        return F8Rows(vals, inv)
    return y


def distances(q_raw: np.ndarray, stored: np.ndarray, block: int = 4096) -> np.ndarray:
    """All cosine distances, float32 [Q, N]. q_raw: float32 [Q,D] un-normalised queries; stored:
    normalize_rows output (float32 or float16)."""
    qn = normalize_rows(q_raw, "f32").astype(np.float64)
    out = np.empty((qn.shape[0], stored.shape[0]), dtype=np.float32)
    inv = getattr(stored, "inv", None)
    if isinstance(stored, F8Rows) and inv is None:
        raise ValueError("F8Rows lost their inverse norms (use row slices / concat_rows)")
    vals = np.asarray(stored)
    for r0 in range(0, vals.shape[0], block):
        c = vals[r0:r0 + block].astype(np.float32).astype(np.float64)
        for qi in range(qn.shape[0]):
            dot = canon_sum(c * qn[qi])
            if inv is not None:
                dot = dot * inv[r0:r0 + block].astype(np.float64)   # fp8 rows: the represented row is values * inv
            out[qi, r0:r0 + block] = (1.0 - dot).astype(np.float32)
    return out


def query(q_raw: np.ndarray, stored: np.ndarray, labels: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> (labels int64 [Q,k], distances float32 [Q,k], counts int32 [Q]); unused slots -1 / +inf."""
    q_raw = np.asarray(q_raw, dtype=np.float32)
    if q_raw.ndim == 1:
        q_raw = q_raw[None]
    labels = np.asarray(labels, dtype=np.int64)
    Q, N = q_raw.shape[0], stored.shape[0]
    out_l = np.full((Q, k), -1, dtype=np.int64)
    out_d = np.full((Q, k), np.inf, dtype=np.float32)
    cnt = np.full((Q,), min(k, N), dtype=np.int32)
    if N == 0:
        return out_l, out_d, cnt
    d = distances(q_raw, stored)
    for qi in range(Q):
        # a NaN distance (a zero-norm or non-finite row or query: 0/0 in the normalisation) is never a result — as in
        # retrieval_oracle.c; the count says how many results there are
        live = np.nonzero(~np.isnan(d[qi]))[0]
        kk = min(k, live.size)
        cnt[qi] = kk
        if kk == 0:
            continue
        dl = d[qi][live]
        if kk < live.size:
            # every row tied with the kk-th smallest distance must be kept for the label tie-break
            kth = np.partition(dl, kk - 1)[kk - 1]
            idx = live[np.nonzero(dl <= kth)[0]]
        else:
            idx = live
        order = np.lexsort((labels[idx], d[qi][idx]))[:kk]
        sel = idx[order]
        out_l[qi, :kk] = labels[sel]
        out_d[qi, :kk] = d[qi][sel]
    return out_l, out_d, cnt


def similarity_from_distance(d):
    """backend/app/main.py:782: similarity = 1 - distance / 2, evaluated in Python floats (float64)."""
    return [1 - (float(x) / 2) for x in d]


def blend(img: np.ndarray, txt: np.ndarray, weight_image: float) -> np.ndarray:
    """search_multimodal's combination (backend/app/main.py:852-860) with canonical norms:
    i^ = normalize(i), t^ = normalize(t); c = f32(w)*i^ + f32(1-w)*t^ (float32 multiply, multiply, add, the
    way numpy evaluates `python_float * float32_array`); out = normalize(c)."""
    i_n = normalize_rows(img, "f32")
    t_n = normalize_rows(txt, "f32")
    w_i = np.float32(weight_image)
    w_t = np.float32(1.0 - weight_image)
    c = (w_i * i_n).astype(np.float32) + (w_t * t_n).astype(np.float32)
    return normalize_rows(c.astype(np.float32), "f32")


def merge_shards(dist: np.ndarray, labels: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """dist [S,Q,k], labels [S,Q,k] -> global (labels [Q,k], dist [Q,k], count [Q]) by (dist asc, label asc)."""
    S, Q, kk = dist.shape
    out_l = np.full((Q, k), -1, dtype=np.int64)
    out_d = np.full((Q, k), np.inf, dtype=np.float32)
    cnt = np.zeros((Q,), dtype=np.int32)
    for q in range(Q):
        d = dist[:, q, :].reshape(-1)
        l = labels[:, q, :].reshape(-1)
        ok = l >= 0
        d, l = d[ok], l[ok]
        order = np.lexsort((l, d))[:k]
        n = order.shape[0]
        out_l[q, :n] = l[order]
        out_d[q, :n] = d[order]
        cnt[q] = n
    return out_l, out_d, cnt

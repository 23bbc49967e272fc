"""ctypes wrapper of oracle/libmmiss_oracle.so (retrieval_oracle.c). TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmmiss_oracle.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = C.CDLL(path)
        _LIB.mo_normalize_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        _LIB.mo_query.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.mo_query_f8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.mo_blend.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p]
        for f in (_LIB.mo_normalize_rows, _LIB.mo_query, _LIB.mo_query_f8, _LIB.mo_blend):
            f.restype = None
    return _LIB


def normalize_rows(x: np.ndarray, dtype: str = "f32") -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim == 1:
        x = x[None]
    f16 = dtype in ("f16", "float16", 1)
    out = np.empty(x.shape, dtype=np.float16 if f16 else np.float32)
    _lib().mo_normalize_rows(x.ctypes.data, x.shape[0], x.shape[1], 1 if f16 else 0, out.ctypes.data)
    return out


def query(q_raw: np.ndarray, stored: np.ndarray, labels: np.ndarray, k: int):
    q = np.ascontiguousarray(q_raw, dtype=np.float32)
    if q.ndim == 1:
        q = q[None]
    inv = getattr(stored, "inv", None)   # retrieval_oracle.F8Rows: fp8 rows carry one inverse norm each
    stored = np.ascontiguousarray(np.asarray(stored))
    labels = np.ascontiguousarray(labels, dtype=np.int64)
    Q, D = q.shape
    ol = np.empty((Q, k), dtype=np.int64)
    od = np.empty((Q, k), dtype=np.float32)
    oc = np.empty((Q,), dtype=np.int32)
    if inv is not None:
        inv = np.ascontiguousarray(inv, dtype=np.float32)
        assert stored.dtype == np.float32 and inv.shape == (stored.shape[0],)
        _lib().mo_query_f8(q.ctypes.data, Q, stored.ctypes.data, inv.ctypes.data, stored.shape[0], D, labels.ctypes.data, k,
                           ol.ctypes.data, od.ctypes.data, oc.ctypes.data)
        return ol, od, oc
    _lib().mo_query(q.ctypes.data, Q, stored.ctypes.data, 1 if stored.dtype == np.float16 else 0, stored.shape[0], D,
                    labels.ctypes.data, k, ol.ctypes.data, od.ctypes.data, oc.ctypes.data)
    return ol, od, oc


def blend(img: np.ndarray, txt: np.ndarray, w: float) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.float32)
    txt = np.ascontiguousarray(txt, dtype=np.float32)
    if img.ndim == 1:
        img, txt = img[None], txt[None]
    out = np.empty_like(img)
    _lib().mo_blend(img.ctypes.data, txt.ctypes.data, float(w), img.shape[0], img.shape[1], out.ctypes.data)
    return out

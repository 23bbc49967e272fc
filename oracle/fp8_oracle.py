"""CPU restatement (numpy) of the number formats of the fp8 encode path. TEST INFRASTRUCTURE ONLY.

The reference has no fp8 path (it runs fp32 on the CPU, backend/app/utils.py:77,97); the fp8 towers are held to the same
bar as the bf16 ones against the fp32 oracle (oracle/clip_oracle.py). What is restated here are the published formats the
HIP kernels (csrc/gemm_fp8.h) must implement exactly, so that their arithmetic can be checked element by element:

  OCP FP8 E4M3 ("e4m3fn"): 1 sign, 4 exponent (bias 7), 3 mantissa bits; no infinities; S.1111.111 = NaN; subnormals
      m/8 * 2^-6; largest finite 1.75 * 2^8 = 448  (OCP 8-bit Floating Point Specification v1.0, table 1)
  E8M0 block scale: value 2^(byte - 127)  (OCP Microscaling Formats v1.0, MXFP8: one scale per 32 elements)

Pinned against torch.float8_e4m3fn (an independent implementation of the same format) in tests/test_fp8_oracle_cpu.py.
"""
from __future__ import annotations

import numpy as np


def e4m3_table() -> np.ndarray:
    """float32 value of every e4m3 byte (NaN for 0x7f / 0xff)."""
    t = np.empty(256, np.float32)
    for b in range(256):
        s = -1.0 if b & 0x80 else 1.0
        e, m = (b >> 3) & 0xF, b & 7
        if e == 15 and m == 7:
            v = np.nan
        elif e == 0:
            v = s * (m / 8.0) * 2.0 ** -6
        else:
            v = s * (1.0 + m / 8.0) * 2.0 ** (e - 7)
        t[b] = v
    return t


_TAB = e4m3_table()
_POS = _TAB[:127].astype(np.float64)  # codes 0..126 ascending: 0 .. 448


def e4m3_decode(b: np.ndarray) -> np.ndarray:
    return _TAB[np.asarray(b, np.uint8)]


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """Round to nearest, ties to even, |x| <= 448 (callers scale first); NaN -> 0x7F / 0xFF (the format's NaN). -> uint8."""
    x = np.asarray(x, np.float64)
    a = np.abs(x)
    hi = np.clip(np.searchsorted(_POS, a, side="left"), 0, 126)
    lo = np.clip(hi - 1, 0, 126)
    dl, dh = a - _POS[lo], _POS[hi] - a
    pick_hi = (dh < dl) | ((dh == dl) & (hi % 2 == 0))
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    code = np.where(np.isnan(x), np.uint8(0x7F), code).astype(np.uint8)
    return np.where(np.signbit(x), code | 0x80, code).astype(np.uint8)


def e8m0_for(amax: np.ndarray):
    """(byte, multiplier 2^-(byte-127)) with amax * multiplier <= 448: byte = ceil(log2(amax / 448)) + 127, in float32
    arithmetic exactly as the kernels do (exponent field + (mantissa != 0))."""
    r = (np.maximum(np.asarray(amax, np.float32), np.float32(1e-30)) * np.float32(1.0 / 448.0)).astype(np.float32)
    u = r.view(np.uint32)
    e = (u >> 23).astype(np.int64) + ((u & 0x7FFFFF) != 0)
    e = np.clip(e, 1, 254)
    inv = ((254 - e).astype(np.uint32) << 23).view(np.float32)
    return e.astype(np.uint8), inv


def scale_row_bytes(K: int) -> int:
    return 16 * ((K + 511) // 512)


def scale_offset(b):
    """byte offset of k-block b (= k // 32) inside a permuted scale row (csrc/gemm_fp8.h)."""
    b = np.asarray(b)
    return (b >> 4) * 16 + (b & 3) * 4 + ((b >> 2) & 3)


def permute_scales(s: np.ndarray) -> np.ndarray:
    """[M, K/32] E8M0 bytes in natural block order -> the permuted [M, scale_row_bytes(K)] layout (pad bytes = 127)."""
    M, nb = s.shape
    out = np.full((M, scale_row_bytes(nb * 32)), 127, np.uint8)
    out[:, scale_offset(np.arange(nb))] = s
    return out


def unpermute_scales(p: np.ndarray, K: int) -> np.ndarray:
    return p[:, scale_offset(np.arange(K // 32))]


def mx_quantize(y: np.ndarray, block: int = 32):
    """float32 [M, K] -> (e4m3 bytes [M,K], E8M0 bytes [M, K/32] natural order) with one scale per `block` columns
    (block = 32, or 64 = the FC1 epilogue's granularity: both 32-blocks of a 64-column group share the scale)."""
    y = np.asarray(y, np.float32)
    M, K = y.shape
    g = np.abs(y).reshape(M, K // block, block).max(axis=2)
    e, inv = e8m0_for(g)
    q = e4m3_encode((y.reshape(M, K // block, block) * inv[:, :, None]).astype(np.float32)).reshape(M, K)
    return q, np.repeat(e, block // 32, axis=1)


def mx_dequantize(q: np.ndarray, e: np.ndarray) -> np.ndarray:
    """e4m3 bytes [M,K] + E8M0 [M, K/32] natural order -> float64 [M,K]."""
    M, K = q.shape
    sc = np.exp2(e.astype(np.float64) - 127.0)
    return (e4m3_decode(q).astype(np.float64).reshape(M, K // 32, 32) * sc[:, :, None]).reshape(M, K)


def quantize_weights(w: np.ndarray):
    """bf16-valued float32 [N,K] -> (e4m3 [N,K], scale f32 [N] = amax / 448), as quantize_weights_fp8_kernel."""
    w = np.asarray(w, np.float32)
    amax = np.abs(w).max(axis=1)
    sc = np.where(amax > 0, amax * np.float32(1.0 / 448.0), np.float32(1.0)).astype(np.float32)
    inv = (np.float32(1.0) / sc).astype(np.float32)
    v = np.clip((w * inv[:, None]).astype(np.float32), -448.0, 448.0)
    return e4m3_encode(v), sc

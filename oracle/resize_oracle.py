"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the resize step of CLIPImageProcessor
(HF:image_processing_clip.py:23-34, called by the reference at backend/app/utils.py:76): resize so the SHORTEST edge
is S with PIL's BICUBIC filter, then centre-crop S x S.

The arithmetic lives in a third-party dependency, Pillow (`Image.resize`, C file src/libImaging/Resample.c; the
reference pins only `Pillow>=10` in requirements.txt; 12.2 is installed here). Its published algorithm, restated:

  * per axis, `precompute_coeffs`: scale = in/out, filterscale = max(scale, 1), support = 2 * filterscale (bicubic),
    for every output index xx: center = (xx + 0.5) * scale, taps xmin = int(center - support + 0.5) clipped to >= 0,
    xmax = int(center + support + 0.5) clipped to <= in; w(x) = bicubic((x + xmin - center + 0.5) / filterscale),
    normalised by their sum; all in IEEE double;
  * 8-bit path: weights become 22-bit fixed point, round-half-away-from-zero (`normalize_coeffs_8bpc`); a pass
    computes clip8((2^21 + sum pixel * k) >> 22) per channel; horizontal pass first, then vertical, with the
    intermediate image rounded to uint8.

Pinned against the real Pillow in tests/test_resize_oracle_cpu.py (bit-exact on a sweep of sizes, up- and
down-scaling), so parity of the GPU kernel with this file is parity with what the reference calls.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """-> (ksize, bounds int[out,2] = (xmin, count), kk int32[out, ksize] fixed point). Box = the whole axis."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            pre = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + pre * (1 << PRECISION_BITS)) if pre < 0 else int(0.5 + pre * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, first: int, count: int) -> np.ndarray:
    """Resample axis 0 of uint8 img [n_in, m, c] to output indices first .. first+count-1 -> uint8 [count, m, c]."""
    out = np.empty((count,) + img.shape[1:], np.uint8)
    for i in range(count):
        xmin, n = bounds[first + i]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += img[xmin + x].astype(np.int64) * int(kk[first + i, x])
        # int32 arithmetic in C: the sums stay far inside int32, so int64 here is the same value
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def output_geometry(h: int, w: int, size: int):
    """Shortest edge -> size, long edge int(size * long / short); centre-crop offsets. -> (new_h, new_w, top, left)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    return new_h, new_w, (new_h - size) // 2, (new_w - size) // 2


def resize_bicubic_u8(rgb: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """uint8 [H,W,3] -> uint8 [new_h,new_w,3], the whole resized image (Image.resize((new_w,new_h), BICUBIC))."""
    h, w, _ = rgb.shape
    _, bx, kx = precompute_coeffs(w, new_w)
    _, by, ky = precompute_coeffs(h, new_h)
    tmp = _pass(np.ascontiguousarray(rgb.transpose(1, 0, 2)), bx, kx, 0, new_w).transpose(1, 0, 2)  # horizontal
    return _pass(np.ascontiguousarray(tmp), by, ky, 0, new_h)                                       # vertical


def resize_crop_u8(rgb: np.ndarray, size: int = 224) -> np.ndarray:
    """uint8 [H,W,3] -> uint8 [size,size,3]: the CLIPImageProcessor resize + centre crop. Computes only the window."""
    h, w, _ = rgb.shape
    new_h, new_w, top, left = output_geometry(h, w, size)
    if new_h < size or new_w < size:
        raise ValueError("centre crop larger than the resized image")
    _, bx, kx = precompute_coeffs(w, new_w)
    _, by, ky = precompute_coeffs(h, new_h)
    tmp = _pass(np.ascontiguousarray(rgb.transpose(1, 0, 2)), bx, kx, left, size).transpose(1, 0, 2)
    return _pass(np.ascontiguousarray(tmp), by, ky, top, size)

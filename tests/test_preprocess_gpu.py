"""SURVEY §8(f) N2 on the GPU: CLIPImageProcessor's resize (shortest edge, bicubic) + centre crop as HIP kernels,
bit-exact against oracle/resize_oracle.py (itself pinned bit-exact against Pillow in test_resize_oracle_cpu.py) and
against the committed HF-processor goldens; then the raw-RGB encode entry point against the uint8 one."""
import os

import numpy as np
import pytest

import mmiss_amd  # noqa: F401
from mmiss_amd.encoder import VIT_B32, ClipEncoder, ClipShape
from oracle import clip_oracle as co
from oracle import resize_oracle as ro

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SIZES = [(480, 640), (640, 480), (224, 224), (225, 224), (224, 300), (300, 224), (100, 80), (37, 91), (1024, 768),
         (333, 1000), (1200, 900), (224, 1792), (17, 17), (231, 229), (1, 1), (3, 500)]


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    if seed % 2:
        yy, xx = np.mgrid[0:h, 0:w]
        return np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy * 3 + xx * 5) % 256)], -1).astype(np.uint8)
    return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.fixture(scope="module")
def b32():
    enc = ClipEncoder(VIT_B32, max_batch_image=8, max_batch_text=8)  # no weights: the resize needs none
    yield enc
    enc.close()


def test_mixed_batch_bit_exact(b32):
    imgs = [_img(h, w, i) for i, (h, w) in enumerate(SIZES)]  # 16 images > max_batch_image: two chunks
    got = b32.resize_crop_rgb(imgs)
    assert got.shape == (len(imgs), 224, 224, 3) and got.dtype == np.uint8
    for i, im in enumerate(imgs):
        want = ro.resize_crop_u8(im, 224)
        assert np.array_equal(got[i], want), f"image {i} {im.shape}: {np.abs(got[i].astype(int) - want).max()}"


def test_extremes_and_saturation(b32):
    # black/white checkers drive the negative lobes of the filter past [0,255]: clip8 must saturate like Pillow's
    h, w = 400, 600
    yy, xx = np.mgrid[0:h, 0:w]
    chk = (((yy // 3 + xx // 2) % 2) * 255).astype(np.uint8)
    imgs = [np.stack([chk, 255 - chk, chk], -1), np.zeros((50, 60, 3), np.uint8), np.full((300, 900, 3), 255, np.uint8)]
    got = b32.resize_crop_rgb(imgs)
    for i, im in enumerate(imgs):
        assert np.array_equal(got[i], ro.resize_crop_u8(im, 224))


def test_golden_hf_processor(b32):
    g = np.load(os.path.join(ROOT, "tests", "golden", "preprocess.npz"))
    crops = b32.resize_crop_rgb([g["wide_u8"], g["tall_u8"]])
    px = co.normalize_u8(crops)
    assert np.abs(px[0] - g["wide_pixels"]).max() < 2e-6
    assert np.abs(px[1] - g["tall_pixels"]).max() < 2e-6


def test_other_crop_size_and_empty():
    s = co.TINY
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=4, max_batch_text=4)
    imgs = [_img(90, 70, 2), _img(64, 64, 3), _img(65, 200, 4), _img(33, 40, 5), _img(500, 64, 6)]
    got = enc.resize_crop_rgb(imgs)
    for i, im in enumerate(imgs):
        assert np.array_equal(got[i], ro.resize_crop_u8(im, s.v_image))
    assert enc.resize_crop_rgb([]).shape == (0, s.v_image, s.v_image, 3)
    enc.close()


def test_encode_rgb_equals_encode_u8_of_the_crops():
    s = co.TINY
    W = co.init_weights(s, seed=3)
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=4, max_batch_text=4)
    enc.load_state_dict(W)
    imgs = [_img(90 + 7 * i, 70 + 11 * i, i) for i in range(6)]
    crops = np.stack([ro.resize_crop_u8(im, s.v_image) for im in imgs])
    a = enc.encode_image_rgb(imgs)
    b = enc.encode_image(crops)
    assert np.array_equal(a, b)
    want = co.embed_images(co.normalize_u8(crops), W, s)
    assert (1.0 - (a * want).sum(1)).max() < 1e-3  # north_star tolerance: 1e-3 cosine
    enc.close()


def test_bad_arguments(b32):
    with pytest.raises(ValueError):
        b32.resize_crop_rgb([np.zeros((4, 4), np.uint8)])
    with pytest.raises(RuntimeError):  # edges are limited to 65536
        b32.resize_crop_rgb([np.zeros((1, 70000, 3), np.uint8)])
    import ctypes as C
    from mmiss_amd import _lib

    blob = np.zeros(10 * 10 * 3, np.uint8)
    off = np.array([8], np.int64)  # runs past the end of the blob
    hs, ws = np.array([10], np.int32), np.array([10], np.int32)
    out = np.zeros((1, 224, 224, 3), np.uint8)
    rc = _lib.load().mmiss_resize_crop_rgb(b32._h, _lib.ptr(blob), blob.size, _lib.ptr(off), _lib.ptr(hs), _lib.ptr(ws), 1,
                                           _lib.ptr(out))
    assert rc != 0

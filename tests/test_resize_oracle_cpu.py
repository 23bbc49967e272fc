"""Pins oracle/resize_oracle.py (the restatement of Pillow's 8-bit bicubic resample) against the real Pillow that the
reference's CLIPProcessor calls (backend/app/utils.py:76 -> HF:image_processing_clip.py:23-34), bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clip_oracle as co  # noqa: E402
from oracle import resize_oracle as ro  # noqa: E402

Image = pytest.importorskip("PIL.Image")

SIZES = [(480, 640), (640, 480), (224, 224), (225, 224), (224, 300), (300, 224), (100, 80), (37, 91), (1024, 768),
         (333, 1000), (2000, 1500), (224, 1792), (17, 17), (231, 229)]


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if seed % 2:  # smooth content as well as noise
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
    return base


@pytest.mark.parametrize("hw", SIZES)
def test_full_resize_equals_pillow(hw):
    h, w = hw
    rgb = _img(h, w, h * 7 + w)
    new_h, new_w, _, _ = ro.output_geometry(h, w, 224)
    want = np.asarray(Image.fromarray(rgb).resize((new_w, new_h), resample=Image.BICUBIC))
    got = ro.resize_bicubic_u8(rgb, new_h, new_w)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("hw", SIZES)
def test_window_equals_clip_processor_crop(hw):
    h, w = hw
    rgb = _img(h, w, h * 3 + w + 1)
    want = co.crop_u8(Image.fromarray(rgb), 224)
    assert np.array_equal(ro.resize_crop_u8(rgb, 224), want)


def test_other_target_sizes():
    rgb = _img(500, 375, 5)
    for s in (32, 336):
        assert np.array_equal(ro.resize_crop_u8(rgb, s), co.crop_u8(Image.fromarray(rgb), s))


def test_golden_hf_processor_outputs():
    """tests/golden/preprocess.npz holds raw uint8 images and what transformers' CLIPImageProcessor made of them."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "preprocess.npz"))
    for name in ("wide", "tall"):
        crop = ro.resize_crop_u8(g[name + "_u8"], 224)
        got = co.normalize_u8(crop[None])[0]
        assert np.abs(got - g[name + "_pixels"]).max() < 2e-6

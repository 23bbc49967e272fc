"""Host-side steps before the GPU path: the CLIP BPE tokenizer against transformers' CLIPTokenizer on a synthetic
vocabulary (the real vocab.json / merges.txt do not exist offline), and image preprocessing against fixtures."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _synthetic_vocab():
    import mmiss_amd  # noqa: F401
    from mmiss_amd.preprocess import _bytes_to_unicode

    chars = list(_bytes_to_unicode().values())
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    merges = []

    def merge(a, b):
        merges.append(f"{a} {b}")
        vocab.setdefault(a + b, len(vocab))

    for a, b in [("r", "e"), ("re", "d</w>"), ("d", "r"), ("dr", "i"), ("dri", "l"), ("dril", "l</w>"), ("t", "h"),
                 ("th", "e</w>"), ("i", "n"), ("in", "g</w>"), ("o", "r"), ("a", "n"), ("an", "g"), ("ang", "e</w>"),
                 ("or", "ange</w>"), ("1", "</w>"), ("'", "s</w>"), ("!", "!</w>"), ("c", "o"), ("co", "r"), ("cor", "d"),
                 ("l", "e"), ("le", "s"), ("les", "s</w>"), ("cord", "less</w>")]:
        merge(a, b)
    vocab["</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    return vocab, merges


TEXTS = ["red drill", "The  ORANGE\tdrill!!", "a cordless drill's battery 18V", "naïve café — ünïcode ✓", "", "   ",
         "drill " * 40, "it's 2 drills, isn't it?", "<|endoftext|> inside"]


def test_bpe_matches_hf_tokenizer_on_synthetic_vocab():
    transformers = pytest.importorskip("transformers")
    from mmiss_amd.preprocess import ClipBPETokenizer

    vocab, merges = _synthetic_vocab()
    mine = ClipBPETokenizer(vocab, merges)
    hf = transformers.CLIPTokenizer(vocab=vocab, merges=[tuple(m.split()) for m in merges])
    for max_length in (77, 16):
        ours = mine(TEXTS, max_length)
        theirs = hf(TEXTS, padding="max_length", max_length=max_length, truncation=True, return_tensors="np")["input_ids"]
        np.testing.assert_array_equal(ours, theirs.astype(np.int32))
    assert ours.dtype == np.int32 and (ours[:, 0] == vocab["<|startoftext|>"]).all()


def test_processor_without_vocab_fails_loudly_and_preprocess_matches_fixture():
    from PIL import Image
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import VIT_B32
    from mmiss_amd.preprocess import ClipProcessor

    proc = ClipProcessor(VIT_B32)
    with pytest.raises(RuntimeError, match="vocabulary"):
        proc.tokenize(["red drill"])
    g = np.load(os.path.join(G, "preprocess.npz"))
    imgs = [Image.fromarray(g["wide_u8"]), Image.fromarray(g["tall_u8"])]
    px = proc.preprocess_images(imgs)
    assert px.shape == (2, 3, 224, 224) and px.dtype == np.float32
    np.testing.assert_allclose(px[0], g["wide_pixels"], atol=1e-6)
    np.testing.assert_allclose(px[1], g["tall_pixels"], atol=1e-6)
    u8 = proc.crop_images_u8(imgs)
    assert u8.shape == (2, 224, 224, 3) and u8.dtype == np.uint8
    gray = proc.preprocess_images([Image.fromarray(g["wide_u8"][..., 0])])  # mode "L" uploads (main.py:140-143)
    assert gray.shape == (1, 3, 224, 224)


def test_drill_set_crops_reproduce(tmp_path):
    """The committed u8 crops of the reference's six sample images give the fixture's HF pixel values back."""
    from oracle import clip_oracle as co

    g = np.load(os.path.join(G, "drill_set.npz"))
    assert g["crops_u8"].shape == (6, 224, 224, 3) and len(g["names"]) == 6
    assert float(g["hf_pixel_max_abs_diff"]) < 1e-6
    assert co.normalize_u8(g["crops_u8"]).shape == (6, 3, 224, 224)

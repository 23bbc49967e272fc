"""MMISS_CLIP_CHECKPOINT -> load_clip_model() -> generate_clip_embedding(): the route INTEGRATION.md tells a maintainer
of the reference to take (backend/app/utils.py:41-45,59-102), on a synthetic HF-layout directory."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


@pytest.mark.parametrize("dtype", ["float32", "float16", "bfloat16"])
def test_checkpoint_route_equals_state_dict_route_and_oracle(tmp_path, monkeypatch, dtype):
    from PIL import Image
    import mmiss_amd  # noqa: F401
    from mmiss_amd import utils
    from mmiss_amd.encoder import ClipEncoder, ClipShape, iter_safetensors_f32
    from oracle import clip_oracle as co
    from ckpt_fixture import tiny_longclip_shape, write_checkpoint
    from test_tokenizer_preprocess_cpu import _synthetic_vocab

    vocab, _ = _synthetic_vocab()
    shape = tiny_longclip_shape(len(vocab), vocab["<|endoftext|>"])
    W = co.init_weights(shape, seed=17)
    write_checkpoint(str(tmp_path), shape, W, dtype)
    monkeypatch.setenv("MMISS_CLIP_CHECKPOINT", str(tmp_path))
    monkeypatch.delenv("MMISS_CLIP_RANDOM_INIT", raising=False)
    utils.set_clip_model(None, None)
    try:
        model, processor = utils.load_clip_model()
        assert model.shape.t_ctx == 248 and processor.max_length == 248     # position table rows, not config.json's 77
        assert utils.load_clip_model()[0] is model                          # cached (utils.py:33-35)
        rng = np.random.Generator(np.random.Philox(5))
        img = Image.fromarray(rng.integers(0, 256, size=(70, 90, 3), dtype=np.uint8))
        out = utils.generate_clip_embedding(image=img, text="The orange drill!!", model=model, processor=processor)
        assert out["image"].shape == (1, shape.proj_dim) and out["text"].shape == (1, shape.proj_dim)
        # the same weights (as stored: rounded to the file's dtype) through load_state_dict
        Wr = {k: v for k, v in iter_safetensors_f32(os.path.join(str(tmp_path), "model.safetensors")) if k in W}
        ref = ClipEncoder(ClipShape.from_any(shape), max_batch_image=4, max_batch_text=4)
        ref.load_state_dict(Wr)
        px_u8 = model.resize_crop_rgb(processor.rgb_arrays([img]))
        ids = processor.tokenize(["The orange drill!!"])
        np.testing.assert_array_equal(out["image"], ref.encode_image(px_u8))
        np.testing.assert_array_equal(out["text"], ref.encode_text(ids))
        # ... and the fp32 oracle on those weights
        assert (1 - _cos(out["image"], co.embed_images(co.normalize_u8(px_u8), Wr, shape))).max() < 1e-3
        assert (1 - _cos(out["text"], co.embed_texts(ids, Wr, shape))).max() < 1e-3
    finally:
        utils.set_clip_model(None, None)

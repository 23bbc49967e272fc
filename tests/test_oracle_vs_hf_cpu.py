"""Live pin of the oracle against the third-party implementation the reference calls (transformers.CLIPModel),
run in this container when transformers is importable. Follows backend/app/utils.py:76-79,88-99."""
import numpy as np
import pytest

from oracle import clip_oracle as co

transformers = pytest.importorskip("transformers")
torch = pytest.importorskip("torch")


def _hf(shape, W):
    from transformers import CLIPConfig, CLIPModel

    cfg = CLIPConfig(
        text_config=dict(hidden_size=shape.t_hidden, intermediate_size=shape.t_mlp, num_hidden_layers=shape.t_layers,
                         num_attention_heads=shape.t_heads, vocab_size=shape.t_vocab, max_position_embeddings=shape.t_ctx,
                         eos_token_id=shape.eos_token_id, bos_token_id=shape.eos_token_id - 1, pad_token_id=1,
                         projection_dim=shape.proj_dim),
        vision_config=dict(hidden_size=shape.v_hidden, intermediate_size=shape.v_mlp, num_hidden_layers=shape.v_layers,
                           num_attention_heads=shape.v_heads, image_size=shape.v_image, patch_size=shape.v_patch,
                           projection_dim=shape.proj_dim),
        projection_dim=shape.proj_dim)
    m = CLIPModel(cfg).eval()
    sd = m.state_dict()
    assert set(W) <= set(sd), "oracle weight keys must be HF state_dict keys"
    m.load_state_dict({k: (torch.from_numpy(W[k]).reshape(v.shape) if k in W else v) for k, v in sd.items()})
    return m


@pytest.mark.parametrize("seed", [0, 3])
def test_tiny_matches_hf(seed):
    s = co.TINY
    W = co.init_weights(s, seed)
    m = _hf(s, W)
    rng = np.random.Generator(np.random.Philox(seed + 50))
    px = rng.standard_normal((6, 3, s.v_image, s.v_image), dtype=np.float32)
    ids = co.synthetic_text_ids(6, s.t_ctx, s.t_vocab, s.eos_token_id, seed=seed + 60)
    with torch.no_grad():
        hi = m.get_image_features(pixel_values=torch.from_numpy(px)).pooler_output
        hi = (hi / hi.norm(dim=1, keepdim=True)).numpy()
        t = torch.from_numpy(ids).long()
        mask = torch.zeros_like(t)
        for r, e in enumerate(co.eos_positions(ids, s.eos_token_id)):
            mask[r, : e + 1] = 1
        ht = m.get_text_features(input_ids=t, attention_mask=mask).pooler_output
        ht = (ht / ht.norm(dim=1, keepdim=True)).numpy()
    np.testing.assert_allclose(co.embed_images(px, W, s), hi, atol=3e-6)
    np.testing.assert_allclose(co.embed_texts(ids, W, s), ht, atol=3e-6)


def test_legacy_eos_id_2_uses_argmax_pooling():
    import dataclasses

    s = dataclasses.replace(co.TINY, eos_token_id=2)
    W = co.init_weights(s, 1)
    m = _hf(s, W)
    rng = np.random.Generator(np.random.Philox(9))
    ids = rng.integers(3, 900, size=(4, s.t_ctx)).astype(np.int32)
    ids[:, 0] = 0
    for r, p in enumerate([5, 9, 15, 2]):
        ids[r, p] = 999  # the largest id marks the pooled position
    with torch.no_grad():
        ht = m.get_text_features(input_ids=torch.from_numpy(ids).long()).pooler_output.numpy()
    np.testing.assert_allclose(co.text_features(ids, W, s), ht, atol=3e-5)

"""oracle/clip_oracle_torch.py (the PyTorch-CPU fp32 restatement bench.py times as the CPU baseline) against the numpy
oracle, which is itself pinned against transformers.CLIPModel."""
import dataclasses

import numpy as np
import pytest

from oracle import clip_oracle as co


def test_torch_restatement_equals_numpy_oracle():
    pytest.importorskip("torch")
    from oracle import clip_oracle_torch as ct

    for s in (co.TINY, dataclasses.replace(co.TINY, v_patch=14, v_image=56, t_ctx=24)):
        W = co.init_weights(s, seed=3)
        Wt = ct.to_torch(W)
        rng = np.random.Generator(np.random.Philox(4))
        px = rng.standard_normal((5, 3, s.v_image, s.v_image), dtype=np.float32)
        ids = co.synthetic_text_ids(6, s.t_ctx, s.t_vocab, s.eos_token_id, seed=5)
        np.testing.assert_allclose(ct.embed_images(px, Wt, s).numpy(), co.embed_images(px, W, s), atol=3e-6)
        np.testing.assert_allclose(ct.embed_texts(ids, Wt, s).numpy(), co.embed_texts(ids, W, s), atol=3e-6)

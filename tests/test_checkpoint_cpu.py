"""The checkpoint route of load_clip_model() (backend/app/utils.py:41-45 of the reference): everything that can be
checked without a GPU — config.json with omitted default keys, the hidden_act guard, bf16 / f16 safetensors, the
position-table override, vocab files next to the weights."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _vocab():
    from test_tokenizer_preprocess_cpu import _synthetic_vocab

    return _synthetic_vocab()


def test_config_with_omitted_defaults_and_act_guard():
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipShape, VIT_B32

    assert ClipShape.from_hf_config({}) == VIT_B32                      # an all-default config.json is ViT-B/32
    assert ClipShape.from_hf_config({"text_config": None, "vision_config": {}}) == VIT_B32
    s = ClipShape.from_hf_config({"projection_dim": 768, "vision_config": {"hidden_size": 1024, "num_hidden_layers": 24,
                                  "num_attention_heads": 16, "intermediate_size": 4096, "patch_size": 14},
                                  "text_config": {"hidden_size": 768, "num_attention_heads": 12, "intermediate_size": 3072,
                                                  "max_position_embeddings": 248}})
    from mmiss_amd.encoder import LONGCLIP_L14
    assert s == LONGCLIP_L14
    with pytest.raises(ValueError, match="quick_gelu"):
        ClipShape.from_hf_config({"vision_config": {"hidden_act": "gelu"}})
    with pytest.raises(ValueError, match="quick_gelu"):
        ClipShape.from_hf_config({"text_config": {"hidden_act": "gelu"}})


def test_from_hf_config_matches_transformers_config_object():
    transformers = pytest.importorskip("transformers")
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipShape, VIT_B32

    assert ClipShape.from_hf_config(transformers.CLIPConfig()) == VIT_B32
    cfg = transformers.CLIPConfig()
    cfg.text_config.max_position_embeddings = 248                        # the reference's override, utils.py:41-42
    assert ClipShape.from_hf_config(cfg).t_ctx == 248


@pytest.mark.parametrize("dtype", ["float32", "float16", "bfloat16"])
def test_safetensors_of_every_dtype_widen_to_f32(tmp_path, dtype):
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import iter_safetensors_f32, safetensors_shapes
    from oracle import clip_oracle as co
    from ckpt_fixture import tiny_longclip_shape, write_checkpoint

    vocab, _ = _vocab()
    shape = tiny_longclip_shape(len(vocab), vocab["<|endoftext|>"])
    W = co.init_weights(shape, seed=3)
    write_checkpoint(str(tmp_path), shape, W, dtype)
    path = os.path.join(str(tmp_path), "model.safetensors")
    shapes = safetensors_shapes(path)
    assert shapes["text_model.embeddings.position_embedding.weight"] == (248, shape.t_hidden)
    got = dict(iter_safetensors_f32(path))
    assert set(W) <= set(got)
    tol = {"float32": 0.0, "float16": 2 ** -11, "bfloat16": 2 ** -8}[dtype]
    for k, v in W.items():
        assert got[k].dtype == np.float32 and got[k].shape == v.shape
        assert np.abs(got[k] - v).max() <= tol * max(np.abs(v).max(), 1e-30) + 1e-7, k


def test_processor_from_checkpoint_directory_tokenizes_to_248(tmp_path):
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipShape
    from mmiss_amd.preprocess import ClipProcessor
    from oracle import clip_oracle as co
    from ckpt_fixture import tiny_longclip_shape, write_checkpoint

    vocab, _ = _vocab()
    shape = tiny_longclip_shape(len(vocab), vocab["<|endoftext|>"])
    write_checkpoint(str(tmp_path), shape, co.init_weights(shape, seed=3))
    with open(os.path.join(str(tmp_path), "config.json")) as f:
        cfg = json.load(f)
    assert "max_position_embeddings" not in cfg["text_config"]            # the fixture really omits defaults
    s = ClipShape.from_hf_config(cfg)
    assert s.t_ctx == 77 and s.v_patch == 14 and s.ln_eps == 1e-5          # 77 until the weights' table says otherwise
    proc = ClipProcessor.from_directory(str(tmp_path), ClipShape.from_any(shape), max_length=248)
    ids = proc.tokenize(["red drill"])
    assert ids.shape == (1, 248) and ids[0, 0] == vocab["<|startoftext|>"] and ids[0, 3] == vocab["<|endoftext|>"]

"""A synthetic HF-layout CLIP checkpoint directory (config.json, model.safetensors, vocab.json, merges.txt) — what
`CLIPModel.save_pretrained` + `CLIPProcessor.save_pretrained` leave on disk, at a tiny LongCLIP-like geometry. The
reference loads such a directory by hub name (backend/app/utils.py:41-45); offline it must be local."""
import json
import os

import numpy as np


def tiny_longclip_shape(vocab_size: int, eos_id: int):
    import dataclasses
    from oracle import clip_oracle as co

    # ViT-L/14-style: patch 14, an odd token count, and the 248-token text table of the reference (utils.py:17)
    return dataclasses.replace(co.TINY, v_patch=14, v_image=56, t_ctx=248, t_vocab=vocab_size, eos_token_id=eos_id)


def write_checkpoint(path: str, shape, W, dtype: str = "float32", stock_ctx_in_config: bool = True) -> None:
    """dtype: float32 | float16 | bfloat16 (the LongCLIP hub checkpoint is f16/f32; bf16 exports exist too).
    config.json omits every key that equals the HF default, as `to_diff_dict` does, and (stock_ctx_in_config) keeps
    max_position_embeddings at 77 although the table has shape.t_ctx rows — the reference overrides it in code."""
    import torch
    from safetensors.torch import save_file
    from test_tokenizer_preprocess_cpu import _synthetic_vocab

    os.makedirs(path, exist_ok=True)
    tdt = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}[dtype]
    tensors = {k: torch.from_numpy(np.ascontiguousarray(v)).to(tdt) for k, v in W.items()}
    tensors["logit_scale"] = torch.tensor(2.6592).to(tdt)                       # unused on this path, present in real files
    tensors["text_model.embeddings.position_ids"] = torch.arange(shape.t_ctx).unsqueeze(0)  # older exports carry it
    save_file(tensors, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    cfg = {
        "architectures": ["CLIPModel"], "model_type": "clip", "projection_dim": shape.proj_dim, "torch_dtype": dtype,
        "text_config": {"hidden_size": shape.t_hidden, "intermediate_size": shape.t_mlp, "num_attention_heads": shape.t_heads,
                        "num_hidden_layers": shape.t_layers, "vocab_size": shape.t_vocab, "eos_token_id": shape.eos_token_id,
                        "model_type": "clip_text_model"},          # hidden_act / layer_norm_eps / max_position_embeddings omitted
        "vision_config": {"hidden_size": shape.v_hidden, "intermediate_size": shape.v_mlp, "num_attention_heads": shape.v_heads,
                          "num_hidden_layers": shape.v_layers, "patch_size": shape.v_patch, "image_size": shape.v_image,
                          "model_type": "clip_vision_model"},
    }
    if not stock_ctx_in_config:
        cfg["text_config"]["max_position_embeddings"] = shape.t_ctx
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    vocab, merges = _synthetic_vocab()
    with open(os.path.join(path, "vocab.json"), "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)
    with open(os.path.join(path, "merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(merges) + "\n")

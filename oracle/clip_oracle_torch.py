"""CPU restatement of the same CLIP arithmetic as oracle/clip_oracle.py, in PyTorch-CPU float32 — the framework the
reference itself runs on (backend/app/utils.py:41-45,76-79,88-99: torch CPU tensors, fp32, model never moved to a device).

TEST INFRASTRUCTURE ONLY: used by bench.py's `cpu_baseline` leg ("the build's own PyTorch-CPU fp32
restatement ... bs = 1 (the reference's actual behaviour) and bs = 32 / 256, one thread and all cores") and pinned against
the numpy oracle (which is pinned against transformers.CLIPModel) in tests/test_oracle_torch_cpu.py.
Follows HF:modeling_clip.py:138-218 (embeddings), :259-383 (attention, MLP, layer), :613-656 / :513-586 (towers),
:674-675 (projections), HF:activations.py:117-123 (QuickGELU).
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def to_torch(W: Dict[str, np.ndarray]):
    import torch

    return {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in W.items()}


def _layer(x, W, p, heads, eps, causal):
    import torch
    import torch.nn.functional as F

    B, T, d = x.shape
    hd = d // heads
    h = F.layer_norm(x, (d,), W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], eps)
    a = p + "self_attn."
    q = F.linear(h, W[a + "q_proj.weight"], W[a + "q_proj.bias"]).view(B, T, heads, hd).transpose(1, 2)
    k = F.linear(h, W[a + "k_proj.weight"], W[a + "k_proj.bias"]).view(B, T, heads, hd).transpose(1, 2)
    v = F.linear(h, W[a + "v_proj.weight"], W[a + "v_proj.bias"]).view(B, T, heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    if causal:
        s = s + torch.full((T, T), float("-inf")).triu(1)
    ctx = (torch.softmax(s, dim=-1, dtype=torch.float32) @ v).transpose(1, 2).reshape(B, T, d)
    x = x + F.linear(ctx, W[a + "out_proj.weight"], W[a + "out_proj.bias"])
    h = F.layer_norm(x, (d,), W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], eps)
    h = F.linear(h, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"])
    h = h * torch.sigmoid(1.702 * h)
    return x + F.linear(h, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"])


def embed_images(pixels, W, shape):
    """pixels float32 [B,3,S,S] (numpy or torch) + to_torch(weights) -> unit rows float32 [B, proj] (torch)."""
    import torch
    import torch.nn.functional as F

    s = shape
    with torch.no_grad():
        px = torch.as_tensor(pixels, dtype=torch.float32)
        B = px.shape[0]
        pe = F.conv2d(px, W["vision_model.embeddings.patch_embedding.weight"], stride=s.v_patch).flatten(2).transpose(1, 2)
        cls = W["vision_model.embeddings.class_embedding"].expand(B, 1, -1)
        x = torch.cat([cls, pe], dim=1) + W["vision_model.embeddings.position_embedding.weight"]
        x = F.layer_norm(x, (s.v_hidden,), W["vision_model.pre_layrnorm.weight"], W["vision_model.pre_layrnorm.bias"], s.ln_eps)
        for i in range(s.v_layers):
            x = _layer(x, W, f"vision_model.encoder.layers.{i}.", s.v_heads, s.ln_eps, False)
        pooled = F.layer_norm(x[:, 0], (s.v_hidden,), W["vision_model.post_layernorm.weight"], W["vision_model.post_layernorm.bias"], s.ln_eps)
        f = F.linear(pooled, W["visual_projection.weight"])
        return f / f.norm(dim=1, keepdim=True)       # backend/app/utils.py:78


def embed_texts(ids, W, shape):
    import torch
    import torch.nn.functional as F

    s = shape
    with torch.no_grad():
        t = torch.as_tensor(np.asarray(ids)).long()
        B, T = t.shape
        x = W["text_model.embeddings.token_embedding.weight"][t] + W["text_model.embeddings.position_embedding.weight"][:T]
        for i in range(s.t_layers):
            x = _layer(x, W, f"text_model.encoder.layers.{i}.", s.t_heads, s.ln_eps, True)
        pos = t.argmax(dim=-1) if s.eos_token_id == 2 else (t == s.eos_token_id).int().argmax(dim=-1)
        pooled = F.layer_norm(x[torch.arange(B), pos], (s.t_hidden,), W["text_model.final_layer_norm.weight"],
                              W["text_model.final_layer_norm.bias"], s.ln_eps)
        f = F.linear(pooled, W["text_projection.weight"])
        return f / f.norm(dim=1, keepdim=True)       # backend/app/utils.py:98

#!/usr/bin/env python3
"""CPU emulation (torch, fp32 arithmetic on dequantised values) of the ViT-L/14 vision tower with its four projections on
block-scaled fp8, in two forms: (A) the shipped one — LayerNorm output quantised to MXFP8; (B) LayerNorm FOLDED into the GEMM —
the raw bf16 residual rows quantised to MXFP8, gamma folded into the fp8 weights, y = rstd (acc - mean c) + b'. Prints
1 - cos of the embeddings against the fp32 oracle. Decides whether (B) — which removes 46 of 47 LayerNorm launches per
encode — holds the 1e-3 bar before anything is built."""
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import clip_oracle as co

torch.set_num_threads(8)


def bf16(x):
    return x.to(torch.bfloat16).float()


def mx_quant(x):
    """rows x K -> dequantised MXFP8 (one E8M0 scale per 32 columns, e4m3 codes): gemm_fp8.h mx_scale_of"""
    r, k = x.shape
    b = x.reshape(r, k // 32, 32)
    amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30) / 448.0
    e = torch.ceil(torch.log2(amax)).clamp(-126, 127)
    s = torch.exp2(e)
    q = (b / s).to(torch.float8_e4m3fn).float()
    return (q * s).reshape(r, k)


def w_quant(w):
    """[N, K] -> dequantised e4m3 with one f32 scale per output channel"""
    s = w.abs().amax(dim=1, keepdim=True).clamp_min(1e-30) / 448.0
    return (w / s).to(torch.float8_e4m3fn).float() * s


def ln_stats(x, eps):
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return mean, 1.0 / torch.sqrt(var + eps)


def tower(px, W, s, mode):
    T = lambda k: torch.from_numpy(W[k])
    B = px.shape[0]
    pw = T("vision_model.embeddings.patch_embedding.weight").reshape(s.v_hidden, -1)
    patches = torch.from_numpy(co.patchify(px, s.v_patch)) @ pw.T
    cls = T("vision_model.embeddings.class_embedding").expand(B, 1, s.v_hidden)
    x = torch.cat([cls, patches], dim=1) + T("vision_model.embeddings.position_embedding.weight")
    m, r = ln_stats(x, s.ln_eps)
    x = (x - m) * r * T("vision_model.pre_layrnorm.weight") + T("vision_model.pre_layrnorm.bias")
    d, H = s.v_hidden, s.v_heads
    x = x.reshape(-1, d)
    if mode != "fp32":
        x = bf16(x)

    def proj(xin, g, b_ln, w, bias):
        """LayerNorm(xin) @ w^T + bias in the tower's arithmetic"""
        m, r = ln_stats(xin, s.ln_eps)
        if mode == "fp32":
            return ((xin - m) * r * g + b_ln) @ w.T + bias
        if mode == "A":
            return mx_quant((xin - m) * r * g + b_ln) @ w_quant(w).T + bias
        wq = w_quant(w * g)                      # gamma folded, then quantised
        c = wq.sum(dim=1)                        # of the QUANTISED weights: what the matrix cores multiply
        bp = bias + w @ b_ln                     # (f32, exact weights: a constant vector)
        return r * (mx_quant(xin) @ wq.T - m * c) + bp

    def plain(xin, w, bias):
        if mode == "fp32":
            return xin @ w.T + bias
        return mx_quant(xin) @ w_quant(w).T + bias

    for i in range(s.v_layers):
        p = f"vision_model.encoder.layers.{i}."
        wqkv = torch.cat([T(p + f"self_attn.{n}.weight") for n in ("q_proj", "k_proj", "v_proj")])
        bqkv = torch.cat([T(p + f"self_attn.{n}.bias") for n in ("q_proj", "k_proj", "v_proj")])
        qkv = proj(x, T(p + "layer_norm1.weight"), T(p + "layer_norm1.bias"), wqkv, bqkv)
        if mode != "fp32":
            qkv = bf16(qkv)
        q, k, v = [t.reshape(B, -1, H, 64).transpose(1, 2) for t in qkv.reshape(B, -1, 3 * d).split(d, dim=-1)]
        pr = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1)
        ctx = (pr @ v).transpose(1, 2).reshape(-1, d)
        x = x + plain(ctx, T(p + "self_attn.out_proj.weight"), T(p + "self_attn.out_proj.bias"))
        if mode != "fp32":
            x = bf16(x)
        h = proj(x, T(p + "layer_norm2.weight"), T(p + "layer_norm2.bias"), T(p + "mlp.fc1.weight"), T(p + "mlp.fc1.bias"))
        h = h * torch.sigmoid(1.702 * h)
        x = x + plain(h, T(p + "mlp.fc2.weight"), T(p + "mlp.fc2.bias"))
        if mode != "fp32":
            x = bf16(x)
    x0 = x.reshape(B, -1, d)[:, 0]
    m, r = ln_stats(x0, s.ln_eps)
    pooled = (x0 - m) * r * T("vision_model.post_layernorm.weight") + T("vision_model.post_layernorm.bias")
    y = pooled @ T("visual_projection.weight").T
    return torch.nn.functional.normalize(y, dim=-1)


def main():
    s = dataclasses.replace(co.LONGCLIP_L14, t_layers=1, t_vocab=1000, eos_token_id=999)
    for name, seed, outl in (("seeded Gaussian weights", 0, False), ("seed 71 + outlier channels (+300 / -180)", 71, True), ("seed 5", 5, False)):
        W = co.init_weights(s, seed=seed)
        if outl:
            pos = W["vision_model.embeddings.position_embedding.weight"].copy()
            pos[:, 31] += 300.0
            pos[:, 700] -= 180.0
            W["vision_model.embeddings.position_embedding.weight"] = pos
        rng = np.random.Generator(np.random.Philox(72))
        px = rng.standard_normal((4, 3, 224, 224), dtype=np.float32)
        with torch.no_grad():
            ref = tower(px, W, s, "fp32")
            a = tower(px, W, s, "A")
            b = tower(px, W, s, "B")
        ca = (1 - (a * ref).sum(-1)).max().item()
        cb = (1 - (b * ref).sum(-1)).max().item()
        print(f"{name}: 1 - cos vs fp32: (A) LayerNorm then MXFP8 {ca:.2e}   (B) folded, raw rows as MXFP8 {cb:.2e}", flush=True)


if __name__ == "__main__":
    main()

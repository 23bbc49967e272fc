"""CPU restatement (numpy, float32) of the CLIP arithmetic on the reference's embedding path.

TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is imported by the product package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker.

What it restates (reference paths relative to /root/reference; "HF:" =
transformers/models/clip/, the third-party package the reference's arithmetic lives in — the
reference pins it only as `transformers>=4.30.0`, requirements.txt:2; this container has 5.15.0):

  generate_clip_embedding, image branch  backend/app/utils.py:73-80
      processor(images=...)              HF:image_processing_clip.py:23-34        -> preprocess_image
      model.get_image_features           HF:modeling_clip.py:719-753,613-656      -> image_features
      / norm(dim=1, keepdim=True)        backend/app/utils.py:78                   -> l2_normalize
  generate_clip_embedding, text branch   backend/app/utils.py:83-100
      model.get_text_features            HF:modeling_clip.py:683-715,513-586      -> text_features
      / norm                             backend/app/utils.py:98
  search_multimodal's blend              backend/app/main.py:852-860               -> blend_reference

Pinning: the reference has no tests or golden vectors for this path, and its own
modules cannot be imported here (chromadb / rembg / imagehash are absent: ordinary
ModuleNotFoundError). The oracle is therefore pinned against the third-party implementation the
reference calls — transformers.CLIPModel run in THIS container with seeded weights, following the
reference's call sequence — by tools/make_goldens.py (fixtures in tests/golden/) and re-checked by
tests/test_oracle_vs_hf.py whenever transformers is importable.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, Optional

import numpy as np

F32 = np.float32

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=F32)  # transformers/utils/constants.py:5
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=F32)  # transformers/utils/constants.py:6


@dataclass
class ClipShape:
    """Shape of both towers. Defaults = HF CLIPConfig() = ViT-B/32 (HF:configuration_clip.py:47-54,97-105,160)."""

    v_hidden: int = 768
    v_layers: int = 12
    v_heads: int = 12
    v_mlp: int = 3072
    v_patch: int = 32
    v_image: int = 224
    t_hidden: int = 512
    t_layers: int = 12
    t_heads: int = 8
    t_mlp: int = 2048
    t_vocab: int = 49408
    t_ctx: int = 77
    proj_dim: int = 512
    eos_token_id: int = 49407
    ln_eps: float = 1e-5

    @property
    def v_tokens(self) -> int:
        return (self.v_image // self.v_patch) ** 2 + 1

    def to_dict(self):
        return asdict(self)


VIT_B32 = ClipShape()
# ViT-L/14 towers with the reference HEAD's 248-token text table (backend/app/utils.py:16-17,41-42)
LONGCLIP_L14 = ClipShape(v_hidden=1024, v_layers=24, v_heads=16, v_mlp=4096, v_patch=14, v_image=224,
                         t_hidden=768, t_layers=12, t_heads=12, t_mlp=3072, t_ctx=248, proj_dim=768)
# ... and the same geometry two layers deep (patch 14 / 257 tokens / width 1024 / 16 heads; 248-token text tower of width
# 768): what tests/golden/clip_l14_2layer.npz pins against transformers (tools/make_goldens.py)
LONGCLIP_L14_2L = ClipShape(v_hidden=1024, v_layers=2, v_heads=16, v_mlp=4096, v_patch=14, v_image=224,
                            t_hidden=768, t_layers=2, t_heads=12, t_mlp=3072, t_ctx=248, proj_dim=768)
# small shape with the same structure (head_dim 64, dims multiples of 128) for fast parity tests
TINY = ClipShape(v_hidden=128, v_layers=2, v_heads=2, v_mlp=256, v_patch=32, v_image=64,
                 t_hidden=128, t_layers=2, t_heads=2, t_mlp=256, t_vocab=1000, t_ctx=16, proj_dim=128,
                 eos_token_id=999)


# ------------------------------------------------------------------------------------------------ weights
def init_weights(shape: ClipShape, seed: int = 0) -> Dict[str, np.ndarray]:
    """The build's own seeded weight generator, HF state_dict key layout (SURVEY.md §8 a-W).

    Matrix stds follow HF's init table (HF:modeling_clip.py:404-437); biases and LayerNorm affine
    parameters are made non-trivial (HF zero/one-initialises them, which would hide bias bugs).
    Real checkpoints load through the same keys.
    """
    rng = np.random.Generator(np.random.Philox(seed))
    w: Dict[str, np.ndarray] = {}

    def normal(name, shp, std):
        w[name] = (rng.standard_normal(shp, dtype=np.float32) * np.float32(std)).astype(F32)

    def ln(prefix, d):
        w[prefix + ".weight"] = (1.0 + 0.1 * rng.standard_normal(d, dtype=np.float32)).astype(F32)
        w[prefix + ".bias"] = (0.1 * rng.standard_normal(d, dtype=np.float32)).astype(F32)

    def tower(prefix, d, layers, mlp):
        in_std = d ** -0.5 * (2 * layers) ** -0.5
        out_std = d ** -0.5
        fc_std = (2 * d) ** -0.5
        for i in range(layers):
            p = f"{prefix}.encoder.layers.{i}."
            for nm in ("q_proj", "k_proj", "v_proj"):
                normal(p + f"self_attn.{nm}.weight", (d, d), in_std)
                normal(p + f"self_attn.{nm}.bias", (d,), 0.02)
            normal(p + "self_attn.out_proj.weight", (d, d), out_std)
            normal(p + "self_attn.out_proj.bias", (d,), 0.02)
            ln(p + "layer_norm1", d)
            ln(p + "layer_norm2", d)
            normal(p + "mlp.fc1.weight", (mlp, d), fc_std)
            normal(p + "mlp.fc1.bias", (mlp,), 0.02)
            normal(p + "mlp.fc2.weight", (d, mlp), in_std)
            normal(p + "mlp.fc2.bias", (d,), 0.02)

    s = shape
    normal("vision_model.embeddings.class_embedding", (s.v_hidden,), s.v_hidden ** -0.5)
    normal("vision_model.embeddings.patch_embedding.weight", (s.v_hidden, 3, s.v_patch, s.v_patch), 0.02)
    normal("vision_model.embeddings.position_embedding.weight", (s.v_tokens, s.v_hidden), 0.02)
    ln("vision_model.pre_layrnorm", s.v_hidden)  # (sic) HF's key
    tower("vision_model", s.v_hidden, s.v_layers, s.v_mlp)
    ln("vision_model.post_layernorm", s.v_hidden)
    normal("visual_projection.weight", (s.proj_dim, s.v_hidden), s.v_hidden ** -0.5)
    normal("text_model.embeddings.token_embedding.weight", (s.t_vocab, s.t_hidden), 0.02)
    normal("text_model.embeddings.position_embedding.weight", (s.t_ctx, s.t_hidden), 0.02)
    tower("text_model", s.t_hidden, s.t_layers, s.t_mlp)
    ln("text_model.final_layer_norm", s.t_hidden)
    normal("text_projection.weight", (s.proj_dim, s.t_hidden), s.t_hidden ** -0.5)
    return w


# ------------------------------------------------------------------------------------------------ ops
def layer_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    """F.layer_norm over the last dim, biased variance, eps inside the sqrt (HF:modeling_clip.py:358-360)."""
    x = x.astype(F32)
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * g + b).astype(F32)


def quick_gelu(x: np.ndarray) -> np.ndarray:
    """x * sigmoid(1.702 x)  (HF:activations.py:117-123; CLIP's hidden_act, configuration_clip.py:54,105)."""
    return (x / (F32(1.0) + np.exp(-F32(1.702) * x))).astype(F32)


def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    y = x @ w.T
    if b is not None:
        y = y + b
    return y.astype(F32)


def attention(x: np.ndarray, W: Dict[str, np.ndarray], p: str, heads: int, causal: bool) -> np.ndarray:
    """CLIPAttention (HF:modeling_clip.py:280-335) with eager_attention_forward (:259-277):
    softmax(q k^T * head_dim^-0.5 + mask) v, softmax in fp32, then out_proj."""
    B, T, d = x.shape
    hd = d // heads
    q = linear(x, W[p + "q_proj.weight"], W[p + "q_proj.bias"]).reshape(B, T, heads, hd).transpose(0, 2, 1, 3)
    k = linear(x, W[p + "k_proj.weight"], W[p + "k_proj.bias"]).reshape(B, T, heads, hd).transpose(0, 2, 1, 3)
    v = linear(x, W[p + "v_proj.weight"], W[p + "v_proj.bias"]).reshape(B, T, heads, hd).transpose(0, 2, 1, 3)
    s = (q @ k.transpose(0, 1, 3, 2)) * F32(hd ** -0.5)
    if causal:  # HF:modeling_clip.py:543-548 — additive -inf above the diagonal
        s = s + np.triu(np.full((T, T), -np.inf, dtype=F32), k=1)
    s = s - s.max(axis=-1, keepdims=True)
    e = np.exp(s)
    pr = (e / e.sum(axis=-1, keepdims=True)).astype(F32)
    ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(B, T, d)
    return linear(ctx, W[p + "out_proj.weight"], W[p + "out_proj.bias"])


def encoder_layer(x, W, p, heads, eps, causal):
    """CLIPEncoderLayer (HF:modeling_clip.py:353-383): pre-LN residual block."""
    h = layer_norm(x, W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], eps)
    x = x + attention(h, W, p + "self_attn.", heads, causal)
    h = layer_norm(x, W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], eps)
    h = quick_gelu(linear(h, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"]))
    return (x + linear(h, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"])).astype(F32)


def patchify(pixels: np.ndarray, patch: int) -> np.ndarray:
    """[B,3,S,S] -> [B, G*G, 3*P*P] with k = c*P*P + ky*P + kx: the im2col of a stride-P, kernel-P conv."""
    B, C, S, _ = pixels.shape
    G = S // patch
    x = pixels.reshape(B, C, G, patch, G, patch).transpose(0, 2, 4, 1, 3, 5)
    return np.ascontiguousarray(x).reshape(B, G * G, C * patch * patch)


def image_features(pixels: np.ndarray, W: Dict[str, np.ndarray], shape: ClipShape = VIT_B32, taps: Optional[dict] = None):
    """CLIPModel.get_image_features(...).pooler_output (HF:modeling_clip.py:719-753 -> :613-656):
    patch conv (no bias) -> [CLS; patches] + pos -> pre_layrnorm -> L layers -> token 0 ->
    post_layernorm -> visual_projection (no bias)."""
    s = shape
    pixels = np.asarray(pixels, dtype=F32)
    B = pixels.shape[0]
    pw = W["vision_model.embeddings.patch_embedding.weight"].reshape(s.v_hidden, -1)
    patches = patchify(pixels, s.v_patch) @ pw.T  # HF:modeling_clip.py:209-211
    cls = np.broadcast_to(W["vision_model.embeddings.class_embedding"], (B, 1, s.v_hidden))
    x = np.concatenate([cls, patches], axis=1) + W["vision_model.embeddings.position_embedding.weight"]
    x = layer_norm(x, W["vision_model.pre_layrnorm.weight"], W["vision_model.pre_layrnorm.bias"], s.ln_eps)
    if taps is not None:
        taps[0] = x.copy()
    for i in range(s.v_layers):
        x = encoder_layer(x, W, f"vision_model.encoder.layers.{i}.", s.v_heads, s.ln_eps, causal=False)
        if taps is not None:
            taps[i + 1] = x.copy()
    pooled = layer_norm(x[:, 0, :], W["vision_model.post_layernorm.weight"], W["vision_model.post_layernorm.bias"], s.ln_eps)
    if taps is not None:
        taps["pooled"] = pooled.copy()
    return linear(pooled, W["visual_projection.weight"])


def eos_positions(ids: np.ndarray, eos_token_id: int) -> np.ndarray:
    """HF:modeling_clip.py:561-581: legacy eos_token_id == 2 -> argmax(ids); else first position equal to eos."""
    ids = np.asarray(ids)
    if eos_token_id == 2:
        return ids.argmax(axis=-1)
    return (ids == eos_token_id).astype(np.int32).argmax(axis=-1)


def text_features(ids: np.ndarray, W: Dict[str, np.ndarray], shape: ClipShape = VIT_B32, taps: Optional[dict] = None):
    """CLIPModel.get_text_features(...).pooler_output (HF:modeling_clip.py:683-715 -> :513-586).
    The padding mask the reference passes (processor(..., padding="max_length"), utils.py:88) only
    removes keys AFTER the pooled EOS row's causal horizon, so it cannot change the pooled row; it is
    therefore not modelled (tests/test_oracle_vs_hf.py checks this against HF with the mask)."""
    s = shape
    ids = np.asarray(ids)
    B, T = ids.shape
    x = W["text_model.embeddings.token_embedding.weight"][ids] + W["text_model.embeddings.position_embedding.weight"][:T]
    x = x.astype(F32)
    if taps is not None:
        taps[0] = x.copy()
    for i in range(s.t_layers):
        x = encoder_layer(x, W, f"text_model.encoder.layers.{i}.", s.t_heads, s.ln_eps, causal=True)
        if taps is not None:
            taps[i + 1] = x.copy()
    pos = eos_positions(ids, s.eos_token_id)
    pooled = layer_norm(x[np.arange(B), pos], W["text_model.final_layer_norm.weight"], W["text_model.final_layer_norm.bias"], s.ln_eps)
    if taps is not None:
        taps["pooled"] = pooled.copy()
    return linear(pooled, W["text_projection.weight"])


def l2_normalize(x: np.ndarray) -> np.ndarray:
    """x / x.norm(dim=1, keepdim=True), fp32, no epsilon (backend/app/utils.py:78,98)."""
    x = np.asarray(x, dtype=F32)
    return (x / np.sqrt((x * x).sum(axis=1, keepdims=True, dtype=F32))).astype(F32)


def embed_images(pixels, W, shape=VIT_B32):
    return l2_normalize(image_features(pixels, W, shape))


def embed_texts(ids, W, shape=VIT_B32):
    return l2_normalize(text_features(ids, W, shape))


# ------------------------------------------------------------------------------------------------ preprocessing
def preprocess_image(img, size: int = 224) -> np.ndarray:
    """CLIPImageProcessor (HF:image_processing_clip.py:23-34), PIL path: convert RGB -> resize so the
    SHORTEST edge is `size` (bicubic) -> centre crop size x size -> * 1/255 -> (x - mean) / std.
    Returns float32 [3, size, size]."""
    from PIL import Image

    img = img.convert("RGB")
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    img = img.resize((new_w, new_h), resample=Image.BICUBIC)
    left, top = (new_w - size) // 2, (new_h - size) // 2
    img = img.crop((left, top, left + size, top + size))
    a = np.asarray(img, dtype=np.uint8).astype(F32) * F32(1.0 / 255.0)
    a = (a - CLIP_MEAN) / CLIP_STD
    return np.ascontiguousarray(a.transpose(2, 0, 1)).astype(F32)


def crop_u8(img, size: int = 224) -> np.ndarray:
    """The same resize + centre crop, stopped before the float conversion: uint8 [size, size, 3]."""
    from PIL import Image

    img = img.convert("RGB")
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    img = img.resize((new_w, new_h), resample=Image.BICUBIC)
    left, top = (new_w - size) // 2, (new_h - size) // 2
    return np.asarray(img.crop((left, top, left + size, top + size)), dtype=np.uint8)


def normalize_u8(u8: np.ndarray) -> np.ndarray:
    """uint8 [B,S,S,3] -> float32 [B,3,S,S], rescale then normalise, the op order of the HF processor."""
    a = u8.astype(F32) * F32(1.0 / 255.0)
    a = (a - CLIP_MEAN) / CLIP_STD
    return np.ascontiguousarray(a.transpose(0, 3, 1, 2)).astype(F32)


# ------------------------------------------------------------------------------------------------ glue
def blend_reference(img: np.ndarray, txt: np.ndarray, weight_image: float) -> np.ndarray:
    """search_multimodal's combination exactly as numpy evaluates it in the reference
    (backend/app/main.py:852-860), vectorised over rows."""
    img = np.asarray(img, dtype=F32)
    txt = np.asarray(txt, dtype=F32)
    i_n = img / np.linalg.norm(img, axis=-1, keepdims=True)
    t_n = txt / np.linalg.norm(txt, axis=-1, keepdims=True)
    c = weight_image * i_n + (1 - weight_image) * t_n
    return (c / np.linalg.norm(c, axis=-1, keepdims=True)).astype(F32)


def synthetic_text_ids(n: int, ctx: int, vocab: int, eos: int, seed: int, bos: Optional[int] = None) -> np.ndarray:
    """BASELINE config 3 inputs: BOS, random body, EOS at a random position in [2, ctx-1], EOS padding
    (CLIP pads with the EOS id). Body ids stay below bos/eos so the first EOS is the pooled row."""
    rng = np.random.Generator(np.random.Philox(seed))
    bos = eos - 1 if bos is None else bos
    ids = np.full((n, ctx), eos, dtype=np.int32)
    ids[:, 0] = bos
    pos = rng.integers(2, ctx, size=n)
    body = rng.integers(0, min(bos, eos), size=(n, ctx), dtype=np.int32)
    for r in range(n):
        ids[r, 1:pos[r]] = body[r, 1:pos[r]]
    return ids

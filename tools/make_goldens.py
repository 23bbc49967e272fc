#!/usr/bin/env python3
"""Generate tests/golden/*.npz — the pins of the oracle.

Run in the BUILD container (needs `transformers`, and /root/reference for the drill-set images):

    python tools/make_goldens.py

The reference cannot be imported here (backend/app/utils.py imports chromadb and rembg at module top:
ModuleNotFoundError) and holds no tests or golden vectors of its own, so the vectors are produced by the
third-party code the reference calls, following the reference's call sequence:

    inputs  = processor(images=image, return_tensors="pt")           backend/app/utils.py:76
    feats   = model.get_image_features(**inputs)                      backend/app/utils.py:77   (.pooler_output in
                                                                      transformers 5.x)
    emb     = feats / feats.norm(dim=1, keepdim=True)                 backend/app/utils.py:78
    (text: utils.py:88,97-98 with padding="max_length" -> attention_mask passed to get_text_features)

with `model = transformers.CLIPModel(CLIPConfig(...))` holding the build's seeded weights
(oracle.clip_oracle.init_weights — regenerated from the seed by the tests; 605 MB of weights cannot be
committed). No pretrained weights or CLIP vocabulary exist offline, so 'red drill' is represented by
explicit surrogate ids.

Only DATA is written (inputs + expected outputs), never source text.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

import torch  # noqa: E402
from transformers import CLIPConfig, CLIPImageProcessor, CLIPModel  # noqa: E402

from oracle import clip_oracle as co  # noqa: E402


def hf_model(shape: co.ClipShape, W):
    cfg = CLIPConfig(
        text_config=dict(hidden_size=shape.t_hidden, intermediate_size=shape.t_mlp, num_hidden_layers=shape.t_layers,
                         num_attention_heads=shape.t_heads, vocab_size=shape.t_vocab,
                         max_position_embeddings=shape.t_ctx, eos_token_id=shape.eos_token_id,
                         bos_token_id=shape.eos_token_id - 1, pad_token_id=1, projection_dim=shape.proj_dim),
        vision_config=dict(hidden_size=shape.v_hidden, intermediate_size=shape.v_mlp, num_hidden_layers=shape.v_layers,
                           num_attention_heads=shape.v_heads, image_size=shape.v_image, patch_size=shape.v_patch,
                           projection_dim=shape.proj_dim),
        projection_dim=shape.proj_dim)
    m = CLIPModel(cfg).eval()
    sd = m.state_dict()
    m.load_state_dict({k: (torch.from_numpy(W[k]).reshape(v.shape) if k in W else v) for k, v in sd.items()})
    return m


@torch.no_grad()
def reference_embeddings(m, shape, pixels=None, ids=None):
    """The reference's generate_clip_embedding arithmetic (utils.py:76-79, 88-99) on HF objects."""
    out = {}
    if pixels is not None:
        feats = m.get_image_features(pixel_values=torch.from_numpy(pixels)).pooler_output
        out["image"] = (feats / feats.norm(dim=1, keepdim=True)).cpu().numpy()
        out["image_raw"] = feats.cpu().numpy()
    if ids is not None:
        t = torch.from_numpy(ids).long()
        eos = co.eos_positions(ids, shape.eos_token_id)
        mask = torch.zeros_like(t)
        for r in range(t.shape[0]):
            mask[r, : eos[r] + 1] = 1  # the tokenizer's attention_mask: 1 up to and including the first EOS
        feats = m.get_text_features(input_ids=t, attention_mask=mask).pooler_output
        out["text"] = (feats / feats.norm(dim=1, keepdim=True)).cpu().numpy()
    return out


@torch.no_grad()
def hf_vision_hidden_states(m, pixels):
    vo = m.vision_model(pixel_values=torch.from_numpy(pixels), output_hidden_states=True)
    return [h.cpu().numpy() for h in vo.hidden_states]


def l14_two_layers():
    """The reference checkpoint's GEOMETRY (backend/app/utils.py:16-17,41-45: zer0int/LongCLIP-GmP-ViT-L-14 — ViT-L/14
    vision tower, 248-row text position table, projection 768), two layers deep so that the file stays small and the
    float32 forward takes seconds: patch 14 -> 257 tokens, width 1024, 16 heads, mlp 4096; text width 768, 12 heads,
    248 positions. Embeddings + the vision hidden states behind pre-LN and behind layer 2 (first image only)."""
    s = co.LONGCLIP_L14_2L
    W = co.init_weights(s, seed=0)
    m = hf_model(s, W)
    rng = np.random.Generator(np.random.Philox(501))
    px = rng.standard_normal((4, 3, s.v_image, s.v_image), dtype=np.float32)
    ids = co.synthetic_text_ids(4, s.t_ctx, s.t_vocab, s.eos_token_id, seed=502, bos=49406)
    e = reference_embeddings(m, s, px, ids)
    hs = hf_vision_hidden_states(m, px[:1])   # (first image only: 1 MB per tap)
    np.savez_compressed(os.path.join(OUT, "clip_l14_2layer.npz"), weight_seed=0, pixel_seed=501, ids=ids,
                        image=e["image"], image_raw=e["image_raw"], text=e["text"],
                        vis_hidden_0=hs[0].astype(np.float32), vis_hidden_last=hs[-1].astype(np.float32))
    print("clip_l14_2layer.npz", os.path.getsize(os.path.join(OUT, "clip_l14_2layer.npz")) // 1024, "KiB; eos at",
          co.eos_positions(ids, s.eos_token_id).tolist())


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "l14":   # only the file added in round 5 (the others are unchanged)
        l14_two_layers()
        return
    l14_two_layers()

    # ---------------------------------------------------------------- tiny shape: embeddings + intermediates
    s = co.TINY
    W = co.init_weights(s, seed=0)
    m = hf_model(s, W)
    rng = np.random.Generator(np.random.Philox(101))
    px = rng.standard_normal((8, 3, s.v_image, s.v_image), dtype=np.float32)
    ids = co.synthetic_text_ids(8, s.t_ctx, s.t_vocab, s.eos_token_id, seed=102)
    e = reference_embeddings(m, s, px, ids)
    hs = hf_vision_hidden_states(m, px)  # hs[0] = after pre_layrnorm, hs[i] = after layer i
    np.savez_compressed(os.path.join(OUT, "clip_tiny.npz"), weight_seed=0, pixel_seed=101, ids=ids,
                        image=e["image"], image_raw=e["image_raw"], text=e["text"],
                        vis_hidden_0=hs[0], vis_hidden_1=hs[1], vis_hidden_last=hs[-1])

    # ---------------------------------------------------------------- ViT-B/32: embeddings
    s = co.VIT_B32
    W = co.init_weights(s, seed=0)
    m = hf_model(s, W)
    rng = np.random.Generator(np.random.Philox(201))
    px = rng.standard_normal((4, 3, 224, 224), dtype=np.float32)
    ids = co.synthetic_text_ids(4, 77, s.t_vocab, s.eos_token_id, seed=202, bos=49406)
    e = reference_embeddings(m, s, px, ids)
    np.savez_compressed(os.path.join(OUT, "clip_b32.npz"), weight_seed=0, pixel_seed=201, ids=ids,
                        image=e["image"], text=e["text"])

    # ---------------------------------------------------------------- drill set (BASELINE config 1)
    ref_images = "/root/reference/images"
    if os.path.isdir(ref_images):
        from PIL import Image

        names = sorted(os.listdir(ref_images))
        proc = CLIPImageProcessor()  # defaults = CLIP preprocessing, PIL backend
        crops, hf_px = [], []
        for n in names:
            img = Image.open(os.path.join(ref_images, n))
            if img.mode not in ("RGB", "L"):
                img = img.convert("RGB")  # backend/app/main.py:140-143
            hf_px.append(proc(images=img, return_tensors="np")["pixel_values"][0])
            crops.append(co.crop_u8(img))
        hf_px = np.stack(hf_px).astype(np.float32)
        crops = np.stack(crops)
        # surrogate token ids for the query 'red drill' (no vocabulary offline): BOS, two ids, EOS, padding
        q_ids = np.full((1, 77), 49407, dtype=np.int32)
        q_ids[0, :4] = [49406, 736, 16451, 49407]
        e = reference_embeddings(m, s, hf_px, q_ids)
        np.savez_compressed(os.path.join(OUT, "drill_set.npz"), names=np.array(names), crops_u8=crops,
                            hf_pixel_max_abs_diff=np.abs(co.normalize_u8(crops) - hf_px).max(),
                            image=e["image"], text=e["text"], query_ids=q_ids,
                            cosine=e["image"] @ e["image"].T, text_image_cosine=e["text"] @ e["image"].T)
        print("drill set:", names, "oracle preprocess vs HF processor max|diff| =",
              float(np.abs(co.normalize_u8(crops) - hf_px).max()))

    # ---------------------------------------------------------------- preprocessing on a synthetic odd-sized image
    from PIL import Image

    rng = np.random.Generator(np.random.Philox(301))
    arr = rng.integers(0, 256, size=(301, 517, 3), dtype=np.uint8)
    hf = CLIPImageProcessor()(images=Image.fromarray(arr), return_tensors="np")["pixel_values"][0]
    arr2 = rng.integers(0, 256, size=(640, 233, 3), dtype=np.uint8)
    hf2 = CLIPImageProcessor()(images=Image.fromarray(arr2), return_tensors="np")["pixel_values"][0]
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), wide_u8=arr, wide_pixels=hf.astype(np.float32),
                        tall_u8=arr2, tall_pixels=hf2.astype(np.float32))

    # ---------------------------------------------------------------- retrieval: independent float64 brute force
    rng = np.random.Generator(np.random.Philox(401))
    N, D = 4096, 512
    corpus = rng.standard_normal((N, D), dtype=np.float32)
    queries = rng.standard_normal((8, D), dtype=np.float32)
    c64 = corpus.astype(np.float64)
    q64 = queries.astype(np.float64)
    dist = 1.0 - (q64 @ c64.T) / (np.linalg.norm(q64, axis=1)[:, None] * np.linalg.norm(c64, axis=1)[None])
    order = np.argsort(dist, axis=1, kind="stable")[:, :10]
    np.savez_compressed(os.path.join(OUT, "retrieval.npz"), corpus_seed=401, N=N, D=D, queries=queries,
                        top10_ids=order.astype(np.int64), top10_dist=np.take_along_axis(dist, order, axis=1))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()

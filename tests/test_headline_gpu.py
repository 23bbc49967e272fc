"""The BASELINE.json configurations THEMSELVES under the oracle (VERDICT r1 "What's weak" #2): the code path bench.py
times — ViT-B/32 at batch 256, where run_layers switches to the folded-LayerNorm GEMMs with 160/192-row tiles — the
text tower at 256 x 77, the HF-generated golden vectors against GPU output, and the retrieval configurations at their
per-GPU sizes (Q = 1000 prompts vs 1M rows, Q = 1024 blended queries vs a 1.25M-row shard).

Reference call sites: backend/app/utils.py:76-79,97-98 (embed + normalise), backend/app/main.py:761-765 (query),
:852-860 (blend)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COS_TOL = 1e-3  # BASELINE.json north_star: "within 1e-3 cosine of the reference CPU path"


def _cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def _kernels_of(fn):
    """Run fn() with every libmmiss launch bracketed; returns (result, {kernel class: launches})."""
    from mmiss_amd import _lib

    _lib.prof_filter(None, 1)
    _lib.prof_reset()
    _lib.prof_enable(True)
    try:
        out = fn()
    finally:
        _lib.prof_enable(False)
    return out, {p["kernel"]: p["launches"] for p in _lib.prof_read()}


@pytest.fixture(scope="module")
def b32_256():
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    W = co.init_weights(co.VIT_B32, seed=0)
    enc = ClipEncoder(ClipShape.from_any(co.VIT_B32), max_batch_image=256, max_batch_text=256)  # bench.py's handle
    enc.load_state_dict(W)
    small = ClipEncoder(ClipShape.from_any(co.VIT_B32), max_batch_image=16, max_batch_text=16)
    small.load_state_dict(W)
    yield enc, small, W, co
    enc.close()
    small.close()


def test_b32_bs256_default_path_is_the_folded_one_and_matches_the_oracle(b32_256):
    """configs[1]: 256 images in one call = 12800 rows -> ln_mode 2 chosen automatically (12 partial statistics per row,
    BM 160/192 tiles with the fold epilogue, bf16(x) as the A operand). 24 of the 256 embeddings against the fp32
    oracle, all 256 against the bs-16 path (separate LayerNorm kernels, 128-row / split-K tiles)."""
    enc, small, W, co = b32_256
    s = co.VIT_B32
    rng = np.random.Generator(np.random.Philox(1234))
    px = rng.standard_normal((256, 3, 224, 224), dtype=np.float32)
    # rows with a large mean / std ratio: the single-pass variance and the bf16(x) - mean * c cancellation are the risk
    px[5] = px[5] * 0.05 + 4.0
    px[77] = px[77] * 3.0 - 2.5
    px[200] = np.abs(px[200]) * 2.0
    out, kern = _kernels_of(lambda: enc.encode_image(px))
    assert kern.get("gemm_bf16_lnfold_bias_p256", 0) == 12 and kern.get("gemm_bf16_lnfold_qgelu_p256", 0) == 11, kern
    assert "layernorm" in kern and kern["layernorm"] <= 3, kern   # pre-LN, pruned last layer's LN2, head: no per-layer LN pass
    sub = np.concatenate([[5, 77, 200], np.arange(0, 256, 13)])[:24]
    ref = co.embed_images(px[sub], W, s)
    d = 1 - _cos(out[sub], ref)
    assert d.max() < COS_TOL, d
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-5
    # the same images 16 at a time: <= 800 rows per call -> ln_mode 0
    parts, kern16 = _kernels_of(lambda: np.concatenate([small.encode_image(px[i:i + 16]) for i in range(0, 256, 16)]))
    assert "gemm_bf16_lnfold_bias" not in kern16 and "gemm_bf16_lnfold_bias_p256" not in kern16, kern16
    assert (1 - _cos(parts, out)).max() < 2e-4  # two bf16 roundings of the same fp32 function (f32 residual stream at 16 rows,
    # bf16 stream at 256: measured ~6e-5)
    again = enc.encode_image(px)
    np.testing.assert_array_equal(out, again)  # deterministic


def test_b32_bs256_outlier_hidden_channels(b32_256):
    """Real CLIP checkpoints carry residual channels in the hundreds. Two channels of the ViT-B/32 residual stream are
    pushed to +300 / -180 through the position table; the bs-256 folded path must still meet the tolerance."""
    from mmiss_amd.encoder import ClipEncoder, ClipShape

    _, _, W0, co = b32_256
    s = co.VIT_B32
    W = dict(W0)
    pos = W["vision_model.embeddings.position_embedding.weight"].copy()
    pos[:, 31] += 300.0
    pos[:, 500] -= 180.0
    W["vision_model.embeddings.position_embedding.weight"] = pos
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=256, max_batch_text=8)
    enc.load_state_dict(W)
    rng = np.random.Generator(np.random.Philox(4321))
    px = rng.standard_normal((256, 3, 224, 224), dtype=np.float32)
    out, kern = _kernels_of(lambda: enc.encode_image(px))
    assert kern.get("gemm_bf16_lnfold_bias_p256", 0) == 12, kern
    sub = np.arange(3, 256, 32)
    d = 1 - _cos(out[sub], co.embed_images(px[sub], W, s))
    assert d.max() < COS_TOL, d
    enc.close()


def test_b32_bs256_fp8_setting_on_the_persistent_kernels_holds_the_bar(b32_256):
    """Round 6: ViT-B/32 — the metric's own model — under set_precision("fp8") at batch 256: QKV, FC1 (K = 768:
    three K-tile pairs per tile) run on the persistent block-scaled fp8 GEMM, FC2 and the out-projection (A = the 50-key
    attention's MXFP8 output) on the fp8 tile kernel. 1 - cos against the fp32 oracle (24 images incl. three with a large mean / std ratio), against the
    transformers golden vectors (four images inside a batch of 256) and against the bf16 path, all at north_star's 1e-3; the
    setting stays an opt-in and `value` stays the bf16 step (include/mmiss.h)."""
    enc, small, W, co = b32_256
    s = co.VIT_B32
    rng = np.random.Generator(np.random.Philox(1234))
    px = rng.standard_normal((256, 3, 224, 224), dtype=np.float32)
    px[5] = px[5] * 0.05 + 4.0
    px[77] = px[77] * 3.0 - 2.5
    px[200] = np.abs(px[200]) * 2.0
    g = np.load(os.path.join(G, "clip_b32.npz"))
    gpx = np.random.Generator(np.random.Philox(int(g["pixel_seed"]))).standard_normal((4, 3, 224, 224), dtype=np.float32)
    big = np.random.Generator(np.random.Philox(9)).standard_normal((256, 3, 224, 224), dtype=np.float32)
    big[[0, 100, 200, 255]] = gpx
    ref16 = enc.encode_image(px)
    enc.set_precision("fp8")
    try:
        out, kern = _kernels_of(lambda: enc.encode_image(px))
        gold = enc.encode_image(big)[[0, 100, 200, 255]]
        again = enc.encode_image(px)
    finally:
        enc.set_precision("bf16")
    # 12 QKV + 11 FC1 on the persistent kernel (450 / 600 tiles of 256 x 256); 11 FC2 + 11 out-projections (150 such tiles: one short
    # round, measured slower there) on the BM x 128 tile kernel; the pruned last layer's three GEMMs on the 128-row bf16 kernels;
    # 11 attentions -> MXFP8
    assert kern.get("gemm_fp8_bias_p256", 0) == 12 and kern.get("gemm_fp8_qgelu_mx_p256", 0) == 11, kern
    assert kern.get("gemm_fp8_bias_resid16", 0) == 22 and kern.get("attention_mx", 0) == 11 and kern.get("attention", 0) == 1, kern
    assert set(k for k in kern if k.startswith("gemm_fp8")) == {"gemm_fp8_bias_p256", "gemm_fp8_qgelu_mx_p256", "gemm_fp8_bias_resid16"}, kern
    sub = np.concatenate([[5, 77, 200], np.arange(0, 256, 13)])[:24]
    d = 1 - _cos(out[sub], co.embed_images(px[sub], W, s))
    d16 = 1 - _cos(out, ref16)
    dg = 1 - _cos(gold, g["image"])
    print("ViT-B/32 bs 256, fp8 setting: 1 - cos vs fp32 oracle %.2e, vs the bf16 path %.2e, vs the transformers golden %.2e" % (d.max(), d16.max(), dg.max()))
    assert d.max() < COS_TOL, d
    assert dg.max() < COS_TOL, dg
    assert d16.max() < COS_TOL
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-5
    np.testing.assert_array_equal(out, again)  # deterministic


FP8_B32_OUTLIER_MEASURED_BOUND = 2e-3   # NOT a parity bar: see the test below


def test_b32_bs256_fp8_setting_with_outlier_hidden_channels_is_outside_the_tolerance(b32_256):
    """The +300 / -180 residual channels of test_b32_bs256_outlier_hidden_channels under set_precision("fp8"), MEASURED: 1.15e-3
    from the fp32 oracle — OUTSIDE north_star's 1e-3 (the bf16 default holds it: 5e-5, the test above). An e4m3 weight keeps 3
    mantissa bits at any magnitude; the weight column that meets a LayerNorm output of ~24 carries a rounding error as large as
    the whole signal of the 766 ordinary channels, in every QKV / FC1 of 12 layers, and ViT-B/32's width (768) averages it over
    fewer columns than ViT-L/14's (1024, 24 layers: 6.7e-4, inside the bar and asserted at it). No parity claim is made for
    this regime: include/mmiss.h says so, MMISS_PREC_FP8 stays an opt-in to be verified per checkpoint against the bf16 path,
    and `value` is the bf16 step. The bound below only catches a regression of the measured figure."""
    from mmiss_amd.encoder import ClipEncoder, ClipShape

    _, _, W0, co = b32_256
    s = co.VIT_B32
    W = dict(W0)
    pos = W["vision_model.embeddings.position_embedding.weight"].copy()
    pos[:, 31] += 300.0
    pos[:, 500] -= 180.0
    W["vision_model.embeddings.position_embedding.weight"] = pos
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=256, max_batch_text=8, precision="fp8")
    enc.load_state_dict(W)
    rng = np.random.Generator(np.random.Philox(4321))
    px = rng.standard_normal((256, 3, 224, 224), dtype=np.float32)
    out, kern = _kernels_of(lambda: enc.encode_image(px))
    assert kern.get("gemm_fp8_bias_p256", 0) == 12, kern
    sub = np.arange(3, 256, 32)
    ref = co.embed_images(px[sub], W, s)
    d = 1 - _cos(out[sub], ref)
    enc.set_precision("bf16")
    d16 = 1 - _cos(enc.encode_image(px)[sub], ref)
    enc.close()
    print("ViT-B/32 bs 256, outlier channels +300 / -180: 1 - cos vs fp32 oracle  fp8 setting %.2e (outside 1e-3: opt-in only), bf16 %.2e"
          % (d.max(), d16.max()))
    assert d16.max() < COS_TOL, d16
    assert d.max() < FP8_B32_OUTLIER_MEASURED_BOUND, d


def test_b32_text_tower_256x77_default_path(b32_256):
    """configs[2] shapes: 256 prompts x 77 tokens = 19712 rows -> folded path, causal attention, first-EOS pooling."""
    enc, small, W, co = b32_256
    s = co.VIT_B32
    ids = co.synthetic_text_ids(256, 77, s.t_vocab, s.eos_token_id, seed=2, bos=49406)
    ids[0, 76] = s.eos_token_id
    ids[0, 1:76] = np.minimum(ids[0, 1:76], s.eos_token_id - 2)   # one prompt fills the whole context (no trimming)
    out, kern = _kernels_of(lambda: enc.encode_text(ids))
    assert kern.get("gemm_bf16_lnfold_bias_p256", 0) == 12, kern
    sub = np.arange(0, 256, 16)
    d = 1 - _cos(out[sub], co.embed_texts(ids[sub], W, s))
    assert d.max() < COS_TOL, d
    parts = np.concatenate([small.encode_text(ids[i:i + 16], trim_padding=False) for i in range(0, 256, 16)])
    assert (1 - _cos(parts, out)).max() < 2e-4  # two bf16 roundings of the same fp32 function (measured ~6e-5)


def test_hf_goldens_b32_against_gpu_output(b32_256):
    """tests/golden/clip_b32.npz holds transformers.CLIPModel's own output (tools/make_goldens.py, the reference's call
    sequence utils.py:76-79,88-99): GPU embeddings against THOSE vectors, not against the oracle."""
    enc, small, W, co = b32_256
    g = np.load(os.path.join(G, "clip_b32.npz"))
    assert int(g["weight_seed"]) == 0
    px = np.random.Generator(np.random.Philox(int(g["pixel_seed"]))).standard_normal((4, 3, 224, 224), dtype=np.float32)
    for e in (enc, small):
        assert (1 - _cos(e.encode_image(px), g["image"])).max() < COS_TOL
        assert (1 - _cos(e.encode_text(g["ids"]), g["text"])).max() < COS_TOL
    # inside a full batch of 256 (folded path) the four golden images still meet the bar
    rng = np.random.Generator(np.random.Philox(9))
    big = rng.standard_normal((256, 3, 224, 224), dtype=np.float32)
    big[[0, 100, 200, 255]] = px
    out = enc.encode_image(big)
    assert (1 - _cos(out[[0, 100, 200, 255]], g["image"])).max() < COS_TOL


def test_hf_golden_l14_geometry_against_gpu_output():
    """tests/golden/clip_l14_2layer.npz: transformers' own output at the reference checkpoint's GEOMETRY (backend/app/utils.py:
    16-17,41-45 — ViT-L/14 vision tower: patch 14, 257 tokens, width 1024, 16 heads; 248-position text tower of width 768;
    projection 768), two layers deep. GPU embeddings against THOSE vectors — bf16 and the fp8 setting, four images in one
    call (1028 rows) and inside the config's batch of 128 (32 896 rows: the persistent GEMMs, the long attention, the bf16
    residual stream), the four 248-token prompts alone and inside a batch of 64."""
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    g = np.load(os.path.join(G, "clip_l14_2layer.npz"))
    s = co.LONGCLIP_L14_2L
    W = co.init_weights(s, int(g["weight_seed"]))
    px = np.random.Generator(np.random.Philox(int(g["pixel_seed"]))).standard_normal((4, 3, 224, 224), dtype=np.float32)
    rng = np.random.Generator(np.random.Philox(19))
    big = rng.standard_normal((128, 3, 224, 224), dtype=np.float32)
    where = [0, 41, 90, 127]
    big[where] = px
    ids = g["ids"]
    ids_big = co.synthetic_text_ids(64, s.t_ctx, s.t_vocab, s.eos_token_id, seed=77, bos=49406)
    twhere = [0, 21, 40, 63]
    ids_big[twhere] = ids
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=128, max_batch_text=64)
    enc.load_state_dict(W)
    res = {}
    for prec in ("bf16", "fp8"):
        enc.set_precision(prec)
        (small_i, k_small) = _kernels_of(lambda: enc.encode_image(px))
        (big_i, k_big) = _kernels_of(lambda: enc.encode_image(big))
        res[prec] = ((1 - _cos(small_i, g["image"])).max(), (1 - _cos(big_i[where], g["image"])).max(),
                     (1 - _cos(enc.encode_text(ids), g["text"])).max(),
                     (1 - _cos(enc.encode_text(ids_big)[twhere], g["text"])).max())
        if prec == "fp8":
            assert any(k.startswith("gemm_fp8") for k in k_big), k_big
    enc.close()
    print("L/14 geometry, 2 layers, vs the transformers golden: 1 - cos (4 images, in 128, 4 texts, in 64)",
          {k: ["%.2e" % x for x in v] for k, v in res.items()})
    for prec, v in res.items():
        assert max(v) < COS_TOL, (prec, v)


def test_l14_width_fp8_outlier_hidden_channels():
    """real CLIP-L checkpoints carry residual channels in the hundreds, and the fp8 setting had only been
    measured on Gaussian seeded weights (5.5e-4 of the 1e-3 bar). The +300 / -180 position-table channels of
    test_b32_bs256_outlier_hidden_channels on a 6-layer tower of the L/14 geometry (width 1024, 257 tokens), bf16 and
    set_precision("fp8"), at 16 images per call (4112 rows) and at the config's 128 (32 896 rows)."""
    import dataclasses
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    s = dataclasses.replace(co.LONGCLIP_L14, v_layers=6, t_layers=1, t_vocab=1000, eos_token_id=999)
    W = co.init_weights(s, seed=61)
    pos = W["vision_model.embeddings.position_embedding.weight"].copy()
    pos[:, 31] += 300.0
    pos[:, 700] -= 180.0
    W["vision_model.embeddings.position_embedding.weight"] = pos
    rng = np.random.Generator(np.random.Philox(62))
    px = rng.standard_normal((128, 3, 224, 224), dtype=np.float32)
    sub = [0, 5, 15, 64, 127]
    ref = co.embed_images(px[sub], W, s)
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=128, max_batch_text=2)
    enc.load_state_dict(W)
    res = {}
    for prec in ("bf16", "fp8"):
        enc.set_precision(prec)
        o16 = enc.encode_image(px[:16])
        (o128, kern) = _kernels_of(lambda: enc.encode_image(px))
        res[prec] = ((1 - _cos(o16[[0, 5, 15]], ref[:3])).max(), (1 - _cos(o128[sub], ref)).max())
        if prec == "fp8":
            print("fp8 kernels with outlier channels:", {k: v for k, v in kern.items() if "gemm" in k})
    enc.close()
    print("outlier channels at L/14 width, 6 layers: 1 - cos vs oracle (16 per call, 128 per call)",
          {k: ["%.2e" % x for x in v] for k, v in res.items()})
    for prec, v in res.items():
        assert max(v) < COS_TOL, (prec, v)


def test_l14_full_depth_outlier_channels_measured():
    """The same +300 / -180 residual channels through ALL 24 layers of the ViT-L/14 tower at the config's batch (128 images,
    first 4 checked): BOTH settings are held to north_star's 1e-3 (round 6; the fp8 bound was a looser 5e-3 before). Every
    e4m3 rounding of a weight column that meets a channel of magnitude 300 is amplified 300-fold against the other 1022
    columns, and 24 layers of it add up: measured 6.7e-4 on these seeded weights, a margin of 1.5 — a checkpoint whose fp8 gap
    comes out above the bar belongs under set_precision("bf16"), the default. Seeded Gaussian weights without
    outliers: 5.5e-4 (test_fp8_gpu.py)."""
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co
    import dataclasses

    s = dataclasses.replace(co.LONGCLIP_L14, t_layers=1, t_vocab=1000, eos_token_id=999)
    W = co.init_weights(s, seed=71)
    pos = W["vision_model.embeddings.position_embedding.weight"].copy()
    pos[:, 31] += 300.0
    pos[:, 700] -= 180.0
    W["vision_model.embeddings.position_embedding.weight"] = pos
    rng = np.random.Generator(np.random.Philox(72))
    px = rng.standard_normal((128, 3, 224, 224), dtype=np.float32)
    ref = co.embed_images(px[:4], W, s)
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=128, max_batch_text=2)
    enc.load_state_dict(W)
    d = {}
    for prec in ("bf16", "fp8"):
        enc.set_precision(prec)
        d[prec] = float((1 - _cos(enc.encode_image(px)[:4], ref)).max())
    enc.close()
    print("outlier channels, ViT-L/14 at FULL depth (24 layers), 128 per call: 1 - cos vs oracle bf16 %.2e, fp8 %.2e" % (d["bf16"], d["fp8"]))
    assert d["bf16"] < COS_TOL, d
    assert d["fp8"] < COS_TOL, d   # (round 6: held to north_star's tolerance — measured 6.7e-4 on these seeded weights; was 5e-3)


def test_drill_set_config0_on_the_gpu(b32_256):
    """BASELINE configs[0]: the reference's six sample images + the query 'red drill' (surrogate ids), seeded ViT-B/32
    weights, through the real towers and the real index: embeddings vs the HF vectors, the 6 x 6 cosine matrix, the
    text-image cosines, and the ranking wherever the HF gap exceeds twice the encoder's error on these cosines."""
    from mmiss_amd.index import FlatIndex
    from oracle import retrieval_oracle as ro

    enc, small, W, co = b32_256
    g = np.load(os.path.join(G, "drill_set.npz"))
    img = small.encode_image(g["crops_u8"])         # uint8 crops: rescale + normalise fused into patchify (utils.py:76)
    txt = small.encode_text(g["query_ids"])
    assert (1 - _cos(img, g["image"])).max() < COS_TOL
    assert (1 - _cos(txt, g["text"])).max() < COS_TOL
    np.testing.assert_allclose(img @ img.T, g["cosine"], atol=2e-3)
    got = (txt @ img.T)[0]
    want = g["text_image_cosine"][0]
    err = float(np.abs(got - want).max())
    assert err < 2e-3, err
    for i in range(6):
        for j in range(6):
            if want[i] - want[j] > 2 * err:
                assert got[i] > got[j], (i, j, want, got)
    # collection.query on the 6 rows (chromadb's exact brute-force regime, < 100 rows): bit-equal to the oracle on the
    # same embedding bits; similarity = 1 - d/2 (main.py:782)
    idx = FlatIndex(512, "f32")
    labels = np.arange(6, dtype=np.int64)
    idx.add(img, labels)
    lab, dist, cnt = idx.query(txt, 6)
    ol, od, oc = ro.query(txt, ro.normalize_rows(img, "f32"), labels, 6)
    np.testing.assert_array_equal(lab, ol)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
    np.testing.assert_allclose(1 - dist[0], got[lab[0]], atol=1e-6)
    idx.close()


def _host_corpus(n, d, seed):
    rng = np.random.Generator(np.random.Philox(seed))
    return rng.standard_normal((n, d), dtype=np.float32)


def _add_in_chunks(idx, rows, labels, chunk=250_000):
    for r0 in range(0, rows.shape[0], chunk):
        idx.add(rows[r0:r0 + chunk], labels[r0:r0 + chunk])


def test_config2_1000_b32_prompts_against_1m_rows(b32_256):
    """configs[2] at full size on one GPU: 1000 random 77-token prompts -> ViT-B/32 text tower (4 chunks of <= 256) ->
    cosine top-10 over a 1M x 512 f16 index (strip score-GEMM path, Q > 128). 16 embeddings against the fp32 oracle;
    ids + distance bits of 8 queries against the C restatement of the retrieval oracle on the same query bits."""
    from mmiss_amd.index import FlatIndex
    from oracle import retrieval_oracle_c as roc

    enc, _, W, co = b32_256
    s = co.VIT_B32
    ids = co.synthetic_text_ids(1000, 77, s.t_vocab, s.eos_token_id, seed=2, bos=49406)
    q = enc.encode_text(ids)
    sub = np.arange(0, 1000, 63)
    assert (1 - _cos(q[sub], co.embed_texts(ids[sub], W, s))).max() < COS_TOL
    N, D = 1_000_000, 512
    c = _host_corpus(N, D, seed=3)
    labels = np.arange(N, dtype=np.int64)
    idx = FlatIndex(D, "f16", capacity=N)
    _add_in_chunks(idx, c, labels)
    lab, dist, cnt = idx.query(q, 10)
    assert (cnt == 10).all() and (np.diff(dist, axis=1) >= 0).all()
    stored = roc.normalize_rows(c, "f16")
    del c
    qs = np.arange(5, 1000, 131)
    ol, od, oc = roc.query(q[qs], stored, labels, 10)
    np.testing.assert_array_equal(lab[qs], ol)
    np.testing.assert_array_equal(dist[qs].view(np.uint32), od.view(np.uint32))
    # the same queries one at a time take the streaming scan path: identical bits
    for j in (5, 136):
        l1, d1, _ = idx.query(q[j:j + 1], 10)
        np.testing.assert_array_equal(l1[0], lab[j])
        np.testing.assert_array_equal(d1[0].view(np.uint32), dist[j].view(np.uint32))
    idx.close()


def test_config3_q1024_blended_against_a_1p25m_row_shard():
    """configs[3] per-GPU size: one of the 8 row shards of the 10M x 512 f16 index (1.25M rows, global labels), a batch
    of 1024 multimodal queries normalize(0.5 i + 0.5 t) (main.py:852-860), top-10. ids + distance bits of 8 queries
    against the C oracle; strip length 2 is what the tile count selects here (bench at 10M rows: up to 19)."""
    from mmiss_amd.index import FlatIndex, blend
    from oracle import retrieval_oracle_c as roc

    N, D, Q, k = 1_250_000, 512, 1024, 10
    shard = 3
    c = _host_corpus(N, D, seed=4 + shard)
    labels = np.arange(shard * N, (shard + 1) * N, dtype=np.int64)
    idx = FlatIndex(D, "f16", capacity=N)
    _add_in_chunks(idx, c, labels)
    qi, qt = _host_corpus(Q, D, seed=50), _host_corpus(Q, D, seed=51)
    q = blend(qi, qt, 0.5)
    np.testing.assert_array_equal(q.view(np.uint32), roc.blend(qi, qt, 0.5).view(np.uint32))
    (lab, dist, cnt), kern = _kernels_of(lambda: idx.query(q, k))
    assert "score_gemm_f16" in kern, kern
    assert (cnt == k).all() and (np.diff(dist, axis=1) >= 0).all()
    assert lab.min() >= shard * N and lab.max() < (shard + 1) * N
    stored = roc.normalize_rows(c, "f16")
    del c
    qs = np.arange(7, Q, 137)
    ol, od, oc = roc.query(q[qs], stored, labels, k)
    np.testing.assert_array_equal(lab[qs], ol)
    np.testing.assert_array_equal(dist[qs].view(np.uint32), od.view(np.uint32))
    idx.close()


@pytest.mark.parametrize("N,D,seed", [(10_000_000, 512, 4), (6_250_000, 768, 6)])
def test_full_size_index_against_the_oracle(N, D, seed):
    """The metric's second half at its own size — 10M x 512 f16 — and one GPU's shard of configs[4] — 6.25M x 768 f16 —
    were only ever TIMED; the largest index under the oracle was 1.25M rows. Here: rows generated on
    the device chunk by chunk (bench.py's recipe), 4 queries through the streaming scan (Q = 1 and Q = 4) and as the first
    rows of a 1024-query batch through the score GEMM with strips of up to 19 tiles. The C oracle cannot walk 10M rows in
    test time, so it gets a candidate set that provably contains the true top-10 and is chosen WITHOUT this library: plain
    torch f32 scores of every row (a brute-force matmul per chunk) and every row within 2e-3 of the 10th best of them — the
    f16 storage rounding moves a score by < 5e-4, f32 accumulation by < 1e-5 — their STORED rows pulled to the host
    (mmiss_index_get, bit-exact against the oracle's normalisation in tests/test_index_gpu.py) and ranked canonically by
    oracle/libmmiss_oracle.so: labels and distance bits must be identical on all three paths."""
    import torch

    from mmiss_amd.index import FlatIndex
    from oracle import retrieval_oracle_c as roc

    k, chunk = 10, 1_250_000
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(seed)
    q = torch.randn(1024, D, device=dev, generator=torch.Generator(device=dev).manual_seed(seed + 1))
    qn4 = torch.nn.functional.normalize(q[:4].double(), dim=1).float()
    idx = FlatIndex(D, "f16", capacity=N)
    scores = torch.empty(4, N, device=dev)
    for r0 in range(0, N, chunk):
        n = min(chunk, N - r0)
        x = torch.randn(n, D, device=dev, generator=gen)
        idx.add(x, np.arange(r0, r0 + n, dtype=np.int64) * 2 + 1)        # labels != rows
        scores[:, r0:r0 + n] = qn4 @ torch.nn.functional.normalize(x, dim=1).T
        del x
    labels = np.arange(N, dtype=np.int64) * 2 + 1
    got = {}
    for name, qq in (("scan_q1", q[:1]), ("scan_q4", q[:4]), ("gemm_q1024", q)):
        (lab, dist, cnt), kern = _kernels_of(lambda: idx.query(qq.contiguous(), k))
        assert ("score_gemm_f16" in kern) == (name == "gemm_q1024"), (name, kern)
        got[name] = (lab[:4].cpu().numpy(), dist[:4].cpu().numpy())
    st = idx.guard_stats()
    q_host = q[:4].cpu().numpy()
    for j in range(4):
        s10 = torch.topk(scores[j], k).values[-1]
        cand = torch.nonzero(scores[j] >= s10 - 2e-3).flatten().cpu().numpy()
        assert k <= cand.size <= 20000, cand.size
        stored = idx.get(labels[cand]).astype(np.float16)                # (stored f16 rows come back widened to f32: lossless)
        ol, od, oc = roc.query(q_host[j:j + 1], stored, labels[cand], k)
        for name, (lab, dist) in got.items():
            if name == "scan_q1" and j > 0:
                continue
            np.testing.assert_array_equal(lab[j], ol[0], err_msg=f"{name} query {j}")
            np.testing.assert_array_equal(dist[j].view(np.uint32), od[0].view(np.uint32), err_msg=f"{name} query {j}")
    print(f"\n[full size {N} x {D}] candidates per query from torch scores: ok; guard stats {st}")
    idx.close()


def test_b32_bs256_fold_epilogue_on_the_256_tile(b32_256):
    """The folded-LayerNorm epilogue on the 256 x 256 phase-pipelined tile (options gemm_256_fold / gemm_256_fold_mlp;
    off by default, tools/option_ab.py decides): same bar against the oracle, and against the default tiles the same
    fp32 function with the same bf16 operand roundings (only the f32 summation order inside a K-tile differs)."""
    from mmiss_amd import _lib

    enc, small, W, co = b32_256
    s = co.VIT_B32
    rng = np.random.Generator(np.random.Philox(77))
    px = rng.standard_normal((256, 3, 224, 224), dtype=np.float32)
    px[9] = px[9] * 0.05 + 4.0
    base, kern_p = _kernels_of(lambda: enc.encode_image(px))   # default: the persistent 256 x 256 kernel (gemm_bf16_p256.h)
    assert kern_p.get("gemm_bf16_lnfold_bias_p256", 0) == 12 and kern_p.get("gemm_bf16_lnfold_qgelu_p256", 0) == 11, kern_p
    _lib.set_option("gemm_p256", 0)
    _lib.set_option("gemm_256_fold", 2304)
    _lib.set_option("gemm_256_fold_mlp", 1)
    try:
        out, kern = _kernels_of(lambda: enc.encode_image(px))
        _lib.set_option("gemm_256_fold", 0)
        _lib.set_option("gemm_256_fold_mlp", 0)
        out128, kern128 = _kernels_of(lambda: enc.encode_image(px))    # ... and the 128-column tiles of rounds 1-2
    finally:
        _lib.set_option("gemm_p256", 1)
        _lib.set_option("gemm_256_fold", -1)
        _lib.set_option("gemm_256_fold_mlp", 0)
    assert kern128.get("gemm_bf16_lnfold_bias", 0) == 12 and kern128.get("gemm_bf16_lnfold_qgelu", 0) == 11, kern128
    assert (1 - _cos(out128, base)).max() < 5e-5
    assert kern.get("gemm_bf16_lnfold_bias", 0) == 12 and kern.get("gemm_bf16_lnfold_qgelu", 0) == 11, kern
    sub = np.concatenate([[9], np.arange(0, 256, 17)])
    assert (1 - _cos(out[sub], co.embed_images(px[sub], W, s))).max() < COS_TOL
    assert (1 - _cos(out, base)).max() < 5e-5   # (a last-bit f32 difference can flip a bf16 rounding of the residual stream)


def test_b32_bs256_bf16_residual_stream_and_the_f32_alternative(b32_256):
    """Default at 12 800 rows: the residual stream itself in bf16 (the residual GEMMs read-modify-write the bf16 rows; every
    add rounds the stream to 8 significant bits). Precision "bf16-f32resid" keeps the f32 stream of round 1. Both against
    the fp32 oracle at the 1e-3 bar; the measured distances are printed (CPU simulation of the same roundings: 3-8e-5
    against 2e-6 .. 2e-5)."""
    enc, small, W, co = b32_256
    s = co.VIT_B32
    rng = np.random.Generator(np.random.Philox(78))
    px = rng.standard_normal((256, 3, 224, 224), dtype=np.float32)
    px[9] = px[9] * 0.05 + 4.0
    ids = co.synthetic_text_ids(256, 77, s.t_vocab, s.eos_token_id, seed=5, bos=49406)
    out_i, kern = _kernels_of(lambda: enc.encode_image(px))
    out_t = enc.encode_text(ids)
    np.testing.assert_array_equal(out_i, enc.encode_image(px))
    # (round 3: out-projection and FC2 of this shape run on the 160 x 256 tile of gemm_bf16_p160.h)
    assert kern.get("gemm_bf16_bias_resid16_p160_k768", 0) == 11 and kern.get("gemm_bf16_bias_resid16_p160_k3072", 0) == 11, kern
    enc.set_precision("bf16-f32resid")
    try:
        f32_i, kern32 = _kernels_of(lambda: enc.encode_image(px))
        f32_t = enc.encode_text(ids)
    finally:
        enc.set_precision("bf16")
    assert kern32.get("gemm_bf16_bias_resid_k768", 0) == 11 and "gemm_bf16_bias_resid16_k768" not in kern32, kern32
    sub = np.concatenate([[9], np.arange(0, 256, 17)])
    ref_i, ref_t = co.embed_images(px[sub], W, s), co.embed_texts(ids[sub], W, s)
    d16 = (1 - _cos(out_i[sub], ref_i)).max(), (1 - _cos(out_t[sub], ref_t)).max()
    d32 = (1 - _cos(f32_i[sub], ref_i)).max(), (1 - _cos(f32_t[sub], ref_t)).max()
    print("1 - cos vs oracle, bf16 residual stream: image %.2e text %.2e; f32 stream: image %.2e text %.2e" % (d16 + d32))
    assert max(d16) < 2e-4, d16          # (the bar of the path is COS_TOL = 1e-3; this pins the measured level)
    assert max(d32) < 6e-5, d32          # (text tower: 3.2e-5 with bf16 operands alone)


def test_b32_bs256_the_160x256_tile_is_bit_identical_to_the_128_column_kernels(b32_256):
    """Round 3: out-projection, FC2 and the patch-embedding GEMM of the bs-256 image encode (and out-projection / FC2 of the
    256 x 77 text encode) run on the 160 x 256 tile of gemm_bf16_p160.h. Same k order inside every accumulator, same epilogue
    arithmetic: the embeddings are the SAME BITS as with option gemm_p160 = 0 (the 128-column kernels of rounds 1-2)."""
    from mmiss_amd import _lib

    enc, small, W, co = b32_256
    s = co.VIT_B32
    rng = np.random.Generator(np.random.Philox(79))
    px = rng.standard_normal((256, 3, 224, 224), dtype=np.float32)
    ids = co.synthetic_text_ids(256, 77, s.t_vocab, s.eos_token_id, seed=6, bos=49406)
    out_i, kern = _kernels_of(lambda: enc.encode_image(px))
    out_t, kern_t = _kernels_of(lambda: enc.encode_text(ids))
    assert kern.get("gemm_bf16_patch_p160", 0) == 1 and kern.get("gemm_bf16_bias_resid16_p160_k3072", 0) == 11, kern
    assert kern_t.get("gemm_bf16_bias_resid16_p160_k2048", 0) == 11 and kern_t.get("gemm_bf16_bias_resid16_p160_k512", 0) == 11, kern_t
    _lib.set_option("gemm_p160", 0)
    try:
        old_i, kern0 = _kernels_of(lambda: enc.encode_image(px))
        old_t = enc.encode_text(ids)
    finally:
        _lib.set_option("gemm_p160", 1)
    assert kern0.get("gemm_bf16_patch", 0) == 1 and kern0.get("gemm_bf16_bias_resid16_k3072", 0) == 11 and not any("p160" in k for k in kern0), kern0
    np.testing.assert_array_equal(out_i.view(np.uint32), old_i.view(np.uint32))
    np.testing.assert_array_equal(out_t.view(np.uint32), old_t.view(np.uint32))


def test_l14_full_depth_bf16_residual_stream_folded_and_separate_layernorm():
    """ViT-L/14 geometry (hidden 1024) at 24 images = 6168 token rows, full depth, bf16 residual stream. Round 4: LayerNorm is
    folded into the QKV / FC1 GEMMs here too (the persistent kernel takes FINISHED row statistics from ln_finalize_kernel:
    the raw partials of a 256-row tile no longer fit beside its staging buffers at K = 1024); option ln_fold_1024 = 0 gives
    the separate LayerNorm kernels of round 3 back. Both, the f32-stream alternative and the fp8 setting (fp8 out-projection
    on an MXFP8 attention output) against the fp32 oracle at the 1e-3 bar."""
    from mmiss_amd import _lib
    import dataclasses
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    s = dataclasses.replace(co.LONGCLIP_L14, t_layers=1, t_vocab=1000, eos_token_id=999)  # vision tower only
    W = co.init_weights(s, seed=61)
    rng = np.random.Generator(np.random.Philox(62))
    px = rng.standard_normal((24, 3, 224, 224), dtype=np.float32)
    sub = np.array([0, 7, 13, 23])
    ref = co.embed_images(px[sub], W, s)
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=24, max_batch_text=2)
    enc.load_state_dict(W)
    outf, kernf = _kernels_of(lambda: enc.encode_image(px))
    assert kernf.get("gemm_bf16_lnfold_bias_p256", 0) == 24 and kernf.get("gemm_bf16_lnfold_qgelu_p256", 0) == 23, kernf
    assert kernf.get("ln_finalize", 0) == 47 and "layernorm16" not in kernf, kernf
    _lib.set_option("ln_fold_1024", 0)
    try:
        out, kern = _kernels_of(lambda: enc.encode_image(px))
    finally:
        _lib.set_option("ln_fold_1024", 1)
    assert kern.get("gemm_bf16_bias_resid16_k1024", 0) == 23 and kern.get("gemm_bf16_bias_resid16_k4096", 0) == 23, kern
    assert kern.get("layernorm16", 0) == 47, kern
    enc.set_precision("bf16-f32resid")
    out32, kern32 = _kernels_of(lambda: enc.encode_image(px))
    assert "layernorm16" not in kern32 and kern32.get("gemm_bf16_bias_resid_k1024", 0) == 23, kern32
    # fp8 GEMMs (QKV / FC1 / FC2) on the same bf16 stream: the same 1e-3 bar (tests/test_fp8_gpu.py)
    enc.set_precision("fp8")
    out8, kern8 = _kernels_of(lambda: enc.encode_image(px))
    enc.close()
    assert kern8.get("gemm_fp8_bias_resid16", 0) == 46 and kern8.get("layernorm16_mxfp8", 0) == 47, kern8   # FC2 + out-projection
    assert kern8.get("attention_mx", 0) == 23 and "gemm_bf16_bias_resid16_k1024" not in kern8, kern8
    d16f, d16, d32, d8 = ((1 - _cos(o[sub], ref)).max() for o in (outf, out, out32, out8))
    print("L/14 24 layers, 6168 rows: 1 - cos vs oracle bf16 stream folded LN %.2e / separate LN %.2e, f32 stream %.2e, "
          "fp8 GEMMs on the bf16 stream %.2e" % (d16f, d16, d32, d8))
    assert d16f < 3e-4 and d16 < 3e-4 and d32 < 3e-5, (d16f, d16, d32)   # (bar: COS_TOL = 1e-3; CPU simulation 3-8e-5 / 2-3e-6)
    assert d8 < 1e-3, d8


def test_config1_100k_distinct_images_ingested_then_queried_back(b32_256):
    """BASELINE configs[1] as SURVEY 8(d) defines it (the reference's ingest-then-query flow, backend/app/main.py:1124-1162 ->
    :761-765): 100 000 DISTINCT images — pixels N(0,1) generated on the device in 391 batches of 256 from seed 1234 + batch —
    encoded at bs 256, every embedding added to the index, every embedding queried back top-10. Every row must rank itself
    first; a 64-query subset must carry the C oracle's ids and distance bits (f16 rows, the bench's index type, and f32)."""
    import torch
    from mmiss_amd.index import FlatIndex
    from oracle import retrieval_oracle_c as roc

    enc, _, W, co = b32_256
    N, D, B = 100_000, 512, 256
    emb = torch.empty((N, D), dtype=torch.float32, device="cuda")
    for c in range((N + B - 1) // B):
        n = min(B, N - c * B)
        g = torch.Generator(device="cuda").manual_seed(1234 + c)
        px = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
        enc.encode_image(px[:n], out=emb[c * B:c * B + n])
    torch.cuda.synchronize()
    host = emb.cpu().numpy()
    assert np.isfinite(host).all() and np.abs(np.linalg.norm(host, axis=1) - 1).max() < 1e-5
    # two batches against the fp32 oracle (first and last: the 160-image tail batch takes other tile shapes)
    for c, rows in ((0, [0, 100, 255]), (390, [0, 159])):
        g = torch.Generator(device="cuda").manual_seed(1234 + c)
        px = torch.randn(B, 3, 224, 224, device="cuda", generator=g).cpu().numpy()
        ref = co.embed_images(px[rows], W, co.VIT_B32)
        assert (1 - _cos(host[[c * B + r for r in rows]], ref)).max() < COS_TOL
    labels = np.arange(N, dtype=np.int64)
    sub = np.arange(0, N, N // 64)[:64]
    for dtype, self_tol in (("f16", 1e-4), ("f32", 1e-6)):   # (f16 rows: the storage rounding puts a row ~2e-5 from itself)
        idx = FlatIndex(D, dtype, capacity=N)
        idx.add(emb, labels)
        lab = np.empty((N, 10), dtype=np.int64)
        dist = np.empty((N, 10), dtype=np.float32)
        for i in range(0, N, 1024):
            l, d, c = idx.query(emb[i:i + 1024], 10)
            lab[i:i + 1024], dist[i:i + 1024] = l.cpu().numpy(), d.cpu().numpy()
            assert (c.cpu().numpy() == 10).all()
        gap = dist[:, 1] - dist[:, 0]
        print(f"config 1, {dtype} rows: self-distance max {dist[:, 0].max():.2e}, smallest gap to the runner-up {gap.min():.2e}, "
              f"guard stats {idx.guard_stats()}")
        stored = roc.normalize_rows(host, dtype)
        ol, od, oc = roc.query(host[sub], stored, labels, 10)
        np.testing.assert_array_equal(lab[sub], ol)
        np.testing.assert_array_equal(dist[sub].view(np.uint32), od.view(np.uint32))
        assert (lab[:, 0] == labels).all(), np.nonzero(lab[:, 0] != labels)[0][:10]
        assert dist[:, 0].max() < self_tol and (np.diff(dist, axis=1) >= 0).all()
        idx.close()

"""CLIP towers on the MI355X vs oracle/clip_oracle.py (numpy fp32, itself pinned against transformers'
CLIPModel): embeddings within 1e-3 cosine (north_star tolerance), intermediates bisected per layer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COS_TOL = 1e-3  # BASELINE.json north_star: "within 1e-3 cosine of the reference CPU path"


@pytest.fixture(scope="module")
def tiny():
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    W = co.init_weights(co.TINY, seed=0)
    enc = ClipEncoder(ClipShape.from_any(co.TINY), max_batch_image=8, max_batch_text=8)
    enc.record_taps(True)
    used, ignored = enc.load_state_dict(W)
    assert ignored == 0 and used == len(W)
    return enc, W, co


def _cos(a, b):
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def test_tiny_image_tower_layer_by_layer(tiny):
    enc, W, co = tiny
    s = co.TINY
    rng = np.random.Generator(np.random.Philox(5))
    px = rng.standard_normal((5, 3, s.v_image, s.v_image), dtype=np.float32)
    out = enc.encode_image(px)
    taps = {}
    ref = co.l2_normalize(co.image_features(px, W, s, taps))
    T, d = s.v_tokens, s.v_hidden
    for l in range(s.v_layers + 1):
        got = enc.tap(0, l, 5 * T * d).reshape(5, T, d)
        err = np.abs(got - taps[l]).max()
        scale = np.abs(taps[l]).max()
        assert err < 0.03 * scale, f"layer {l}: err {err} scale {scale}"
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-5
    assert (1 - _cos(out, ref)).max() < COS_TOL


def test_tiny_text_tower_and_short_sequences(tiny):
    enc, W, co = tiny
    s = co.TINY
    ids = co.synthetic_text_ids(7, s.t_ctx, s.t_vocab, s.eos_token_id, seed=3)
    out = enc.encode_text(ids)
    ref = co.embed_texts(ids, W, s)
    assert (1 - _cos(out, ref)).max() < COS_TOL
    # a shorter T (fewer pad columns) must give the same embeddings when every row's EOS is inside
    eos = co.eos_positions(ids, s.eos_token_id)
    Tcut = int(eos.max()) + 1
    out2 = enc.encode_text(ids[:, :Tcut])
    assert (1 - _cos(out2, ref)).max() < COS_TOL


def test_last_layer_pruning_matches_full_computation(tiny):
    """Without taps the last layer's out-proj / MLP run on the pooled rows only; rows do not mix after attention,
    so the embeddings must equal the unpruned path (the `tiny` fixture records taps = unpruned)."""
    from mmiss_amd.encoder import ClipEncoder, ClipShape

    enc, W, co = tiny
    s = co.TINY
    pruned = ClipEncoder(ClipShape.from_any(s), max_batch_image=8, max_batch_text=8)
    pruned.load_state_dict(W)
    rng = np.random.Generator(np.random.Philox(77))
    px = rng.standard_normal((7, 3, s.v_image, s.v_image), dtype=np.float32)
    ids = co.synthetic_text_ids(7, s.t_ctx, s.t_vocab, s.eos_token_id, seed=78)
    from mmiss_amd import _lib

    # identical arithmetic on both sides (separate LayerNorm kernels): the pruned tail must reproduce the full computation
    _lib.set_option("skinny_fold", 0)
    try:
        np.testing.assert_allclose(pruned.encode_image(px), enc.encode_image(px), atol=1e-6)
        np.testing.assert_allclose(pruned.encode_text(ids), enc.encode_text(ids), atol=1e-6)
    finally:
        _lib.set_option("skinny_fold", 1)
    # default at this size: LayerNorm folded into the skinny GEMMs; the pruned tail still runs its B rows through a LayerNorm
    # kernel (bf16(LN(x)) W vs the folded bf16(x) (W gamma) form): two bf16 roundings of the same function
    assert (1 - _cos(pruned.encode_image(px), enc.encode_image(px))).max() < 1e-4
    assert (1 - _cos(pruned.encode_text(ids), enc.encode_text(ids))).max() < 1e-4

    for mode in (2,):  # the pruned tail is the same whatever the LayerNorm mode
        pruned.set_fuse_ln(mode)
        assert (1 - _cos(pruned.encode_image(px), co.embed_images(px, W, s))).max() < COS_TOL
    assert (1 - _cos(pruned.encode_text(ids), co.embed_texts(ids, W, s))).max() < COS_TOL


def test_layernorm_modes_agree(tiny):
    """LayerNorm1/2 as separate kernels (0), normalised during operand staging (1) and folded algebraically into
    weights + GEMM epilogue (2, the default): same embeddings to rounding, all within the bar of the oracle."""
    from mmiss_amd.encoder import ClipEncoder, ClipShape

    enc, W, co = tiny
    s = co.TINY
    rng = np.random.Generator(np.random.Philox(91))
    px = rng.standard_normal((6, 3, s.v_image, s.v_image), dtype=np.float32) * 3 + 1.5  # non-zero row means
    ids = co.synthetic_text_ids(6, s.t_ctx, s.t_vocab, s.eos_token_id, seed=92)
    ref_i, ref_t = co.embed_images(px, W, s), co.embed_texts(ids, W, s)
    from mmiss_amd import _lib

    outs = []
    for mode in (0, 2):  # (mode 1, LayerNorm during operand staging, was removed in round 4)
        e = ClipEncoder(ClipShape.from_any(s), max_batch_image=8, max_batch_text=8)
        e.load_state_dict(W)
        e.set_fuse_ln(mode)
        a, t = e.encode_image(px), e.encode_text(ids)
        assert (1 - _cos(a, ref_i)).max() < COS_TOL and (1 - _cos(t, ref_t)).max() < COS_TOL, mode
        outs.append((a, t))
    for a, t in outs[1:]:
        assert (1 - _cos(a, outs[0][0])).max() < 3e-5 and (1 - _cos(t, outs[0][1])).max() < 3e-5


def test_folded_layernorm_with_outlier_channels():
    """Real CLIP checkpoints carry a few hidden channels whose activations are two orders of magnitude above the rest.
    The folded LayerNorm feeds bf16(x), not bf16(LN(x)), to the GEMMs and takes the variance from single-pass partial
    sums - both must hold up there. Two channels of the residual stream are pushed to ~+-60 through the position table;
    folded, separate and the fp32 oracle must still agree within the north_star tolerance."""
    import dataclasses
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    s = dataclasses.replace(co.TINY, v_layers=4)
    W = co.init_weights(s, seed=11)
    pos = W["vision_model.embeddings.position_embedding.weight"].copy()
    pos[:, 3] += 60.0
    pos[:, 77] -= 45.0
    W["vision_model.embeddings.position_embedding.weight"] = pos
    rng = np.random.Generator(np.random.Philox(13))
    px = rng.standard_normal((6, 3, s.v_image, s.v_image), dtype=np.float32)
    ref = co.embed_images(px, W, s)
    outs = {}
    for mode in (0, 2):
        e = ClipEncoder(ClipShape.from_any(s), max_batch_image=8, max_batch_text=8)
        e.load_state_dict(W)
        e.set_fuse_ln(mode)
        outs[mode] = e.encode_image(px)
        assert (1 - _cos(outs[mode], ref)).max() < COS_TOL, mode
        e.close()
    assert (1 - _cos(outs[0], outs[2])).max() < 1e-4


def test_patch14_padded_k_and_odd_token_count():
    """ViT-L/14-style geometry: patch 14 -> 3*14*14 = 588 is padded to 640 for the MFMA K loop; T = 17 tokens."""
    import dataclasses
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    s = dataclasses.replace(co.TINY, v_patch=14, v_image=56)
    W = co.init_weights(s, seed=5)
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=4, max_batch_text=4)
    enc.load_state_dict(W)
    rng = np.random.Generator(np.random.Philox(79))
    px = rng.standard_normal((3, 3, 56, 56), dtype=np.float32)
    assert (1 - _cos(enc.encode_image(px), co.embed_images(px, W, s))).max() < COS_TOL


def test_wide_mlp_takes_the_256_tile_and_matches():
    """FC1 with N >= 4096 (the ViT-L/14 MLP width) and more than 512 rows can run on the 256x256 phase-pipelined tile
    (option gemm_256); its per-element k order is that of the 128-column tile, so the embeddings must not change by a bit."""
    import dataclasses
    from mmiss_amd import _lib
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    s = dataclasses.replace(co.TINY, v_patch=14, v_image=56, v_mlp=4096)  # 17 tokens x 32 images = 544 rows
    W = co.init_weights(s, seed=6)
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=32, max_batch_text=4)
    enc.load_state_dict(W)
    rng = np.random.Generator(np.random.Philox(80))
    px = rng.standard_normal((32, 3, 56, 56), dtype=np.float32)
    base = enc.encode_image(px)                 # default: the 128-column tile (faster since the banded tile order)
    assert (1 - _cos(base, co.embed_images(px, W, s))).max() < COS_TOL
    _lib.set_option("gemm_256", 4096)
    try:
        got = enc.encode_image(px)
    finally:
        _lib.set_option("gemm_256", 0)
    np.testing.assert_array_equal(got, base)


def test_chunking_beyond_max_batch(tiny):
    enc, W, co = tiny
    s = co.TINY
    rng = np.random.Generator(np.random.Philox(8))
    px = rng.standard_normal((19, 3, s.v_image, s.v_image), dtype=np.float32)  # max_batch_image = 8
    out = enc.encode_image(px)
    ref = co.embed_images(px, W, s)
    assert (1 - _cos(out, ref)).max() < COS_TOL
    one = enc.encode_image(px[4:5])
    np.testing.assert_allclose(one[0], out[4], atol=2e-6)  # batch-size invariance


def test_u8_fused_preprocess_matches_float_path(tiny):
    enc, W, co = tiny
    s = co.TINY
    rng = np.random.Generator(np.random.Philox(9))
    u8 = rng.integers(0, 256, size=(4, s.v_image, s.v_image, 3), dtype=np.uint8)
    a = enc.encode_image(u8)
    b = enc.encode_image(co.normalize_u8(u8))
    assert (1 - _cos(a, b)).max() < 1e-5
    assert (1 - _cos(a, co.embed_images(co.normalize_u8(u8), W, s))).max() < COS_TOL


def test_device_tensor_io(tiny):
    import torch

    enc, W, co = tiny
    s = co.TINY
    rng = np.random.Generator(np.random.Philox(10))
    px = rng.standard_normal((3, 3, s.v_image, s.v_image), dtype=np.float32)
    out = enc.encode_image(torch.from_numpy(px).cuda())
    assert out.is_cuda
    ref = enc.encode_image(px)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-6)


def test_errors_are_loud(tiny):
    from mmiss_amd.encoder import ClipEncoder, ClipShape

    enc, W, co = tiny
    with pytest.raises(ValueError):
        enc.encode_image(np.zeros((1, 3, 32, 32), np.float32))
    with pytest.raises(RuntimeError):
        enc.encode_text(np.zeros((1, co.TINY.t_ctx + 1), np.int32))
    fresh = ClipEncoder(ClipShape.from_any(co.TINY), max_batch_image=2, max_batch_text=2)
    with pytest.raises(RuntimeError):  # weights missing -> finalize fails
        fresh.load_state_dict({k: v for k, v in list(W.items())[:10]})


@pytest.fixture(scope="module")
def b32():
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    W = co.init_weights(co.VIT_B32, seed=0)
    enc = ClipEncoder(ClipShape.from_any(co.VIT_B32), max_batch_image=16, max_batch_text=16)
    enc.load_state_dict(W)
    return enc, W, co


def test_b32_image_and_text_embeddings(b32):
    enc, W, co = b32
    s = co.VIT_B32
    rng = np.random.Generator(np.random.Philox(1))
    px = rng.standard_normal((6, 3, 224, 224), dtype=np.float32)
    out = enc.encode_image(px)
    ref = co.embed_images(px, W, s)
    assert (1 - _cos(out, ref)).max() < COS_TOL
    ids = co.synthetic_text_ids(6, 77, s.t_vocab, s.eos_token_id, seed=2, bos=49406)
    out_t = enc.encode_text(ids)
    ref_t = co.embed_texts(ids, W, s)
    assert (1 - _cos(out_t, ref_t)).max() < COS_TOL


def test_b32_every_gemm_regime_agrees_with_the_oracle(b32):
    """The same 16 images at batch 1 (weight-streaming GEMMs), 5 (skinny for the narrow outputs, tiled for the wide
    ones), and 16 (800 rows: three-buffer 8-wave tiles, split-K FC2 and patch embedding): every regime within the
    north_star tolerance of the oracle and of each other."""
    enc, W, co = b32
    s = co.VIT_B32
    rng = np.random.Generator(np.random.Philox(31))
    px = rng.standard_normal((16, 3, 224, 224), dtype=np.float32)
    ref = co.embed_images(px, W, s)
    whole = enc.encode_image(px)
    assert (1 - _cos(whole, ref)).max() < COS_TOL
    fives = np.concatenate([enc.encode_image(px[i:i + 5]) for i in range(0, 16, 5)])
    ones = np.concatenate([enc.encode_image(px[i:i + 1]) for i in range(16)])
    for other in (fives, ones):
        assert (1 - _cos(other, ref)).max() < COS_TOL
        assert (1 - _cos(other, whole)).max() < 2e-5  # only the summation order of the GEMMs differs
    again = enc.encode_image(px)
    np.testing.assert_array_equal(whole, again)  # every path is deterministic (fixed-order split-K reduction)


def test_longclip_l14_geometry_two_layers():
    """The reference HEAD's model family (ViT-L/14 towers, 248-token text table, 768-d joint space; utils.py:16-17)
    with the depth cut to 2 layers to keep the oracle fast: d=1024/16 heads/T=257 vision, d=768/12 heads/T=248 text."""
    import dataclasses
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    s = dataclasses.replace(co.LONGCLIP_L14, v_layers=2, t_layers=2, t_vocab=2000, eos_token_id=1999)
    W = co.init_weights(s, seed=21)
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=4, max_batch_text=4)
    enc.load_state_dict(W)
    rng = np.random.Generator(np.random.Philox(22))
    px = rng.standard_normal((3, 3, 224, 224), dtype=np.float32)
    out = enc.encode_image(px)
    assert out.shape == (3, 768)
    assert (1 - _cos(out, co.embed_images(px, W, s))).max() < COS_TOL
    ids = co.synthetic_text_ids(3, 248, s.t_vocab, s.eos_token_id, seed=23)
    assert (1 - _cos(enc.encode_text(ids), co.embed_texts(ids, W, s))).max() < COS_TOL


def test_concurrent_queries_and_updates_on_one_collection():
    """The reference mutates the collection from a background task while routes read it (main.py:410,1030): one writer
    thread and two reader threads on the same FlatCollection must not corrupt results."""
    import threading
    import mmiss_amd  # noqa: F401
    from mmiss_amd.collection import FlatCollection

    col = FlatCollection("threads", autosave=False)
    rng = np.random.Generator(np.random.Philox(31))
    base = rng.standard_normal((64, 128), dtype=np.float32)
    col.add(ids=[f"b{i}" for i in range(64)], embeddings=base, metadatas=[{"n": i} for i in range(64)])
    errors = []

    def writer():
        try:
            for i in range(40):
                col.add(ids=[f"w{i}"], embeddings=rng.standard_normal((1, 128), dtype=np.float32), metadatas=[{"n": 100 + i}])
                col.update(ids=[f"b{i}"], metadatas=[{"touched": True}])
                if i % 5 == 4:
                    col.delete(ids=[f"w{i - 2}"])
        except Exception as e:  # pragma: no cover
            errors.append(e)

    def reader(seed):
        try:
            for i in range(60):
                j = (seed * 7 + i) % 64
                res = col.query(query_embeddings=base[j:j + 1], n_results=3, include=["metadatas", "distances"])
                assert res["ids"][0][0] == f"b{j}" and res["distances"][0][0] < 1e-5
                assert res["metadatas"][0][0]["n"] == j
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=writer)] + [threading.Thread(target=reader, args=(s_,)) for s_ in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert col.count() == 64 + 40 - 8


def test_text_padding_is_trimmed_without_changing_results(tiny):
    enc, W, co = tiny
    s = co.TINY
    ids = co.synthetic_text_ids(5, s.t_ctx, s.t_vocab, s.eos_token_id, seed=41)
    ids[:, 6:] = s.eos_token_id          # short prompts padded to the full context, as the reference does
    ids[:, 5] = s.eos_token_id
    a = enc.encode_text(ids)              # runs at T = 6
    b = enc.encode_text(ids, trim_padding=False)
    assert (1 - _cos(a, b)).max() < 1e-6
    assert (1 - _cos(a, co.embed_texts(ids, W, s))).max() < COS_TOL


def test_single_request_runs_without_layernorm_launches(b32):
    """One image / one prompt at a time (the reference's regime, backend/app/utils.py:76-77,88-97): every GEMM of a layer is on
    the weight-streaming skinny kernel, LayerNorm1/2 are folded into the QKV / FC1 weights and the residual GEMMs' epilogues
    keep the bf16 rows + per-16-column statistics up to date — no per-layer LayerNorm launch is left. Same bar against the
    oracle as every other path, and against the separate-LayerNorm form of the same request."""
    from mmiss_amd import _lib

    enc, W, co = b32
    s = co.VIT_B32
    rng = np.random.Generator(np.random.Philox(41))
    px = rng.standard_normal((2, 3, 224, 224), dtype=np.float32)
    px[1] = px[1] * 0.05 + 4.0                      # a row with a large mean / std ratio
    ids = co.synthetic_text_ids(1, 77, s.t_vocab, s.eos_token_id, seed=42, bos=49406)

    def kernels_of(fn):
        _lib.prof_reset(); _lib.prof_enable(True)
        try:
            out = fn()
        finally:
            _lib.prof_enable(False)
        return out, {p["kernel"]: p["launches"] for p in _lib.prof_read()}

    out1, k1 = kernels_of(lambda: enc.encode_image(px[:1]))
    assert k1.get("gemm_skinny_lnfold_bias", 0) == 12 and k1.get("gemm_skinny_lnfold_qgelu", 0) == 11, k1
    assert k1.get("layernorm", 0) <= 4, k1          # pre-LN, the pruned last layer's LN2, the head — no per-layer launches
    # round 5: CLS rows + pre-LN + the mode's entry statistics are ONE launch (prelayernorm_skinny_kernel): no row_stats launch,
    # and the same bits as the three separate kernels (option embed_fused = 0)
    assert "row_stats" not in k1, k1
    _lib.set_option("embed_fused", 0)
    try:
        out1_unfused, k1u = kernels_of(lambda: enc.encode_image(px[:1]))
    finally:
        _lib.set_option("embed_fused", 1)
    assert k1u.get("row_stats", 0) == 1, k1u
    np.testing.assert_array_equal(out1, out1_unfused)
    out2 = enc.encode_image(px[1:2])
    out_t, kt = kernels_of(lambda: enc.encode_text(ids))
    assert kt.get("gemm_skinny_lnfold_bias", 0) == 12, kt
    ref = co.embed_images(px, W, s)
    d_img = (1 - _cos(np.concatenate([out1, out2]), ref)).max()
    d_txt = (1 - _cos(out_t, co.embed_texts(ids, W, s))).max()
    _lib.set_option("skinny_fold", 0)
    try:
        plain1, kp = kernels_of(lambda: enc.encode_image(px[:1]))
    finally:
        _lib.set_option("skinny_fold", 1)
    assert "gemm_skinny_lnfold_bias" not in kp and kp.get("layernorm", 0) >= 24, kp
    print("single request, folded skinny GEMMs: 1 - cos vs oracle image %.2e text %.2e; vs separate LayerNorm %.2e"
          % (d_img, d_txt, (1 - _cos(out1, plain1)).max()))
    assert d_img < COS_TOL and d_txt < COS_TOL
    assert (1 - _cos(out1, plain1)).max() < 2e-5


def test_pinned_staging_ring_delivers_the_default_path_bits():
    """Option pinned_stage = 1 (csrc/host_stager.h: a ring of pinned blocks filled by worker threads) is a different ROUTE for
    host-resident inputs of 4 MB and more, not different arithmetic: float32 pixels, uint8 crops and raw RGB uploads must give
    the bytes of the default route (hipMemcpyAsync from the caller's memory). Sizes chosen so that a chunk spans several ring
    blocks (1 MB blocks, 3 threads: slices, wrap-around of the four slots, a ragged last block) and several chunks per call."""
    import dataclasses

    import mmiss_amd  # noqa: F401
    from mmiss_amd import _lib
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from oracle import clip_oracle as co

    s = dataclasses.replace(co.TINY, v_image=224)          # 7 x 7 patches + CLS; 602 KB of float32 per image
    W = co.init_weights(s, seed=11)
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=32, max_batch_text=2)
    enc.load_state_dict(W)
    rng = np.random.Generator(np.random.Philox(12))
    px = rng.standard_normal((75, 3, 224, 224), dtype=np.float32)                   # 45 MB: chunks of 32, 32, 11 images
    u8 = rng.integers(0, 256, size=(75, 224, 224, 3), dtype=np.uint8)               # 11 MB; 4.8 MB per full chunk
    raw = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in [(480, 640), (300, 500), (224, 224), (700, 333)] * 12]   # 28 MB
    got = {}
    try:
        for ring in (0, 1):
            _lib.set_option("pinned_stage", ring)
            _lib.set_option("stage_block_mb", 1)
            _lib.set_option("stage_threads", 3)
            got[ring] = (enc.encode_image(px), enc.encode_image(u8), enc.encode_image_rgb(raw))
    finally:
        _lib.set_option("pinned_stage", 0)
        _lib.set_option("stage_block_mb", 16)
        _lib.set_option("stage_threads", 0)
    for a, b in zip(got[0], got[1]):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    ref = co.embed_images(px[[0, 40, 74]], W, s)
    assert (1 - _cos(got[1][0][[0, 40, 74]], ref)).max() < COS_TOL
    # a second handle with out-of-range knobs: clamped (a zero block would never advance the copy loop), same bits
    enc2 = ClipEncoder(ClipShape.from_any(s), max_batch_image=32, max_batch_text=2)
    enc2.load_state_dict(W)
    try:
        _lib.set_option("pinned_stage", 1)
        _lib.set_option("stage_block_mb", 0)
        _lib.set_option("stage_threads", 1000)
        again = enc2.encode_image(px)
    finally:
        _lib.set_option("pinned_stage", 0)
        _lib.set_option("stage_block_mb", 16)
        _lib.set_option("stage_threads", 0)
    assert np.array_equal(again.view(np.uint32), got[0][0].view(np.uint32))
    enc.close()
    enc2.close()

"""Flat cosine index on the MI355X vs oracle/retrieval_oracle.py: ids (labels) and ranking bit-exact,
distances bit-exact (same canonical fp64 arithmetic), across storage dtypes, batch sizes, k regimes
(fused top-k, paged top-k), ties, mutation and edge cases."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import mmiss_amd  # noqa: F401
    from mmiss_amd.index import FlatIndex, blend, merge_topk
    from oracle import retrieval_oracle as ro

    return FlatIndex, blend, merge_topk, ro


def _corpus(n, d, seed):
    rng = np.random.Generator(np.random.Philox(seed))
    return rng.standard_normal((n, d), dtype=np.float32)


def _check(idx, ro, stored, labels, q, k):
    lab, dist, cnt = idx.query(q, k)
    ol, od, oc = ro.query(q, stored, labels, k)
    np.testing.assert_array_equal(cnt, oc)
    np.testing.assert_array_equal(lab, ol)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("dtype", ["f32", "f16", "f8"])
@pytest.mark.parametrize("N,D", [(1, 128), (15, 128), (1000, 128), (5000, 512), (20011, 768)])
def test_query_matches_oracle(mods, dtype, N, D):
    FlatIndex, _, _, ro = mods
    c = _corpus(N, D, seed=N + D)
    labels = np.arange(N, dtype=np.int64) * 3 + 5
    idx = FlatIndex(D, dtype)
    idx.add(c, labels)
    assert idx.count() == N
    stored = ro.normalize_rows(c, dtype)
    head = stored[: min(N, 64)]   # (fp8 rows: get() returns the vectors the rows represent, values x inverse norm)
    np.testing.assert_array_equal(idx.get(labels[: min(N, 64)]), head.represented() if dtype == "f8" else head.astype(np.float32))
    for Q, k in [(1, 1), (1, 10), (3, 10), (17, 10), (40, 5), (2, 24), (70, 10)]:
        q = _corpus(Q, D, seed=1000 + Q + k)
        _check(idx, ro, stored, labels, q, k)


@pytest.mark.parametrize("N,D,Q", [(20011, 512, 200), (70001, 768, 300), (40000, 256, 1024)])
def test_fp8_rows_on_the_score_gemm(mods, N, D, Q):
    """Round 4: from two 128-query tiles on, an fp8 index takes the strip score GEMM too (its e4m3 codes widened to f16 in the
    kernel's operand load; 64-byte rows in LDS, one ds_read_b64 per fragment) instead of re-streaming the index once per 64
    queries. ids + distance bits against the oracle's restatement of the fp8 rounding; the same queries through the
    streaming scan (option score_f8_gemm = 0) must return the same bits; 40 000 x 256 x 1024 = the threshold-filtered form."""
    from mmiss_amd import _lib

    FlatIndex, _, _, ro = mods
    c = _corpus(N, D, seed=N + D + 1)
    labels = np.arange(N, dtype=np.int64) * 3 + 5
    idx = FlatIndex(D, "f8")
    idx.add(c, labels)
    stored = ro.normalize_rows(c, "f8")
    q = _corpus(Q, D, seed=2000 + Q)
    _lib.prof_filter(None, 1); _lib.prof_reset(); _lib.prof_enable(True)
    if Q == 1024:
        _lib.set_option("score_filter", 2)
    try:
        lab, dist, cnt = idx.query(q, 10)
    finally:
        _lib.set_option("score_filter", 1)
        _lib.prof_enable(False)
    kern = {p["kernel"] for p in _lib.prof_read()}
    assert "score_gemm_f8" in kern and "scan_topk_f8" not in kern, kern
    if Q == 1024:
        assert "score_gemm_f8_sample" in kern, kern
    sub = np.arange(0, Q, max(1, Q // 24))
    ol, od, oc = ro.query(q[sub], stored, labels, 10)
    np.testing.assert_array_equal(lab[sub], ol)
    np.testing.assert_array_equal(dist[sub].view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(cnt[sub], oc)
    _lib.set_option("score_f8_gemm", 0)
    try:
        l2, d2, _ = idx.query(q, 10)
    finally:
        _lib.set_option("score_f8_gemm", 1)
    np.testing.assert_array_equal(l2, lab)
    np.testing.assert_array_equal(d2.view(np.uint32), dist.view(np.uint32))
    idx.close()


@pytest.mark.parametrize("dtype", ["f32", "f16", "f8"])
def test_large_k_paging_and_k_beyond_count(mods, dtype):
    FlatIndex, _, _, ro = mods
    N, D = 3000, 128
    c = _corpus(N, D, seed=11)
    labels = np.arange(N, dtype=np.int64)
    idx = FlatIndex(D, dtype)
    idx.add(c, labels)
    stored = ro.normalize_rows(c, dtype)
    q = _corpus(3, D, seed=12)
    for k in (25, 100, 1000):
        _check(idx, ro, stored, labels, q, k)
    small = FlatIndex(D, dtype)
    small.add(c[:6], labels[:6])
    # the UI's "All" asks for 1000 results of a 6-image collection (main.py:757): not an error
    _check(small, ro, stored[:6], labels[:6], q, 1000)
    _check(small, ro, stored[:6], labels[:6], q[:1], 10)


def test_ties_break_by_label_and_duplicates(mods):
    FlatIndex, _, _, ro = mods
    D = 128
    base = _corpus(40, D, seed=21)
    c = np.concatenate([base, base, base[:10], base])  # exact duplicates -> exactly equal distances
    N = c.shape[0]
    labels = np.arange(N, dtype=np.int64) + 100
    for dtype in ("f32", "f16"):
        idx = FlatIndex(D, dtype)
        idx.add(c, labels)
        stored = ro.normalize_rows(c, dtype)
        q = base[:5] + 0.01 * _corpus(5, D, seed=22)
        for k in (1, 3, 10, 24, 60):
            _check(idx, ro, stored, labels, q, k)
    # every row identical: the k smallest labels win
    same = np.tile(base[:1], (500, 1))
    idx = FlatIndex(D, "f32")
    idx.add(same, np.arange(500, dtype=np.int64))
    lab, dist, cnt = idx.query(base[:1], 10)
    np.testing.assert_array_equal(lab[0], np.arange(10))


@pytest.mark.parametrize("N,D,Q,k", [(100, 128, 17, 10), (4097, 128, 33, 10), (30000, 512, 256, 10), (50000, 512, 100, 24), (9000, 768, 64, 3)])
def test_batched_query_path_f16(mods, N, D, Q, k):
    """Q > 16 on f16 rows takes the score-GEMM (group max) + select path; same bit-exact contract."""
    FlatIndex, _, _, ro = mods
    c = _corpus(N, D, seed=N + Q)
    # plant near-duplicates of some queries so that several top hits share a 16-row group
    q = _corpus(Q, D, seed=7 * N + Q)
    rng = np.random.Generator(np.random.Philox(N))
    for j in range(min(Q, 8)):
        at = int(rng.integers(0, max(1, N - 40)))
        c[at:at + 5] = q[j] + 0.05 * _corpus(5, D, seed=j + 1)
    labels = np.arange(N, dtype=np.int64) + 11
    idx = FlatIndex(D, "f16")
    idx.add(c, labels)
    _check(idx, ro, ro.normalize_rows(c, "f16"), labels, q, k)


@pytest.mark.parametrize("strip", [1, 3, 5, 32])
def test_score_gemm_strips(mods, strip):
    """Q > 128 scores through the strip-persistent 256x256 kernel: a workgroup walks `strip` consecutive 256-row tiles
    of the index as one K-tile stream. Any strip length (dividing the tile count or not, longer than it or not) must
    give the oracle's answer; 35000 rows = 137 tiles, the last one partly padding."""
    FlatIndex, _, _, ro = mods
    from mmiss_amd import _lib

    N, D, Q, k = 35000, 512, 300, 10
    c = _corpus(N, D, seed=77)
    labels = np.arange(N, dtype=np.int64) * 2 + 1
    idx = FlatIndex(D, "f16")
    idx.add(c, labels)
    stored = ro.normalize_rows(c, "f16")
    q = _corpus(Q, D, seed=78)
    q[:40] = c[1000:1040] + 0.01 * q[:40]  # some queries with a clear nearest row
    _lib.set_option("score_strip", strip)
    try:
        _check(idx, ro, stored, labels, q, k)
        _lib.set_option("score_filter", 2)       # the same strips through the threshold-filtered selection (137 tiles >= 128)
        _check(idx, ro, stored, labels, q, k)
    finally:
        _lib.set_option("score_strip", 0)
        _lib.set_option("score_filter", 1)


def test_batched_path_with_exact_duplicates(mods):
    """Many identical rows spread over several 16-row groups: group maxima tie exactly, the k smallest labels win."""
    FlatIndex, _, _, ro = mods
    N, D, Q = 3000, 128, 20
    c = _corpus(N, D, seed=5)
    q = _corpus(Q, D, seed=6)
    dup = np.concatenate([np.arange(40, 100), np.arange(700, 1000, 7), np.arange(2500, 2600)])
    c[dup] = q[3]                    # > 200 rows identical to query 3
    c[np.arange(1200, 1300)] = c[5]  # another duplicate cluster, not aligned with any query
    labels = np.arange(N, dtype=np.int64)
    idx = FlatIndex(D, "f16")
    idx.add(c, labels)
    stored = ro.normalize_rows(c, "f16")
    for k in (1, 10, 24):
        _check(idx, ro, stored, labels, q, k)


def test_adversarial_order_ascending_similarity(mods):
    """Rows sorted so every later row beats all earlier ones: the running top-k' list is rewritten constantly."""
    FlatIndex, _, _, ro = mods
    N, D = 6000, 128
    c = _corpus(N, D, seed=31)
    q = _corpus(1, D, seed=32)
    sims = (c / np.linalg.norm(c, axis=1, keepdims=True)) @ (q[0] / np.linalg.norm(q[0]))
    c = c[np.argsort(sims)]
    labels = np.arange(N, dtype=np.int64)
    idx = FlatIndex(D, "f16")
    idx.add(c, labels)
    _check(idx, ro, ro.normalize_rows(c, "f16"), labels, q, 10)


@pytest.mark.parametrize("dtype", ["f32", "f8"])
def test_incremental_add_remove_update(mods, dtype):
    FlatIndex, _, _, ro = mods
    D = 128
    c = _corpus(900, D, seed=41)
    labels = np.arange(900, dtype=np.int64)
    idx = FlatIndex(D, dtype)
    for r0 in range(0, 900, 250):
        idx.add(c[r0:r0 + 250], labels[r0:r0 + 250])
    q = _corpus(4, D, seed=42)
    stored = ro.normalize_rows(c, dtype)
    _check(idx, ro, stored, labels, q, 10)
    # labels must keep increasing (they are the tie-break order)
    with pytest.raises(RuntimeError):
        idx.add(c[:1], np.array([5], dtype=np.int64))
    drop = np.array([0, 17, 450, 899, 12345], dtype=np.int64)
    assert idx.remove(drop) == 4
    keep = np.setdiff1d(labels, drop)
    assert idx.count() == keep.shape[0]
    np.testing.assert_array_equal(idx.labels(), keep)
    _check(idx, ro, stored[keep], keep, q, 10)
    newv = _corpus(2, D, seed=43)
    idx.update(np.array([3, 600], dtype=np.int64), newv)
    c2 = c.copy()
    c2[3], c2[600] = newv[0], newv[1]
    _check(idx, ro, ro.normalize_rows(c2, dtype)[keep], keep, q, 10)
    idx.clear()
    assert idx.count() == 0


def test_empty_index_and_errors(mods):
    FlatIndex, _, _, ro = mods
    idx = FlatIndex(128, "f32")
    lab, dist, cnt = idx.query(_corpus(2, 128, seed=1), 5)
    assert (lab == -1).all() and np.isinf(dist).all() and (cnt == 0).all()
    with pytest.raises(RuntimeError):
        FlatIndex(100, "f32")  # dim must be a multiple of 128
    with pytest.raises(ValueError):
        idx.add(np.zeros((2, 64), np.float32), np.arange(2))


@pytest.mark.parametrize("dtype", ["f32", "f16", "f8"])
def test_zero_and_non_finite_vectors_are_never_results(mods, dtype):
    """A zero-norm, NaN or Inf vector has no direction: its normalisation is 0/0, every distance to or from it is NaN. Such a
    ROW is never returned (the count says how many results there are); such a QUERY has no results (count 0, labels -1,
    distances +inf). Same in both oracles (retrieval_oracle.py / .c); the reference's chromadb call has no such case
    (backend/app/utils.py:88-99 always hands over a CLIP embedding)."""
    import warnings

    from oracle import retrieval_oracle_c as rc

    FlatIndex, _, _, ro = mods
    N, D = 300, 128
    c = _corpus(N, D, seed=71)
    c[5] = 0.0
    c[17, 3] = np.nan
    c[40, 9] = np.inf
    c[41, 9] = -np.inf
    labels = np.arange(1000, 1000 + N, dtype=np.int64)
    q = _corpus(7, D, seed=72)
    q[2] = 0.0
    q[4, 100] = np.nan
    q[6, 0] = np.inf
    idx = FlatIndex(D, dtype)
    idx.add(c, labels)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")   # (numpy: 0/0, inf/inf in the oracle's normalisation)
        stored = ro.normalize_rows(c, dtype)
        assert np.isnan(stored[[5, 17, 40, 41]].astype(np.float32)).any(axis=1).all()
        for k in (1, 10, N - 4, N, N + 20):
            lab, dist, cnt = idx.query(q, k)
            ol, od, oc = ro.query(q, stored, labels, k)
            cl, cd, cc = rc.query(q, stored, labels, k)
            np.testing.assert_array_equal(ol, cl)
            np.testing.assert_array_equal(oc, cc)
            np.testing.assert_array_equal(od.view(np.uint32), cd.view(np.uint32))
            np.testing.assert_array_equal(cnt, oc)
            np.testing.assert_array_equal(lab, ol)
            np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
            assert (cnt[[2, 4, 6]] == 0).all() and (cnt[[0, 1, 3, 5]] == min(k, N - 4)).all()
            assert not np.isin(lab, labels[[5, 17, 40, 41]]).any()
    # the batched (score GEMM) path: the same queries repeated past its threshold
    qq = np.concatenate([q] * 12)[:80]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lab, dist, cnt = idx.query(qq, 10)
        ol, od, oc = ro.query(qq, stored, labels, 10)
    np.testing.assert_array_equal(cnt, oc)
    np.testing.assert_array_equal(lab, ol)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))


def test_device_tensors_in_and_out(mods):
    import torch

    FlatIndex, _, _, ro = mods
    N, D = 2048, 512
    c = _corpus(N, D, seed=51)
    labels = np.arange(N, dtype=np.int64)
    idx = FlatIndex(D, "f16")
    idx.add(torch.from_numpy(c).cuda(), labels)
    q = _corpus(9, D, seed=52)
    lab, dist, cnt = idx.query(torch.from_numpy(q).cuda(), 10)
    assert lab.is_cuda and dist.is_cuda
    ol, od, oc = ro.query(q, ro.normalize_rows(c, "f16"), labels, 10)
    np.testing.assert_array_equal(lab.cpu().numpy(), ol)
    np.testing.assert_array_equal(dist.cpu().numpy().view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("dtype", ["f16", "f8"])
def test_save_load_roundtrip(mods, tmp_path, dtype):
    FlatIndex, _, _, ro = mods
    N, D = 700, 128
    c = _corpus(N, D, seed=61)
    labels = np.arange(N, dtype=np.int64) * 2
    idx = FlatIndex(D, dtype)
    idx.add(c, labels)
    p = tmp_path / "shard.mmiss"
    idx.save(str(p))
    again = FlatIndex(D, dtype)
    again.load(str(p))
    assert again.count() == N
    q = _corpus(3, D, seed=62)
    a = idx.query(q, 10)
    b = again.query(q, 10)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    wrong = FlatIndex(D, "f32")
    with pytest.raises(RuntimeError):
        wrong.load(str(p))


@pytest.mark.parametrize("dtype", ["f16", "f8"])
def test_truncated_file_leaves_the_index_as_it_was(mods, tmp_path, dtype):
    """mmiss_index_load checks the header's row count against the file's length before it overwrites anything: a cut-off
    file is refused and the index answers exactly as before (fp8: codes and inverse norms still belong together)."""
    FlatIndex, _, _, ro = mods
    D = 128
    c = _corpus(500, D, seed=63)
    labels = np.arange(500, dtype=np.int64)
    idx = FlatIndex(D, dtype)
    idx.add(c, labels)
    other = FlatIndex(D, dtype)
    other.add(_corpus(400, D, seed=64), np.arange(400, dtype=np.int64) + 7)
    p = tmp_path / "shard.mmiss"
    other.save(str(p))
    blob = p.read_bytes()
    q = _corpus(3, D, seed=65)
    before = idx.query(q, 10)
    for cut in (len(blob) - 1, len(blob) // 2, 40):
        (tmp_path / "cut.mmiss").write_bytes(blob[:cut])
        with pytest.raises(RuntimeError):
            idx.load(str(tmp_path / "cut.mmiss"))
        assert idx.count() == 500
        for x, y in zip(before, idx.query(q, 10)):
            np.testing.assert_array_equal(x, y)
        _check(idx, ro, ro.normalize_rows(c, dtype), labels, q, 10)
    idx.load(str(p))                      # the whole file still loads
    assert idx.count() == 400


def test_removing_every_row_of_an_fp8_index_then_adding_again(mods):
    """mmiss_index_remove down to zero rows keeps the fp8 invariant (inverse norms zero behind `count`); rows added afterwards
    answer like a fresh index."""
    FlatIndex, _, _, ro = mods
    D = 128
    c = _corpus(300, D, seed=66)
    labels = np.arange(300, dtype=np.int64)
    idx = FlatIndex(D, "f8")
    idx.add(c, labels)
    assert idx.remove(labels) == 300 and idx.count() == 0
    lab, dist, cnt = idx.query(_corpus(2, D, seed=67), 5)
    assert (lab == -1).all() and (cnt == 0).all()
    c2 = _corpus(40, D, seed=68)
    l2 = np.arange(40, dtype=np.int64) + 1000
    idx.add(c2, l2)
    _check(idx, ro, ro.normalize_rows(c2, "f8"), l2, _corpus(4, D, seed=69), 10)
    _check(idx, ro, ro.normalize_rows(c2, "f8"), l2, _corpus(200, D, seed=70), 10)   # the score GEMM's 256-row tile reads behind count


def test_blend_matches_oracle_and_reference_formula(mods):
    from oracle import clip_oracle as co

    _, blend, _, ro = mods
    img = _corpus(33, 512, seed=71)
    txt = _corpus(33, 512, seed=72)
    for w in (0.5, 0.3, 0.0, 1.0, 1.7):  # the backend does not clamp weight_image (main.py:299)
        out = blend(img, txt, w)
        np.testing.assert_array_equal(out.view(np.uint32), ro.blend(img, txt, w).view(np.uint32))
        # numpy's own float32 evaluation of main.py:852-860 differs only in the last bits
        np.testing.assert_allclose(out, co.blend_reference(img, txt, w), rtol=0, atol=1e-6)


def test_merge_topk_equals_unsharded(mods):
    FlatIndex, _, merge_topk, ro = mods
    N, D, S, k = 4000, 128, 4, 10
    c = _corpus(N, D, seed=81)
    labels = np.arange(N, dtype=np.int64)
    q = _corpus(6, D, seed=82)
    per = N // S
    ds, ls = [], []
    for s in range(S):
        sh = FlatIndex(D, "f16")
        sh.add(c[s * per:(s + 1) * per], labels[s * per:(s + 1) * per])
        l, d, _ = sh.query(q, k)
        ds.append(d)
        ls.append(l)
    ml, md, mc = merge_topk(np.stack(ds), np.stack(ls))
    ol, od, oc = ro.query(q, ro.normalize_rows(c, "f16"), labels, k)
    np.testing.assert_array_equal(ml, ol)
    np.testing.assert_array_equal(md.view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(mc, oc)


def test_config3_text_queries_against_image_index():
    """BASELINE configs[2] at reduced size: random 77-token prompts -> text tower -> top-10 over a 200k x 512 index."""
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from mmiss_amd.index import FlatIndex
    from oracle import clip_oracle as co
    from oracle import retrieval_oracle as ro

    s = co.TINY
    W = co.init_weights(s, seed=2)
    enc = ClipEncoder(ClipShape.from_any(s), max_batch_text=64, max_batch_image=8)
    enc.load_state_dict(W)
    ids = co.synthetic_text_ids(200, s.t_ctx, s.t_vocab, s.eos_token_id, seed=3)
    q = enc.encode_text(ids)  # [200, 128], chunked over max_batch_text
    assert (1 - (q * co.embed_texts(ids, W, s)).sum(1)).max() < 1e-3
    N = 200_000
    c = _corpus(N, s.proj_dim, seed=4)
    labels = np.arange(N, dtype=np.int64)
    idx = FlatIndex(s.proj_dim, "f16")
    idx.add(c, labels)
    lab, dist, cnt = idx.query(q, 10)
    # the oracle on the SAME query bits: ids and distances bit-exact (checked on a subset to keep the CPU side short)
    sub = np.arange(0, 200, 17)
    ol, od, oc = ro.query(q[sub], ro.normalize_rows(c, "f16"), labels, 10)
    np.testing.assert_array_equal(lab[sub], ol)
    np.testing.assert_array_equal(dist[sub].view(np.uint32), od.view(np.uint32))
    assert (np.diff(dist, axis=1) >= 0).all() and (cnt == 10).all()


def test_config4_sharded_multimodal_batch():
    """BASELINE configs[3] at reduced size: 8 logical row shards, batched (image+text average) queries, per-shard
    top-10, shard merge == the unsharded index, bit for bit."""
    import mmiss_amd  # noqa: F401
    from mmiss_amd.index import FlatIndex, blend, merge_topk
    from oracle import retrieval_oracle as ro

    N, D, S, Q, k = 160_000, 512, 8, 96, 10
    c = _corpus(N, D, seed=14)
    labels = np.arange(N, dtype=np.int64)
    qi, qt = _corpus(Q, D, seed=15), _corpus(Q, D, seed=16)
    q = blend(qi, qt, 0.5)
    np.testing.assert_array_equal(q.view(np.uint32), ro.blend(qi, qt, 0.5).view(np.uint32))
    whole = FlatIndex(D, "f16")
    whole.add(c, labels)
    full = whole.query(q, k)
    per = N // S
    ls, ds = [], []
    for s_ in range(S):
        sh = FlatIndex(D, "f16")
        sh.add(c[s_ * per:(s_ + 1) * per], labels[s_ * per:(s_ + 1) * per])
        l, d, _ = sh.query(q, k)
        ls.append(l)
        ds.append(d)
    ml, md, mc = merge_topk(np.stack(ds), np.stack(ls))
    np.testing.assert_array_equal(ml, full[0])
    np.testing.assert_array_equal(md.view(np.uint32), full[1].view(np.uint32))
    sub = np.arange(0, Q, 13)
    ol, od, _ = ro.query(q[sub], ro.normalize_rows(c, "f16"), labels, k)
    np.testing.assert_array_equal(ml[sub], ol)


def test_full_size_properties_1m_rows():
    """Size-independent properties at 1M x 512 (oracle too slow there): every row finds itself first at distance ~0,
    results are sorted, counts are k, a query batch equals the same queries one by one, f16 and Q>16 paths agree."""
    import torch
    import mmiss_amd  # noqa: F401
    from mmiss_amd.index import FlatIndex

    N, D = 1_000_000, 512
    g = torch.Generator(device="cuda").manual_seed(1)
    rows = torch.randn(N, D, device="cuda", generator=g)
    idx = FlatIndex(D, "f16", capacity=N)
    idx.add(rows, np.arange(N, dtype=np.int64))
    assert idx.count() == N
    probe = torch.randint(0, N, (48,), generator=torch.Generator().manual_seed(2))
    q = rows[probe.cuda()]
    lab, dist, cnt = idx.query(q, 10)                      # Q = 48 -> score-GEMM path
    lab, dist = lab.cpu().numpy(), dist.cpu().numpy()
    assert (lab[:, 0] == probe.numpy()).all()
    assert np.abs(dist[:, 0]).max() < 2e-3                 # f16 storage rounding of a unit vector
    assert (np.diff(dist, axis=1) >= 0).all() and (cnt.cpu().numpy() == 10).all()
    for j in range(0, 48, 9):                              # Q = 1 -> streaming scan path: identical results
        l1, d1, _ = idx.query(q[j:j + 1], 10)
        np.testing.assert_array_equal(l1.cpu().numpy()[0], lab[j])
        np.testing.assert_array_equal(d1.cpu().numpy()[0].view(np.uint32), dist[j].view(np.uint32))
    l16, d16, _ = idx.query(q[:16], 10)                    # Q = 16 -> scan path with a full MFMA tile of queries
    np.testing.assert_array_equal(l16.cpu().numpy(), lab[:16])


@pytest.mark.parametrize("S,k,Q", [(8, 1000, 3), (20, 1000, 2), (5, 2040, 2), (64, 10, 33)])
def test_merge_topk_any_shard_count(mods, S, k, Q):
    """The UI's "All" = 1000 hits (main.py:757) on 8 shards is 8000 entries per query (one LDS pass); 20 shards x 1000 and
    5 x 2040 go through several merge levels. Against the oracle's merge, incl. empty slots and ties across shards."""
    _, _, merge_topk, ro = mods
    rng = np.random.Generator(np.random.Philox(S * k + Q))
    dist = np.sort(rng.random((S, Q, k), dtype=np.float32), axis=2)
    labels = np.empty((S, Q, k), dtype=np.int64)
    for s_ in range(S):   # disjoint label ranges per shard, ascending within a list like a real per-shard result
        labels[s_] = np.sort(rng.choice(10 * k, size=(Q, k), replace=True), axis=1) + s_ * 10 * k
    dist[1, 0, :5] = dist[0, 0, :5]                      # exact ties across shards: the smaller label wins
    labels[S - 1, :, k - 7:] = -1                         # a shard with fewer than k rows
    dist[S - 1, :, k - 7:] = np.inf
    ml, md, mc = merge_topk(dist, labels)
    ol, od, oc = ro.merge_shards(dist, labels, k)
    np.testing.assert_array_equal(ml, ol)
    np.testing.assert_array_equal(md.view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(mc, oc)


@pytest.mark.parametrize("device_io", [False, True])
@pytest.mark.parametrize("force_widen", [0, 1])
def test_query_begin_end_equals_query(mods, device_io, force_widen):
    """mmiss_index_query_begin / _end (the wait split off, other GPU work queued in between) returns the bits query() and the
    oracle return — also when EVERY query goes through the widen pass inside _end (guard_force), behind work the caller
    queued on the same stream after _begin."""
    import torch
    from mmiss_amd import _lib

    FlatIndex, _, _, ro = mods
    N, D, Q, k = 30000, 512, 40, 10
    c = _corpus(N, D, seed=61)
    labels = np.arange(N, dtype=np.int64) * 2 + 5
    idx = FlatIndex(D, "f16")
    idx.add(c, labels)
    q = _corpus(Q, D, seed=62)
    ol, od, oc = ro.query(q, ro.normalize_rows(c, "f16"), labels, k)
    qin = torch.from_numpy(q).cuda() if device_io else q
    busy = torch.randn(2048, 2048, device="cuda")
    _lib.set_option("guard_force", force_widen)
    try:
        before = idx.guard_stats()["widened"]
        h = idx.query_begin(qin, k)
        for _ in range(20):                      # the "next batch's encode": queued behind the first pass, ahead of a widen pass
            busy = busy @ busy * 1e-3
        with pytest.raises(RuntimeError, match="still open"):
            idx.add(c[:4], np.arange(4, dtype=np.int64) + 10 ** 9)
        with pytest.raises(RuntimeError, match="still open"):
            idx.query(qin, k)
        assert idx.count() == N                  # count / labels / guard_stats stay callable
        lab, dist, cnt = h.result()
        assert h.result()[0] is lab              # idempotent
        assert idx.guard_stats()["widened"] - before == (Q if force_widen else 0)
    finally:
        _lib.set_option("guard_force", 0)
    if device_io:
        assert lab.is_cuda
        lab, dist, cnt = lab.cpu().numpy(), dist.cpu().numpy(), cnt.cpu().numpy()
    np.testing.assert_array_equal(lab, ol)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(cnt, oc)
    with pytest.raises(RuntimeError, match="no query was begun"):
        _lib.check(idx._lib.mmiss_index_query_end(idx._h))
    _check(idx, ro, ro.normalize_rows(c, "f16"), labels, q, k)   # and the index is usable again
    torch.cuda.synchronize()


@pytest.mark.parametrize("force_widen", [0, 1])
@pytest.mark.parametrize("device_io", [False, True])
def test_query_next_chains_batches_like_separate_queries(mods, device_io, force_widen):
    """FlatIndex.query_next(pending, queries, k) = pending.result() + query_begin(queries, k) with nothing but two C calls between
    them (round 6: the serving loop of bench.py): a chain of batches of different sizes gives, batch for batch, what query()
    gives — ids, distance bits, counts — with and without the widen pass, host and device tensors."""
    import torch
    from mmiss_amd import _lib

    FlatIndex, _, _, ro = mods
    N, D, k = 30000, 256, 10
    c = _corpus(N, D, seed=91)
    labels = np.arange(N, dtype=np.int64) * 2
    idx = FlatIndex(D, "f16")
    idx.add(c, labels)
    stored = ro.normalize_rows(c, "f16")
    batches = [_corpus(Q, D, seed=92 + i) for i, Q in enumerate((200, 7, 256, 1, 130))]
    batches[2][:50] = c[100:150]                       # some queries that are rows of the index
    _lib.set_option("guard_force", force_widen)
    try:
        to_dev = (lambda a: torch.from_numpy(a).cuda()) if device_io else (lambda a: a)
        pending = idx.query_begin(to_dev(batches[0]), k)
        got = []
        for b in batches[1:]:
            res, pending = idx.query_next(pending, to_dev(b), k)
            got.append(res)
        got.append(pending.result())
    finally:
        _lib.set_option("guard_force", 0)
    for b, (lab, dist, cnt) in zip(batches, got):
        if device_io:
            lab, dist, cnt = lab.cpu().numpy(), dist.cpu().numpy(), cnt.cpu().numpy()
        ol, od, oc = ro.query(b, stored, labels, k)
        np.testing.assert_array_equal(lab, ol)
        np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
        np.testing.assert_array_equal(cnt, oc)
    _check(idx, ro, stored, labels, batches[1], k)      # and the index is free again


def test_an_abandoned_query_handle_does_not_wedge_the_index(mods, tmp_path):
    """a query opened with query_begin whose handle is lost (an exception in the serving loop before
    result()) used to leave every other entry point refusing with MMISS_ERR_STATE until destroy. The handle now aborts the
    query when it is dropped, leaves a `with` block, or on abort(); mmiss_index_query_abort itself is a no-op without an
    open query; add / save / query work afterwards."""
    from mmiss_amd import _lib

    FlatIndex, _, _, ro = mods
    N, D, k = 5000, 256, 10
    c = _corpus(N, D, seed=81)
    labels = np.arange(N, dtype=np.int64)
    idx = FlatIndex(D, "f16")
    idx.add(c, labels)
    q = _corpus(8, D, seed=82)
    _lib.check(idx._lib.mmiss_index_query_abort(idx._h))        # nothing open: fine
    h = idx.query_begin(q, k)
    with pytest.raises(RuntimeError, match="still open"):
        idx.save(str(tmp_path / "i.bin"))
    del h                                                       # the lost handle
    idx.add(c[:3] * 2.0, np.arange(3, dtype=np.int64) + N)
    idx.save(str(tmp_path / "i.bin"))
    with pytest.raises(ZeroDivisionError):
        with idx.query_begin(q, k):                             # an exception between the two halves
            1 / 0
    h = idx.query_begin(q, k)
    h.abort()
    h.abort()                                                   # idempotent
    with pytest.raises(RuntimeError, match="no query was begun"):
        _lib.check(idx._lib.mmiss_index_query_end(idx._h))
    c2 = np.concatenate([c, c[:3] * 2.0])
    lab2 = np.concatenate([labels, np.arange(3, dtype=np.int64) + N])
    _check(idx, ro, ro.normalize_rows(c2, "f16"), lab2, q, k)
    h = idx.query_begin(q, k)                                   # close() with a query still open
    idx.close()
    del h


def test_a_stale_query_handle_cannot_abort_a_newer_query(mods):
    """PendingQuery.abort() / __del__ used to abort whatever query was open on the index. A handle that
    outlives its query (aborted through the index; kept alive by a traceback or a cycle) must not abort the query someone
    else has begun since — its abort() is a no-op and its result() raises; collecting it while the index lock is held by the
    same thread must not deadlock (re-entrant lock)."""
    FlatIndex, _, _, ro = mods
    N, D, k = 3000, 256, 10
    c = _corpus(N, D, seed=91)
    labels = np.arange(N, dtype=np.int64)
    idx = FlatIndex(D, "f16")
    idx.add(c, labels)
    q1, q2 = _corpus(4, D, seed=92), _corpus(6, D, seed=93)
    stale = idx.query_begin(q1, k)
    idx.abort_query()                       # the serving loop gives the open query up through the index
    fresh = idx.query_begin(q2, k)          # ... and someone else begins the next one
    with idx._call_lock:                    # collected by the GC inside a locked call of the same thread: no self-deadlock,
        stale.__del__()                     # and it must NOT close `fresh`
    del stale
    lab, dist, cnt = fresh.result()
    ol, od, oc = ro.query(q2, ro.normalize_rows(c, "f16"), labels, k)
    np.testing.assert_array_equal(lab, ol)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(cnt, oc)
    stale = idx.query_begin(q1, k)
    idx.abort_query()
    with pytest.raises(RuntimeError, match="aborted"):
        stale.result()                      # a result that was never computed is not handed out
    _check(idx, ro, ro.normalize_rows(c, "f16"), labels, q1, k)


@pytest.mark.parametrize("dtype,D,Q", [("f32", 768, 40), ("f32", 768, 70), ("f32", 1024, 40), ("f32", 1024, 70), ("f16", 2048, 40),
                                       ("f8", 2048, 40), ("f32", 1920, 33)])
def test_widen_scan_fits_the_lds_at_every_admitted_dim(mods, dtype, D, Q):
    """the streaming threshold pass of the widen pass sized its query block as 64 queries whenever more than
    32 were flagged, without the 160 KB bound plan_scan applies — f32 rows at the reference's D = 768 need 64 x 3088 B = 193 KB
    and the launch failed. Every query forced through the widen pass (guard_force) at the dims where 64 (or 32) staged queries
    do not fit; the score-GEMM form of the pass is switched off so that the scan takes them."""
    from mmiss_amd import _lib

    FlatIndex, _, _, ro = mods
    N, k = 3001, 10
    c = _corpus(N, D, seed=D + Q)
    labels = np.arange(N, dtype=np.int64) * 2 + 1
    idx = FlatIndex(D, dtype)
    idx.add(c, labels)
    q = _corpus(Q, D, seed=7 * D + Q)
    _lib.set_option("guard_force", 1)
    _lib.set_option("sweep_gemm_min_q", 1 << 20)
    try:
        before = idx.guard_stats()
        _check(idx, ro, ro.normalize_rows(c, dtype), labels, q, k)
        after = idx.guard_stats()
    finally:
        _lib.set_option("guard_force", 0)
        _lib.set_option("sweep_gemm_min_q", 64)
    assert after["widened"] - before["widened"] == Q and after["exhaustive"] == before["exhaustive"]


def test_create_refuses_dims_whose_query_block_cannot_fit(mods):
    """The scan stages 16 queries beside its lists in the CU's 160 KB: f32 rows up to dim 1920, f16 / fp8 rows up to 3968 —
    refused at create time (it used to be a failed launch at the first query)."""
    FlatIndex = mods[0]
    for dtype, D in [("f32", 2048), ("f32", 4096), ("f16", 4096), ("f8", 4096)]:
        with pytest.raises(RuntimeError, match="LDS"):
            FlatIndex(D, dtype)
    for dtype, D in [("f32", 1920), ("f16", 3968)]:
        FlatIndex(D, dtype).close()


def test_fp8_rows_rank_like_the_f32_rows_up_to_their_quantisation(mods):
    """MMISS_F8 storage (e4m3 codes of 128 x + one inverse norm per row, include/mmiss.h): the index is exact with respect to the
    rows it REPRESENTS (the tests above, against the oracle's restatement of the rounding), and what it returns is a cosine
    distance (round 5: the codes alone have norms 0.97 .. 1.03): the represented rows have unit norm, a row queried with itself
    comes back at distance ~1e-7, with its unquantised original within the e4m3 rounding (< 4e-3). Against the unquantised
    rows: a planted near-duplicate is still found first, the top-10 overlaps the f32 index's in >= 8 of 10 on average, and
    the distances agree to what the rows' directions moved by."""
    FlatIndex, _, _, ro = mods
    N, D = 20000, 512
    c = _corpus(N, D, seed=71)
    q = _corpus(16, D, seed=72)
    c[777] = q[3] + 0.05 * _corpus(1, D, seed=73)[0]
    labels = np.arange(N, dtype=np.int64)
    s8, s32 = ro.normalize_rows(c, "f8"), ro.normalize_rows(c, "f32")
    cosrow = (s8.astype(np.float64) * s32).sum(1) / np.linalg.norm(s8.astype(np.float64), axis=1)
    assert (1 - cosrow).max() < 4e-3 and abs(np.linalg.norm(s8.astype(np.float64), axis=1) - 1).max() < 0.03
    assert abs(np.linalg.norm(s8.represented().astype(np.float64), axis=1) - 1).max() < 3e-7
    i8, i32 = FlatIndex(D, "f8"), FlatIndex(D, "f32")
    i8.add(c, labels)
    i32.add(c, labels)
    l8, d8, _ = i8.query(q, 10)
    l32, d32, _ = i32.query(q, 10)
    assert l8[3, 0] == 777 and l32[3, 0] == 777
    overlap = [len(set(l8[i]) & set(l32[i])) for i in range(16)]
    assert min(overlap) >= 7 and np.mean(overlap) >= 8.5, overlap
    assert np.abs(d8 - d32).max() < 2e-2
    # cosine distances: a represented row against itself, and the unquantised originals against their rows
    own = np.arange(100, 164)
    ls, ds, _ = i8.query(i8.get(labels[own]), 1)
    np.testing.assert_array_equal(ls[:, 0], labels[own])
    assert np.abs(ds[:, 0]).max() < 5e-7, ds[:, 0]
    lo, do, _ = i8.query(c[own], 1)
    np.testing.assert_array_equal(lo[:, 0], labels[own])
    assert (do[:, 0] >= -1e-7).all() and do[:, 0].max() < 4e-3, do[:, 0]
    np.testing.assert_allclose(do[:, 0], 1 - cosrow[own], atol=2e-6)

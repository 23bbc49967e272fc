"""The retrieval oracle: numpy restatement == plain-C restatement bit for bit, both close to a straightforward
float64 evaluation; tie-break, padding, blend and shard-merge semantics."""
import numpy as np
import pytest

from oracle import retrieval_oracle as ro


def _rand(n, d, seed):
    return np.random.Generator(np.random.Philox(seed)).standard_normal((n, d), dtype=np.float32)


@pytest.fixture(scope="module")
def rc():
    from oracle import retrieval_oracle_c as rc

    rc._lib()
    return rc


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("N,D,Q,k", [(1, 128, 1, 1), (50, 128, 3, 10), (3000, 512, 5, 10), (700, 768, 2, 100)])
def test_numpy_and_c_oracles_agree_bitwise(rc, dtype, N, D, Q, k):
    c, q = _rand(N, D, N + D), _rand(Q, D, Q + k)
    labels = np.arange(N, dtype=np.int64) * 7 + 3
    a, b = ro.normalize_rows(c, dtype), rc.normalize_rows(c, dtype)
    assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))
    for x, y in zip(ro.query(q, a, labels, k), rc.query(q, b, labels, k)):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("dtype", ["f32", "f16", "f8"])
def test_zero_and_non_finite_vectors_have_no_distance(rc, dtype):
    """0/0 in the normalisation: such a row is never a result, such a query has none — in both restatements."""
    import warnings

    N, D = 60, 128
    c, q = _rand(N, D, 5), _rand(4, D, 6)
    c[3] = 0.0
    c[9, 1] = np.nan
    c[11, 2] = np.inf
    q[1] = 0.0
    q[3, 7] = -np.inf
    labels = np.arange(N, dtype=np.int64) + 100
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        stored = ro.normalize_rows(c, dtype)
        for k in (5, N - 3, N + 4):
            ol, od, oc = ro.query(q, stored, labels, k)
            cl, cd, cc = rc.query(q, stored, labels, k)
            np.testing.assert_array_equal(ol, cl)
            np.testing.assert_array_equal(oc, cc)
            np.testing.assert_array_equal(od.view(np.uint32), cd.view(np.uint32))
            assert list(oc) == [min(k, N - 3), 0, min(k, N - 3), 0]
            assert not np.isin(ol, labels[[3, 9, 11]]).any() and not np.isnan(od).any()
            assert (ol[1] == -1).all() and np.isinf(od[1]).all()


def test_canonical_distance_is_close_to_plain_float64():
    c, q = _rand(2000, 512, 1), _rand(4, 512, 2)
    d = ro.distances(q, ro.normalize_rows(c))
    c64, q64 = c.astype(np.float64), q.astype(np.float64)
    ref = 1 - (q64 @ c64.T) / (np.linalg.norm(q64, axis=1)[:, None] * np.linalg.norm(c64, axis=1)[None])
    np.testing.assert_allclose(d, ref, atol=2e-7)


def test_ties_break_by_label_and_short_results_are_padded():
    base = _rand(10, 128, 5)
    c = np.concatenate([base, base, base])
    labels = np.array(list(range(100, 110)) + list(range(10)) + list(range(50, 60)), dtype=np.int64)
    order = np.argsort(labels)
    c, labels = c[order], labels[order]
    lab, dist, cnt = ro.query(base[:2], ro.normalize_rows(c), labels, 3)
    assert list(lab[0]) == [0, 50, 100] and list(lab[1]) == [1, 51, 101]
    assert dist[0, 0] == dist[0, 1] == dist[0, 2]
    lab, dist, cnt = ro.query(base[:1], ro.normalize_rows(c[:4]), labels[:4], 10)
    assert cnt[0] == 4 and (lab[0, 4:] == -1).all() and np.isinf(dist[0, 4:]).all()
    lab, dist, cnt = ro.query(base[:1], ro.normalize_rows(c[:0].reshape(0, 128)), labels[:0], 5)
    assert cnt[0] == 0 and (lab == -1).all()


def test_similarity_conversion_matches_reference_formula():
    assert ro.similarity_from_distance(np.array([0.0, 1.0, 2.0], np.float32)) == [1.0, 0.5, 0.0]


def test_blend_close_to_reference_numpy_formula_and_c(rc):
    from oracle import clip_oracle as co

    i, t = _rand(9, 512, 11), _rand(9, 512, 12)
    for w in (0.5, 0.25, 0.0, 1.0, -0.3):
        np.testing.assert_array_equal(ro.blend(i, t, w).view(np.uint32), rc.blend(i, t, w).view(np.uint32))
        np.testing.assert_allclose(ro.blend(i, t, w), co.blend_reference(i, t, w), atol=1e-6)


def test_merge_shards_equals_unsharded():
    c, q = _rand(900, 128, 21), _rand(4, 128, 22)
    labels = np.arange(900, dtype=np.int64)
    stored = ro.normalize_rows(c, "f16")
    full = ro.query(q, stored, labels, 10)
    parts = [ro.query(q, stored[s::3], labels[s::3], 10) for s in range(3)]
    ml, md, mc = ro.merge_shards(np.stack([p[1] for p in parts]), np.stack([p[0] for p in parts]), 10)
    np.testing.assert_array_equal(ml, full[0])
    np.testing.assert_array_equal(md, full[1])


def test_fp8_storage_rounding_is_e4m3_of_128x():
    """MMISS_F8 rows (include/mmiss.h): stored value = decode(e4m3(128 y)) / 128, round to nearest even. Hand-checked points,
    the value grid, idempotence, and how far a stored unit row is from the unquantised one."""
    from oracle import fp8_oracle
    from oracle import retrieval_oracle as ro

    # one row whose normalised form is known: (0.6, 0.8, 0, ...): 76.8 -> 80 (grid step 8 in [64, 128)), 102.4 -> 104
    x = np.zeros((1, 128), np.float32)
    x[0, 0], x[0, 1] = 3.0, 4.0
    s = ro.normalize_rows(x, "f8")
    assert s.dtype == np.float32 and s[0, 0] == 80.0 / 128.0 and s[0, 1] == 104.0 / 128.0 and not s[0, 2:].any()
    rng = np.random.Generator(np.random.Philox(3))
    c = rng.standard_normal((500, 512), dtype=np.float32)
    s8, s32 = ro.normalize_rows(c, "f8"), ro.normalize_rows(c, "f32")
    grid = set((fp8_oracle.e4m3_table()[:127] / 128.0).astype(np.float32).tolist())
    assert set(np.abs(s8).ravel().tolist()) <= grid                      # every stored value is an e4m3 value / 128
    again = (fp8_oracle.e4m3_decode(fp8_oracle.e4m3_encode(s8 * np.float32(128))) / np.float32(128)).astype(np.float32)
    np.testing.assert_array_equal(again, s8)                             # idempotent
    rel = np.abs(s8 - s32)[np.abs(s32) > 2.0 ** -6 / 128] / np.abs(s32)[np.abs(s32) > 2.0 ** -6 / 128]
    assert rel.max() <= 2.0 ** -4 + 1e-6                                 # three mantissa bits in the normal range
    nrm = np.linalg.norm(s8.astype(np.float64), axis=1)
    cos = (s8.astype(np.float64) * s32).sum(1) / nrm
    assert abs(nrm - 1).max() < 0.02 and (1 - cos).max() < 2e-3
    # one inverse norm per row (round 5): the REPRESENTED row values * inv has unit norm, so a distance is a cosine distance —
    # the self-query of a stored row returns it first at distance 0 (it was 1 - |values| = up to 0.03 with the codes alone)
    np.testing.assert_allclose(s8.inv, (1.0 / nrm).astype(np.float32), rtol=2e-7)
    assert abs(np.linalg.norm(s8.represented().astype(np.float64), axis=1) - 1).max() < 3e-7
    l, d, _ = ro.query(np.asarray(s8[7:8]), s8, np.arange(500, dtype=np.int64), 3)
    assert l[0, 0] == 7 and abs(d[0, 0]) < 2e-7
    l, d, _ = ro.query(c[7:8], s8, np.arange(500, dtype=np.int64), 3)     # the unquantised original: 1 - cos of the rounding
    assert l[0, 0] == 7 and abs(d[0, 0] - (1 - cos[7])) < 2e-7
    # row slices, index arrays and concat_rows keep the inverse norms in step; a column slice is a plain array
    assert np.array_equal(s8[10:20].inv, s8.inv[10:20]) and np.array_equal(s8[[3, 1]].inv, s8.inv[[3, 1]])
    assert np.array_equal(ro.concat_rows([s8[:5], s8[9:12]]).inv, np.concatenate([s8.inv[:5], s8.inv[9:12]]))
    assert type(s8[:, :8]) is np.ndarray and type(s8[4]) is np.ndarray

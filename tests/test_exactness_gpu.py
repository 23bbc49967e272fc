"""The exactness contract of mmiss_index_query under attack (VERDICT r1 "What's weak" #3): stage 1 ranks by approximate
matrix-core scores and keeps k' > k rows; that is exact only if nothing it left out belongs to the true top-k. These tests
build the cases where it is NOT — more distinct near-tied rows around rank k than the slack k' - k holds — and require
(a) the oracle's ids and distance bits anyway, (b) evidence that the guard fired and widened (mmiss_index_guard_stats).
"top-10 recall = 1.0 vs reference" (BASELINE.json north_star; backend/app/main.py:761-765)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import mmiss_amd  # noqa: F401
    from mmiss_amd import _lib
    from mmiss_amd.index import FlatIndex
    from oracle import retrieval_oracle as ro

    return FlatIndex, ro, _lib


def _randn(n, d, seed):
    return np.random.Generator(np.random.Philox(seed)).standard_normal((n, d), dtype=np.float32)


def _unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def _check(idx, ro, stored, labels, q, k):
    lab, dist, cnt = idx.query(q, k)
    ol, od, oc = ro.query(q, stored, labels, k)
    np.testing.assert_array_equal(cnt, oc)
    np.testing.assert_array_equal(lab, ol)
    np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))


def _near_tie_corpus(N, D, q, n_ties, cos0, step, seed):
    """N random rows, n_ties of them replaced (at scattered positions) by DISTINCT rows whose cosine to q is
    cos0 + j * step: after normalisation and storage rounding their canonical scores sit within the stage-1 error of
    each other, in an order stage 1 cannot see."""
    c = _randn(N, D, seed)
    rng = np.random.Generator(np.random.Philox(seed + 1))
    at = np.sort(rng.choice(N, size=n_ties, replace=False))
    qh = _unit(q.astype(np.float64))
    for j, r in enumerate(at):
        u = rng.standard_normal(D)
        u -= (u @ qh) * qh
        u /= np.linalg.norm(u)
        t = cos0 + j * step
        c[r] = (3.0 * (t * qh + np.sqrt(1 - t * t) * u)).astype(np.float32)  # any norm: rows are normalised at add
    return c, at


@pytest.mark.parametrize("dtype", ["f32", "f16"])
@pytest.mark.parametrize("Q", [1, 5, 40])
def test_more_distinct_near_ties_than_the_slack(mods, dtype, Q):
    """60 distinct rows within 60 * 2e-8 = 1.2e-6 (f32 rows) / the f16 storage noise (~1e-5, f16 rows) of each other at
    the top of every query's ranking; k = 10 -> k' = 16 candidates (groups) survive stage 1. Q = 1, 5: streaming scan;
    Q = 40: score GEMM + group-max selection (f16) / scan with several query tiles (f32)."""
    FlatIndex, ro, _ = mods
    N, D, k = 20000, 512, 10
    qs = _randn(Q, D, seed=700 + Q)
    c = None
    for j in range(min(Q, 3)):  # the first three queries each get their own cluster of near-ties
        cj, at = _near_tie_corpus(N, D, qs[j], 60, 0.9, 2e-8, seed=710 + 10 * j)
        if c is None:
            c = cj
        else:
            c[at] = cj[at]
    labels = np.arange(N, dtype=np.int64) * 2 + 7
    idx = FlatIndex(D, dtype)
    idx.add(c, labels)
    stored = ro.normalize_rows(c, dtype)
    before = idx.guard_stats()
    _check(idx, ro, stored, labels, qs, k)
    after = idx.guard_stats()
    assert after["queries"] - before["queries"] == Q
    assert after["widened"] - before["widened"] >= min(Q, 3), (before, after)   # the guard saw it
    for kk in (1, 16, 24, 50):  # the tie cluster straddles every one of these ranks as well
        _check(idx, ro, stored, labels, qs[: min(Q, 3)], kk)
    idx.close()


@pytest.mark.parametrize("dtype", ["f32", "f16", "f8"])
def test_thousands_of_exact_duplicates(mods, dtype):
    """5000 rows identical to the query spread over the index (a photo uploaded again and again): approximate scores tie
    EXACTLY, far more ties than any candidate page holds; the k smallest labels must win, for k = 10 and the UI's
    'All' = 1000 (main.py:757)."""
    FlatIndex, ro, _ = mods
    N, D = 30000, 128
    c = _randn(N, D, seed=31)
    q = _randn(2, D, seed=32)
    rng = np.random.Generator(np.random.Philox(33))
    dup = np.sort(rng.choice(N, size=5000, replace=False))
    c[dup] = q[0]
    labels = np.arange(N, dtype=np.int64)
    idx = FlatIndex(D, dtype)
    idx.add(c, labels)
    stored = ro.normalize_rows(c, dtype)
    for k in (10, 1000):
        lab, dist, cnt = idx.query(q, k)
        np.testing.assert_array_equal(lab[0], dup[:k])
        _check(idx, ro, stored, labels, q, k)
    st = idx.guard_stats()
    assert st["widened"] >= 1 and st["rounds"] >= 1 and st["swept_rows"] >= 5000
    idx.close()


@pytest.mark.parametrize("dtype", ["f32", "f16", "f8"])
def test_widen_pass_alone_reproduces_the_oracle(mods, dtype):
    """guard_force = 1 sends EVERY query through the widen pass (one threshold pass over the index + the exact re-rank of
    what it collected): it must return exactly what the ordinary path returns — the oracle's ids and distance bits — for
    every Q / k regime: Q <= 16 / 32 / 64 on the streaming scan's three query-tile widths, Q = 70 on the score GEMM (f16)."""
    FlatIndex, ro, _lib = mods
    N, D = 9000, 256
    c = _randn(N, D, seed=41)
    c[100:140] = c[7]                      # a duplicate cluster for good measure
    labels = np.arange(N, dtype=np.int64) * 3
    idx = FlatIndex(D, dtype)
    idx.add(c, labels)
    stored = ro.normalize_rows(c, dtype)
    _lib.set_option("guard_force", 1)
    try:
        for Q, k in [(1, 1), (1, 10), (17, 10), (70, 10), (3, 24), (2, 100), (2, 1000)]:
            before = idx.guard_stats()["widened"]
            _check(idx, ro, stored, labels, _randn(Q, D, seed=50 + Q + k), k)
            assert idx.guard_stats()["widened"] - before == Q
    finally:
        _lib.set_option("guard_force", 0)
    small = FlatIndex(D, dtype)             # fewer rows than k': stage 1 leaves nothing out, nothing to widen
    small.add(c[:12], labels[:12])
    _lib.set_option("guard_force", 1)
    try:
        _check(small, ro, stored[:12], labels[:12], _randn(3, D, seed=60), 10)
        assert small.guard_stats()["widened"] == 0
    finally:
        _lib.set_option("guard_force", 0)
    idx.close()
    small.close()


def test_guard_stays_quiet_on_random_data(mods):
    """On a random corpus the 10th and the 16th best score are ~5e-3 apart, eight times the error bound (6e-4 for f16 rows
    at D = 512): the guard must (almost) never fire — it is a proof obligation, not a second pass."""
    FlatIndex, ro, _ = mods
    N, D = 100_000, 512
    c = _randn(N, D, seed=91)
    idx = FlatIndex(D, "f16", capacity=N)
    idx.add(c, np.arange(N, dtype=np.int64))
    for Q, seed in [(1, 1), (16, 2), (256, 3), (700, 4)]:
        idx.query(_randn(Q, D, seed=900 + seed), 10)
    st = idx.guard_stats()
    assert st["queries"] == 1 + 16 + 256 + 700
    assert st["widened"] <= 2, st           # expectation ~4e-5 per query
    idx.close()


def test_filtered_selection_overflow_is_handed_to_the_widen_pass(mods):
    """Q > 128 over >= 128 index tiles takes the threshold-filtered selection: tau_q comes from the first 1/16 of the rows.
    Here the rows get MORE similar to the first queries the later they come, so tau_q is far too low, their candidate lists
    overflow, and the guard must route exactly those queries through the widen pass; every query still equals the oracle.
    Then the same with a deliberately tiny list capacity on random data, and with the filter switched off."""
    FlatIndex, ro, _lib = mods
    N, D, Q, k = 60000, 256, 200, 10
    c = _randn(N, D, seed=71)
    q = _randn(Q, D, seed=72)
    ramp = (np.arange(N, dtype=np.float32) / N)[:, None]
    for j in range(3):
        sel = slice(j, N, 3)                      # every third row drifts towards query j, more and more
        c[sel] += 6.0 * ramp[sel] * _unit(q[j])[None, :] * np.sqrt(D)
    labels = np.arange(N, dtype=np.int64)
    idx = FlatIndex(D, "f16", capacity=N)
    idx.add(c, labels)
    stored = ro.normalize_rows(c, "f16")
    sub = np.concatenate([np.arange(6), np.arange(6, Q, 23)])
    _lib.set_option("score_filter", 2)            # (by default only from Q x N = 1e9 scores)
    try:
        lab, dist, cnt = idx.query(q, k)
    finally:
        _lib.set_option("score_filter", 1)
    st = idx.guard_stats()
    assert st["widened"] >= 3, st                 # the three drifting queries overflowed
    ol, od, oc = ro.query(q[sub], stored, labels, k)
    np.testing.assert_array_equal(lab[sub], ol)
    np.testing.assert_array_equal(dist[sub].view(np.uint32), od.view(np.uint32))
    full = (lab.copy(), dist.copy())
    for opts in ({"score_filter": 2, "score_filter_cap": 8}, {"score_filter": 0}):
        for o, v in opts.items():
            _lib.set_option(o, v)
        try:
            l2, d2, _ = idx.query(q, k)
        finally:
            _lib.set_option("score_filter", 1)
            _lib.set_option("score_filter_cap", 2048)
        np.testing.assert_array_equal(l2, full[0])
        np.testing.assert_array_equal(d2.view(np.uint32), full[1].view(np.uint32))
    idx.close()


@pytest.mark.parametrize("dtype", ["f16", "f32", "f8"])
def test_a_plateau_of_250k_duplicates_ends_in_the_exhaustive_pass(mods, dtype):
    """more rows within eps of the k-th score than the widen pass's list holds (8192): 250 000 byte-identical
    rows (a placeholder image uploaded over and over) among 300 000. The query must come back — the oracle's ids (the k smallest labels of the plateau) and distance
    bits — through the exhaustive canonical pass, for the query on the plateau AND for its neighbours in the same batch."""
    FlatIndex, ro, _ = mods
    N, D = 300_000, 128
    c = _randn(N, D, seed=900)
    rng = np.random.Generator(np.random.Philox(901))
    dup = np.sort(rng.choice(N, size=250_000, replace=False))
    c[dup] = c[dup[0]]
    labels = np.arange(N, dtype=np.int64) * 2 + 1
    idx = FlatIndex(D, dtype)
    idx.add(c, labels)
    stored = ro.normalize_rows(c, dtype)
    q = np.concatenate([c[dup[0]][None], _randn(3, D, seed=902)])
    for k in (10, 100):
        before = idx.guard_stats()
        lab, dist, cnt = idx.query(q, k)
        after = idx.guard_stats()
        np.testing.assert_array_equal(lab[0], labels[dup[:k]])
        assert after["exhaustive"] - before["exhaustive"] >= 1, (before, after)
        ol, od, oc = ro.query(q, stored, labels, k)
        np.testing.assert_array_equal(cnt, oc)
        np.testing.assert_array_equal(lab, ol)
        np.testing.assert_array_equal(dist.view(np.uint32), od.view(np.uint32))
    idx.close()


@pytest.mark.parametrize("dtype,Q", [("f16", 256), ("f16", 40), ("f32", 40), ("f8", 20), ("f8", 256)])
def test_a_clustered_index_where_every_query_is_widened(mods, dtype, Q):
    """BASELINE configs[1] queries each embedding against the index OF those embeddings, and random-weight embeddings sit at
    pairwise cosine ~0.99: the 10th and the 16th best score are closer than the error bound for EVERY query (VERDICT r3
    weak #2). Rows = one common direction + 10 % noise; the queries are index rows (self-match first, distance ~0). Every
    query must be widened by ONE threshold pass (rounds == calls), none may end in the exhaustive pass, and ids + distance
    bits equal the oracle's. Q = 256 (f16, f8): the score-GEMM threshold pass; the others: the streaming-scan one."""
    FlatIndex, ro, _ = mods
    N, D, k = 40000, 512, 10
    centre = _unit(_randn(1, D, seed=1300))[0]
    c = centre[None, :] + 0.1 * _randn(N, D, seed=1301) / np.sqrt(D)     # |noise| ~ 0.1: pairwise cosine ~ 0.99 +- 4e-4
    labels = np.arange(N, dtype=np.int64) * 3 + 1
    idx = FlatIndex(D, dtype, capacity=N)
    idx.add(c, labels)
    stored = ro.normalize_rows(c, dtype)
    sel = np.random.Generator(np.random.Philox(1302)).choice(N, size=Q, replace=False)
    q = c[sel]
    before = idx.guard_stats()
    lab, dist, cnt = idx.query(q, k)
    after = idx.guard_stats()
    # (fp8 rows: the storage rounding itself spreads the scores by ~1e-3, a few queries are proven by the first pass)
    assert after["widened"] - before["widened"] >= Q * (0.5 if dtype == "f8" else 0.9), (before, after)
    assert after["rounds"] - before["rounds"] == 1 and after["exhaustive"] == before["exhaustive"], (before, after)
    assert after["swept_rows"] - before["swept_rows"] >= Q * k
    # configs[1]'s own check: every embedding finds ITSELF first. fp8 rows carry their inverse norm (round 5), so what comes
    # back is a cosine distance: the row's own unquantised vector lies within the e4m3 rounding of its direction (1 - cos
    # < 4e-3, typically 7e-4), its neighbours 1e-2 away.
    assert (dist[:, 0] < (4e-3 if dtype == "f8" else 1e-4)).all(), dist[:, 0].max()
    np.testing.assert_array_equal(lab[:, 0], labels[sel])
    sub = np.arange(0, Q, max(1, Q // 16))
    ol, od, oc = ro.query(q[sub], stored, labels, k)
    np.testing.assert_array_equal(lab[sub], ol)
    np.testing.assert_array_equal(dist[sub].view(np.uint32), od.view(np.uint32))
    np.testing.assert_array_equal(cnt[sub], oc)
    idx.close()

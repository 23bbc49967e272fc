"""How far the canonical float64 evaluation of the retrieval oracle lies from a float32 evaluation in the order the
reference's dependency uses (oracle/chroma_float32.py: restated from chromadb's / chroma-hnswlib's published formulas, not
run against them — both are absent here, so this pins nothing; it bounds what a user switching over can see change):
distances within 1e-6 of each other, the same ten ids wherever the tenth and eleventh distance are further apart than that."""
import numpy as np
import pytest

from oracle import chroma_float32 as cf
from oracle import retrieval_oracle as ro

TOL = 1e-6


def _rand(n, d, seed):
    return np.random.Generator(np.random.Philox(seed)).standard_normal((n, d), dtype=np.float32)


def _clustered(n, d, seed, spread):
    """unit rows around one direction: pairwise cosine ~ 1 - spread^2 (what real CLIP embeddings of one class look like)"""
    rng = np.random.Generator(np.random.Philox(seed))
    centre = rng.standard_normal((1, d), dtype=np.float32)
    centre /= np.linalg.norm(centre)
    return (centre + spread * rng.standard_normal((n, d), dtype=np.float32) / np.sqrt(d)).astype(np.float32)


@pytest.mark.parametrize("corpus", ["random", "clustered"])
@pytest.mark.parametrize("D", [512, 768])
def test_float32_orders_agree_with_the_canonical_distances(corpus, D):
    N, Q, k = 3000, 6, 10
    c = _rand(N, D, 11) if corpus == "random" else _clustered(N, D, 12, 0.5)
    q = _rand(Q, D, 13) if corpus == "random" else _clustered(Q, D, 14, 0.5)
    labels = np.arange(N, dtype=np.int64)
    canon = ro.distances(q, ro.normalize_rows(c, "f32"))          # float32 [Q, N], canonical float64 arithmetic
    lab, dist, _ = ro.query(q, ro.normalize_rows(c, "f32"), labels, k)
    for name, d32 in (("brute force", cf.brute_force_cosine(q, c)), ("hnswlib scalar", cf.hnswlib_cosine(q, c, 1)),
                      ("hnswlib 16 lanes", cf.hnswlib_cosine(q, c, 16))):
        assert np.abs(d32.astype(np.float64) - canon.astype(np.float64)).max() <= TOL, name
        ids, dd = cf.topk(d32, k + 1)
        for qi in range(Q):
            full = np.sort(canon[qi])
            gap_ok = full[k] - full[k - 1] > 2 * TOL
            # inside the top-k two rows closer than the tolerance may swap places: compare as sets, and the order where gaps allow
            if gap_ok:
                assert set(ids[qi, :k].tolist()) == set(lab[qi].tolist()), (name, qi)
            clear = np.diff(full[:k + 1]) > 2 * TOL
            if clear.all():
                np.testing.assert_array_equal(ids[qi, :k], lab[qi], err_msg=name)
            np.testing.assert_allclose(dd[qi, :k], dist[qi], atol=TOL, err_msg=name)


def test_similarity_shown_to_the_user_moves_by_half_of_that():
    """backend/app/main.py:782: similarity = 1 - distance / 2 — a 1e-6 distance difference is 5e-7 of similarity, below the
    3 decimals the UI prints."""
    d = np.float32(0.25)
    a, b = ro.similarity_from_distance([d])[0], ro.similarity_from_distance([d + np.float32(TOL)])[0]
    assert abs(a - b) <= 0.51 * TOL + 1e-9

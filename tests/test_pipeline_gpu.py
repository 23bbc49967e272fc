"""Several batches in flight (mmiss_amd.pipeline.BatchLanes): independent encoder handles on their own streams and host
threads, ONE shared index handle. The C-ABI promise under test (include/mmiss.h, SURVEY §8b "Threading"): handles are
re-entrant across handles, a handle serialises its own calls — so lanes must return exactly what one-batch-at-a-time
returns (the reference's regime, backend/app/main.py:177-232), bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_lanes_equal_one_batch_at_a_time():
    import torch
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipEncoder, ClipShape, random_state_dict
    from mmiss_amd.index import FlatIndex
    from mmiss_amd.pipeline import BatchLanes

    TINY = ClipShape(v_hidden=128, v_layers=2, v_heads=2, v_mlp=256, v_patch=32, v_image=64, t_hidden=128, t_layers=2,
                     t_heads=2, t_mlp=256, t_vocab=1000, t_ctx=16, proj_dim=128, eos_token_id=999)
    B, N, k, n_batches = 24, 5000, 10, 14
    W = random_state_dict(TINY, seed=3)
    encs = []
    for _ in range(2):
        e = ClipEncoder(TINY, max_batch_image=B, max_batch_text=4)
        e.load_state_dict(W)
        encs.append(e)
    g = torch.Generator(device="cuda").manual_seed(11)
    rows = torch.randn(N, 128, device="cuda", generator=g)
    idx = FlatIndex(128, "f16")
    idx.add(rows, np.arange(N, dtype=np.int64) * 5)
    S = TINY.v_image
    batches = [torch.randn(B, 3, S, S, device="cuda", generator=g) for _ in range(n_batches)]

    def one(enc, px):
        emb = enc.encode_image(px)
        lab, dist, cnt = idx.query(emb, k)
        return emb.cpu().numpy(), lab.cpu().numpy(), dist.cpu().numpy(), cnt.cpu().numpy()

    want = [one(encs[0], px) for px in batches]
    torch.cuda.synchronize()
    with BatchLanes(2, lambda lane, px: one(encs[lane], px)) as lanes:
        got = lanes.map(batches)
        again = lanes.map(batches[:3])           # a second wave through the same lanes
    assert len(got) == n_batches and len(again) == 3
    for w, g_ in zip(want + want[:3], got + again):
        for a, b in zip(w, g_):
            assert a.dtype == b.dtype and a.shape == b.shape
            np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                          b.view(np.uint32) if b.dtype == np.float32 else b)
    for e in encs:
        e.close()
    idx.close()


def test_lane_errors_surface_in_drain():
    import mmiss_amd  # noqa: F401
    from mmiss_amd.pipeline import BatchLanes

    def work(lane, item):
        if item == 3:
            raise RuntimeError("batch 3 is broken")
        return item * 2

    with BatchLanes(2, work) as lanes:
        with pytest.raises(RuntimeError, match="batch 3"):
            lanes.map(list(range(8)))
        assert lanes.map([5, 6]) == [10, 12]     # the lanes survive and start clean

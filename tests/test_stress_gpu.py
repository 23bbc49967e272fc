"""Handle lifecycle under churn: encoders and indexes created, used once and destroyed in a tight loop from two threads
while a long-lived encoder keeps the GPU busy. Every result is checked against the oracle — a stale or half-cleared buffer
(e.g. a clear enqueued on the null stream landing after a weight upload on the handle's own non-blocking stream) shows up
as a finite but wrong embedding. The reference creates its model once (backend/app/utils.py:27-49), but tests, reloads
(uvicorn reload=True, run.py:10-14) and multi-handle servers do not."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_create_use_destroy_churn():
    import mmiss_amd  # noqa: F401
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from mmiss_amd.index import FlatIndex
    from oracle import clip_oracle as co
    from oracle import retrieval_oracle as ro

    s = co.TINY
    Ws = [co.init_weights(s, seed=k) for k in range(3)]
    rng = np.random.Generator(np.random.Philox(3))
    px = rng.standard_normal((4, 3, s.v_image, s.v_image), dtype=np.float32)
    ids = co.synthetic_text_ids(3, s.t_ctx, s.t_vocab, s.eos_token_id, seed=4)
    refs = [(co.embed_images(px, W, s), co.embed_texts(ids, W, s)) for W in Ws]
    big = ClipEncoder(ClipShape.from_any(s), max_batch_image=64, max_batch_text=8)   # keeps kernels in flight meanwhile
    big.load_state_dict(Ws[0])
    bpx = rng.standard_normal((64, 3, s.v_image, s.v_image), dtype=np.float32)
    errors = []
    stop = threading.Event()

    def busy():
        try:
            while not stop.is_set():
                big.encode_image(bpx)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    def churn(tid):
        try:
            for it in range(12):
                k = (it + tid) % 3
                enc = ClipEncoder(ClipShape.from_any(s), max_batch_image=4, max_batch_text=4,
                                  precision="fp8" if (it + tid) % 4 == 3 else "bf16")
                enc.load_state_dict(Ws[k])
                img, txt = enc.encode_image(px), enc.encode_text(ids)
                enc.close()
                ci = (img * refs[k][0]).sum(1)
                ct = (txt * refs[k][1]).sum(1)
                assert (1 - ci).max() < 1e-3 and (1 - ct).max() < 1e-3, (tid, it, ci, ct)
                idx = FlatIndex(s.proj_dim, "f16" if it % 2 else "f32")
                labels = np.arange(4, dtype=np.int64) + 10 * it
                idx.add(img, labels)
                lab, dist, cnt = idx.query(txt, 3)
                ol, od, oc = ro.query(txt, ro.normalize_rows(img, "f16" if it % 2 else "f32"), labels, 3)
                idx.close()
                assert np.array_equal(lab, ol) and np.array_equal(dist.view(np.uint32), od.view(np.uint32))
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=busy)] + [threading.Thread(target=churn, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads[1:]:
        t.join()
    stop.set()
    threads[0].join()
    big.close()
    assert not errors, errors

"""End to end on the MI355X through the reference-facing surface: FastAPI stand-in -> mmiss_amd.utils / search ->
libmmiss (tiny seeded CLIP, real HIP kernels, real flat index), the reference's six drill images as uploads
(BASELINE configs[0] shape of test: ingest, then image / text / multimodal queries)."""
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def stack(tmp_path):
    pytest.importorskip("fastapi")
    pytest.importorskip("httpx")
    from starlette.testclient import TestClient
    import mmiss_amd  # noqa: F401
    from mmiss_amd import api, collection, search, utils
    from mmiss_amd.encoder import ClipEncoder, ClipShape
    from mmiss_amd.preprocess import ClipBPETokenizer, ClipProcessor
    from oracle import clip_oracle as co
    from test_tokenizer_preprocess_cpu import _synthetic_vocab

    import dataclasses
    vocab, merges = _synthetic_vocab()
    shape = dataclasses.replace(co.TINY, v_image=224, t_vocab=len(vocab), eos_token_id=vocab["<|endoftext|>"], t_ctx=32)
    W = co.init_weights(shape, seed=11)
    enc = ClipEncoder(ClipShape.from_any(shape), max_batch_image=8, max_batch_text=8)
    enc.load_state_dict(W)
    proc = ClipProcessor(ClipShape.from_any(shape), ClipBPETokenizer(vocab, merges), max_length=32)
    utils.set_clip_model(enc, proc)
    col = collection.PersistentClient(path=str(tmp_path)).create_collection("image-match", {"hnsw:space": "cosine"})
    search.set_collection(col)
    yield TestClient(api.create_app()), enc, proc, W, shape, col, co, str(tmp_path)
    utils.set_clip_model(None, None)
    search.set_collection(None)


def _png_bytes(arr):
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="PNG")
    return buf.getvalue()


def test_ingest_and_query_drill_set(stack):
    client, enc, proc, W, shape, col, co, path = stack
    g = np.load(os.path.join(G, "drill_set.npz"))
    crops = g["crops_u8"]  # 224x224 crops of the reference's sample images
    ids = []
    for i in range(6):
        r = client.post("/api/upload", files={"file": (str(g["names"][i]) + ".png", _png_bytes(crops[i]), "image/png")},
                        data={"description": str(g["names"][i])})
        assert r.status_code == 200, r.text
        ids.append(r.json()["metadata"]["id"])
    assert col.count() == 6
    # embeddings stored in the index == oracle embeddings of the same pixels (within the 1e-3 cosine bar)
    ref = co.embed_images(co.normalize_u8(crops), W, shape)
    got = col.get(ids=ids, include=["embeddings"])["embeddings"]
    assert (1 - (got * ref).sum(1)).max() < 1e-3
    # image query: the uploaded image itself ranks first with similarity ~1; ranking follows the oracle's cosines
    r = client.post("/api/search/image", files={"file": ("q.png", _png_bytes(crops[4]), "image/png")}, data={"limit": "6"})
    res = r.json()["results"]
    assert [x["id"] for x in res][0] == ids[4] and abs(res[0]["similarity_score"] - 1) < 1e-4
    oracle_order = np.argsort(-(ref @ ref[4]), kind="stable")
    assert [x["id"] for x in res] == [ids[j] for j in oracle_order]
    np.testing.assert_allclose([x["similarity_score"] for x in res], (1 + (ref @ ref[4])[oracle_order]) / 2, atol=5e-4)
    # text and multimodal queries run through the BPE tokenizer + text tower + blend kernel
    r = client.post("/api/search/text", data={"query": "red drill", "limit": "3"})
    assert r.status_code == 200 and len(r.json()["results"]) == 3
    tq = co.embed_texts(proc.tokenize(["red drill"]), W, shape)
    want = np.argsort(-(ref @ tq[0]), kind="stable")[:3]
    assert [x["id"] for x in r.json()["results"]] == [ids[j] for j in want]
    r = client.post("/api/search/multimodal", files={"file": ("q.png", _png_bytes(crops[0]), "image/png")},
                    data={"query": "orange drill", "weight_image": "0.7", "limit": "2"})
    assert r.status_code == 200 and r.json()["results"][0]["id"] == ids[0]
    # the collection persisted itself: a fresh client sees the same rows and answers identically
    from mmiss_amd import collection

    again = collection.PersistentClient(path=path).get_collection("image-match")
    assert again.count() == 6
    a = col.query(query_embeddings=ref[2:3], n_results=6)
    b = again.query(query_embeddings=ref[2:3], n_results=6)
    assert a["ids"] == b["ids"] and a["distances"] == b["distances"]

"""The FastAPI stand-in for the four hot routes (backend/app/main.py:124-350): form fields, response shapes, 409 on a
duplicate, limit / filters handling, 500 on errors. Model and collection are oracle-backed stand-ins (no GPU here)."""
import io
import json

import numpy as np
import pytest

fastapi = pytest.importorskip("fastapi")
pytest.importorskip("httpx")

from fakes import OracleEncoder, OracleIndex  # noqa: E402
from oracle import clip_oracle as co  # noqa: E402
from oracle import retrieval_oracle as ro  # noqa: E402


class _Proc:
    def __init__(self, shape):
        self.shape = shape

    def preprocess_images(self, images):
        return np.stack([co.preprocess_image(im, self.shape.v_image) for im in images])

    def rgb_arrays(self, images):
        return [np.asarray(im.convert("RGB") if hasattr(im, "convert") else im, dtype=np.uint8) for im in images]

    def tokenize(self, texts):
        s = self.shape
        return np.stack([co.synthetic_text_ids(1, s.t_ctx, s.t_vocab, s.eos_token_id, seed=sum(map(ord, t)) % 9973)[0] for t in texts])


def _png(seed, size=(90, 70)):
    from PIL import Image

    rng = np.random.Generator(np.random.Philox(seed))
    buf = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, size=(size[1], size[0], 3), dtype=np.uint8)).save(buf, format="PNG")
    return buf.getvalue()


@pytest.fixture()
def client(monkeypatch):
    from starlette.testclient import TestClient
    import mmiss_amd  # noqa: F401
    from mmiss_amd import api, collection, search, utils

    monkeypatch.setattr(collection, "FlatIndex", OracleIndex)
    monkeypatch.setattr(search, "blend", lambda i, t, w: ro.blend(i, t, w))
    utils.set_clip_model(OracleEncoder(co.TINY), _Proc(co.TINY))
    search.set_collection(collection.FlatCollection("t"))
    yield TestClient(api.create_app())
    utils.set_clip_model(None, None)
    search.set_collection(None)


def test_multipart_parser_handles_repeated_fields_and_files():
    import mmiss_amd  # noqa: F401
    from mmiss_amd.api import parse_form

    boundary = "XyZ"
    body = (b"--XyZ\r\nContent-Disposition: form-data; name=\"filters\"\r\n\r\nhas cord\r\n"
            b"--XyZ\r\nContent-Disposition: form-data; name=\"filters\"\r\n\r\nis red\r\n"
            b"--XyZ\r\nContent-Disposition: form-data; name=\"file\"; filename=\"a b.png\"\r\nContent-Type: image/png\r\n\r\n\x89PNG\r\n\x00\xff\r\n"
            b"--XyZ--\r\n")
    fields, files = parse_form(f"multipart/form-data; boundary={boundary}", body)
    assert fields["filters"] == ["has cord", "is red"]
    assert files["file"] == ("a b.png", b"\x89PNG\r\n\x00\xff")
    fields, files = parse_form("application/x-www-form-urlencoded", b"query=red+drill&limit=0")
    assert fields == {"query": ["red drill"], "limit": ["0"]} and files == {}


def test_upload_then_duplicate_then_searches(client):
    imgs = [_png(i) for i in range(4)]
    ids = []
    for i, data in enumerate(imgs):
        r = client.post("/api/upload", files={"file": (f"drill {i}.png", data, "image/png")},
                        data={"description": f"drill number {i}", "custom_metadata": "{}"})
        assert r.status_code == 200 and r.json()["success"] is True
        meta = r.json()["metadata"]
        assert meta["filename"] == f"drill {i}.png" and meta["id"].startswith("img_")
        ids.append(meta["id"])
    dup = client.post("/api/upload", files={"file": ("again.png", imgs[1], "image/png")})
    assert dup.status_code == 409 and dup.json()["error"] == "Duplicate image" and dup.json()["metadata"]["id"] == ids[1]

    r = client.post("/api/search/image", files={"file": ("q.png", imgs[2], "image/png")}, data={"limit": "3"})
    assert r.status_code == 200
    res = r.json()["results"]
    assert len(res) == 3 and res[0]["id"] == ids[2] and abs(res[0]["similarity_score"] - 1.0) < 1e-5
    assert all("url" in x and "thumbnail_url" in x for x in res)
    assert len(client.post("/api/search/image", files={"file": ("q.png", imgs[2], "image/png")}, data={"limit": "0"}).json()["results"]) == 4

    r = client.post("/api/search/text", data={"query": "red drill", "limit": "2"})
    assert r.status_code == 200 and len(r.json()["results"]) == 2
    assert client.post("/api/search/text", data={"limit": "2"}).status_code == 422  # `query` is required (Form(...))

    r = client.post("/api/search/multimodal", files={"file": ("q.png", imgs[0], "image/png")},
                    data={"query": "drill", "weight_image": "1.0", "limit": "1"})
    assert r.status_code == 200 and r.json()["results"][0]["id"] == ids[0]


def test_filters_are_applied_after_the_query(client):
    import mmiss_amd  # noqa: F401
    from mmiss_amd import search

    imgs = [_png(10 + i) for i in range(3)]
    ids = [client.post("/api/upload", files={"file": (f"{i}.png", d, "image/png")}).json()["metadata"]["id"] for i, d in enumerate(imgs)]
    col = search._collection()
    col.update(ids=[ids[0]], metadatas=[{"filter_results_json": json.dumps({"is red": "Yes ", "has cord": "no"})}])
    col.update(ids=[ids[1]], metadatas=[{"filter_results_json": json.dumps({"is red": "yes", "has cord": "yes"})}])
    col.update(ids=[ids[2]], metadatas=[{"filter_results_json": "not json"}])
    r = client.post("/api/search/text", data={"query": "drill", "limit": "10", "filters": ["is red"]})
    assert sorted(x["id"] for x in r.json()["results"]) == sorted(ids[:2])
    r = client.post("/api/search/text", data={"query": "drill", "limit": "10", "filters": ["is red", "has cord"]})
    assert [x["id"] for x in r.json()["results"]] == [ids[1]]
    r = client.post("/api/search/text", data={"query": "  ", "limit": "10", "filters": ["has cord"]})  # filter-only browse
    assert [x["id"] for x in r.json()["results"]] == [ids[1]]


def test_errors_become_500_json(client):
    r = client.post("/api/search/image", files={"file": ("q.png", b"not an image", "image/png")})
    assert r.status_code == 500 and r.json()["success"] is False and "error" in r.json()
    r = client.post("/api/upload", files={"file": ("x.png", _png(99), "image/png")}, data={"remove_bg": "true"})
    assert r.status_code == 500


def test_perceptual_hash_ids():
    """`img_` + 16 hex digits as generate_image_hash (main.py:581-585): stable for the same pixels in another container,
    different for different pictures."""
    from PIL import Image
    import mmiss_amd  # noqa: F401
    from mmiss_amd.api import image_id_for

    yy, xx = np.mgrid[0:120, 0:160]
    a = np.stack([(np.sin(xx / 17.0) + np.cos(yy / 11.0)) * 60 + 128, (xx * yy / 80.0) % 255, (xx + yy) / 2.0], -1)
    im = Image.fromarray(a.clip(0, 255).astype(np.uint8))
    one = image_id_for(im)
    assert one.startswith("img_") and len(one) == 4 + 16 and int(one[4:], 16) >= 0
    buf = io.BytesIO()
    im.save(buf, format="PNG")
    assert image_id_for(Image.open(io.BytesIO(buf.getvalue()))) == one
    other = Image.fromarray(np.random.Generator(np.random.Philox(3)).integers(0, 256, (120, 160, 3), dtype=np.uint8))
    assert image_id_for(other) != one

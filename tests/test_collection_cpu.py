"""Host bookkeeping of FlatCollection / PersistentClient (ids, metadata, persistence, chroma-like return shapes)
with the numeric index replaced by an oracle-backed stand-in (no GPU here)."""
import os

import numpy as np
import pytest

from fakes import OracleIndex


@pytest.fixture()
def colmod(monkeypatch):
    import mmiss_amd  # noqa: F401
    from mmiss_amd import collection

    monkeypatch.setattr(collection, "FlatIndex", OracleIndex)
    return collection


def _vecs(n, d=128, seed=0):
    return np.random.Generator(np.random.Philox(seed)).standard_normal((n, d), dtype=np.float32)


def test_add_query_get_update_delete_roundtrip(colmod):
    col = colmod.FlatCollection("t", metadata={"hnsw:space": "cosine"})
    v = _vecs(6)
    ids = [f"img_{i}" for i in range(6)]
    metas = [{"filename": f"{i}.jpg", "n": i} for i in range(6)]
    for i in range(6):  # the reference adds one image at a time (main.py:735-740), embeddings as python lists
        col.add(ids=[ids[i]], embeddings=[v[i].tolist()], metadatas=[metas[i]], documents=[f"caption {i}"])
    assert col.count() == 6
    res = col.query(query_embeddings=[v[2].tolist()], n_results=1000, include=["metadatas", "distances"])
    assert res["ids"][0][0] == "img_2" and len(res["ids"][0]) == 6  # n_results > count is not an error
    assert res["distances"][0][0] < 1e-6 and res["distances"][0] == sorted(res["distances"][0])
    assert res["metadatas"][0][0]["n"] == 2 and res["documents"] is None
    got = col.get(ids=["img_3", "missing"], include=["metadatas"])
    assert got["ids"] == ["img_3"] and got["metadatas"][0]["filename"] == "3.jpg"
    assert col.get(include=[])["ids"] == ids and col.get(include=[])["metadatas"] is None
    col.update(ids=["img_3"], metadatas=[{"filter_results_json": "{}"}])
    assert col.get(ids=["img_3"])["metadatas"][0] == {"filename": "3.jpg", "n": 3, "filter_results_json": "{}"}
    with pytest.raises(colmod.DuplicateIDError):
        col.add(ids=["img_1"], embeddings=[v[1].tolist()])
    col.delete(ids=["img_2", "nope"])
    assert col.count() == 5 and "img_2" not in col.get(include=[])["ids"]
    res = col.query(query_embeddings=[v[2].tolist()], n_results=3)
    assert "img_2" not in res["ids"][0] and len(res["ids"][0]) == 3
    col.delete()
    assert col.count() == 0 and col.query(query_embeddings=[v[0].tolist()], n_results=5)["ids"] == [[]]


def test_rejects_non_cosine_and_dimension_mismatch(colmod):
    with pytest.raises(ValueError):
        colmod.FlatCollection("x", metadata={"hnsw:space": "l2"})
    col = colmod.FlatCollection("t")
    col.add(ids=["a"], embeddings=_vecs(1, 128))
    with pytest.raises(ValueError):
        col.add(ids=["b"], embeddings=_vecs(1, 256))
    with pytest.raises(ValueError):
        col.query(query_embeddings=_vecs(1, 256), n_results=1)


def test_persistent_client_reopens_collection(colmod, tmp_path):
    client = colmod.PersistentClient(path=str(tmp_path))
    assert client.list_collections() == []
    col = client.create_collection("image-match", metadata={"hnsw:space": "cosine"})
    v = _vecs(4)
    col.add(ids=list("abcd"), embeddings=v, metadatas=[{"i": i} for i in range(4)], documents=list("wxyz"))
    col.delete(ids=["b"])
    again = colmod.PersistentClient(path=str(tmp_path))
    assert again.list_collections() == ["image-match"]
    col2 = again.get_collection("image-match")
    assert col2.count() == 3 and col2.get(include=["documents"])["documents"] == ["w", "y", "z"]
    r1 = col.query(query_embeddings=v[3:4], n_results=3)
    r2 = col2.query(query_embeddings=v[3:4], n_results=3)
    assert r1["ids"] == r2["ids"] and r1["distances"] == r2["distances"]
    col2.add(ids=["e"], embeddings=_vecs(1, seed=9))  # labels keep increasing after a reload
    with pytest.raises(ValueError):
        again.create_collection("image-match")
    with pytest.raises(ValueError):
        again.get_collection("other")


def test_mutations_are_journaled_not_resnapshotted(colmod, tmp_path):
    """An upload must cost O(1) on disk, not a rewrite of the whole index (chromadb persists incrementally too): add / update
    / delete append to the journal of the current generation; reopening replays it; a torn last line is an uncommitted
    mutation; the journal is compacted into a new snapshot once it outgrows a quarter of the collection."""
    import os

    client = colmod.PersistentClient(path=str(tmp_path))
    col = client.create_collection("image-match", metadata={"hnsw:space": "cosine"})
    gen0 = col._index_gen
    v = _vecs(40, seed=3)
    for i in range(40):
        col.add(ids=[f"img_{i}"], embeddings=[v[i].tolist()], metadatas=[{"n": i}], documents=[f"c{i}"])
    col.update(ids=["img_5"], metadatas=[{"filter_results_json": "{}"}])
    col.update(ids=["img_6"], embeddings=_vecs(1, seed=77))
    col.delete(ids=["img_7", "img_8"])
    assert col._index_gen == gen0                                   # no new snapshot: 43 journal records
    files = sorted(os.listdir(tmp_path))
    assert f"image-match.journal.{gen0}.jsonl" in files and f"image-match.journal.{gen0}.f32" in files
    assert sum(1 for _ in open(tmp_path / f"image-match.journal.{gen0}.jsonl")) == 43
    assert os.path.getsize(tmp_path / f"image-match.journal.{gen0}.f32") == 41 * 128 * 4
    again = colmod.PersistentClient(path=str(tmp_path)).get_collection("image-match")
    assert again.count() == 38 and again.get(ids=["img_5"])["metadatas"][0] == {"n": 5, "filter_results_json": "{}"}
    q = _vecs(3, seed=9)
    r1, r2 = col.query(query_embeddings=q, n_results=10), again.query(query_embeddings=q, n_results=10)
    assert r1["ids"] == r2["ids"] and r1["distances"] == r2["distances"] and r1["metadatas"] == r2["metadatas"]
    assert again.query(query_embeddings=_vecs(1, seed=77), n_results=1)["ids"] == [["img_6"]]
    # a crash in the middle of the next append: the half-written line is ignored
    with open(tmp_path / f"image-match.journal.{gen0}.jsonl", "a") as f:
        f.write('{"op": "add", "ids": ["img_x"], "meta')
    third = colmod.PersistentClient(path=str(tmp_path)).get_collection("image-match")
    assert third.count() == 38 and "img_x" not in third.get(include=[])["ids"]
    # compaction: past max(4096, n/4) journaled rows the collection snapshots itself and starts a fresh journal
    big = _vecs(4200, seed=5)
    third.add(ids=[f"b{i}" for i in range(4200)], embeddings=big)
    assert third._index_gen == gen0 + 1 and third._journal_rows == 0
    assert not os.path.exists(tmp_path / f"image-match.journal.{gen0}.jsonl")
    fourth = colmod.PersistentClient(path=str(tmp_path)).get_collection("image-match")
    assert fourth.count() == 38 + 4200
    assert fourth.query(query_embeddings=big[17:18], n_results=1)["ids"] == [["b17"]]


@pytest.mark.parametrize("tail", ['{"op": "add", "ids": ["img_x"], "meta', '{"op": "delete", "ids": ["img_3"]}'])
def test_mutations_after_a_torn_journal_tail_survive_the_next_restart(colmod, tmp_path, tail):
    """replay used to stop at a torn last line but leave its bytes in the file; the next acknowledged mutation
    was then glued to them, unparsable, and lost (with everything after it) at the following restart. Now the journal is cut
    back to the last committed record before new writes are accepted. Second case: a complete record whose newline was
    cut off counts as committed and gets its newline back."""
    client = colmod.PersistentClient(path=str(tmp_path))
    col = client.create_collection("image-match", metadata={"hnsw:space": "cosine"})
    v = _vecs(12, seed=11)
    for i in range(5):
        col.add(ids=[f"img_{i}"], embeddings=[v[i].tolist()], metadatas=[{"n": i}])
    jl = tmp_path / f"image-match.journal.{col._index_gen}.jsonl"
    with open(jl, "a") as f:
        f.write(tail)                                    # crash in the middle of (or right behind) the next append
    committed = 5 if tail.endswith("meta") else 4        # (the complete delete record is a committed mutation)
    second = colmod.PersistentClient(path=str(tmp_path)).get_collection("image-match")
    assert second.count() == committed
    assert open(jl, "rb").read().endswith(b"}\n")        # the file ends behind a committed record again
    for i in range(5, 10):
        second.add(ids=[f"img_{i}"], embeddings=[v[i].tolist()], metadatas=[{"n": i}])
    assert second.count() == committed + 5
    third = colmod.PersistentClient(path=str(tmp_path)).get_collection("image-match")
    assert third.count() == committed + 5                # (was 5: the five acknowledged adds were lost)
    assert third.get(ids=["img_9"])["metadatas"] == [{"n": 9}]
    assert third.query(query_embeddings=v[7:8], n_results=1)["ids"] == [["img_7"]]


@pytest.mark.parametrize("torn_bytes", [1, 6, 515])
def test_a_torn_vector_journal_tail_is_cut_off(colmod, tmp_path, torn_bytes):
    """a crash in the middle of the VECTOR write (the large one) leaves image-match.journal.N.f32 with a size that
    is not a multiple of 4 and no commit line for those bytes: np.memmap(float32) used to raise and the collection could not
    be opened at all. Replay now maps the whole floats only, cuts the file back to the end of the last committed record's
    vectors, and later appends land 4-aligned behind it and survive the next restart."""
    client = colmod.PersistentClient(path=str(tmp_path))
    col = client.create_collection("image-match", metadata={"hnsw:space": "cosine"})
    v = _vecs(12, seed=21)
    for i in range(5):
        col.add(ids=[f"img_{i}"], embeddings=[v[i].tolist()], metadatas=[{"n": i}])
    jv = tmp_path / f"image-match.journal.{col._index_gen}.f32"
    good = os.path.getsize(jv)
    with open(jv, "ab") as f:
        f.write(b"\x7f" * torn_bytes)                    # the torn vectors of an add that never got its commit line
    second = colmod.PersistentClient(path=str(tmp_path)).get_collection("image-match")
    assert second.count() == 5 and os.path.getsize(jv) == good
    for i in range(5, 9):
        second.add(ids=[f"img_{i}"], embeddings=[v[i].tolist()], metadatas=[{"n": i}])
    assert os.path.getsize(jv) == good // 5 * 9
    third = colmod.PersistentClient(path=str(tmp_path)).get_collection("image-match")
    assert third.count() == 9
    assert third.query(query_embeddings=v[7:8], n_results=1)["ids"] == [["img_7"]]
    assert third.query(query_embeddings=v[2:3], n_results=1)["ids"] == [["img_2"]]


@pytest.mark.parametrize("torn_bytes", [3, 512, 700])
def test_torn_vectors_of_a_generations_first_mutation_are_cut_off(colmod, tmp_path, torn_bytes):
    """the crash may hit the FIRST journaled mutation of a generation — torn bytes in the vector file and no
    .jsonl file at all. Replay used to return early on the missing commit log, the next add recorded an offset behind the
    orphan bytes (unaligned, or owned by nobody), and the restart after that one declared its record torn and cut the whole log
    back to nothing: every mutation acknowledged after the crash was lost."""
    client = colmod.PersistentClient(path=str(tmp_path))
    col = client.create_collection("image-match", metadata={"hnsw:space": "cosine"})
    v = _vecs(8, seed=31)
    jl, jv = col._journal_paths(col._index_gen)
    assert not os.path.exists(jl)
    with open(jv, "wb") as f:
        f.write(b"\x7f" * torn_bytes)                    # the first add's vectors, torn; its commit line never written
    second = colmod.PersistentClient(path=str(tmp_path)).get_collection("image-match")
    assert second.count() == 0 and os.path.getsize(jv) == 0
    for i in range(4):
        second.add(ids=[f"img_{i}"], embeddings=[v[i].tolist()], metadatas=[{"n": i}])
    assert os.path.getsize(jv) == 4 * 128 * 4
    third = colmod.PersistentClient(path=str(tmp_path)).get_collection("image-match")
    assert third.count() == 4                            # (was 0)
    assert third.query(query_embeddings=v[2:3], n_results=1)["ids"] == [["img_2"]]
    third.add(ids=["img_4"], embeddings=[v[4].tolist()])
    assert colmod.PersistentClient(path=str(tmp_path)).get_collection("image-match").count() == 5

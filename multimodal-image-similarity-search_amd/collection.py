"""FlatCollection — the chromadb Collection surface the reference uses, over the in-HBM FlatIndex.

Call sites mirrored (paths relative to the reference root):
  client.create_collection(name, metadata={"hnsw:space": "cosine"})     backend/app/utils.py:127-130
  collection.add(ids, embeddings, metadatas, documents)                  backend/app/main.py:735-740
  collection.query(query_embeddings, n_results, include)                 backend/app/main.py:761-765
  collection.get(ids=..., include=[...]) / get(include=[])               backend/app/main.py:533,556,563-566,631-634
  collection.update(ids, metadatas)                                      backend/app/main.py:503-510,1030-1033
  collection.delete(ids)                                                 backend/app/main.py:1065-1069
  collection.count()                                                     init_db.py:58

Only the embedding arithmetic runs on the GPU (FlatIndex); ids, metadatas and documents are host-side
bookkeeping exactly as they are bookkeeping inside chromadb's sqlite. String ids map to int64 labels
handed out in insertion order; labels are the deterministic tie-break of equal distances.
"""
from __future__ import annotations

import json
import logging
import os
import threading
from typing import Dict, List, Optional, Sequence

import numpy as np

from .index import FlatIndex

logger = logging.getLogger("image-match")
MAX_N_RESULTS = 2040  # mmiss_index_query's per-call limit on k


class DuplicateIDError(ValueError):
    """Same condition chromadb reports for add() of an id that already exists."""


def _as_list(x):
    if x is None:
        return None
    if isinstance(x, (str, bytes)):
        return [x]
    return list(x)


class FlatCollection:
    def __init__(self, name: str = "image-match", dim: Optional[int] = None, dtype: str = "f32", device: int = 0,
                 metadata: Optional[dict] = None, persist_dir: Optional[str] = None, autosave: bool = True):
        self.name = name
        self.metadata = dict(metadata or {"hnsw:space": "cosine"})
        space = self.metadata.get("hnsw:space", "cosine")
        if space != "cosine":
            raise ValueError(f"FlatCollection implements the cosine space only (got {space!r})")
        self._dim = dim
        self._dtype = dtype
        self._device = device
        self._index: Optional[FlatIndex] = None
        self._ids: List[str] = []              # row order == label order
        self._labels: List[int] = []
        self._by_id: Dict[str, int] = {}       # id -> label
        self._meta: Dict[int, Optional[dict]] = {}
        self._docs: Dict[int, Optional[str]] = {}
        self._next_label = 0
        self._lock = threading.RLock()         # 1 writer + readers (the reference has a background updater, main.py:410)
        self._persist_dir = persist_dir
        self._autosave = autosave
        if persist_dir:
            os.makedirs(persist_dir, exist_ok=True)
            if os.path.exists(self._meta_path()):
                self._load()

    # ------------------------------------------------------------------ persistence (replaces chroma_data/)
    def _meta_path(self):
        return os.path.join(self._persist_dir, f"{self.name}.meta.json")

    def _index_path(self, gen: int = 0):
        """Generation 0 is the legacy single-file name; later generations carry their number, and the metadata file
        names the generation it belongs to - so replacing the metadata file is the one atomic switch of a save."""
        suffix = "" if gen == 0 else f".{gen}"
        return os.path.join(self._persist_dir, f"{self.name}.index{suffix}.mmiss")

    def _journal_paths(self, gen: int):
        base = os.path.join(self._persist_dir, f"{self.name}.journal.{gen}")
        return base + ".jsonl", base + ".f32"

    def persist(self) -> None:
        """Full snapshot (compaction): metadata file + a new index generation; the journal of the previous generation is
        dropped. O(N) — called on create, explicitly, and automatically when the journal has grown past a quarter of the
        collection (see _saved): ordinary add / update / delete only append to the journal."""
        if not self._persist_dir:
            return
        with self._lock:
            old_gen = getattr(self, "_index_gen", 0)
            new_gen = old_gen + 1
            blob = {
                "name": self.name, "metadata": self.metadata, "dim": self._dim, "dtype": self._dtype,
                "next_label": self._next_label, "ids": self._ids, "labels": self._labels,
                "metadatas": [self._meta.get(l) for l in self._labels],
                "documents": [self._docs.get(l) for l in self._labels],
                "index_gen": new_gen,
            }
            tmp = self._meta_path() + ".tmp"
            with open(tmp, "w") as f:
                json.dump(blob, f)
                f.flush()
                os.fsync(f.fileno())
            if self._index is not None:
                self._index.save(self._index_path(new_gen))   # a new file: the previous generation stays intact
            os.replace(tmp, self._meta_path())                 # the atomic switch
            self._index_gen = new_gen
            self._journal_rows = 0
            for path in (self._index_path(old_gen),) + self._journal_paths(old_gen):
                try:
                    os.remove(path)
                except FileNotFoundError:
                    pass

    def _journal(self, record: dict, vectors: Optional[np.ndarray] = None) -> None:
        """Append one mutation to the journal of the current generation: the vectors (raw float32, as handed to add /
        update — replay re-normalises them, bit-identically) first, then ONE json line that is the commit record. O(batch),
        where a full snapshot per uploaded image would be O(N) (10 GB per /api/upload at 10M x 512 f16)."""
        jl, jv = self._journal_paths(getattr(self, "_index_gen", 0))
        if vectors is not None:
            with open(jv, "ab") as f:
                record["offset"] = f.tell()
                record["rows"] = int(vectors.shape[0])
                record["dim"] = int(vectors.shape[1])
                f.write(np.ascontiguousarray(vectors, dtype=np.float32).tobytes())
                f.flush()
                os.fsync(f.fileno())
        with open(jl, "a") as f:
            f.write(json.dumps(record) + "\n")
            f.flush()
            os.fsync(f.fileno())
        self._journal_rows = getattr(self, "_journal_rows", 0) + max(1, len(record.get("ids", [])))

    def _replay_journal(self) -> None:
        jl, jv = self._journal_paths(self._index_gen)
        if not os.path.exists(jl):
            # No commit line exists in this generation. A crash inside the FIRST journaled mutation's vector write still leaves
            # bytes in the vector file that no record owns: drop them, or the next add records an offset behind (or inside)
            # them and the restart after that one cuts the whole log back to nothing.
            if os.path.exists(jv) and os.path.getsize(jv) > 0:
                with open(jv, "r+b") as f:
                    f.truncate(0)
                    f.flush()
                    os.fsync(f.fileno())
            return
        # The vector file may end in a torn write (a crash inside f.write of the vectors — the large write, so the likelier one
        # to tear): its size need not be a multiple of 4. Map only the whole floats; the tail is cut off below.
        vec_bytes = os.path.getsize(jv) if os.path.exists(jv) else 0
        vec = np.memmap(jv, dtype=np.float32, mode="r", shape=(vec_bytes // 4,)) if vec_bytes >= 4 else None
        vec_end = 0           # byte offset just behind the last committed record's vectors
        autosave, self._autosave = self._autosave, False   # replay must not journal again
        good_end = 0          # byte offset just behind the last committed record
        torn = needs_newline = False
        try:
            with open(jl, "rb") as f:
                for raw in f:
                    try:
                        rec = json.loads(raw.decode("utf-8"))
                        if not isinstance(rec, dict) or "op" not in rec:
                            raise ValueError("not a journal record")
                    except ValueError:
                        torn = True                          # a torn last line: the mutation was never committed
                        break
                    emb = None
                    if "offset" in rec:
                        o, n, d = rec["offset"] // 4, rec["rows"], rec["dim"]
                        if vec is None or (o + n * d) > vec.shape[0]:
                            torn = True                      # the commit line survived, its vectors did not: not committed
                            break
                        if rec["offset"] % 4:
                            torn = True                      # (written behind a torn tail by a build without the cut below)
                            break
                        emb = np.array(vec[o:o + n * d]).reshape(n, d)
                        vec_end = max(vec_end, (o + n * d) * 4)
                    if rec["op"] == "add":
                        self.add(rec["ids"], emb, rec.get("metadatas"), rec.get("documents"))
                    elif rec["op"] == "update":
                        self.update(rec["ids"], emb, rec.get("metadatas"), rec.get("documents"), _replace_metadata=True)
                    elif rec["op"] == "delete":
                        self.delete(rec["ids"])
                    self._journal_rows = getattr(self, "_journal_rows", 0) + max(1, len(rec.get("ids", [])))
                    good_end += len(raw)
                    needs_newline = not raw.endswith(b"\n")   # (a complete record whose newline was cut off)
        finally:
            self._autosave = autosave
        # Leave the file ending in a newline behind the last committed record: later appends would otherwise be glued to
        # the torn bytes, become unparsable, and every mutation acknowledged after the crash would be lost at the next restart.
        if torn or needs_newline:
            with open(jl, "r+b") as f:
                f.truncate(good_end)
                if needs_newline:
                    f.seek(good_end)
                    f.write(b"\n")
                f.flush()
                os.fsync(f.fileno())
        # ... and the vector file ending behind the last committed record's vectors: the next append records f.tell() as its
        # offset, which must be 4-aligned and must not sit behind bytes no record owns.
        if vec_bytes > vec_end:
            del vec
            with open(jv, "r+b") as f:
                f.truncate(vec_end)
                f.flush()
                os.fsync(f.fileno())

    def _load(self) -> None:
        with open(self._meta_path()) as f:
            blob = json.load(f)
        self.metadata = blob.get("metadata", self.metadata)
        self._dim = blob["dim"]
        self._dtype = blob.get("dtype", self._dtype)
        self._next_label = blob["next_label"]
        self._ids = list(blob["ids"])
        self._labels = [int(x) for x in blob["labels"]]
        self._by_id = dict(zip(self._ids, self._labels))
        self._meta = dict(zip(self._labels, blob["metadatas"]))
        self._docs = dict(zip(self._labels, blob["documents"]))
        self._index_gen = int(blob.get("index_gen", 0))
        if self._dim is not None and self._labels:
            self._ensure_index(self._dim)
            self._index.load(self._index_path(self._index_gen))
            if self._index.count() != len(self._labels):
                raise RuntimeError("collection files are inconsistent (row count differs from id count)")
        self._journal_rows = 0
        self._replay_journal()

    def _saved(self, record: Optional[dict] = None, vectors: Optional[np.ndarray] = None):
        """After a mutation (called under the lock): append it to the journal; compact into a full snapshot once the
        journal holds more than max(4096, a quarter of the collection) rows."""
        if not (self._autosave and self._persist_dir):
            return
        if record is None:
            self.persist()
            return
        self._journal(record, vectors)
        if self._journal_rows > max(4096, len(self._ids) // 4):
            self.persist()

    # ------------------------------------------------------------------ helpers
    def _ensure_index(self, dim: int) -> FlatIndex:
        if self._index is None:
            self._dim = int(dim)
            self._index = FlatIndex(self._dim, self._dtype, self._device)
        elif dim != self._dim:
            raise ValueError(f"embedding dimension {dim} does not match the collection's {self._dim}")
        return self._index

    @staticmethod
    def _embeddings_array(embeddings) -> np.ndarray:
        arr = np.asarray(embeddings, dtype=np.float32)
        if arr.ndim == 1:
            arr = arr[None]
        if arr.ndim != 2:
            raise ValueError("embeddings must be a list of vectors")
        return np.ascontiguousarray(arr)

    # ------------------------------------------------------------------ chroma surface
    def count(self) -> int:
        with self._lock:
            return len(self._ids)

    def add(self, ids, embeddings, metadatas=None, documents=None) -> None:
        ids = _as_list(ids)
        emb = self._embeddings_array(embeddings)  # the reference passes python lists of floats (main.py:687)
        metadatas = _as_list(metadatas) if metadatas is not None else [None] * len(ids)
        documents = _as_list(documents) if documents is not None else [None] * len(ids)
        if not (len(ids) == emb.shape[0] == len(metadatas) == len(documents)):
            raise ValueError("ids, embeddings, metadatas and documents differ in length")
        with self._lock:
            if len(set(ids)) != len(ids):
                raise DuplicateIDError("duplicate ids in add()")
            dup = [i for i in ids if i in self._by_id]
            if dup:
                raise DuplicateIDError(f"ids already in collection: {dup[:3]}")
            index = self._ensure_index(emb.shape[1])
            labels = np.arange(self._next_label, self._next_label + len(ids), dtype=np.int64)
            index.add(emb, labels)
            self._next_label += len(ids)
            for i, lab in zip(ids, labels.tolist()):
                self._ids.append(i)
                self._labels.append(lab)
                self._by_id[i] = lab
            for lab, m, d in zip(labels.tolist(), metadatas, documents):
                self._meta[lab] = dict(m) if m is not None else None
                self._docs[lab] = d
            self._saved({"op": "add", "ids": ids, "metadatas": metadatas, "documents": documents}, emb)

    def query(self, query_embeddings=None, n_results: int = 10, include: Sequence[str] = ("metadatas", "documents", "distances"),
              **_unused) -> dict:
        if query_embeddings is None:
            raise ValueError("FlatCollection.query needs query_embeddings (text embedding functions are out of scope)")
        q = self._embeddings_array(query_embeddings)
        include = list(include)
        with self._lock:
            Q = q.shape[0]
            n = len(self._ids)
            if n == 0 or self._index is None:
                labs = np.full((Q, 0), -1, dtype=np.int64)
                dist = np.zeros((Q, 0), dtype=np.float32)
                cnt = np.zeros((Q,), dtype=np.int32)
            else:
                if q.shape[1] != self._dim:
                    raise ValueError(f"query dimension {q.shape[1]} does not match the collection's {self._dim}")
                k = max(1, min(int(n_results), n))  # n_results > count is not an error ("All" = 1000, main.py:757)
                if k > MAX_N_RESULTS:  # one mmiss_index_query call ranks at most 2040 rows per query: clamp, do not fail
                    logger.warning(f"n_results={n_results} clamped to {MAX_N_RESULTS}")
                    k = MAX_N_RESULTS
                labs, dist, cnt = self._index.query(q, k)
            label_to_id = dict(zip(self._labels, self._ids))
            out = {"ids": [], "distances": None, "metadatas": None, "documents": None, "embeddings": None,
                   "uris": None, "data": None, "included": include}
            rows = []
            for qi in range(Q):
                ls = [int(x) for x in labs[qi, : int(cnt[qi])]]
                rows.append(ls)
                out["ids"].append([label_to_id[l] for l in ls])
            if "distances" in include:
                out["distances"] = [[float(x) for x in dist[qi, : int(cnt[qi])]] for qi in range(Q)]
            if "metadatas" in include:
                out["metadatas"] = [[self._meta.get(l) for l in ls] for ls in rows]
            if "documents" in include:
                out["documents"] = [[self._docs.get(l) for l in ls] for ls in rows]
            if "embeddings" in include:
                out["embeddings"] = [self._index.get(np.asarray(ls, dtype=np.int64)) for ls in rows]
            return out

    def get(self, ids=None, include: Sequence[str] = ("metadatas", "documents"), limit: Optional[int] = None,
            offset: Optional[int] = None, **_unused) -> dict:
        include = list(include)
        with self._lock:
            if ids is None:
                sel = list(zip(self._ids, self._labels))
            else:
                sel = [(i, self._by_id[i]) for i in _as_list(ids) if i in self._by_id]  # missing ids are skipped
            if offset:
                sel = sel[int(offset):]
            if limit is not None:
                sel = sel[: int(limit)]
            out = {"ids": [i for i, _ in sel], "metadatas": None, "documents": None, "embeddings": None,
                   "uris": None, "data": None, "included": include}
            if "metadatas" in include:
                out["metadatas"] = [self._meta.get(l) for _, l in sel]
            if "documents" in include:
                out["documents"] = [self._docs.get(l) for _, l in sel]
            if "embeddings" in include:
                labs = np.asarray([l for _, l in sel], dtype=np.int64)
                out["embeddings"] = self._index.get(labs) if (self._index is not None and labs.size) else np.zeros((0, self._dim or 0), np.float32)
            return out

    def update(self, ids, embeddings=None, metadatas=None, documents=None, _replace_metadata: bool = False) -> None:
        ids = _as_list(ids)
        with self._lock:
            missing = [i for i in ids if i not in self._by_id]
            if missing:
                raise ValueError(f"ids not in collection: {missing[:3]}")
            labs = [self._by_id[i] for i in ids]
            if metadatas is not None:
                for lab, m in zip(labs, _as_list(metadatas)):
                    if m is not None:
                        merged = {} if _replace_metadata else dict(self._meta.get(lab) or {})
                        merged.update(m)  # chroma merges keys on update
                        self._meta[lab] = merged
            if documents is not None:
                for lab, d in zip(labs, _as_list(documents)):
                    self._docs[lab] = d
            emb = None
            if embeddings is not None:
                emb = self._embeddings_array(embeddings)
                self._index.update(np.asarray(labs, dtype=np.int64), emb)
            # the journal records the RESULTING metadata (replay replaces instead of merging: idempotent)
            self._saved({"op": "update", "ids": ids,
                         "metadatas": [self._meta.get(l) for l in labs] if metadatas is not None else None,
                         "documents": [self._docs.get(l) for l in labs] if documents is not None else None}, emb)

    def delete(self, ids=None) -> None:
        with self._lock:
            if ids is None:
                ids = list(self._ids)
            ids = [i for i in _as_list(ids) if i in self._by_id]
            if not ids:
                return
            labs = [self._by_id[i] for i in ids]
            self._index.remove(np.asarray(labs, dtype=np.int64))
            gone = set(labs)
            keep = [(i, l) for i, l in zip(self._ids, self._labels) if l not in gone]
            self._ids = [i for i, _ in keep]
            self._labels = [l for _, l in keep]
            for i, l in zip(ids, labs):
                del self._by_id[i]
                self._meta.pop(l, None)
                self._docs.pop(l, None)
            self._saved({"op": "delete", "ids": ids})

    def peek(self, limit: int = 10) -> dict:
        return self.get(limit=limit)


class PersistentClient:
    """Minimal stand-in for chromadb.PersistentClient(path) as the reference drives it (utils.py:113-130)."""

    def __init__(self, path: str = "chroma_data", device: int = 0, dtype: str = "f32"):
        self.path = path
        self.device = device
        self.dtype = dtype
        os.makedirs(path, exist_ok=True)
        self._open: Dict[str, FlatCollection] = {}

    def list_collections(self) -> List[str]:
        names = {n[: -len(".meta.json")] for n in os.listdir(self.path) if n.endswith(".meta.json")}
        return sorted(names | set(self._open))

    def get_collection(self, name: str) -> FlatCollection:
        if name in self._open:
            return self._open[name]
        if name not in self.list_collections():
            raise ValueError(f"Collection {name} does not exist.")
        col = FlatCollection(name, persist_dir=self.path, device=self.device, dtype=self.dtype)
        self._open[name] = col
        return col

    def create_collection(self, name: str, metadata: Optional[dict] = None) -> FlatCollection:
        if name in self.list_collections():
            raise ValueError(f"Collection {name} already exists.")
        col = FlatCollection(name, metadata=metadata, persist_dir=self.path, device=self.device, dtype=self.dtype)
        col.persist()
        self._open[name] = col
        return col

    def get_or_create_collection(self, name: str, metadata: Optional[dict] = None) -> FlatCollection:
        return self.get_collection(name) if name in self.list_collections() else self.create_collection(name, metadata)

    def delete_collection(self, name: str) -> None:
        self._open.pop(name, None)
        for fn in os.listdir(self.path):   # the metadata file and every index generation of this collection
            if fn == name + ".meta.json" or fn == name + ".meta.json.tmp" or (
                    fn.startswith(name + ".index") and fn.endswith(".mmiss")) or fn.startswith(name + ".journal."):
                os.remove(os.path.join(self.path, fn))

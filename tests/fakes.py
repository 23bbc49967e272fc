"""Oracle-backed stand-ins used by the CPU tests to exercise HOST logic (bookkeeping, collectives, wrappers) where
no GPU exists. They live under tests/ and are never imported by the product."""
import numpy as np

from oracle import clip_oracle as co
from oracle import retrieval_oracle as ro


class OracleIndex:
    """Same surface as mmiss_amd.index.FlatIndex, arithmetic by the retrieval oracle."""

    def __init__(self, dim, dtype="f32", device=0, capacity=0):
        self.dim, self.dtype = int(dim), dtype
        self.rows = np.zeros((0, dim), np.float16 if dtype in ("f16", 1) else np.float32)
        self.labs = np.zeros((0,), np.int64)

    def add(self, vecs, labels):
        labels = np.asarray(labels, np.int64).reshape(-1)
        if self.labs.size and labels[0] <= self.labs[-1] or np.any(np.diff(labels) <= 0):
            raise RuntimeError("labels must be strictly increasing")
        new = ro.normalize_rows(np.asarray(vecs, np.float32), self.dtype)
        self.rows = ro.concat_rows([self.rows, new]) if self.rows.shape[0] else new
        self.labs = np.concatenate([self.labs, labels])

    def update(self, labels, vecs):
        for l, v in zip(np.asarray(labels).reshape(-1), np.asarray(vecs, np.float32).reshape(-1, self.dim)):
            i, new = int(np.nonzero(self.labs == l)[0][0]), ro.normalize_rows(v, self.dtype)
            np.asarray(self.rows)[i] = np.asarray(new)[0]
            if getattr(new, "inv", None) is not None:
                self.rows.inv[i] = new.inv[0]

    def remove(self, labels):
        keep = ~np.isin(self.labs, np.asarray(labels, np.int64))
        n = int((~keep).sum())
        self.rows, self.labs = self.rows[keep], self.labs[keep]
        return n

    def clear(self):
        self.rows, self.labs = self.rows[:0], self.labs[:0]

    def count(self):
        return int(self.labs.size)

    def labels(self):
        return self.labs.copy()

    def get(self, labels):
        idx = [int(np.nonzero(self.labs == l)[0][0]) for l in np.asarray(labels).reshape(-1)]
        rows = self.rows[idx]
        return rows.represented() if hasattr(rows, "represented") else rows.astype(np.float32)

    def query(self, q, k):
        q = np.asarray(q, np.float32)
        if q.ndim == 1:
            q = q[None]
        return ro.query(q, self.rows, self.labs, k)

    def save(self, path):
        np.savez(path + ".npz", rows=self.rows, labs=self.labs)

    def load(self, path):
        z = np.load(path + ".npz")
        self.rows, self.labs = z["rows"], z["labs"]


class OracleEncoder:
    """Same surface as mmiss_amd.encoder.ClipEncoder, arithmetic by the CLIP oracle."""

    def __init__(self, shape=co.TINY, seed=0):
        self.shape = shape
        self.W = co.init_weights(shape, seed)

    def encode_image(self, pixels, out=None):
        px = np.asarray(pixels)
        if px.dtype == np.uint8:
            px = co.normalize_u8(px)
        return co.embed_images(px, self.W, self.shape)

    def encode_image_rgb(self, images, out=None):
        from oracle import resize_oracle as ro

        crops = np.stack([ro.resize_crop_u8(np.asarray(im, dtype=np.uint8), self.shape.v_image) for im in images])
        return self.encode_image(crops)

    def encode_text(self, ids, out=None):
        return co.embed_texts(np.asarray(ids), self.W, self.shape)

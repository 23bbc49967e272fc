"""Host-side steps immediately before the GPU path: CLIP image preprocessing and CLIP BPE tokenisation.

What they mirror: `processor(images=image, return_tensors="pt")` and `processor(text=[text], padding="max_length",
max_length=MAX_TOKEN_LENGTH, truncation=True)` in backend/app/utils.py:76,88 of the reference, i.e.
HF CLIPImageProcessor (HF:image_processing_clip.py:23-34) and CLIPTokenizer (HF:tokenization_clip.py:57-124).

Decode/resize stay on the host (PIL). The rescale+normalise can also run fused into the GPU patchify kernel:
`crop_images_u8` stops at the uint8 crop and ClipEncoder.encode_image accepts that uint8 batch directly.
"""
from __future__ import annotations

import json
import os
import unicodedata
from functools import lru_cache
from typing import Dict, List, Optional, Sequence

import numpy as np

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)  # transformers/utils/constants.py:5
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)  # transformers/utils/constants.py:6


def _resize_crop(img, size: int):
    """convert RGB -> resize SHORTEST edge to `size` (bicubic) -> centre crop size x size; returns a PIL image."""
    from PIL import Image

    img = img.convert("RGB")
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(size * long / short)
    new_w, new_h = (size, new_long) if w <= h else (new_long, size)
    img = img.resize((new_w, new_h), resample=Image.BICUBIC)
    left, top = (new_w - size) // 2, (new_h - size) // 2
    return img.crop((left, top, left + size, top + size))


# ------------------------------------------------------------------------------------------------ tokenizer
@lru_cache()
def _bytes_to_unicode() -> Dict[int, str]:
    """The GPT-2 byte -> printable unicode table used by byte-level BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


class ClipBPETokenizer:
    """CLIP's byte-level BPE (HF:tokenization_clip.py:57-124): NFC -> collapse whitespace -> lowercase; split with the
    CLIP pattern; bytes -> unicode table; BPE merges with the `</w>` end-of-word suffix; wrap in
    `<|startoftext|>` ... `<|endoftext|>`; pad with `<|endoftext|>`; truncate keeping the end token."""

    PATTERN = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""

    def __init__(self, vocab: Dict[str, int], merges: Sequence[str], bos_token="<|startoftext|>", eos_token="<|endoftext|>"):
        import regex

        self.encoder = dict(vocab)
        pairs = [tuple(m.split()) for m in merges if m and not m.startswith("#version")]
        self.ranks = {p: i for i, p in enumerate(pairs)}
        self.byte_encoder = _bytes_to_unicode()
        self.pat = regex.compile(self.PATTERN)
        self.ws = regex.compile(r"\s+")
        self.bos_token, self.eos_token = bos_token, eos_token
        self.bos_id, self.eos_id = self.encoder[bos_token], self.encoder[eos_token]
        self.unk_id = self.eos_id
        self._cache: Dict[str, List[str]] = {}

    @classmethod
    def from_files(cls, vocab_file: str, merges_file: str) -> "ClipBPETokenizer":
        with open(vocab_file, encoding="utf-8") as f:
            vocab = json.load(f)
        with open(merges_file, encoding="utf-8") as f:
            merges = f.read().split("\n")
        return cls(vocab, merges)

    def _bpe(self, token: str) -> List[str]:
        if token in self._cache:
            return self._cache[token]
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            best, best_rank = None, None
            for i in range(len(word) - 1):
                r = self.ranks.get((word[i], word[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (word[i], word[i + 1]), r
            if best is None:
                break
            merged, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and (word[i], word[i + 1]) == best:
                    merged.append(word[i] + word[i + 1])
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        self._cache[token] = word
        return word

    def encode(self, text: str) -> List[int]:
        text = self.ws.sub(" ", unicodedata.normalize("NFC", text)).lower()
        ids: List[int] = []
        for tok in self.pat.findall(text):
            if tok == self.bos_token or tok == self.eos_token:
                ids.append(self.encoder[tok])
                continue
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder.get(piece, self.unk_id) for piece in self._bpe(tok))
        return ids

    def __call__(self, texts: Sequence[str], max_length: int) -> np.ndarray:
        out = np.full((len(texts), max_length), self.eos_id, dtype=np.int32)  # pad token = <|endoftext|>
        for r, t in enumerate(texts):
            ids = [self.bos_id] + self.encode(t)[: max_length - 2] + [self.eos_id]
            out[r, : len(ids)] = ids
        return out


# ------------------------------------------------------------------------------------------------ processor
class ClipProcessor:
    """The opaque `processor` half of load_clip_model()'s pair."""

    def __init__(self, shape, tokenizer: Optional[ClipBPETokenizer] = None, max_length: Optional[int] = None):
        self.shape = shape
        self.image_size = int(shape.v_image)
        self.tokenizer = tokenizer
        self.max_length = int(max_length or shape.t_ctx)

    @classmethod
    def from_directory(cls, path: str, shape, max_length: Optional[int] = None) -> "ClipProcessor":
        vocab, merges = os.path.join(path, "vocab.json"), os.path.join(path, "merges.txt")
        tok = ClipBPETokenizer.from_files(vocab, merges) if os.path.exists(vocab) and os.path.exists(merges) else None
        return cls(shape, tok, max_length)

    @staticmethod
    def rgb_arrays(images: Sequence) -> list:
        """PIL images (or uint8 arrays) -> list of uint8 [H,W,3] arrays: `convert("RGB")`, the only host step left
        before ClipEncoder.encode_image_rgb (do_convert_rgb of the HF processor)."""
        out = []
        for im in images:
            if hasattr(im, "convert"):
                im = im.convert("RGB")
            out.append(np.asarray(im, dtype=np.uint8))
        return out

    def crop_images_u8(self, images: Sequence) -> np.ndarray:
        """-> uint8 [B,S,S,3]; the GPU applies x/255 and (x-mean)/std inside the patchify kernel."""
        return np.stack([np.asarray(_resize_crop(im, self.image_size), dtype=np.uint8) for im in images])

    def preprocess_images(self, images: Sequence) -> np.ndarray:
        """-> float32 [B,3,S,S], the CLIPImageProcessor output."""
        u8 = self.crop_images_u8(images)
        a = u8.astype(np.float32) * np.float32(1.0 / 255.0)
        a = (a - CLIP_MEAN) / CLIP_STD
        return np.ascontiguousarray(a.transpose(0, 3, 1, 2)).astype(np.float32)

    def tokenize(self, texts: Sequence[str], max_length: Optional[int] = None) -> np.ndarray:
        if self.tokenizer is None:
            raise RuntimeError(
                "no CLIP vocabulary available (vocab.json / merges.txt were not found next to the checkpoint); "
                "pass ready-made input_ids to generate_clip_embeddings(input_ids=...) instead")
        return self.tokenizer(list(texts), int(max_length or self.max_length))

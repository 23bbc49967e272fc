"""Host-side mirror of the reference's backend/app/utils.py for the hot path (same names, argument meaning
and error behaviour), running on the MI355X through libmmiss.

  load_clip_model()            backend/app/utils.py:27-49   -> (ClipEncoder, ClipProcessor), cached
  generate_clip_embedding()    backend/app/utils.py:59-102  -> {"image": f32[1,D], "text": f32[1,D]}, unit rows
  init_chromadb()              backend/app/utils.py:104-137 -> FlatCollection (cosine space, persistent)
  remove_background()          backend/app/utils.py:51-57   -> out of scope (rembg / U2-Net is a different model)

Where the weights come from: the reference names a hub model (`CLIP_MODEL_ID`); this build never touches the
network. `MMISS_CLIP_CHECKPOINT` must point at a local HF-layout directory (config.json + model.safetensors
[+ vocab.json, merges.txt]). For benchmarks and tests `MMISS_CLIP_RANDOM_INIT=<seed>` builds seeded random
weights of `MMISS_CLIP_SHAPE` (vit-b-32 | longclip-l-14). With neither set load_clip_model() raises.
"""
from __future__ import annotations

import dataclasses
import json
import logging
import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from .collection import FlatCollection, PersistentClient
from .encoder import LONGCLIP_L14, VIT_B32, ClipEncoder, ClipShape, random_state_dict, safetensors_shapes
from .preprocess import ClipProcessor

logger = logging.getLogger("image-match")

# Model constants (backend/app/utils.py:16-17)
CLIP_MODEL_ID = "zer0int/LongCLIP-GmP-ViT-L-14"
MAX_TOKEN_LENGTH = 248

# ChromaDB constants (backend/app/utils.py:20-21)
COLLECTION_NAME = os.getenv("COLLECTION_NAME", "image-match")
CHROMA_PERSIST_DIR = os.getenv("CHROMA_PERSIST_DIR", "chroma_data")

_SHAPES = {"vit-b-32": VIT_B32, "longclip-l-14": LONGCLIP_L14}

# Cache for models to avoid reloading (backend/app/utils.py:24-25)
_clip_model: Optional[ClipEncoder] = None
_clip_processor: Optional[ClipProcessor] = None


def _device() -> int:
    return int(os.getenv("MMISS_DEVICE", os.getenv("LOCAL_RANK", "0")))


def load_clip_model():
    """Load the CLIP towers onto the GPU and the host-side processor, cached (utils.py:27-49)."""
    global _clip_model, _clip_processor
    if _clip_model is not None and _clip_processor is not None:
        logger.info("Using cached CLIP model")
        return _clip_model, _clip_processor
    start_time = time.time()
    ckpt = os.getenv("MMISS_CLIP_CHECKPOINT")
    seed = os.getenv("MMISS_CLIP_RANDOM_INIT")
    if ckpt:
        with open(os.path.join(ckpt, "config.json")) as f:
            cfg = json.load(f)
        # the reference overrides the text context to MAX_TOKEN_LENGTH (utils.py:41-42); honour what the
        # checkpoint's position table actually holds
        shape = ClipShape.from_hf_config(cfg)
        weights = os.path.join(ckpt, "model.safetensors")
        pos = safetensors_shapes(weights).get("text_model.embeddings.position_embedding.weight")
        if pos is not None and int(pos[0]) != shape.t_ctx:  # e.g. a LongCLIP table (248 rows) under a stock config.json (77)
            shape = dataclasses.replace(shape, t_ctx=int(pos[0]))
        model = ClipEncoder(shape, device=_device())
        model.load_safetensors(weights)
        processor = ClipProcessor.from_directory(ckpt, shape, max_length=min(MAX_TOKEN_LENGTH, shape.t_ctx))
    elif seed is not None:
        shape = _SHAPES[os.getenv("MMISS_CLIP_SHAPE", "vit-b-32")]
        model = ClipEncoder(shape, device=_device())
        model.load_state_dict(random_state_dict(shape, int(seed)))
        processor = ClipProcessor(shape, tokenizer=None, max_length=min(MAX_TOKEN_LENGTH, shape.t_ctx))
    else:
        raise RuntimeError(
            f"CLIP weights for {CLIP_MODEL_ID!r} are not available offline: set MMISS_CLIP_CHECKPOINT to a local "
            "HF-layout checkpoint directory (or MMISS_CLIP_RANDOM_INIT=<seed> for seeded random weights)")
    _clip_model, _clip_processor = model, processor
    logger.info(f"CLIP model loaded in {time.time() - start_time:.2f} seconds")
    return _clip_model, _clip_processor


def set_clip_model(model: ClipEncoder, processor: ClipProcessor) -> None:
    """Install an already-built (model, processor) pair as the cached one (tests, benchmarks)."""
    global _clip_model, _clip_processor
    _clip_model, _clip_processor = model, processor


def remove_background(image):
    raise NotImplementedError("remove_background (rembg / U2-Net) is outside the embed-and-retrieve hot path")


def generate_clip_embedding(image=None, text: Optional[str] = None, model=None, processor=None) -> Dict[str, np.ndarray]:
    """Generate image and/or text embeddings (utils.py:59-102). Raises on failure, like the reference;
    the search wrappers catch and return []."""
    if model is None or processor is None:
        model, processor = load_clip_model()
    result = {}
    if image is not None:
        start_time = time.time()
        # processor(images=image) + get_image_features + "/ norm" -> f32 [1, D]: the processor's resize / crop /
        # rescale / normalise run on the GPU in front of the tower; only convert("RGB") stays on the host
        result["image"] = np.asarray(model.encode_image_rgb(processor.rgb_arrays([image])))
        logger.info(f"Image embedding generated in {time.time() - start_time:.2f} seconds")
    if text is not None:
        start_time = time.time()
        ids = processor.tokenize([text])                          # padding="max_length", truncation=True
        result["text"] = np.asarray(model.encode_text(ids))
        logger.info(f"Text embedding generated in {time.time() - start_time:.2f} seconds")
    return result


def generate_clip_embeddings(images: Optional[Sequence] = None, texts: Optional[Sequence[str]] = None, input_ids=None,
                             model=None, processor=None) -> Dict[str, np.ndarray]:
    """Batched form of generate_clip_embedding (the reference loops one item at a time, main.py:1124-1162):
    images -> f32 [B, D], texts (or ready-made input_ids [B, T]) -> f32 [B, D]."""
    if model is None or processor is None:
        model, processor = load_clip_model()
    result = {}
    if images is not None:
        result["image"] = np.asarray(model.encode_image_rgb(processor.rgb_arrays(list(images))))
    if texts is not None:
        result["text"] = np.asarray(model.encode_text(processor.tokenize(list(texts))))
    elif input_ids is not None:
        result["text"] = np.asarray(model.encode_text(np.asarray(input_ids)))
    return result


def init_chromadb() -> FlatCollection:
    """Open (or create) the persistent cosine collection (utils.py:104-137)."""
    logger.info("Initializing flat index collection...")
    os.makedirs(CHROMA_PERSIST_DIR, exist_ok=True)
    client = PersistentClient(path=CHROMA_PERSIST_DIR, device=_device())
    try:
        if COLLECTION_NAME in client.list_collections():
            collection = client.get_collection(name=COLLECTION_NAME)
            logger.info(f"Using existing collection: {COLLECTION_NAME}")
        else:
            collection = client.create_collection(name=COLLECTION_NAME, metadata={"hnsw:space": "cosine"})
            logger.info(f"Created new collection: {COLLECTION_NAME}")
    except Exception as e:
        logger.error(f"Error with collection: {e}")
        raise
    return collection

"""Host-side mirror of the search / ingest orchestration on the hot path of the reference's
backend/app/main.py (same names, argument meaning, return layout and error behaviour):

  search_similar(embedding, limit)                             main.py:748-805
  search_by_text(query_text, limit)                            main.py:807-827
  search_multimodal(image, query_text, weight_image, limit)    main.py:829-867
  process_image(...) — the embedding + collection.add portion  main.py:685-687,733-744

plus batched forms (the reference is batch-1 everywhere; BASELINE configs 2-4 are batched).
Like the reference, the wrappers catch every exception, log it and return [].
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import utils
from .collection import DuplicateIDError
from .index import blend

logger = logging.getLogger("image-match")

# module-global collection, as in the reference (main.py:77,530)
collection = None


def set_collection(col) -> None:
    global collection
    collection = col


def _collection():
    global collection
    if collection is None:
        collection = utils.init_chromadb()
    return collection


def _results_from_query(results: dict, qi: int) -> List[Dict]:
    """main.py:768-801 for one query row."""
    similar_images = []
    if not results or "ids" not in results or not results["ids"]:
        return []
    result_ids = results["ids"][qi]
    result_metadatas = results["metadatas"][qi]
    result_distances = results["distances"][qi]
    # cosine distance: 0 = identical, 2 = opposite; similarity 1 = identical, 0 = opposite (main.py:782)
    similarities = [1 - (distance / 2) for distance in result_distances]
    for i, img_id in enumerate(result_ids):
        metadata = result_metadatas[i]
        result_metadata = metadata.copy() if metadata is not None else {}
        result_metadata["similarity_score"] = similarities[i]
        if "url" not in result_metadata:
            result_metadata["url"] = f"/static/processed/{img_id}.png"
        if "thumbnail_url" not in result_metadata:
            result_metadata["thumbnail_url"] = f"/static/processed/{img_id}.png"
        similar_images.append(result_metadata)
    return similar_images


def search_similar(embedding: np.ndarray, limit: int = 10) -> List[Dict]:
    """Search for similar images using an embedding (main.py:748-805)."""
    try:
        actual_limit = 1000 if limit <= 0 else limit  # "All" option (limit of 0), main.py:757
        results = _collection().query(query_embeddings=[np.asarray(embedding, dtype=np.float32).tolist()],
                                      n_results=actual_limit, include=["metadatas", "distances"])
        out = _results_from_query(results, 0)
        logger.info(f"Found {len(out)} similar images")
        return out
    except Exception as e:
        logger.error(f"Error searching for similar images: {e}")
        return []


def search_similar_batch(embeddings: np.ndarray, limit: int = 10) -> List[List[Dict]]:
    """[Q, D] embeddings -> one result list per query, one index pass for the whole batch."""
    try:
        actual_limit = 1000 if limit <= 0 else limit
        q = np.asarray(embeddings, dtype=np.float32)
        results = _collection().query(query_embeddings=q, n_results=actual_limit, include=["metadatas", "distances"])
        return [_results_from_query(results, qi) for qi in range(q.shape[0])]
    except Exception as e:
        logger.error(f"Error searching for similar images: {e}")
        return []


def search_by_text(query_text: str, limit: int = 10) -> List[Dict]:
    """Search for images using a text query (main.py:807-827)."""
    try:
        model, processor = utils.load_clip_model()
        embedding_result = utils.generate_clip_embedding(text=query_text, model=model, processor=processor)
        text_embedding = embedding_result["text"][0]
        return search_similar(embedding=text_embedding, limit=limit)
    except Exception as e:
        logger.error(f"Error in text search: {e}")
        return []


def search_multimodal(image, query_text: str, weight_image: float = 0.5, limit: int = 10) -> List[Dict]:
    """Search using both image and text with a weighted combination (main.py:829-867); weight_image is not
    clamped, as in the backend route."""
    try:
        model, processor = utils.load_clip_model()
        image_embedding = utils.generate_clip_embedding(image=image, model=model, processor=processor)["image"][0]
        text_embedding = utils.generate_clip_embedding(text=query_text, model=model, processor=processor)["text"][0]
        # normalise both, weighted sum, normalise again (main.py:852-860) — one GPU kernel
        combined_embedding = blend(image_embedding[None], text_embedding[None], weight_image)[0]
        return search_similar(embedding=combined_embedding, limit=limit)
    except Exception as e:
        logger.error(f"Error in multimodal search: {e}")
        return []


def search_multimodal_batch(image_embeddings: np.ndarray, text_embeddings: np.ndarray, weight_image: float = 0.5,
                            limit: int = 10) -> List[List[Dict]]:
    """BASELINE config 4: a batch of (image, text) embedding pairs, blended and searched in one pass."""
    try:
        return search_similar_batch(blend(image_embeddings, text_embeddings, weight_image), limit)
    except Exception as e:
        logger.error(f"Error in multimodal search: {e}")
        return []


def process_image(image, image_id: str, metadata: Optional[Dict[str, Any]] = None, document: str = ""):
    """The embedding + collection.add portion of process_image (main.py:631-640,685-687,733-744): returns
    (metadata, True) when stored, (existing_metadata, False) for a duplicate id (the caller answers 409).
    Hashing, captioning, background removal and file writes are outside the hot path; the caller supplies the id."""
    col = _collection()
    existing = col.get(ids=[image_id], include=["metadatas"])
    if existing and existing["ids"]:
        return existing["metadatas"][0], False
    model, processor = utils.load_clip_model()
    embedding_result = utils.generate_clip_embedding(image, model=model, processor=processor)
    embedding = embedding_result["image"][0].tolist()
    metadata = dict(metadata or {})
    metadata.setdefault("id", image_id)
    try:
        col.add(ids=[image_id], embeddings=[embedding], metadatas=[metadata], documents=[document])
    except DuplicateIDError:
        # a concurrent upload of the same image won the race between the check above and this add: same answer as the
        # check would have given (the caller's 409), not a 500
        existing = col.get(ids=[image_id], include=["metadatas"])
        return (existing["metadatas"][0] if existing["ids"] else metadata), False
    return metadata, True


def process_images(images: Sequence, image_ids: Sequence[str], metadatas: Optional[Sequence[dict]] = None,
                   documents: Optional[Sequence[str]] = None) -> int:
    """Batched ingest: one encoder pass for all new images, one collection.add."""
    col = _collection()
    have = set(col.get(ids=list(image_ids), include=[])["ids"])
    todo = [i for i, iid in enumerate(image_ids) if iid not in have]
    if not todo:
        return 0
    emb = utils.generate_clip_embeddings(images=[images[i] for i in todo])["image"]
    metas = [dict((metadatas[i] if metadatas else None) or {"id": image_ids[i]}) for i in todo]
    docs = [(documents[i] if documents else "") for i in todo]
    col.add(ids=[image_ids[i] for i in todo], embeddings=emb, metadatas=metas, documents=docs)
    return len(todo)

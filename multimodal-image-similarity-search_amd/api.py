"""FastAPI stand-in for the FOUR hot routes of the reference's backend (backend/app/main.py:124-350):

    POST /api/upload              file, description, custom_metadata, remove_bg        main.py:124-175
    POST /api/search/image        file, filters[], limit                               main.py:177-232
    POST /api/search/text         query, filters[], limit                              main.py:234-293
    POST /api/search/multimodal   file, query, weight_image, filters[], limit          main.py:295-350

Same form-field names, same response shapes (`{"success": True, "metadata": ...}`, 409 with the duplicate's
metadata, `{"results": [...]}` with `similarity_score`, 500 `{"success": False, "error": ...}` on any exception), same
post-query yes/no filter on `filter_results_json` (main.py:202-222), `limit <= 0` meaning "All" (main.py:757).
This is SURVEY.md §8(f) N1: the caller of the hot path, kept thin; everything numeric goes through
mmiss_amd.utils / mmiss_amd.search. The other ten routes (metadata editing, Moondream filters, reset, static
files) are out of scope.

The reference declares `Form`/`File` parameters, which need the `python-multipart` package; it is not installed in
this image, so the routes read the raw body and parse multipart/form-data with the standard library.
Image ids: the reference uses a perceptual hash (`imagehash.phash`, main.py:581-585; the package is absent here);
`phash_hex` restates its published algorithm, so ids have the reference's form and the same image uploaded twice — in any
lossless container — is a duplicate (409), as there.
"""
import hashlib
import json
import logging
from email.parser import BytesParser
from email.policy import HTTP
from io import BytesIO
from typing import Dict, List, Optional, Tuple
from urllib.parse import parse_qs

from . import search, utils

logger = logging.getLogger("image-match")


def parse_form(content_type: str, body: bytes) -> Tuple[Dict[str, List[str]], Dict[str, Tuple[str, bytes]]]:
    """multipart/form-data or application/x-www-form-urlencoded -> (fields: name -> [values], files: name -> (filename, bytes))."""
    fields: Dict[str, List[str]] = {}
    files: Dict[str, Tuple[str, bytes]] = {}
    ctype = (content_type or "").lower()
    if ctype.startswith("multipart/form-data"):
        msg = BytesParser(policy=HTTP).parsebytes(
            b"Content-Type: " + content_type.encode("latin-1") + b"\r\nMIME-Version: 1.0\r\n\r\n" + body)
        for part in msg.iter_parts():
            name = part.get_param("name", header="content-disposition")
            if name is None:
                continue
            filename = part.get_filename()
            payload = part.get_payload(decode=True) or b""
            if filename is not None:
                files[name] = (filename, payload)
            else:
                fields.setdefault(name, []).append(payload.decode(part.get_content_charset() or "utf-8", "replace"))
    elif ctype.startswith("application/x-www-form-urlencoded"):
        for k, v in parse_qs(body.decode("utf-8", "replace"), keep_blank_values=True).items():
            fields[k] = list(v)
    return fields, files


def _first(fields, name, default=None):
    v = fields.get(name)
    return v[0] if v else default


def _as_bool(s) -> bool:
    return str(s).strip().lower() in ("1", "true", "yes", "on")


def apply_filters(results: List[dict], filters: Optional[List[str]]) -> List[dict]:
    """Keep r iff every selected filter answered "yes" in r["filter_results_json"] (main.py:202-222)."""
    if not filters:
        return results
    kept = []
    for r in results:
        filter_results = {}
        if "filter_results_json" in r:
            try:
                filter_results = json.loads(r["filter_results_json"])
            except (json.JSONDecodeError, TypeError):
                logger.warning(f"Error parsing filter_results_json for image {r.get('id')}")
        if all(str(filter_results.get(f, "")).lower().strip() == "yes" for f in filters):
            kept.append(r)
    return kept


def phash_hex(image, hash_size: int = 8, highfreq_factor: int = 4) -> str:
    """Perceptual hash, restated from the published algorithm of `imagehash.phash` (the third-party package behind the
    reference's generate_image_hash, main.py:581-585; absent here, so this restatement is unpinned): grayscale, LANCZOS
    resize to 32x32, 2-D DCT-II, the 8x8 low-frequency corner thresholded at its median, bits row-major -> 16 hex digits.
    Bookkeeping, not similarity arithmetic: it only decides which uploads count as duplicates (HTTP 409)."""
    import numpy as np
    from PIL import Image
    from scipy.fft import dct

    size = hash_size * highfreq_factor
    pixels = np.asarray(image.convert("L").resize((size, size), Image.Resampling.LANCZOS), dtype=np.float64)
    low = dct(dct(pixels, axis=0, type=2), axis=1, type=2)[:hash_size, :hash_size]
    bits = (low > np.median(low)).flatten()
    width = (bits.size + 3) // 4
    return "{:0>{width}x}".format(int("".join("1" if b else "0" for b in bits), 2), width=width)


def image_id_for(image) -> str:
    """`img_<phash>` as generate_image_hash (main.py:581-585); falls back to a pixel digest without scipy."""
    try:
        return "img_" + phash_hex(image)
    except ImportError:
        rgb = image.convert("RGB")
        return "img_" + hashlib.sha1(rgb.tobytes() + str(rgb.size).encode()).hexdigest()[:16]


def create_app():
    """Build the FastAPI app. Model and collection come from mmiss_amd.utils / mmiss_amd.search (lazy, cached)."""
    from fastapi import FastAPI, Request
    from fastapi.responses import JSONResponse
    from PIL import Image

    app = FastAPI(title="mmiss_amd hot-path stand-in")

    async def _form(request: Request):
        return parse_form(request.headers.get("content-type", ""), await request.body())

    def _open(files, convert_all: bool):
        if "file" not in files:
            raise ValueError("field 'file' is required")
        image = Image.open(BytesIO(files["file"][1]))
        if convert_all:
            return image.convert("RGB")                      # main.py:189,313
        return image if image.mode in ("RGB", "L") else image.convert("RGB")  # main.py:140-143

    def _error(e: Exception):
        return JSONResponse(status_code=500, content={"success": False, "error": str(e)})

    @app.post("/api/upload")
    async def upload_image(request: Request):
        try:
            fields, files = await _form(request)
            image = _open(files, convert_all=False)
            filename = files["file"][0]
            description = _first(fields, "description")
            custom_metadata = _first(fields, "custom_metadata")
            if _as_bool(_first(fields, "remove_bg", "false")):
                raise NotImplementedError("remove_bg is outside the embed-and-retrieve hot path")
            image_id = image_id_for(image)
            metadata = {"id": image_id, "filename": filename, "description": description or "",
                        "custom_metadata": custom_metadata or "", "url": f"/static/processed/{image_id}.png",
                        "thumbnail_url": f"/static/processed/{image_id}.png", "filter_results_json": "{}"}
            metadata, success = search.process_image(image, image_id, metadata, document=description or "")
            if success:
                return {"success": True, "metadata": metadata}
            return JSONResponse(status_code=409, content={
                "success": False, "error": "Duplicate image",
                "message": "This image already exists in the database", "metadata": metadata})
        except Exception as e:  # same catch-all as the reference (main.py:170-175)
            logger.error(f"Upload error: {e}")
            return _error(e)

    @app.post("/api/search/image")
    async def search_by_image(request: Request):
        try:
            fields, files = await _form(request)
            image = _open(files, convert_all=True)
            limit = int(_first(fields, "limit", 10))
            model, processor = utils.load_clip_model()
            embedding_result = utils.generate_clip_embedding(image=image, model=model, processor=processor)
            results = search.search_similar(embedding=embedding_result["image"][0], limit=limit)
            return {"results": apply_filters(results, fields.get("filters"))}
        except Exception as e:
            logger.error(f"Error in image search: {e}")
            return _error(e)

    @app.post("/api/search/text")
    async def search_by_text_route(request: Request):
        try:
            fields, _ = await _form(request)
            if "query" not in fields:
                return JSONResponse(status_code=422, content={"detail": "field 'query' is required"})
            query = fields["query"][0]
            limit = int(_first(fields, "limit", 10))
            filters = fields.get("filters")
            if not query.strip() and filters:
                # filter-only browse (main.py:245-249,1225-1242): every stored image, no vector search
                got = search._collection().get(include=["metadatas"])
                n = 1000 if limit <= 0 else limit
                results = [dict(m or {}) for m in got["metadatas"]][:n]
            else:
                results = search.search_by_text(query_text=query, limit=limit)
            return {"results": apply_filters(results, filters)}
        except Exception as e:
            logger.error(f"Error in text search: {e}")
            return _error(e)

    @app.post("/api/search/multimodal")
    async def search_multimodal_route(request: Request):
        try:
            fields, files = await _form(request)
            image = _open(files, convert_all=True)
            if "query" not in fields:
                return JSONResponse(status_code=422, content={"detail": "field 'query' is required"})
            weight_image = float(_first(fields, "weight_image", 0.5))  # not clamped, as in the backend route
            limit = int(_first(fields, "limit", 10))
            results = search.search_multimodal(image=image, query_text=fields["query"][0], weight_image=weight_image,
                                               limit=limit)
            return {"results": apply_filters(results, fields.get("filters"))}
        except Exception as e:
            logger.error(f"Error in multimodal search: {e}")
            return _error(e)

    return app

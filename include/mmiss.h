/*
 * mmiss.h — C-ABI of libmmiss.so, the MI355X (gfx950) embed-and-retrieve hot path.
 *
 * This is the drop-in boundary for ONE path of parsakhaz/multimodal-image-similarity-search:
 *   image / text -> CLIP tower -> L2-normalise -> cosine top-k over a flat in-HBM index.
 * The reference has no FFI of its own (it is pure Python over transformers + chromadb); each entry
 * point below names the reference call site whose arithmetic it replaces (paths relative to the
 * reference root; "HF:" = transformers/models/clip/).
 *
 * Conventions
 *   - plain C types only: pointers, sizes, opaque handles. No torch / HIP types in any signature.
 *   - every data pointer may be a HOST pointer or a DEVICE (HBM) pointer of the handle's GPU; the
 *     library detects which (hipPointerGetAttributes) and stages host data itself. Outputs likewise.
 *   - caller allocates every input and output buffer; the library owns only its handles and the HBM
 *     behind them (weights, index rows, workspaces). It never retains caller memory past a call.
 *   - every function returns an int status: 0 = MMISS_OK, <0 = error class; mmiss_last_error() returns
 *     a thread-local message. The Python shim turns non-zero into RuntimeError so the reference's
 *     `except Exception` paths (backend/app/main.py:803-805,825-827,865-867) behave the same.
 *   - a handle is internally serialised (one mutex + one HIP stream per handle); different handles are
 *     independent. Index mutation is safe against a concurrent query on the same handle.
 *   - there is NO CPU fallback anywhere in this library: with no usable gfx950 device every compute
 *     entry point fails with MMISS_ERR_HIP.
 */
#ifndef MMISS_H
#define MMISS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMISS_ABI_VERSION 1

enum {
    MMISS_OK = 0,
    MMISS_ERR_ARG = -1,         /* bad argument (null, shape, range)                 */
    MMISS_ERR_HIP = -2,         /* HIP runtime / device failure                      */
    MMISS_ERR_STATE = -3,       /* call order (e.g. encode before finalize)          */
    MMISS_ERR_NOMEM = -4,       /* host or device allocation failed                  */
    MMISS_ERR_UNSUPPORTED = -5, /* shape/dtype the kernels do not cover              */
    MMISS_ERR_IO = -6           /* file read/write failed (index save/load)          */
};

/* index storage dtypes. MMISS_F8: one byte per element, OCP e4m3 with the fixed scale 2^7 — a code stands for the value
 * decode(byte) / 128; a unit-norm row has |x_i| <= 1, so 128 x_i fits e4m3 and a typical component keeps its 3 mantissa
 * bits — plus ONE float per row, the inverse of the canonical norm of the row's values (+0.8 % bytes at dim 512; derived from
 * the codes, so save files do not hold it). The row the index REPRESENTS is values x inverse norm: a unit vector, so what an
 * fp8 index returns is a cosine distance, 1 - cos(query, represented row) (the reference turns it into
 * similarity = 1 - d / 2, backend/app/main.py:782, on that premise; round 4 returned 1 - |values| cos with |values| within
 * 3 % of 1). Distances are exact (canonical fp64, x the inverse norm) with respect to the represented rows, as for f16; the
 * represented rows themselves are 2^-4-coarse in every component: a row's direction lies within 1 - cos <= 4e-3 of the
 * vector that was added, so an fp8 index ranks like the f32 / f16 index up to that (a row is found first by its own
 * unquantised vector as long as its neighbours are further away than that; mmiss_index_get returns the represented rows).
 * Up to 128 queries per call take the streaming scan (half the bytes of f16 per row), from 129 on the score GEMM with the
 * codes widened to f16 in its operand load; both scale every score by the row's inverse norm. */
enum { MMISS_F32 = 0, MMISS_F16 = 1, MMISS_F8 = 2 };

typedef struct mmiss_encoder mmiss_encoder;
typedef struct mmiss_index mmiss_index;

/* ---------------------------------------------------------------- misc ------------------------- */
int mmiss_abi_version(void);
const char* mmiss_last_error(void);
/* number of visible HIP devices; does not create a context on any of them */
int mmiss_device_count(int* count);

/* ---------------------------------------------------------------- encoder ---------------------- */
/*
 * Shape-parametric CLIP (two towers + projections). Defaults of HF CLIPConfig() are ViT-B/32
 * (HF:configuration_clip.py:47-54,97-105,160). The reference HEAD loads a ViT-L/14 LongCLIP with
 * text_ctx = 248 (backend/app/utils.py:16-17,41-45); both are instances of this struct.
 * head_dim = hidden / heads must be 64; hidden and mlp must be multiples of 128.
 */
typedef struct mmiss_clip_config {
    int32_t struct_size;   /* = sizeof(mmiss_clip_config), ABI guard */
    int32_t v_hidden, v_layers, v_heads, v_mlp, v_patch, v_image; /* 768,12,12,3072,32,224 */
    int32_t t_hidden, t_layers, t_heads, t_mlp, t_vocab, t_ctx;   /* 512,12, 8,2048,49408,77 */
    int32_t proj_dim;      /* 512 */
    int32_t eos_token_id;  /* 49407; the legacy value 2 selects argmax(ids) pooling (HF:modeling_clip.py:561-581) */
    float ln_eps;          /* 1e-5 */
    int32_t max_batch_image; /* workspace sizing; larger batches are processed in chunks of this */
    int32_t max_batch_text;
} mmiss_clip_config;

/* replaces load_clip_model()'s model half — backend/app/utils.py:27-49 */
int mmiss_encoder_create(const mmiss_clip_config* cfg, int device, mmiss_encoder** out);
int mmiss_encoder_destroy(mmiss_encoder* enc);

/*
 * Weights arrive by their HF state_dict key (SURVEY.md §8 a-W), host float32, row-major, e.g.
 *   "vision_model.encoder.layers.3.self_attn.q_proj.weight"  [768,768]
 *   "text_model.embeddings.token_embedding.weight"            [49408,512]
 * Unknown keys ("logit_scale", "*.position_ids") are accepted and ignored (returns MMISS_OK and sets
 * *used = 0 when used != NULL). The library converts GEMM weights to bf16 and fuses q/k/v.
 * replaces CLIPModel.from_pretrained(...)'s tensor load — backend/app/utils.py:44
 */
int mmiss_encoder_set_weight(mmiss_encoder* enc, const char* hf_key, const float* data, int64_t numel, int* used);
/* checks that every tensor of both towers was supplied; must precede encode calls */
int mmiss_encoder_finalize(mmiss_encoder* enc);
/*
 * Arithmetic of the towers' large GEMMs. MMISS_PREC_BF16 (default): bf16 operands, f32 accumulation; in calls of at
 * least ~6000 token rows (hidden <= 768: LayerNorm folded into the GEMMs) the residual stream between the layers is
 * bf16 as well, as in any bf16 deployment of the model (measured 1 - cos vs the fp32 reference arithmetic: 5e-5; with an
 * f32 stream 5e-6). MMISS_PREC_BF16_F32RESID: bf16 operands, the residual stream f32 at every batch size (the round-1
 * behaviour; 4-5 % slower at 256 images). MMISS_PREC_FP8 (BASELINE.json configs[4] "ViT-L/14 fp8 MFMA encode"):
 * the QKV, FC1 and FC2 projections of the VISION tower run on the block-scaled fp8 matrix cores (OCP e4m3 operands, E8M0
 * block scales on the activations, per-output-channel scales on the weights, f32 accumulation) in calls of at least 1024
 * token rows — and, in calls on the bf16 residual stream (>= ~6000 token rows), the out-projection too, its A operand being
 * the attention kernel's MXFP8 output; the attention arithmetic, LayerNorm statistics and the projection head keep their
 * precision, and the residual stream is the same as under MMISS_PREC_BF16: bf16 in calls of at least ~6000 token rows, f32
 * below. Measured 1 - cos vs the fp32 reference arithmetic: 5e-4 at full ViT-L/14 depth, 6e-4 on ViT-B/32 at batch 256
 * (seeded Gaussian weights; asserted at 1e-3). NOT covered by that bar: a checkpoint with residual OUTLIER channels. An e4m3
 * weight keeps three mantissa bits at any magnitude, so the weight column that meets a channel of magnitude 300 carries a
 * rounding error comparable to the signal of all ordinary channels: with +300 / -180 channels planted, ViT-L/14 (width 1024)
 * measures 6.7e-4 — inside — but ViT-B/32 (width 768) 1.15e-3 — OUTSIDE the tolerance (tests/test_headline_gpu.py prints
 * both; MMISS_PREC_BF16 holds 5e-5 on the same weights). MMISS_PREC_FP8 is therefore an OPT-IN, to be verified per checkpoint
 * against the default path (bench.py prints that gap); the default is MMISS_PREC_BF16. The TEXT tower stays on the bf16
 * kernels under this setting: its fp8 form measures 3.3-3.9e-3, outside the 1e-3 tolerance (three mantissa bits put ~5 %
 * noise on every GEMM output and the text stream is built almost entirely from GEMM outputs).
 * mmiss_encoder_set_tower_precision switches ONE tower (MMISS_TOWER_VISION / MMISS_TOWER_TEXT) to MMISS_PREC_BF16 or
 * MMISS_PREC_FP8 explicitly; fp8 on the text tower is an opt-in outside the tolerance the other settings are held to.
 * Both may be called before or after finalize. The reference runs fp32 on the CPU (backend/app/utils.py:77,97); every
 * default setting is held to the same bar against it (1 - cos <= 1e-3, asserted in tests/test_fp8_gpu.py).
 */
enum { MMISS_PREC_BF16 = 0, MMISS_PREC_FP8 = 1, MMISS_PREC_BF16_F32RESID = 2 };
enum { MMISS_TOWER_VISION = 0, MMISS_TOWER_TEXT = 1 };
int mmiss_encoder_set_precision(mmiss_encoder* enc, int32_t precision);
int mmiss_encoder_set_tower_precision(mmiss_encoder* enc, int32_t tower, int32_t precision);
/*
 * use_own != 0 (the default after create): calls run on the handle's private stream and return after the
 * work has finished (host-synchronous). use_own == 0: calls are enqueued on the caller's hipStream_t
 * (passed as void*; NULL = the legacy default stream) and return without synchronising unless an
 * input or output lives in host memory. Changing the stream waits for the work the handle's LAST call left on the
 * stream being left (its workspaces are reused), not for anything else the caller has queued there since.
 * Threads: every call locks its handle, but the stream is handle state — a thread that sets the stream and then calls
 * must keep other threads off that handle in between (the Python wrappers hold a per-handle lock around the pair);
 * different handles never interact (several batches in flight: one encoder handle per host thread / stream,
 * mmiss_amd/pipeline.py).
 */
int mmiss_encoder_set_stream(mmiss_encoder* enc, void* hip_stream, int32_t use_own);

/*
 * pixels: float32 [B,3,S,S] NCHW, already CLIP-normalised (the output of CLIPImageProcessor,
 *         HF:image_processing_clip.py:23-34). out: float32 [B,proj_dim], rows unit-norm.
 * replaces model.get_image_features(**inputs) + "/ norm" — backend/app/utils.py:77-78
 *          (HF:modeling_clip.py:719-753,613-656)
 */
int mmiss_encode_image(mmiss_encoder* enc, const float* pixels, int32_t B, float* out);

/*
 * pixels_u8: uint8 [B,S,S,3] HWC RGB, already resized+centre-cropped to SxS; the kernel applies
 * x/255, (x-mean)/std (HF:image_processing_clip.py:23-34; mean/std transformers/utils/constants.py:5-6)
 * in the patchify prologue. Same output as mmiss_encode_image.
 */
int mmiss_encode_image_u8(mmiss_encoder* enc, const uint8_t* pixels_u8, int32_t B, float* out);

/*
 * Raw decoded images of ANY size (what PIL hands the processor at backend/app/utils.py:76): B tightly packed RGB8
 * images (row stride 3*W, no padding) laid end to end in `rgb` (host or device, rgb_bytes long); image b starts at
 * byte offsets[b] and is heights[b] x widths[b] (host arrays). The library does the CLIPImageProcessor geometry on
 * the GPU — resize so the shortest edge is S = v_image with PIL's bicubic filter (long edge int(S*long/short)),
 * centre crop S x S (HF:image_processing_clip.py:23-34) — bit-identical to Pillow's 8-bit Image.resize, then
 * rescale/normalise/encode as mmiss_encode_image_u8. JPEG/PNG decoding and convert("RGB") stay with the caller.
 *   mmiss_resize_crop_rgb   out_u8: uint8 [B,S,S,3] (host or device) — the crops themselves;
 *   mmiss_encode_image_rgb  out: float32 [B,proj_dim] unit rows.
 * Errors: sizes outside 1..65536, bytes outside the blob -> MMISS_ERR_ARG; a downscale needing more than 4096 filter
 * taps per output pixel -> MMISS_ERR_UNSUPPORTED.
 */
int mmiss_resize_crop_rgb(mmiss_encoder* enc, const uint8_t* rgb, int64_t rgb_bytes, const int64_t* offsets,
                          const int32_t* heights, const int32_t* widths, int32_t B, uint8_t* out_u8);
int mmiss_encode_image_rgb(mmiss_encoder* enc, const uint8_t* rgb, int64_t rgb_bytes, const int64_t* offsets,
                           const int32_t* heights, const int32_t* widths, int32_t B, float* out);

/*
 * ids: int32 [B,T], T <= t_ctx, rows = BOS ... EOS then padding (CLIPTokenizer output,
 *      backend/app/utils.py:88). out: float32 [B,proj_dim], rows unit-norm. The pooled row is the
 *      first EOS position, so the padding mask cannot change the result (causal attention).
 * replaces model.get_text_features(**inputs) + "/ norm" — backend/app/utils.py:97-98
 *          (HF:modeling_clip.py:683-715,513-586)
 */
int mmiss_encode_text(mmiss_encoder* enc, const int32_t* ids, int32_t B, int32_t T, float* out);

/*
 * Debug tap used by the parity tests to bisect against the oracle's intermediates.
 * tower: 0 = vision, 1 = text. what: 0 = embeddings (after pre-LN for vision), 1..L = residual
 * stream after layer i, 100 = pooled+post-LN row (bf16 widened), 101 = projected (pre-normalise).
 * Copies min(cap, available) floats of the LAST encode call's buffer; *written gets the count.
 */
int mmiss_encoder_tap(mmiss_encoder* enc, int tower, int what, float* out, int64_t cap, int64_t* written);

/* ---------------------------------------------------------------- flat index ------------------- */
/*
 * Flat cosine index resident in HBM: rows are L2-normalised at add time and stored as f32, f16 or fp8 (e4m3).
 * dim: a multiple of 128; f32 rows up to 1920, f16 / fp8 rows up to 3968 (the scan stages a 16-query block beside its
 * lists in one CU's 160 KB of LDS) — anything else is MMISS_ERR_UNSUPPORTED here, not a failed launch at the first query.
 * replaces chromadb's collection with metadata {"hnsw:space":"cosine"} — backend/app/utils.py:104-137
 */
int mmiss_index_create(int32_t dim, int32_t storage_dtype, int device, int64_t capacity_hint, mmiss_index** out);
int mmiss_index_destroy(mmiss_index* idx);
/* same contract as mmiss_encoder_set_stream */
int mmiss_index_set_stream(mmiss_index* idx, void* hip_stream, int32_t use_own);

/*
 * vecs: float32 [n,dim] (any norm > 0). labels: int64 [n], strictly increasing and greater than every
 * label already stored (they are the tie-break order; the Collection shim hands out sequence numbers).
 * replaces collection.add(ids, embeddings, ...) — backend/app/main.py:735-740
 */
int mmiss_index_add(mmiss_index* idx, const float* vecs, const int64_t* labels, int64_t n);
/* overwrite the rows of existing labels (labels: host int64 [n]) — collection.update(..., embeddings=) */
int mmiss_index_update(mmiss_index* idx, const int64_t* labels, const float* vecs, int64_t n);
/* stable removal (row order = label order is preserved). labels: host. — collection.delete(ids), main.py:1069 */
int mmiss_index_remove(mmiss_index* idx, const int64_t* labels, int64_t n, int64_t* removed);
int mmiss_index_clear(mmiss_index* idx);
/* collection.count() — init_db.py:58 */
int mmiss_index_count(mmiss_index* idx, int64_t* count);
/* stored (normalised, storage-rounded) rows widened to f32 [n,dim]; labels: host. Missing label -> MMISS_ERR_ARG */
int mmiss_index_get(mmiss_index* idx, const int64_t* labels, int64_t n, float* out);
/* all labels in row order (host int64 [count]) */
int mmiss_index_labels(mmiss_index* idx, int64_t* out, int64_t cap);

/*
 * queries: float32 [Q,dim] (any norm > 0; normalised internally). k >= 1.
 * out_labels int64 [Q,k], out_dist float32 [Q,k] = cosine distance 1 - cos, ascending; ties by label
 * ascending. out_count int32 [Q] = min(k, count); unused slots hold label -1 / distance +inf.
 * k larger than count is not an error (the UI's "All" sends 1000 — backend/app/main.py:757).
 * A vector without a direction — zero norm, or a NaN / Inf component: 0/0 in the normalisation — has no distance to
 * anything: such a ROW is stored but never returned (out_count then counts the rows that have a distance), such a QUERY has
 * no results (out_count 0). Not an error; the reference never sends one (backend/app/utils.py:88-99 always hands chromadb a
 * CLIP embedding). tests/test_index_gpu.py: test_zero_and_non_finite_vectors_are_never_results, all three storage types.
 * Synchronisation: the exactness guard (below) decides on the host whether any query must be widened (the last kernel
 * leaves one flag per query in a pinned host block; no copy-engine operation), so the call waits for its own work on the
 * stream it runs on — also on a caller's stream with device outputs (mmiss_index_set_stream(.., use_own = 0)), and it
 * cannot be captured into a HIP graph (the encode calls can). mmiss_index_query_begin / _end split that wait off.
 * replaces collection.query(query_embeddings, n_results, include=["metadatas","distances"]) —
 *          backend/app/main.py:761-765
 */
int mmiss_index_query(mmiss_index* idx, const float* queries, int32_t Q, int32_t k,
                      int64_t* out_labels, float* out_dist, int32_t* out_count);
/*
 * The same query in two halves, for a caller that keeps the GPU fed while it waits (a serving loop: queue the NEXT batch's
 * encode between the two calls, bench.py does). _begin queues the first pass on the index's stream and returns without
 * waiting; _end waits for exactly that work (an event, not the stream: whatever the caller queued behind it keeps running),
 * runs the guard's widen pass for the queries that need it, and fills host outputs. mmiss_index_query = _begin + _end.
 * Between the two calls `queries` and the output buffers must stay valid, and every other call on this index except
 * count / labels / guard_stats fails with MMISS_ERR_STATE (the scratch buffers belong to the open query); _end without
 * _begin -> MMISS_ERR_STATE. No reference analogue: collection.query is synchronous (backend/app/main.py:761-765).
 */
int mmiss_index_query_begin(mmiss_index* idx, const float* queries, int32_t Q, int32_t k,
                            int64_t* out_labels, float* out_dist, int32_t* out_count);
int mmiss_index_query_end(mmiss_index* idx);
/*
 * Give up a query opened with _begin (the caller lost interest: an exception in the serving loop between the two halves):
 * waits for the queued first pass — it still writes the output buffers, which must stay valid until this returns — and
 * re-opens the handle for other calls. No results are delivered. Without an open query: MMISS_OK, nothing happens.
 */
int mmiss_index_query_abort(mmiss_index* idx);
/*
 * Exactness accounting since the index was created ("top-10 recall = 1.0" is proven per query, not assumed): the first
 * pass ranks by approximate matrix-core scores and keeps k' > k rows; a query whose k-th exact score c_k does not clear the
 * best score that pass may have left out by more than the arithmetic's error bound eps_q is widened: ONE threshold pass
 * over the index for all such queries of the call collects every row whose approximate score reaches c_k - eps_q (every
 * row that can still belong to the top-k), and the exact re-rank of those rows is the answer.
 * out[0] = queries served, out[1] = queries widened, out[2] = widen passes run (one per call that had to widen),
 * out[3] = extra passes over the index those cost (1 per widen pass on the score GEMM, one per 64 widened queries on the
 * streaming scan). No reference analogue (chromadb's HNSW is approximate above 100 rows).
 */
int mmiss_index_guard_stats(mmiss_index* idx, int64_t out[4]);
/*
 * The same four counters and: out[4] = the queries whose threshold pass collected more than 8192 rows and went through the
 * exhaustive canonical pass instead (the canonical distance of every row, the k best by (distance, label)) — a plateau of
 * rows within the rounding bound of the k-th score, e.g. > 10^4 copies of one placeholder image; exact for any data, one
 * pass over the index and one host selection per such query. out[5] = rows the threshold passes collected and re-ranked.
 * out[6..7] are reserved (0).
 */
int mmiss_index_guard_stats_ex(mmiss_index* idx, int64_t out[8]);

/* persistence of rows + labels (replaces chroma_data/, backend/app/utils.py:21,113) */
int mmiss_index_save(mmiss_index* idx, const char* path);
int mmiss_index_load(mmiss_index* idx, const char* path);

/* ---------------------------------------------------------------- glue kernels ----------------- */
/*
 * out[q] = normalize(w * normalize(img[q]) + (1-w) * normalize(txt[q])), float32 [Q,dim].
 * w is applied as numpy does for `python_float * float32_array`: (float)w and (float)(1.0 - w).
 * replaces search_multimodal's blend — backend/app/main.py:852-860
 */
int mmiss_blend(int device, void* hip_stream, const float* img, const float* txt, double w,
                int32_t Q, int32_t dim, float* out);

/*
 * Merge S per-shard result lists (as produced by mmiss_index_query on each shard, then all-gathered)
 * into the global top-k: dist float32 [S,Q,k], labels int64 [S,Q,k] -> out [Q,k], ordered by
 * (distance asc, label asc), label -1 entries ignored. Any S >= 1 and k <= 4096: up to 8192 entries per query are
 * merged in one pass (8 shards x the UI's "All" = 1000 hits, backend/app/main.py:757), more in several levels.
 * No reference analogue (the reference is single-process); this is the one exchange step of the row-sharded
 * index.
 */
int mmiss_merge_topk(int device, void* hip_stream, const float* dist, const int64_t* labels,
                     int32_t S, int32_t Q, int32_t k, float* out_dist, int64_t* out_labels, int32_t* out_count);

/* ---------------------------------------------------------------- kernel timing ---------------- */
/*
 * When enabled every kernel launch of this library is bracketed by HIP events on its stream and
 * accumulated per kernel class. mmiss_prof_read drains finished events and writes a JSON array
 *   [{"kernel": "...", "launches": n, "ms": total, "flops": algorithmic, "bytes": algorithmic}, ...]
 * into buf (NUL-terminated, truncated to cap). Used by bench.py for the roofline object.
 */
int mmiss_prof_enable(int on);
/* restrict the bracketing to one kernel class and to every stride-th launch of it (kernel = NULL or "" lifts the
 * restriction). Lets bench.py time the dominant kernel INSIDE its timed region at negligible cost. */
int mmiss_prof_filter(const char* kernel, int stride);
int mmiss_prof_reset(void);
int mmiss_prof_read(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* MMISS_H */

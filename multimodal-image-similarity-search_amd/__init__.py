"""mmiss_amd — MI355X-native embed-and-retrieve hot path of parsakhaz/multimodal-image-similarity-search.

Host-side mirror of the reference's function boundary (backend/app/utils.py, backend/app/main.py:748-867)
over the C-ABI library libmmiss.so (include/mmiss.h). Import as ``mmiss_amd`` (see /mmiss_amd.py).

  mmiss_amd.utils       load_clip_model, generate_clip_embedding, init_chromadb, constants
  mmiss_amd.search      search_similar, search_by_text, search_multimodal, process_image (embedding + add part)
  mmiss_amd.collection  FlatCollection — the chromadb Collection surface the reference uses
  mmiss_amd.encoder     ClipEncoder — batched image / text encode on the GPU
  mmiss_amd.index       FlatIndex — in-HBM cosine index (single shard)
  mmiss_amd.sharded     ShardedIndex — row-sharded index, one process per GPU, RCCL all-gather of per-shard top-k
"""
__version__ = "0.1.0"

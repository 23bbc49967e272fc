/*
 * abi_demo.c — a plain C consumer of libmmiss.so (no Python, no HIP headers): the drop-in boundary of
 * include/mmiss.h exercised end to end on a small CLIP shape with pseudo-random weights.
 *
 *   gcc -O2 -std=c99 -I include examples/abi_demo.c -L multimodal-image-similarity-search_amd -lmmiss \
 *       -Wl,-rpath,$PWD/multimodal-image-similarity-search_amd -lm -o /tmp/abi_demo && /tmp/abi_demo
 *
 * It follows the reference's call sequence — load the model (backend/app/utils.py:27-49), embed images and a
 * prompt (utils.py:59-102), add to the cosine collection (main.py:735-740), query it (main.py:761-765), blend
 * image and text queries (main.py:852-860) — through the C entry points that replace those call sites.
 * Exit code 0 = every check passed.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mmiss.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int _rc = (call);                                                                    \
        if (_rc != MMISS_OK) {                                                               \
            fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, _rc, mmiss_last_error()); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float frand(void) { /* uniform in [-1, 1) */
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((rng_state >> 40) / 8388608.0 - 1.0);
}

static int set_tensor(mmiss_encoder* enc, const char* key, long long numel, float scale, float offset) {
    float* buf = (float*)malloc(sizeof(float) * (size_t)numel);
    if (!buf) return MMISS_ERR_NOMEM;
    for (long long i = 0; i < numel; ++i) buf[i] = offset + scale * frand();
    int used = 0;
    int rc = mmiss_encoder_set_weight(enc, key, buf, numel, &used);
    free(buf);
    if (rc == MMISS_OK && !used) {
        fprintf(stderr, "weight %s was not recognised\n", key);
        return MMISS_ERR_ARG;
    }
    return rc;
}

static int set_tower(mmiss_encoder* enc, const char* prefix, int d, int layers, int mlp) {
    char key[256];
    static const char* proj[] = {"q_proj", "k_proj", "v_proj", "out_proj"};
    for (int l = 0; l < layers; ++l) {
        for (int p = 0; p < 4; ++p) {
            snprintf(key, sizeof key, "%s.encoder.layers.%d.self_attn.%s.weight", prefix, l, proj[p]);
            if (set_tensor(enc, key, (long long)d * d, 0.08f, 0.f)) return 1;
            snprintf(key, sizeof key, "%s.encoder.layers.%d.self_attn.%s.bias", prefix, l, proj[p]);
            if (set_tensor(enc, key, d, 0.02f, 0.f)) return 1;
        }
        for (int n = 1; n <= 2; ++n) {
            snprintf(key, sizeof key, "%s.encoder.layers.%d.layer_norm%d.weight", prefix, l, n);
            if (set_tensor(enc, key, d, 0.1f, 1.f)) return 1;
            snprintf(key, sizeof key, "%s.encoder.layers.%d.layer_norm%d.bias", prefix, l, n);
            if (set_tensor(enc, key, d, 0.1f, 0.f)) return 1;
        }
        snprintf(key, sizeof key, "%s.encoder.layers.%d.mlp.fc1.weight", prefix, l);
        if (set_tensor(enc, key, (long long)mlp * d, 0.06f, 0.f)) return 1;
        snprintf(key, sizeof key, "%s.encoder.layers.%d.mlp.fc1.bias", prefix, l);
        if (set_tensor(enc, key, mlp, 0.02f, 0.f)) return 1;
        snprintf(key, sizeof key, "%s.encoder.layers.%d.mlp.fc2.weight", prefix, l);
        if (set_tensor(enc, key, (long long)d * mlp, 0.04f, 0.f)) return 1;
        snprintf(key, sizeof key, "%s.encoder.layers.%d.mlp.fc2.bias", prefix, l);
        if (set_tensor(enc, key, d, 0.02f, 0.f)) return 1;
    }
    return 0;
}

static double row_norm(const float* v, int d) {
    double s = 0;
    for (int i = 0; i < d; ++i) s += (double)v[i] * v[i];
    return sqrt(s);
}

int main(void) {
    int ndev = 0;
    CHECK(mmiss_device_count(&ndev));
    if (mmiss_abi_version() != MMISS_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    printf("libmmiss ABI %d, %d HIP device(s)\n", mmiss_abi_version(), ndev);

    /* ---- load_clip_model(): a small two-tower CLIP (head_dim 64, hidden and mlp multiples of 128) */
    mmiss_clip_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = (int32_t)sizeof cfg;
    cfg.v_hidden = 128; cfg.v_layers = 2; cfg.v_heads = 2; cfg.v_mlp = 256; cfg.v_patch = 32; cfg.v_image = 64;
    cfg.t_hidden = 128; cfg.t_layers = 2; cfg.t_heads = 2; cfg.t_mlp = 256; cfg.t_vocab = 1000; cfg.t_ctx = 16;
    cfg.proj_dim = 128; cfg.eos_token_id = 999; cfg.ln_eps = 1e-5f;
    cfg.max_batch_image = 8; cfg.max_batch_text = 8;
    const int T_IMG = (cfg.v_image / cfg.v_patch) * (cfg.v_image / cfg.v_patch) + 1, D = cfg.proj_dim;

    mmiss_encoder* enc = NULL;
    CHECK(mmiss_encoder_create(&cfg, 0, &enc));
    /* calling encode before finalize must fail loudly, not compute garbage */
    {
        float px1[3 * 64 * 64] = {0}, out1[128];
        if (mmiss_encode_image(enc, px1, 1, out1) != MMISS_ERR_STATE) { fprintf(stderr, "encode before finalize was accepted\n"); return 1; }
    }
    if (set_tensor(enc, "vision_model.embeddings.class_embedding", cfg.v_hidden, 0.09f, 0.f)) return 1;
    if (set_tensor(enc, "vision_model.embeddings.patch_embedding.weight", (long long)cfg.v_hidden * 3 * cfg.v_patch * cfg.v_patch, 0.02f, 0.f)) return 1;
    if (set_tensor(enc, "vision_model.embeddings.position_embedding.weight", (long long)T_IMG * cfg.v_hidden, 0.02f, 0.f)) return 1;
    if (set_tensor(enc, "vision_model.pre_layrnorm.weight", cfg.v_hidden, 0.1f, 1.f)) return 1; /* sic: the real key */
    if (set_tensor(enc, "vision_model.pre_layrnorm.bias", cfg.v_hidden, 0.1f, 0.f)) return 1;
    if (set_tower(enc, "vision_model", cfg.v_hidden, cfg.v_layers, cfg.v_mlp)) return 1;
    if (set_tensor(enc, "vision_model.post_layernorm.weight", cfg.v_hidden, 0.1f, 1.f)) return 1;
    if (set_tensor(enc, "vision_model.post_layernorm.bias", cfg.v_hidden, 0.1f, 0.f)) return 1;
    if (set_tensor(enc, "visual_projection.weight", (long long)D * cfg.v_hidden, 0.09f, 0.f)) return 1;
    if (set_tensor(enc, "text_model.embeddings.token_embedding.weight", (long long)cfg.t_vocab * cfg.t_hidden, 0.02f, 0.f)) return 1;
    if (set_tensor(enc, "text_model.embeddings.position_embedding.weight", (long long)cfg.t_ctx * cfg.t_hidden, 0.02f, 0.f)) return 1;
    if (set_tower(enc, "text_model", cfg.t_hidden, cfg.t_layers, cfg.t_mlp)) return 1;
    if (set_tensor(enc, "text_model.final_layer_norm.weight", cfg.t_hidden, 0.1f, 1.f)) return 1;
    if (set_tensor(enc, "text_model.final_layer_norm.bias", cfg.t_hidden, 0.1f, 0.f)) return 1;
    if (set_tensor(enc, "text_projection.weight", (long long)D * cfg.t_hidden, 0.09f, 0.f)) return 1;
    {   /* keys outside the two towers are accepted and ignored */
        float one = 4.6f; int used = 1;
        CHECK(mmiss_encoder_set_weight(enc, "logit_scale", &one, 1, &used));
        if (used) { fprintf(stderr, "logit_scale should be ignored\n"); return 1; }
    }
    CHECK(mmiss_encoder_finalize(enc));

    /* ---- generate_clip_embedding(image=...): 6 images as float pixels, then the same as raw RGB of odd sizes */
    enum { NIMG = 6 };
    const size_t img_elems = (size_t)3 * cfg.v_image * cfg.v_image;
    float* px = (float*)malloc(sizeof(float) * NIMG * img_elems);
    float* emb = (float*)malloc(sizeof(float) * NIMG * D);
    for (size_t i = 0; i < NIMG * img_elems; ++i) px[i] = 1.5f * frand();
    CHECK(mmiss_encode_image(enc, px, NIMG, emb));
    for (int i = 0; i < NIMG; ++i)
        if (fabs(row_norm(emb + (size_t)i * D, D) - 1.0) > 1e-5) { fprintf(stderr, "image embedding %d is not unit-norm\n", i); return 1; }

    const int32_t hs[2] = {90, 70}, ws[2] = {70, 131};
    int64_t offs[2];
    size_t total = 0;
    for (int i = 0; i < 2; ++i) { offs[i] = (int64_t)total; total += (size_t)hs[i] * ws[i] * 3; }
    uint8_t* rgb = (uint8_t*)malloc(total);
    for (size_t i = 0; i < total; ++i) rgb[i] = (uint8_t)((frand() + 1.f) * 127.5f);
    float raw_emb[2 * 128];
    uint8_t* crops = (uint8_t*)malloc((size_t)2 * cfg.v_image * cfg.v_image * 3);
    CHECK(mmiss_encode_image_rgb(enc, rgb, (int64_t)total, offs, hs, ws, 2, raw_emb));
    CHECK(mmiss_resize_crop_rgb(enc, rgb, (int64_t)total, offs, hs, ws, 2, crops));
    float crop_emb[2 * 128];
    CHECK(mmiss_encode_image_u8(enc, crops, 2, crop_emb));
    if (memcmp(raw_emb, crop_emb, sizeof raw_emb) != 0) { fprintf(stderr, "raw-RGB path != resize + u8 path\n"); return 1; }

    /* ---- generate_clip_embedding(text=...): BOS, tokens, EOS, padding */
    int32_t ids[2 * 16];
    for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 16; ++c) ids[r * 16 + c] = 999;
        ids[r * 16] = 998;
        for (int c = 1; c < 5 + r; ++c) ids[r * 16 + c] = 10 + 7 * c + r;
    }
    float temb[2 * 128];
    CHECK(mmiss_encode_text(enc, ids, 2, 16, temb));
    if (fabs(row_norm(temb, D) - 1.0) > 1e-5) { fprintf(stderr, "text embedding is not unit-norm\n"); return 1; }
    if (mmiss_encode_text(enc, ids, 2, 17, temb) != MMISS_ERR_ARG) { fprintf(stderr, "T > context length was accepted\n"); return 1; }

    /* ---- collection.add / collection.query on the cosine index (f16 rows), then the multimodal blend */
    mmiss_index* idx = NULL;
    CHECK(mmiss_index_create(D, MMISS_F16, 0, 0, &idx));
    enum { NROWS = 5000 };
    float* rows = (float*)malloc(sizeof(float) * NROWS * D);
    int64_t* labels = (int64_t*)malloc(sizeof(int64_t) * NROWS);
    for (size_t i = 0; i < (size_t)NROWS * D; ++i) rows[i] = frand();
    for (int i = 0; i < NROWS; ++i) labels[i] = 100 + 3 * (int64_t)i;
    memcpy(rows + (size_t)1234 * D, emb + (size_t)2 * D, sizeof(float) * D); /* plant image 2 as row 1234 */
    CHECK(mmiss_index_add(idx, rows, labels, NROWS));
    int64_t count = 0;
    CHECK(mmiss_index_count(idx, &count));
    if (count != NROWS) { fprintf(stderr, "count %lld\n", (long long)count); return 1; }
    int64_t got[NIMG * 10];
    float dist[NIMG * 10];
    int32_t cnt[NIMG];
    CHECK(mmiss_index_query(idx, emb, NIMG, 10, got, dist, cnt));
    if (got[2 * 10] != 100 + 3 * 1234 || dist[2 * 10] > 1e-3f || cnt[2] != 10) {
        fprintf(stderr, "planted row not found first: label %lld dist %g\n", (long long)got[2 * 10], dist[2 * 10]);
        return 1;
    }
    for (int q = 0; q < NIMG; ++q)
        for (int j = 1; j < 10; ++j)
            if (dist[q * 10 + j] < dist[q * 10 + j - 1]) { fprintf(stderr, "distances not ascending\n"); return 1; }
    {   /* the same query in two halves: between them every mutating call on the handle is refused */
        int64_t got2[NIMG * 10];
        float dist2[NIMG * 10];
        int32_t cnt2[NIMG];
        const int64_t one = 7;
        CHECK(mmiss_index_query_begin(idx, emb, NIMG, 10, got2, dist2, cnt2));
        if (mmiss_index_add(idx, emb, &one, 1) != MMISS_ERR_STATE) { fprintf(stderr, "add inside an open query was not refused\n"); return 1; }
        CHECK(mmiss_index_query_end(idx));
        if (memcmp(got, got2, sizeof(got)) || memcmp(dist, dist2, sizeof(dist)) || memcmp(cnt, cnt2, sizeof(cnt))) {
            fprintf(stderr, "query_begin / query_end differ from query\n");
            return 1;
        }
        if (mmiss_index_query_end(idx) != MMISS_ERR_STATE) { fprintf(stderr, "query_end without query_begin was not refused\n"); return 1; }
    }
    float blended[2 * 128];
    CHECK(mmiss_blend(0, NULL, emb, temb, 0.5, 2, D, blended));
    if (fabs(row_norm(blended, D) - 1.0) > 1e-5) { fprintf(stderr, "blend is not unit-norm\n"); return 1; }
    CHECK(mmiss_index_query(idx, blended, 2, 5, got, dist, cnt));
    int64_t removed = 0;
    const int64_t kill = 100 + 3 * 1234;
    CHECK(mmiss_index_remove(idx, &kill, 1, &removed));
    CHECK(mmiss_index_query(idx, emb + (size_t)2 * D, 1, 1, got, dist, cnt));
    if (removed != 1 || got[0] == kill) { fprintf(stderr, "removed row still returned\n"); return 1; }

    printf("ok: %d image embeddings, raw-RGB == resize+u8, text, top-10 over %d rows (planted row first, d = %.2e), blend, remove\n",
           NIMG, NROWS, (double)dist[0]);
    free(px); free(emb); free(rgb); free(crops); free(rows); free(labels);
    CHECK(mmiss_index_destroy(idx));
    CHECK(mmiss_encoder_destroy(enc));
    return 0;
}

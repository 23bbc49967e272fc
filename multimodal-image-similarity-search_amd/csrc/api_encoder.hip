// api_encoder.hip — C-ABI of the CLIP towers: weights by HF state_dict key, batched image / text encode.
// Launch sequence per layer follows HF:modeling_clip.py:353-383 (pre-LN transformer block).
#include "common.h"
#include "gemm_bf16.h"
#include "gemm_bf16_256.h"
#include "gemm_bf16_p256.h"
#include "gemm_bf16_p160.h"
#include "gemm_fp8.h"
#include "gemm_fp8_p256.h"
#include "host_stager.h"
#include "encoder_kernels.h"
#include "preprocess_kernels.h"
#include <map>
#include <set>

namespace {

// f32 [R,C] -> bf16 rows of stride ldd (dst pre-zeroed where ldd > C)
__global__ void convert_2d_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t R, int C,
                                       int ldd) {
    const int64_t total = R * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        bf16_t v = (bf16_t)src[i];
        dst[r * ldd + c] = __builtin_bit_cast(uint16_t, v);
    }
}

// The pruned last layer's two gathers in one launch: ctxc row r = ctx row rowmap[r] (bf16), xc row r = the residual row
// rowmap[r] as f32 — copied from x32, or widened from x16 when the stream is bf16. d % 4 == 0.
__global__ void gather_pooled_kernel(const uint16_t* __restrict__ ctx, uint16_t* __restrict__ ctxc, const float* __restrict__ x32,
                                     const uint16_t* __restrict__ x16, float* __restrict__ xc, const int32_t* __restrict__ rowmap,
                                     int n, int d) {
    const int chunks = d >> 2;
    const int64_t total = (int64_t)n * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / chunks), c = (int)(i - (int64_t)r * chunks);
        const size_t so = (size_t)rowmap[r] * d + (size_t)c * 4, dofs = (size_t)r * d + (size_t)c * 4;
        *reinterpret_cast<u32x2*>(ctxc + dofs) = *reinterpret_cast<const u32x2*>(ctx + so);
        f32x4 o;
        if (x16) {
            const u32x2 v = *reinterpret_cast<const u32x2*>(x16 + so);
            o[0] = __uint_as_float(v[0] << 16); o[1] = __uint_as_float(v[0] & 0xFFFF0000u);
            o[2] = __uint_as_float(v[1] << 16); o[3] = __uint_as_float(v[1] & 0xFFFF0000u);
        } else {
            o = *reinterpret_cast<const f32x4*>(x32 + so);
        }
        *reinterpret_cast<f32x4*>(xc + dofs) = o;
    }
}

struct LayerW {
    DevBuf wqkv, bqkv, wo, bo, ln1g, ln1b, ln2g, ln2b, w1, b1, w2, b2;
    DevBuf wqkv_f, cqkv, bqkv_f, w1_f, c1, b1_f;  // LayerNorm folded into the QKV / FC1 weights (finalize)
    DevBuf wqkv8, sqkv, w1_8, s1, w2_8, s2;       // fp8 path: e4m3 weights + per-output-channel f32 scales
    DevBuf wo8, so;                               // ... of the out-projection (used when the attention output is MXFP8)
    DevBuf wqkv8f, sqkvf, cqkv16, w1_8f, s1f, c1_16;  // fp8 path with the LayerNorm folded in (hidden 1024): e4m3 of the gamma-folded weights, scales, f16 row sums
};

struct Tower {
    int hidden = 0, layers = 0, heads = 0, mlp = 0, T = 0;  // T = tokens per item
    std::vector<LayerW> L;
    DevBuf pos;             // f32 [T, hidden]
    DevBuf lnf_g, lnf_b;    // post_layernorm / final_layer_norm
    DevBuf proj;            // bf16 [proj_dim, hidden]
    // workspaces (lazily allocated for max_batch)
    int ws_batch = 0;
    DevBuf x, h, qkv, ctx, u, pooled, proj_out, pool_row, out_stage, taps;
    int pool_B = -1, pool_T = -1;   // vision: pool_row currently holds b * pool_T for b < pool_B (written once per shape)
    bool embed_stats = false;       // this call's embedding stage already left xb + row statistics (layernorm_stats_kernel)
    bool embed_stats16 = false;     // ... xb + the 16-column statistics of the one-request mode (prelayernorm_skinny_kernel)
    DevBuf xc, hc, ctxc, uc;  // compact [Bp, *] buffers of the pooled rows (last-layer pruning)
    DevBuf stats;             // [Mp][hidden/64][2] partial row (sum, sumsq) for the LayerNorm-fused GEMMs
    DevBuf stats_final;       // [Mp][2] finished (mean, rstd): hidden > 768 on the persistent GEMM (ln_finalize_kernel)
    DevBuf xb;                // bf16 copy of the residual stream (A operand of the LayerNorm-folded GEMMs)
    DevBuf splitk;            // f32 partial products of the split-K GEMMs (middle batch sizes), allocated on demand
    DevBuf h8, hs, u8, us;    // fp8 path: MXFP8 LayerNorm output / FC1 output (e4m3 bytes + permuted E8M0 block scales)
    DevBuf ctx8, ctxs;        // fp8 path: MXFP8 attention output (the out-projection's A operand)
    bool fp8_ready = false;   // fp8 weights built for this tower
    bool pooled_compact = false;
    int last_B = 0, last_T = 0;
    int64_t tap_stride = 0;  // floats per recorded tap
};

// where a state_dict tensor lands
struct Slot {
    DevBuf* dst = nullptr;
    int64_t rows = 0, cols = 0;  // source shape as [rows, cols] (1-D tensors: rows = 1)
    int64_t dst_row_off = 0;     // destination row offset (q/k/v fusion)
    int64_t ldd = 0;             // destination row stride in elements
    bool bf16 = false;
};

}  // namespace

struct mmiss_encoder {
    mmiss_clip_config cfg;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t user_stream = nullptr;
    bool has_user_stream = false;
    hipEvent_t done_ev = nullptr;  // end of the last call that returned with work queued on a caller's stream
    bool async_pending = false;
    std::mutex mu;
    bool finalized = false;
    bool record_taps = false;
    // How LayerNorm1/2 reach the QKV / FC1 GEMMs:
    //   0 separate LayerNorm kernels (x f32 -> h bf16);
    //   1 normalised while the f32 rows are staged as the A operand (correct, but SLOWER at B=256: every one of the
    //     18-24 column blocks re-normalises its rows through registers, 342-371 TF);
    //   2 folded algebraically: A = bf16(x) written by the residual epilogues, gamma folded into the weights, the
    //     (mean, rstd) correction applied in the GEMM epilogue — no LayerNorm pass at all. Same precision
    //     (1-cos 5e-6 either way). With the f32 residual rows stored non-temporally (they would push the bf16 copy the
    //     next GEMM reads out of the L2) it is 2-3 % faster than 0 from ~6000 rows up (bs 128 / 192 / 256: 1.95 vs 2.02,
    //     2.46 vs 2.53, 3.19 vs 3.25 ms, tools/ln_mode_ab.py) and slower below (bs 64: 1.39 vs 1.33 ms), where the
    //     weight-streaming and split-K paths it cannot feed are worth more.
    //     Hidden size 1024 (ViT-L/14) loses 4 % with it (16 partial statistics per row in the epilogues, and FC1 gives up
    //     its 256 x 256 tile): 4.54 vs 4.74 k images/s.
    // -1 (default) = automatic: 2 from `ln_fold_min_rows` (6000) rows per call when hidden <= 768, else 0.
    int ln_mode = -1;
    // MMISS_PREC_FP8: the QKV / FC1 / FC2 GEMMs of calls with at least `fp8_min_rows` rows run on the block-scaled fp8
    // MFMA (gemm_fp8.h); out-proj, the pruned last layer, the embeddings and the head stay bf16
    int precision = MMISS_PREC_BF16;
    // Which towers the fp8 setting applies to. MMISS_PREC_FP8 switches the VISION tower only (BASELINE configs[4] is an image
    // encode; its measured 1 - cos vs the fp32 oracle is 5e-4 at full ViT-L/14 depth, inside the 1e-3 tolerance); the text
    // tower stays on the bf16 kernels because its fp8 form measures 3.3-3.9e-3, OUTSIDE the tolerance.
    // mmiss_encoder_set_tower_precision opts a tower in or out explicitly.
    bool fp8_tower[2] = {false, false};   // [0] vision, [1] text

    Tower vis, txt;
    // vision-only
    int G = 0, Kp = 0;
    DevBuf patch_w;   // bf16 [v_hidden, Kp]
    DevBuf cls;       // f32 [v_hidden]
    DevBuf pre_g, pre_b;
    DevBuf patches;   // bf16 [Mpp, Kp]
    DevBuf pix_stage; // host->device staging of pixels
    // raw-RGB input (N2): source blob staging, per-image descriptors, fixed-point taps, windows, uint8 crops
    DevBuf raw_stage, rz_desc, rz_pool, rz_bounds, crop_stage;
    // host inputs larger than one chunk are pipelined: the next chunk's bytes cross PCIe on copy_stream into the second
    // staging buffer while the current chunk is computed (ev_copied / ev_free order the two streams)
    DevBuf stage2[2];
    hipStream_t copy_stream = nullptr;
    HostStager stager;   // pageable host inputs cross PCIe through its ring of pinned blocks (host_stager.h)
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    ResizeDesc* rz_host = nullptr;   // pinned; rz_copied marks the end of its last host->device copy
    size_t rz_host_cap = 0;
    hipEvent_t rz_copied = nullptr;
    // text-only
    DevBuf tok;       // f32 [vocab, t_hidden]
    DevBuf ids_stage;

    DevBuf w_stage;   // set_weight staging
    std::map<std::string, Slot> slots;
    std::set<std::string> seen;

    hipStream_t stream() const { return has_user_stream ? user_stream : own_stream; }
};

namespace {

// hipMemset of device memory is enqueued on the NULL stream and may return before it has run; the handle's own stream is
// non-blocking, i.e. NOT ordered behind the null stream — a kernel of the next call (a weight upload, the first encode)
// could otherwise write the buffer BEFORE the late memset clears it. Seen once as a finite but wrong embedding of the
// first handle created in a busy process. So: wait for the clear before handing the buffer out (allocation time only).
int alloc_zero(DevBuf& b, size_t bytes) {
    MM_TRY(b.alloc(bytes));
    MM_HIP(hipMemsetAsync(b.p, 0, bytes, nullptr));
    MM_HIP(hipStreamSynchronize(nullptr));
    return MMISS_OK;
}

int build_tower(mmiss_encoder* e, Tower& tw, const std::string& prefix, int hidden, int layers, int heads, int mlp,
                int T, int proj_dim, const char* final_ln, const char* proj_key) {
    tw.hidden = hidden; tw.layers = layers; tw.heads = heads; tw.mlp = mlp; tw.T = T;
    tw.L.resize(layers);
    auto add = [&](const std::string& key, DevBuf* dst, int64_t rows, int64_t cols, int64_t row_off, int64_t ldd,
                   bool bf) {
        Slot s; s.dst = dst; s.rows = rows; s.cols = cols; s.dst_row_off = row_off; s.ldd = ldd; s.bf16 = bf;
        e->slots[key] = s;
    };
    const int d = hidden;
    MM_TRY(alloc_zero(tw.pos, (size_t)T * d * 4));
    add(prefix + ".embeddings.position_embedding.weight", &tw.pos, T, d, 0, d, false);
    MM_TRY(alloc_zero(tw.lnf_g, (size_t)d * 4));
    MM_TRY(alloc_zero(tw.lnf_b, (size_t)d * 4));
    add(prefix + "." + final_ln + ".weight", &tw.lnf_g, 1, d, 0, d, false);
    add(prefix + "." + final_ln + ".bias", &tw.lnf_b, 1, d, 0, d, false);
    MM_TRY(alloc_zero(tw.proj, (size_t)proj_dim * d * 2));
    add(proj_key, &tw.proj, proj_dim, d, 0, d, true);
    for (int i = 0; i < layers; ++i) {
        LayerW& L = tw.L[i];
        const std::string lp = prefix + ".encoder.layers." + std::to_string(i) + ".";
        MM_TRY(alloc_zero(L.wqkv, (size_t)3 * d * d * 2));
        MM_TRY(alloc_zero(L.bqkv, (size_t)3 * d * 4));
        const char* qkv[3] = {"q_proj", "k_proj", "v_proj"};
        for (int j = 0; j < 3; ++j) {
            add(lp + "self_attn." + qkv[j] + ".weight", &L.wqkv, d, d, (int64_t)j * d, d, true);
            add(lp + "self_attn." + qkv[j] + ".bias", &L.bqkv, 1, d, 0, 0, false);
            e->slots[lp + "self_attn." + qkv[j] + ".bias"].dst_row_off = (int64_t)j * d;  // element offset (1-D)
        }
        MM_TRY(alloc_zero(L.wo, (size_t)d * d * 2));
        MM_TRY(alloc_zero(L.bo, (size_t)d * 4));
        add(lp + "self_attn.out_proj.weight", &L.wo, d, d, 0, d, true);
        add(lp + "self_attn.out_proj.bias", &L.bo, 1, d, 0, d, false);
        MM_TRY(alloc_zero(L.ln1g, (size_t)d * 4)); MM_TRY(alloc_zero(L.ln1b, (size_t)d * 4));
        MM_TRY(alloc_zero(L.ln2g, (size_t)d * 4)); MM_TRY(alloc_zero(L.ln2b, (size_t)d * 4));
        add(lp + "layer_norm1.weight", &L.ln1g, 1, d, 0, d, false);
        add(lp + "layer_norm1.bias", &L.ln1b, 1, d, 0, d, false);
        add(lp + "layer_norm2.weight", &L.ln2g, 1, d, 0, d, false);
        add(lp + "layer_norm2.bias", &L.ln2b, 1, d, 0, d, false);
        MM_TRY(alloc_zero(L.w1, (size_t)mlp * d * 2)); MM_TRY(alloc_zero(L.b1, (size_t)mlp * 4));
        MM_TRY(alloc_zero(L.w2, (size_t)d * mlp * 2)); MM_TRY(alloc_zero(L.b2, (size_t)d * 4));
        add(lp + "mlp.fc1.weight", &L.w1, mlp, d, 0, d, true);
        add(lp + "mlp.fc1.bias", &L.b1, 1, mlp, 0, mlp, false);
        add(lp + "mlp.fc2.weight", &L.w2, d, mlp, 0, mlp, true);
        add(lp + "mlp.fc2.bias", &L.b2, 1, d, 0, d, false);
    }
    return MMISS_OK;
}

int ensure_tower_ws(mmiss_encoder* e, Tower& tw, int max_batch, int proj_dim) {
    if (tw.ws_batch >= max_batch) return MMISS_OK;
    const int64_t Mp = round_up((int64_t)max_batch * tw.T, 128) + 192;  // room for any tile height (128/160/192)
    const int64_t Bp = round_up(max_batch, 128);
    const int d = tw.hidden;
    MM_TRY(alloc_zero(tw.x, (size_t)Mp * d * 4));
    MM_TRY(alloc_zero(tw.h, (size_t)Mp * d * 2));
    MM_TRY(alloc_zero(tw.qkv, (size_t)Mp * 3 * d * 2));
    MM_TRY(alloc_zero(tw.ctx, (size_t)Mp * d * 2));
    MM_TRY(alloc_zero(tw.u, (size_t)Mp * tw.mlp * 2));
    MM_TRY(alloc_zero(tw.pooled, (size_t)Bp * d * 2));
    MM_TRY(alloc_zero(tw.proj_out, (size_t)Bp * proj_dim * 4));
    MM_TRY(alloc_zero(tw.pool_row, (size_t)max_batch * 4));
    tw.pool_B = tw.pool_T = -1;
    MM_TRY(alloc_zero(tw.out_stage, (size_t)max_batch * proj_dim * 4));
    MM_TRY(alloc_zero(tw.xc, (size_t)Bp * d * 4));
    MM_TRY(alloc_zero(tw.hc, (size_t)Bp * d * 2));
    MM_TRY(alloc_zero(tw.ctxc, (size_t)Bp * d * 2));
    MM_TRY(alloc_zero(tw.uc, (size_t)Bp * tw.mlp * 2));
    MM_TRY(alloc_zero(tw.stats, (size_t)Mp * (d / 16) * 2 * 4));  // (one partial per 64 columns; per 16 in the skinny folded mode)
    MM_TRY(alloc_zero(tw.xb, (size_t)Mp * d * 2));
    if (d > 768) MM_TRY(alloc_zero(tw.stats_final, (size_t)Mp * 2 * 4));
    if (e->fp8_tower[&tw == &e->txt ? 1 : 0]) {
        MM_TRY(alloc_zero(tw.h8, (size_t)Mp * d));
        MM_TRY(alloc_zero(tw.hs, (size_t)Mp * mx_scale_row_bytes(d)));
        MM_TRY(alloc_zero(tw.u8, (size_t)Mp * tw.mlp));
        MM_TRY(alloc_zero(tw.us, (size_t)Mp * mx_scale_row_bytes(tw.mlp)));
        MM_TRY(alloc_zero(tw.ctx8, (size_t)Mp * d));
        MM_TRY(alloc_zero(tw.ctxs, (size_t)Mp * mx_scale_row_bytes(d)));
    }
    if (e->record_taps) {
        tw.tap_stride = Mp * d;
        MM_TRY(alloc_zero(tw.taps, (size_t)(tw.layers + 1) * tw.tap_stride * 4));
    }
    tw.ws_batch = max_batch;
    return MMISS_OK;
}

// One request at a time: all four GEMMs of a layer on the weight-streaming skinny kernel with the LayerNorm folded in (run_layers)
static bool skinny_fold_ok(const Tower& tw, int M) {
    const int d = tw.hidden;
    if (M > 128 || (d % 128) != 0 || mmiss_option("skinny_fold", 1) == 0) return false;
    GemmEpi probe{};
    probe.stats16 = 1;
    return gemm_skinny_ok(MMISS_EPI_LNFOLD_BF16, M, 3 * d, d, probe) && gemm_skinny_ok(MMISS_EPI_LNFOLD_QGELU_BF16, M, tw.mlp, d, probe) &&
           gemm_skinny_ok(MMISS_EPI_BIAS_RESID_F32, M, d, d, probe) && gemm_skinny_ok(MMISS_EPI_BIAS_RESID_F32, M, d, tw.mlp, probe);
}

// the transformer stack shared by both towers; x holds the embeddings on entry
// LayerNorm placement for a call of M token rows: 0 separate kernels, 1 staged (experiments), 2 folded into the GEMMs
static int pick_ln_mode(const mmiss_encoder* e, const Tower& tw, int M, bool& fp8) {
    int mode = e->ln_mode;
    if (mode < 0) {
        const int forced = mmiss_option("ln_mode", -1);
        // (hidden 1024 — ViT-L/14 — folds since round 4: finished row statistics in front of the persistent GEMM; option
        // ln_fold_1024 = 0 gives the separate LayerNorm kernels back)
        const int max_hidden = mmiss_option("ln_fold_1024", 1) != 0 ? 1024 : 768;
        mode = forced >= 0 ? forced : ((M >= mmiss_option("ln_fold_min_rows", 6000) && tw.hidden <= max_hidden) ? 2 : 0);
    }
    fp8 = e->fp8_tower[&tw == &e->txt ? 1 : 0] && tw.fp8_ready && tw.h8.p && M >= mmiss_option("fp8_min_rows", 1024);
    if (fp8) mode = 0;  // the fp8 GEMMs take their A operand from the MXFP8 LayerNorm kernel
    return mode;
}

int run_layers(mmiss_encoder* e, Tower& tw, int B, bool causal, hipStream_t st) {
    const int d = tw.hidden, M = B * tw.T;
    const float eps = e->cfg.ln_eps;
    // per-GEMM tile height (fills the 256 CUs x 2 blocks evenly) and the row count padded to it
    // the LayerNorm-fused / folded GEMMs (ln_mode 1, 2) exist for the 128-column tiles only
    bool fp8 = false;
    const int mode = pick_ln_mode(e, tw, M, fp8);
    const bool plain = mode == 0;
    int bm_qkv = plain ? gemm_pick_variant(M, 3 * d) : gemm_pick_bm(M, 3 * d);
    int bm_d = plain ? gemm_pick_variant(M, d) : gemm_pick_bm(M, d);
    int bm_mlp = plain ? gemm_pick_variant(M, tw.mlp) : gemm_pick_bm(M, tw.mlp);
    // experiment knobs (tools/option_ab.py): force a tile variant per GEMM family
    if (const int f = mmiss_option("gemm_bm_qkv", 0)) bm_qkv = f;
    if (const int f = mmiss_option("gemm_bm_d", 0)) bm_d = f;
    if (const int f = mmiss_option("gemm_bm_mlp", 0)) bm_mlp = f;
    auto padded = [&](int bm) { return (int)round_up(M, bm % 1000); };
    const int bm8_qkv = gemm_pick_bm(M, 3 * d), bm8_d = gemm_pick_bm(M, d), bm8_mlp = gemm_pick_bm(M, tw.mlp);
    // fp8 GEMMs on the persistent 256 x 256 kernel (gemm_fp8_p256.h, round 5) once there is a tile per CU; K % 256 == 0 (round 6:
    // every GEMM of ViT-L/14 AND of ViT-B/32, whose K = 768 is three K-tile pairs). Option gemm_p256_fp8 = 0 turns it off,
    // n > 1 = minimum tile count. The N = 768 GEMMs of ViT-B/32 at 12 800 rows (FC2, out-projection) are 150 tiles of 256 x 256
    // — one round on 150 of the 256 CUs — and measure SLOWER there than on the BM x 128 tile kernel, whose 600+ smaller tiles
    // fill the chip (tools/b32_kernel_table.py, profiles/b32_fp8_r06.txt: 34.1 vs 30.7 us per launch on average, the encode
    // 113.4 vs 117.7 k images/s): they stay on gemm8_kernel. Option gemm_p256_fp8_narrow = the minimum tile count from which
    // such one-round grids take the persistent kernel anyway (0 = never, the default).
    const int p8_min = mmiss_option("gemm_p256_fp8", 1);
    const int p8_narrow = mmiss_option("gemm_p256_fp8_narrow", 0);
    auto p8 = [&](int epi, int N, int K) {
        if (p8_min == 0 || !gemm256p8_ok(epi, (int)round_up(M, 256), N, K)) return false;
        const int64_t tiles = (int64_t)(round_up(M, 256) / 256) * (N / 256);
        if (p8_min == 1 && p8_narrow > 0 && tiles < 256 && tiles >= p8_narrow && M >= 6000) return true;   // (measured at 12 800 rows only)
        return tiles >= (p8_min > 1 ? p8_min : 256);
    };
    auto gemm8_any = [&](int epi, int bm, Gemm8Args g8, int xt = 0) -> int {   // g8.M unset: padded here to the kernel's tile height
        if (xt != 0 || p8(epi, g8.N, g8.K)) { g8.M = (int)round_up(M, 256); return launch_gemm256p8(st, epi, g8, xt); }
        g8.M = (int)round_up(M, bm % 1000);
        return launch_gemm8(st, epi, bm, g8);
    };
    // 256 x 256 phase-pipelined tile for the widest GEMMs. Round 1 used it from N = 4096 (the ViT-L/14 FC1: 336 -> 320 us);
    // with the banded tile order the 128-column kernel now does that GEMM in 292 us (312 us on the 256 x 256 tile, whose
    // 2064 tiles are 8.06 rounds of 256 CUs), so it is off by default. Option gemm_256 = minimum N, 0 = never.
    const int min_n256 = mmiss_option("gemm_256", 0);
    auto use256 = [&](int N) { return plain && min_n256 > 0 && N >= min_n256 && (N % 256) == 0 && (d % 64) == 0 && M > 512; };
    // the same tile with the folded-LayerNorm epilogue, for the QKV GEMM: round 2, inside the bs-256 encode (tools/option_ab2.py)
    // 55.4 -> 52.3 us per QKV launch (2.989 -> 2.960 ms per encode), text tower 42.4 -> 40.7 us. Taken when the tile count
    // fills its last round of 256 CUs to >= 85 % (ViT-B/32 at 12 800 rows: 450 tiles = 88 %; its FC1 would be 600 = 78 % and
    // measures 78 -> 84 us, so that one stays on the 128-column tiles: option gemm_256_fold_mlp). Option gemm_256_fold:
    // -1 = this rule, 0 = never, n > 0 = from N = n whatever the fill.
    const int min_n256f = mmiss_option("gemm_256_fold", -1);
    auto fold256 = [&](int N) {
        if (mode != 2 || min_n256f == 0 || (N % 256) != 0 || (d % 64) != 0 || M <= 512) return false;
        if (min_n256f > 0) return N >= min_n256f;
        const int64_t tiles = (int64_t)((M + 255) / 256) * (N / 256), rounds = (tiles + 255) / 256;
        return tiles * 100 >= rounds * 256 * 85;
    };  // (workspace rows are padded to round_up(M, 128) + 192 >= round_up(M, 256))
    // Persistent form of that tile (gemm_bf16_p256.h, round 3): one workgroup per CU walks its tiles as one K stream with the
    // epilogue in registers. Taken for the wide GEMMs (QKV, FC1) of calls with at least 256 tiles (one per CU); option
    // gemm_p256 = 0 turns it off, n > 1 = minimum tile count.
    const int p256_min = mmiss_option("gemm_p256", 1);
    const bool fold_final = d > 768;   // the folded persistent GEMM takes FINISHED (mean, rstd) per row (tw.stats_final)
    auto p256 = [&](int epi, int N) {
        const bool is_fold = epi == MMISS_EPI_LNFOLD_BF16 || epi == MMISS_EPI_LNFOLD_QGELU_BF16;
        if (p256_min == 0 || !gemm256p_ok(epi, padded(256), N, d, is_fold && fold_final && tw.stats_final.p)) return false;
        return (int64_t)(padded(256) / 256) * (N / 256) >= (p256_min > 1 ? p256_min : 256);
    };
    auto finalize_stats = [&]() -> int {   // partial (sum, sumsq) per 64 columns -> (mean, rstd) per row
        MM_PROF("ln_finalize", st, 4.0 * M * (d / 64), (double)M * ((d / 64) * 8 + 8));
        hipLaunchKernelGGL(ln_finalize_kernel, dim3((M * 8 + 255) / 256), dim3(256), 0, st, tw.stats.as<float>(),
                           tw.stats_final.as<float>(), M, d / 64, d, eps);
        MM_HIP(hipGetLastError());
        return MMISS_OK;
    };
    // The residual GEMMs on the bf16 stream (out-projection, FC2) on the 160 x 256 tile of the two-phase staggered loop
    // (gemm_bf16_p160.h, round 3) when that grid is about one round of the chip, or K is long: isolated, ViT-B/32 bs 256
    // FC2 67.0 -> 61.1 us, out-projection 27.9 -> 25.2, text FC2 47.4 -> 43.0, ViT-L/14 bs 128 FC2 300 -> 274 (its out-projection,
    // 828 tiles of K = 1024, is faster on the 128-column tile: 93 vs 98 us). Option gemm_p160 = 0 turns it off.
    const int p160_opt = mmiss_option("gemm_p160", 1);
    auto p160 = [&](int N, int K) {
        if (p160_opt == 0 || !gemm160p_ok(padded(160), N, K)) return false;
        const int64_t tiles = (int64_t)(padded(160) / 160) * (N / 256);
        return tiles >= 200 && (tiles <= 256 || K >= 2048);
    };
    // split-K scratch for the narrow long-K GEMM (FC2) while its grid is far below the CU count
    if (gemm_splitk_candidate((int64_t)((M + 127) / 128) * (d / GEMM_BN), tw.mlp))
        MM_TRY(tw.splitk.ensure((size_t)8 * (round_up(M, 128) + 192) * d * 4));  // any tile height's row padding
    auto tap = [&](int which) -> int {
        if (e->record_taps && tw.taps.p)
            MM_HIP(hipMemcpyAsync(tw.taps.as<float>() + (size_t)which * tw.tap_stride, tw.x.p, (size_t)M * d * 4,
                                  hipMemcpyDeviceToDevice, st));
        return MMISS_OK;
    };
    MM_TRY(tap(0));
    // Only the pooled row of every item (token 0 / first EOS) leaves the last layer (HF:modeling_clip.py:650-651,
    // 561-581), and rows do not mix after the attention: out-proj, LN2 and the MLP of the LAST layer run on those B
    // rows only (compacted). Kept off while taps are recorded so the tests can compare every row of every layer.
    const bool prune = !e->record_taps;
    tw.pooled_compact = false;
    const bool fuse = mode == 1, fold = mode == 2;
    // bf16 residual stream for the large calls (needs the pruned last layer, i.e. no taps): the residual GEMMs
    // read-modify-write the bf16 rows; in the folded mode these rows are the A operand of the next GEMM as well, in the
    // separate-LayerNorm mode (hidden > 768: ViT-L/14) the LayerNorm kernel reads them. Default under MMISS_PREC_BF16 from
    // `ln_fold_min_rows` rows; MMISS_PREC_BF16_F32RESID or option resid16 = 0 keep the f32 stream (small calls always do).
    // ViT-B/32, 256 images: residual GEMMs 33.8 -> 28.5 us (K = 768), 71.4 -> 66.6 us (K = 3072), encode 3.05 -> 2.91 ms;
    // 1 - cos vs the fp32 oracle 5e-6 -> 5e-5 (tolerance 1e-3).
    const int r16opt = mmiss_option("resid16", -1);
    const bool r16mode = fold || (plain && M >= mmiss_option("ln_fold_min_rows", 6000) && bm_d < 1000);  // (fp8 GEMMs included)
    const bool resid16 = r16mode && prune && tw.layers >= 1 &&
                         (r16opt >= 0 ? r16opt != 0 : e->precision != MMISS_PREC_BF16_F32RESID);
    // One request at a time (all four GEMMs of a layer on the weight-streaming skinny kernel, M <= 128): LayerNorm folded
    // into the QKV / FC1 weights there as well — the skinny residual GEMMs leave a bf16 copy of the new rows and their
    // partial statistics per 16 columns, the skinny QKV / FC1 GEMMs apply (mean, rstd) in their epilogue. 24 of the ~99
    // launches of a ViT-B/32 request disappear. Option skinny_fold = 0 keeps the LayerNorm kernels.
    const bool sfold = plain && !fp8 && skinny_fold_ok(tw, M);
    // (round 6, VERDICT r5 next #8, tried and removed: every skinny launch touching the weight lines of the NEXT skinny launch — one
    // dword per 128-byte line, workgroup L taking blocks L, L + grid, ... so that they land in the L2 of the XCD that reads them
    // next — made the request 3 % SLOWER, 0.4603 against 0.4466 ms per encode, same bits: profiles/single_request_r06.txt)
    const int parts = d / 64;
    const bool have_embed_stats16 = tw.embed_stats16;   // (the one-request embedding stage already wrote xb + the 16-column statistics)
    tw.embed_stats16 = false;
    if (sfold && !have_embed_stats16) {
        MM_PROF("row_stats", st, 3.0 * M * d, 6.0 * M * d);
        hipLaunchKernelGGL(skinny_row_stats16_kernel, dim3((M + 3) / 4), dim3(256), 0, st, tw.x.as<float>(), tw.stats.as<float>(),
                           tw.xb.as<uint16_t>(), M, d);
        MM_HIP(hipGetLastError());
    }
    const bool have_embed_stats = tw.embed_stats;  // (the vision embedding stage's pre-LN already wrote xb + statistics)
    tw.embed_stats = false;
    if ((fuse || fold || resid16) && !have_embed_stats) {  // (separate-LayerNorm mode with a bf16 stream: only the bf16 copy is used)
        MM_PROF("row_stats", st, 3.0 * M * d, (fold ? 6.0 : 4.0) * M * d);
        hipLaunchKernelGGL(row_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, st, tw.x.as<float>(), tw.stats.as<float>(),
                           (fold || resid16) ? tw.xb.as<uint16_t>() : nullptr, M, d, parts);
        MM_HIP(hipGetLastError());
    }
    // fp8 tower with the LayerNorm FOLDED into the QKV / FC1 GEMMs (round 5; hidden 1024 on the persistent fp8 kernel): the residual
    // GEMMs' epilogue leaves the new bf16 rows as MXFP8 — raw — with their 256-column statistics, the QKV / FC1 epilogue applies
    // (mean, rstd): 46 of the 47 LayerNorm launches of a ViT-L/14 encode disappear (tools/fp8_fold_sim.py: the same 1 - cos as
    // LayerNorm-then-quantise). OFF by default (option fp8_ln_fold = 1 turns it on): measured inside the bs-128 encode the 1.03 ms
    // of LayerNorm kernels (HBM-bound, 4.5 TB/s) come back as 1.3 ms of epilogue time in the GEMMs — +12 us per residual GEMM
    // (half again as many bytes stored in the epilogue burst of 256 workgroups in step), +12 / +20 us per QKV / FC1 (two more packed
    // operations per value and a workgroup barrier per tile), with the matrix pipes idle meanwhile: 9.19 k against 9.40 k images/s
    // (tools/l14_fp8_ab.py, tools/l14_kernel_table.py).
    bool fold8 = false;
    if (fp8 && resid16 && !causal && d == 1024 && tw.layers >= 1 && tw.L[0].wqkv8f.p && tw.ctx8.p && attention_mx_ok(tw.T, tw.heads) &&
        mmiss_option("fp8_outproj", 1) != 0 && mmiss_option("fp8_ln_fold", 0) != 0 && p8(MMISS_EPI8_BIAS_BF16, 3 * d, d) &&
        p8(MMISS_EPI8_QGELU_MXFP8, tw.mlp, d) && p8(MMISS_EPI8_BIAS_RESID_BF16, d, d) && p8(MMISS_EPI8_BIAS_RESID_BF16, d, tw.mlp)) {
        Gemm8Args t{};   // (the residual form must not end in a round of half tiles)
        t.M = (int)round_up(M, 256); t.N = d; t.K = d; t.m_valid = M;
        t.ragged = (t.M >= 512 && M > t.M - 256 && M <= t.M - 128 && mmiss_option("gemm_p256_ragged", 1) != 0) ? M - (t.M - 256) : 0;
        t.q_out = tw.h8.as<uint8_t>(); t.q_scale = tw.hs.as<uint8_t>(); t.stats_out = tw.stats.as<float>(); t.ld_qs = mx_scale_row_bytes(d);
        fold8 = gemm256p8_xt_ok(MMISS_EPI8_BIAS_RESID_BF16, 2, t);
    }
    if (fold8) {
        MM_PROF("quant16_mxfp8_stats", st, 3.0 * M * d, 3.0 * M * d);
        hipLaunchKernelGGL(quant16_mxfp8_stats_1024_kernel, dim3((M + 3) / 4), dim3(256), 0, st, tw.xb.as<uint16_t>(), tw.h8.as<uint8_t>(),
                           tw.hs.as<uint8_t>(), tw.stats.as<float>(), M, mx_scale_row_bytes(d));
        MM_HIP(hipGetLastError());
    }
    auto fold8_args = [&](Gemm8Args& g) {   // the consumer side: A = the raw rows as MXFP8, statistics, the bf16 rows for a ragged block
        g.A = tw.h8.as<uint8_t>(); g.As = tw.hs.as<uint8_t>(); g.ld_as = mx_scale_row_bytes(d);
        g.ln_stats = tw.stats.as<float>(); g.x16 = tw.xb.as<uint16_t>(); g.ln_eps = eps;
    };
    auto mxq_args = [&](Gemm8Args& g) {     // the producer side
        g.q_out = tw.h8.as<uint8_t>(); g.q_scale = tw.hs.as<uint8_t>(); g.ld_qs = mx_scale_row_bytes(d); g.stats_out = tw.stats.as<float>();
    };
    for (int l = 0; l < tw.layers; ++l) {
        LayerW& L = tw.L[l];
        GemmEpi ep{};
        ep.out = tw.qkv.p; ep.bias = L.bqkv.as<float>(); ep.ldo = 3 * d; ep.m_valid = M;
        if (fold8) {
            Gemm8Args g{};
            fold8_args(g);
            g.W = L.wqkv8f.as<uint8_t>(); g.wscale = L.sqkvf.as<float>(); g.bias = L.bqkv_f.as<float>(); g.c16 = L.cqkv16.as<uint16_t>();
            g.out = tw.qkv.p; g.ldo = 3 * d; g.N = 3 * d; g.K = d; g.m_valid = M;
            MM_TRY(gemm8_any(MMISS_EPI8_BIAS_BF16, bm8_qkv, g, 1));
        } else if (fold) {
            ep.bias = L.bqkv_f.as<float>(); ep.aux = L.cqkv.as<float>();
            ep.ln_stats = tw.stats.as<float>(); ep.ln_parts = parts; ep.ln_eps = eps;
            if (p256(MMISS_EPI_LNFOLD_BF16, 3 * d)) {
                if (fold_final) { MM_TRY(finalize_stats()); ep.ln_final = tw.stats_final.as<float>(); }
                MM_TRY(launch_gemm256p(st, MMISS_EPI_LNFOLD_BF16, tw.xb.p, L.wqkv_f.p, ep, padded(256), 3 * d, d));
            } else if (fold256(3 * d)) {
                MM_TRY(launch_gemm256(st, MMISS_EPI_LNFOLD_BF16, tw.xb.p, L.wqkv_f.p, ep, padded(256), 3 * d, d));
            } else {
                MM_TRY(launch_gemm_fold(st, MMISS_EPI_LNFOLD_BF16, bm_qkv, tw.xb.p, L.wqkv_f.p, ep, padded(bm_qkv), 3 * d, d));
            }
        } else if (fp8) {
            MM_TRY(launch_layernorm_mxfp8(st, resid16 ? tw.xb.p : tw.x.p, resid16, L.ln1g.as<float>(), L.ln1b.as<float>(), tw.h8.as<uint8_t>(),
                                          tw.hs.as<uint8_t>(), M, d, eps));
            Gemm8Args g{};
            g.A = tw.h8.as<uint8_t>(); g.As = tw.hs.as<uint8_t>(); g.ld_as = mx_scale_row_bytes(d);
            g.W = L.wqkv8.as<uint8_t>(); g.wscale = L.sqkv.as<float>(); g.bias = L.bqkv.as<float>();
            g.out = tw.qkv.p; g.ldo = 3 * d; g.N = 3 * d; g.K = d; g.m_valid = M;
            MM_TRY(gemm8_any(MMISS_EPI8_BIAS_BF16, bm8_qkv, g));
        } else if (sfold) {
            ep.bias = L.bqkv_f.as<float>(); ep.aux = L.cqkv.as<float>();
            ep.ln_stats = tw.stats.as<float>(); ep.ln_parts = d / 16; ep.ln_eps = eps; ep.stats16 = 1;
            MM_TRY(launch_gemm_skinny_fold(st, MMISS_EPI_LNFOLD_BF16, tw.xb.p, L.wqkv_f.p, ep, M, 3 * d, d));
        } else {
            if (resid16) MM_TRY(launch_layernorm16(st, tw.xb.as<uint16_t>(), L.ln1g.as<float>(), L.ln1b.as<float>(), tw.h.p, M, d, eps));
            else
            MM_TRY(launch_layernorm(st, tw.x.as<float>(), L.ln1g.as<float>(), L.ln1b.as<float>(), tw.h.p, true, nullptr, M,
                                    d, eps));
            if (p256(MMISS_EPI_BIAS_BF16, 3 * d)) MM_TRY(launch_gemm256p(st, MMISS_EPI_BIAS_BF16, tw.h.p, L.wqkv.p, ep, padded(256), 3 * d, d));
            else if (use256(3 * d)) MM_TRY(launch_gemm256(st, MMISS_EPI_BIAS_BF16, tw.h.p, L.wqkv.p, ep, padded(256), 3 * d, d));
            else MM_TRY(launch_gemm(st, MMISS_EPI_BIAS_BF16, bm_qkv, tw.h.p, L.wqkv.p, ep, padded(bm_qkv), 3 * d, d));
        }
        // fp8 tower: the attention output leaves the kernel as MXFP8 and the out-projection runs on the fp8 GEMM too (round 4:
        // the long-sequence kernels, ViT-L/14; round 6: the one-pass kernels as well, ViT-B/32's 50 keys; option fp8_outproj = 0:
        // bf16 ctx + bf16 out-projection as in round 3). Not in the pruned last layer (its out-projection is a 128-row bf16
        // GEMM on gathered rows).
        const bool last_pruned = prune && l == tw.layers - 1;
        const bool out8 = fp8 && !causal && !last_pruned && resid16 && L.wo8.p && tw.ctx8.p && attention_mx_ok(tw.T, tw.heads) &&
                          mmiss_option("fp8_outproj", 1) != 0;
        if (out8) MM_TRY(launch_attention_mx(st, tw.qkv.p, tw.ctx8.as<uint8_t>(), tw.ctxs.as<uint8_t>(), mx_scale_row_bytes(d), B, tw.T, tw.heads));
        else
        MM_TRY(launch_attention(st, tw.qkv.p, tw.ctx.p, B, tw.T, tw.heads, causal));
        if (prune && l == tw.layers - 1) {
            const int Bp = (int)round_up(B, 128);
            const int grid = (B * d / 4 + 255) / 256;
            // pooled rows of the attention output (bf16) and of the residual stream (f32, or widened bf16) in ONE launch
            hipLaunchKernelGGL(gather_pooled_kernel, dim3(grid), dim3(256), 0, st, tw.ctx.as<uint16_t>(), tw.ctxc.as<uint16_t>(),
                               resid16 ? nullptr : tw.x.as<float>(), resid16 ? tw.xb.as<uint16_t>() : nullptr,
                               tw.xc.as<float>(), tw.pool_row.as<int32_t>(), B, d);
            MM_HIP(hipGetLastError());
            ep = GemmEpi{};
            ep.out = tw.xc.p; ep.bias = L.bo.as<float>(); ep.ldo = d; ep.m_valid = B;
            MM_TRY(launch_gemm(st, MMISS_EPI_BIAS_RESID_F32, 128, tw.ctxc.p, L.wo.p, ep, Bp, d, d));
            MM_TRY(launch_layernorm(st, tw.xc.as<float>(), L.ln2g.as<float>(), L.ln2b.as<float>(), tw.hc.p, true, nullptr,
                                    B, d, eps));
            ep = GemmEpi{};
            ep.out = tw.uc.p; ep.bias = L.b1.as<float>(); ep.ldo = tw.mlp; ep.m_valid = B;
            MM_TRY(launch_gemm(st, MMISS_EPI_BIAS_QGELU_BF16, 128, tw.hc.p, L.w1.p, ep, Bp, tw.mlp, d));
            ep = GemmEpi{};
            ep.out = tw.xc.p; ep.bias = L.b2.as<float>(); ep.ldo = d; ep.m_valid = B;
            MM_TRY(launch_gemm(st, MMISS_EPI_BIAS_RESID_F32, 128, tw.uc.p, L.w2.p, ep, Bp, d, tw.mlp));
            tw.pooled_compact = true;
            break;
        }
        ep = GemmEpi{};
        ep.out = tw.x.p; ep.bias = L.bo.as<float>(); ep.ldo = d; ep.m_valid = M;
        ep.stats_out = (fuse || fold || sfold) ? tw.stats.as<float>() : nullptr;  // row statistics of the new residual for LN2
        ep.xb_out = (fold || sfold) ? tw.xb.p : nullptr;
        ep.stats16 = sfold ? 1 : 0;
        if (out8) {
            Gemm8Args g{};
            g.A = tw.ctx8.as<uint8_t>(); g.As = tw.ctxs.as<uint8_t>(); g.ld_as = mx_scale_row_bytes(d);
            g.W = L.wo8.as<uint8_t>(); g.wscale = L.so.as<float>(); g.bias = L.bo.as<float>();
            g.out = tw.xb.p; g.ldo = d; g.N = d; g.K = d; g.m_valid = M;
            if (fold8) mxq_args(g);
            MM_TRY(gemm8_any(MMISS_EPI8_BIAS_RESID_BF16, bm8_d, g, fold8 ? 2 : 0));
        } else if (resid16) {  // the bf16 rows ARE the residual stream: read-modify-write in place, no f32 stream
            ep.out = tw.xb.p; ep.xb_out = nullptr;
            if (p160(d, d)) MM_TRY(launch_gemm160p(st, tw.ctx.p, L.wo.p, ep, padded(160), d, d));
            else MM_TRY(launch_gemm_resid16(st, bm_d, tw.ctx.p, L.wo.p, ep, padded(bm_d), d, d));
        } else {
            MM_TRY(launch_gemm(st, MMISS_EPI_BIAS_RESID_F32, bm_d, tw.ctx.p, L.wo.p, ep, padded(bm_d), d, d));
        }
        ep = GemmEpi{};
        ep.out = tw.u.p; ep.bias = L.b1.as<float>(); ep.ldo = tw.mlp; ep.m_valid = M;
        if (fold) {
            ep.bias = L.b1_f.as<float>(); ep.aux = L.c1.as<float>();
            ep.ln_stats = tw.stats.as<float>(); ep.ln_parts = parts; ep.ln_eps = eps;
            if (p256(MMISS_EPI_LNFOLD_QGELU_BF16, tw.mlp)) {
                if (fold_final) { MM_TRY(finalize_stats()); ep.ln_final = tw.stats_final.as<float>(); }
                MM_TRY(launch_gemm256p(st, MMISS_EPI_LNFOLD_QGELU_BF16, tw.xb.p, L.w1_f.p, ep, padded(256), tw.mlp, d));
            } else if (fold256(tw.mlp) && mmiss_option("gemm_256_fold_mlp", 0)) {  // (A/B knob: 78 -> 84 us at 12 800 rows, off)
                MM_TRY(launch_gemm256(st, MMISS_EPI_LNFOLD_QGELU_BF16, tw.xb.p, L.w1_f.p, ep, padded(256), tw.mlp, d));
            } else {
                MM_TRY(launch_gemm_fold(st, MMISS_EPI_LNFOLD_QGELU_BF16, bm_mlp, tw.xb.p, L.w1_f.p, ep, padded(bm_mlp), tw.mlp, d));
            }
        } else if (fp8) {
            // LN2 -> MXFP8 (fold8: folded into FC1), FC1 + QuickGELU -> MXFP8 (per row and 64 columns), FC2 + residual: `u` crosses HBM as 1 byte
            Gemm8Args g{};
            if (fold8) {
                fold8_args(g);
                g.W = L.w1_8f.as<uint8_t>(); g.wscale = L.s1f.as<float>(); g.bias = L.b1_f.as<float>(); g.c16 = L.c1_16.as<uint16_t>();
            } else {
                MM_TRY(launch_layernorm_mxfp8(st, resid16 ? tw.xb.p : tw.x.p, resid16, L.ln2g.as<float>(), L.ln2b.as<float>(), tw.h8.as<uint8_t>(),
                                              tw.hs.as<uint8_t>(), M, d, eps));
                g.A = tw.h8.as<uint8_t>(); g.As = tw.hs.as<uint8_t>(); g.ld_as = mx_scale_row_bytes(d);
                g.W = L.w1_8.as<uint8_t>(); g.wscale = L.s1.as<float>(); g.bias = L.b1.as<float>();
            }
            g.out = tw.u8.p; g.out_scale = tw.us.as<uint8_t>(); g.ld_os = mx_scale_row_bytes(tw.mlp);
            g.ldo = tw.mlp; g.N = tw.mlp; g.K = d; g.m_valid = M;
            MM_TRY(gemm8_any(MMISS_EPI8_QGELU_MXFP8, bm8_mlp, g, fold8 ? 1 : 0));
            g = Gemm8Args{};
            g.A = tw.u8.as<uint8_t>(); g.As = tw.us.as<uint8_t>(); g.ld_as = mx_scale_row_bytes(tw.mlp);
            g.W = L.w2_8.as<uint8_t>(); g.wscale = L.s2.as<float>(); g.bias = L.b2.as<float>();
            g.out = resid16 ? tw.xb.p : tw.x.p; g.ldo = d; g.N = d; g.K = tw.mlp; g.m_valid = M;
            if (fold8) mxq_args(g);
            MM_TRY(gemm8_any(resid16 ? MMISS_EPI8_BIAS_RESID_BF16 : MMISS_EPI8_BIAS_RESID_F32, bm8_d, g, fold8 ? 2 : 0));
            MM_TRY(tap(l + 1));
            continue;
        } else if (sfold) {
            ep.bias = L.b1_f.as<float>(); ep.aux = L.c1.as<float>();
            ep.ln_stats = tw.stats.as<float>(); ep.ln_parts = d / 16; ep.ln_eps = eps; ep.stats16 = 1;
            MM_TRY(launch_gemm_skinny_fold(st, MMISS_EPI_LNFOLD_QGELU_BF16, tw.xb.p, L.w1_f.p, ep, M, tw.mlp, d));
        } else {
            if (resid16) MM_TRY(launch_layernorm16(st, tw.xb.as<uint16_t>(), L.ln2g.as<float>(), L.ln2b.as<float>(), tw.h.p, M, d, eps));
            else
            MM_TRY(launch_layernorm(st, tw.x.as<float>(), L.ln2g.as<float>(), L.ln2b.as<float>(), tw.h.p, true, nullptr, M,
                                    d, eps));
            if (p256(MMISS_EPI_BIAS_QGELU_BF16, tw.mlp)) MM_TRY(launch_gemm256p(st, MMISS_EPI_BIAS_QGELU_BF16, tw.h.p, L.w1.p, ep, padded(256), tw.mlp, d));
            else if (use256(tw.mlp)) MM_TRY(launch_gemm256(st, MMISS_EPI_BIAS_QGELU_BF16, tw.h.p, L.w1.p, ep, padded(256), tw.mlp, d));
            else MM_TRY(launch_gemm(st, MMISS_EPI_BIAS_QGELU_BF16, bm_mlp, tw.h.p, L.w1.p, ep, padded(bm_mlp), tw.mlp, d));
        }
        ep = GemmEpi{};
        ep.out = tw.x.p; ep.bias = L.b2.as<float>(); ep.ldo = d; ep.m_valid = M;
        ep.stats_out = (fuse || fold || sfold) ? tw.stats.as<float>() : nullptr;  // ... and for the next layer's LN1
        ep.xb_out = (fold || sfold) ? tw.xb.p : nullptr;
        ep.stats16 = sfold ? 1 : 0;
        ep.splitk_ws = tw.splitk.as<float>(); ep.splitk_ws_bytes = tw.splitk.bytes;
        if (resid16) {
            ep.out = tw.xb.p; ep.xb_out = nullptr;
            if (p160(d, tw.mlp)) MM_TRY(launch_gemm160p(st, tw.u.p, L.w2.p, ep, padded(160), d, tw.mlp));
            else MM_TRY(launch_gemm_resid16(st, bm_d, tw.u.p, L.w2.p, ep, padded(bm_d), d, tw.mlp));
        } else {
            MM_TRY(launch_gemm(st, MMISS_EPI_BIAS_RESID_F32, bm_d, tw.u.p, L.w2.p, ep, padded(bm_d), d, tw.mlp));
        }
        MM_TRY(tap(l + 1));
    }
    return MMISS_OK;
}

// K8: pooled row -> final LayerNorm -> projection (no bias) -> L2 normalise
int run_head(mmiss_encoder* e, Tower& tw, int B, float* out_dev, hipStream_t st) {
    const int d = tw.hidden, P = e->cfg.proj_dim;
    const int Bp = (int)round_up(B, 128);
    if (tw.pooled_compact)  // the last layer already compacted the pooled rows into xc
        MM_TRY(launch_layernorm(st, tw.xc.as<float>(), tw.lnf_g.as<float>(), tw.lnf_b.as<float>(), tw.pooled.p, true,
                                nullptr, B, d, e->cfg.ln_eps));
    else
        MM_TRY(launch_layernorm(st, tw.x.as<float>(), tw.lnf_g.as<float>(), tw.lnf_b.as<float>(), tw.pooled.p, true,
                                tw.pool_row.as<int32_t>(), B, d, e->cfg.ln_eps));
    GemmEpi ep{};
    ep.out = tw.proj_out.p; ep.ldo = P; ep.m_valid = B;
    MM_TRY(launch_gemm(st, MMISS_EPI_F32, 0, tw.pooled.p, tw.proj.p, ep, Bp, P, d));
    {
        MM_PROF("l2norm_rows", st, 3.0 * B * P, 8.0 * B * P);
        hipLaunchKernelGGL(l2norm_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, st, tw.proj_out.as<float>(), out_dev, B,
                           P, P);
    }
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

int encode_image_chunk(mmiss_encoder* e, const void* pix_dev, bool src_u8, int B, float* out_dev, hipStream_t st) {
    Tower& tw = e->vis;
    // "the embedding stage already left xb + row statistics" is said by THIS call's embedding stage only: a call that failed between
    // setting and consuming the flags must not make the next one fold stale statistics
    tw.embed_stats = tw.embed_stats16 = false;
    const int d = tw.hidden, S = e->cfg.v_image, P = e->cfg.v_patch;
    const int Mpatch = B * e->G * e->G;
    const int bm_p = gemm_pick_variant(Mpatch, d);
    const int Mpp = (int)round_up(Mpatch, bm_p % 1000);
    MM_TRY(launch_im2col(st, pix_dev, src_u8, e->patches.p, B, S, P, e->Kp));
    // One request at a time (the layers will run in the skinny folded mode): CLS rows, pre_layrnorm and the mode's entry
    // statistics are ONE launch behind the patch GEMM (prelayernorm_skinny_kernel) instead of three. Option embed_fused = 0: off.
    bool fp8_l = false;
    const bool fused_embed = !e->record_taps && d <= 1024 && (d % 16) == 0 && pick_ln_mode(e, tw, B * tw.T, fp8_l) == 0 && !fp8_l &&
                             skinny_fold_ok(tw, B * tw.T) && mmiss_option("embed_fused", 1) != 0;
    if (!fused_embed)
        hipLaunchKernelGGL(cls_rows_kernel, dim3((B * d + 255) / 256), dim3(256), 0, st, tw.x.as<float>(),
                           e->cls.as<float>(), tw.pos.as<float>(), B, tw.T, d);
    GemmEpi ep{};
    ep.out = tw.x.p; ep.aux = tw.pos.as<float>(); ep.ldo = d; ep.m_valid = Mpatch; ep.p0 = e->G * e->G; ep.p1 = tw.T;
    if (gemm_splitk_candidate((int64_t)(Mpp / (bm_p % 1000)) * (d / GEMM_BN), e->Kp)) {
        MM_TRY(tw.splitk.ensure((size_t)8 * Mpp * d * 4));
        ep.splitk_ws = tw.splitk.as<float>(); ep.splitk_ws_bytes = tw.splitk.bytes;
    }
    {   // round 3: on the 160 x 256 tile of gemm_bf16_p160.h when that grid is about one round of the chip and K is long
        const int M160 = (int)round_up(Mpatch, 160);
        const int64_t tiles = (int64_t)(M160 / 160) * (d / 256);
        if (mmiss_option("gemm_p160", 1) != 0 && (d % 256) == 0 && gemm160p_ok(M160, d, e->Kp) && tiles >= 200 && tiles <= 256 && e->Kp >= 2048)
            MM_TRY(launch_gemm160p_patch(st, e->patches.p, e->patch_w.p, ep, M160, d, e->Kp));
        else
            MM_TRY(launch_gemm(st, MMISS_EPI_PATCH_F32, bm_p, e->patches.p, e->patch_w.p, ep, Mpp, d, e->Kp));
    }
    // pre_layrnorm, in place on the fp32 residual stream (HF:modeling_clip.py:640); when the layers will want the bf16 copy
    // and the row statistics of the result (folded LayerNorm, bf16 residual stream) the same pass writes them
    {
        const int M = B * tw.T;
        bool fp8 = false;
        const int mode = pick_ln_mode(e, tw, M, fp8);
        const bool want = !e->record_taps && d % 128 == 0 &&
                          (mode == 2 || (mode == 0 && M >= mmiss_option("ln_fold_min_rows", 6000))) &&
                          mmiss_option("prelayernorm_stats", 1) != 0;
        if (fused_embed) {
            MM_PROF("layernorm", st, 12.0 * M * d, (double)M * d * 10);
            hipLaunchKernelGGL(prelayernorm_skinny_kernel, dim3((M + 3) / 4), dim3(256), 0, st, tw.x.as<float>(), e->cls.as<float>(),
                               tw.pos.as<float>(), e->pre_g.as<float>(), e->pre_b.as<float>(), tw.xb.as<uint16_t>(), tw.stats.as<float>(),
                               M, tw.T, d, e->cfg.ln_eps);
            MM_HIP(hipGetLastError());
            tw.embed_stats16 = true;
        } else if (want) {
            MM_PROF("layernorm", st, 10.0 * M * d, (double)M * d * 10);
            hipLaunchKernelGGL(layernorm_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, st, tw.x.as<float>(), e->pre_g.as<float>(),
                               e->pre_b.as<float>(), tw.xb.as<uint16_t>(), tw.stats.as<float>(), M, d, d / 64, e->cfg.ln_eps);
            MM_HIP(hipGetLastError());
            tw.embed_stats = true;
        } else {
            MM_TRY(launch_layernorm(st, tw.x.as<float>(), e->pre_g.as<float>(), e->pre_b.as<float>(), tw.x.p, false, nullptr, M, d,
                                    e->cfg.ln_eps));
        }
    }
    if (tw.pool_B < B || tw.pool_T != tw.T) {  // token 0 of every image: the same table until the shape changes
        const int nb = tw.ws_batch > B ? tw.ws_batch : B;
        hipLaunchKernelGGL(vision_pool_rows_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, tw.pool_row.as<int32_t>(), nb, tw.T);
        tw.pool_B = nb;
        tw.pool_T = tw.T;
    }
    MM_TRY(run_layers(e, tw, B, false, st));
    MM_TRY(run_head(e, tw, B, out_dev, st));
    tw.last_B = B;
    tw.last_T = tw.T;
    return MMISS_OK;
}

int encode_text_chunk(mmiss_encoder* e, const int32_t* ids_dev, int B, int T, float* out_dev, hipStream_t st) {
    Tower& tw = e->txt;
    const int d = tw.hidden;
    const int savedT = tw.T;
    tw.T = T;  // shorter-than-ctx sequences run at their own length (positions 0..T-1)
    hipLaunchKernelGGL(text_embed_kernel, dim3(B), dim3(256), 0, st, ids_dev, e->tok.as<float>(), tw.pos.as<float>(),
                       tw.x.as<float>(), tw.pool_row.as<int32_t>(), T, d, e->cfg.t_vocab, e->cfg.eos_token_id);
    int rc = run_layers(e, tw, B, true, st);
    if (rc == MMISS_OK) rc = run_head(e, tw, B, out_dev, st);
    tw.T = savedT;
    tw.last_B = B;
    tw.last_T = T;
    return rc;
}

int finish_call(mmiss_encoder* e, hipStream_t st, bool must_sync) {
    if (must_sync || !e->has_user_stream) {
        MM_HIP(hipStreamSynchronize(st));
    } else {
        if (!e->done_ev) MM_HIP(hipEventCreateWithFlags(&e->done_ev, hipEventDisableTiming));
        MM_HIP(hipEventRecord(e->done_ev, st));
        e->async_pending = true;
    }
    return MMISS_OK;
}


// Resize + centre-crop images b0 .. b0+nb-1 of a raw RGB8 blob into dst_dev (uint8 [nb,S,S,3], device), on st.
// `staged` != null: the chunk's byte range [its lo, its hi) of the host blob already sits at `staged` in HBM.
// host bytes -> device on `stream`. Default: hipMemcpyAsync from the caller's pageable memory. Option pinned_stage = 1: through
// the handle's ring of pinned blocks filled by parallel host copies (host_stager.h; stage_threads, stage_block_mb) from 4 MB
// on. Measured on the MI355X box (tools/stage_probe.py, profiles/host_staging_r05.txt): once a call holds several batches, so
// that batch k + 1 crosses PCIe while batch k is computed, BOTH forms run at what the link gives — 40-45 GB/s, 4-16 copy
// threads, 8-64 MB blocks alike (f32 pixels 41.6 vs 40.5 GB/s, RGB8 uploads 44.5 vs 44.8) — so the ring is off by default;
// the 26 GB/s of round 4's `pcie_inclusive` was one batch per call: copy and compute in series, not a slow copy.
static int enc_h2d(mmiss_encoder* enc, void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (bytes >= (size_t)(4 << 20) && mmiss_option("pinned_stage", 0) != 0) {
        if (!enc->stager.slot_bytes) {
            const int hw = (int)std::thread::hardware_concurrency();
            int nthr = mmiss_option("stage_threads", 0);
            if (nthr <= 0) nthr = hw >= 2 ? (hw / 2 < 8 ? hw / 2 : 8) : 1;
            if (nthr > (hw > 0 ? hw : 1)) nthr = hw > 0 ? hw : 1;
            if (nthr > 64) nthr = 64;
            int mb = mmiss_option("stage_block_mb", 16);
            mb = mb < 1 ? 1 : (mb > 256 ? 256 : mb);
            MM_TRY(enc->stager.init((size_t)mb << 20, nthr));
        }
        return enc->stager.h2d(dst, src, bytes, stream);
    }
    MM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
    return MMISS_OK;
}

int resize_chunk(mmiss_encoder* e, const uint8_t* rgb, bool rgb_dev, int64_t rgb_bytes, const int64_t* offsets,
                 const int32_t* heights, const int32_t* widths, int b0, int nb, uint8_t* dst_dev, hipStream_t st,
                 const uint8_t* staged = nullptr) {
    const int S = e->cfg.v_image;
    if (!e->rz_copied) MM_HIP(hipEventCreateWithFlags(&e->rz_copied, hipEventDisableTiming));
    else MM_HIP(hipEventSynchronize(e->rz_copied));  // the previous call's descriptor upload has left the buffer
    if ((size_t)nb > e->rz_host_cap) {
        if (e->rz_host) (void)hipHostFree(e->rz_host);
        e->rz_host = nullptr; e->rz_host_cap = 0;
        MM_HIP(hipHostMalloc(reinterpret_cast<void**>(&e->rz_host), sizeof(ResizeDesc) * (size_t)nb, hipHostMallocDefault));
        e->rz_host_cap = (size_t)nb;
    }
    int64_t pool = 0, lo = INT64_MAX, hi = 0;
    int max_ksx = 0;
    for (int i = 0; i < nb; ++i) {
        const int H = heights[b0 + i], W = widths[b0 + i];
        const int64_t off = offsets[b0 + i];
        if (H < 1 || W < 1 || H > (1 << 16) || W > (1 << 16))
            MM_FAIL(MMISS_ERR_ARG, "image %d: size %d x %d outside 1..65536", b0 + i, W, H);
        const int64_t bytes = (int64_t)H * W * 3;
        if (off < 0 || off + bytes > rgb_bytes)
            MM_FAIL(MMISS_ERR_ARG, "image %d: bytes [%lld, %lld) outside the %lld-byte source", b0 + i, (long long)off,
                    (long long)(off + bytes), (long long)rgb_bytes);
        ResizeDesc& d = e->rz_host[i];
        resize_geometry(H, W, S, d);
        if (d.ksx > 4096 || d.ksy > 4096)
            MM_FAIL(MMISS_ERR_UNSUPPORTED, "image %d: %d x %d -> %d needs %d / %d filter taps (limit 4096)", b0 + i, W, H, S,
                    d.ksx, d.ksy);
        d.src_off = off;
        max_ksx = d.ksx > max_ksx ? d.ksx : max_ksx;
        d.kx_off = pool; pool += (int64_t)d.ksx * S;
        d.ky_off = pool; pool += (int64_t)d.ksy * S;
        lo = off < lo ? off : lo;
        hi = off + bytes > hi ? off + bytes : hi;
    }
    const uint8_t* src = rgb;
    if (staged) {
        for (int i = 0; i < nb; ++i) e->rz_host[i].src_off -= lo;
        src = staged;
    } else if (!rgb_dev) {  // stage the byte range this chunk touches
        MM_TRY(e->raw_stage.ensure((size_t)(hi - lo)));
        MM_TRY(enc_h2d(e, e->raw_stage.p, rgb + lo, (size_t)(hi - lo), st));
        for (int i = 0; i < nb; ++i) e->rz_host[i].src_off -= lo;
        src = e->raw_stage.as<uint8_t>();
    }
    MM_TRY(e->rz_desc.ensure(sizeof(ResizeDesc) * (size_t)nb));
    MM_TRY(e->rz_pool.ensure((size_t)pool * 4));
    MM_TRY(e->rz_bounds.ensure((size_t)nb * 4 * S * 4));
    MM_HIP(hipMemcpyAsync(e->rz_desc.p, e->rz_host, sizeof(ResizeDesc) * (size_t)nb, hipMemcpyHostToDevice, st));
    MM_HIP(hipEventRecord(e->rz_copied, st));
    {
        MM_PROF("resize_coeffs", st, 0.0, (double)pool * 4);
        hipLaunchKernelGGL(resize_coeffs_kernel, dim3(nb, 2), dim3(256), 0, st, e->rz_desc.as<ResizeDesc>(),
                           e->rz_pool.as<int32_t>(), e->rz_bounds.as<int32_t>(), S);
        MM_HIP(hipGetLastError());
    }
    {
        MM_PROF("resize_crop", st, 0.0, (double)(hi - lo) + (double)nb * S * S * 3);
        launch_resize_crop(st, max_ksx, src, (rgb_dev && !staged) ? rgb_bytes : hi - lo, e->rz_desc.as<ResizeDesc>(), e->rz_pool.as<int32_t>(),
                           e->rz_bounds.as<int32_t>(), dst_dev, S, nb);
        MM_HIP(hipGetLastError());
    }
    return MMISS_OK;
}

}  // namespace

// ================================================================================================ C-ABI
extern "C" int mmiss_encoder_create(const mmiss_clip_config* cfg, int device, mmiss_encoder** out) {
    if (!cfg || !out) MM_FAIL(MMISS_ERR_ARG, "mmiss_encoder_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(mmiss_clip_config))
        MM_FAIL(MMISS_ERR_ARG, "mmiss_clip_config.struct_size %d != %zu (ABI mismatch)", cfg->struct_size,
                sizeof(mmiss_clip_config));
    auto bad_tower = [](int hidden, int layers, int heads, int mlp) {
        return hidden <= 0 || layers <= 0 || heads <= 0 || hidden != heads * 64 || hidden % 128 || mlp % 128 ||
               hidden > 1024 || mlp <= 0;
    };
    if (bad_tower(cfg->v_hidden, cfg->v_layers, cfg->v_heads, cfg->v_mlp) ||
        bad_tower(cfg->t_hidden, cfg->t_layers, cfg->t_heads, cfg->t_mlp))
        MM_FAIL(MMISS_ERR_UNSUPPORTED,
                "tower shape unsupported: need head_dim 64, hidden %% 128 == 0, hidden <= 1024, mlp %% 128 == 0");
    if (cfg->v_patch <= 0 || cfg->v_image <= 0 || cfg->v_image % cfg->v_patch)
        MM_FAIL(MMISS_ERR_ARG, "image size %d not a multiple of patch %d", cfg->v_image, cfg->v_patch);
    if (cfg->proj_dim <= 0 || cfg->proj_dim % 128) MM_FAIL(MMISS_ERR_UNSUPPORTED, "proj_dim %d must be a multiple of 128", cfg->proj_dim);
    const int G = cfg->v_image / cfg->v_patch;
    if (G * G + 1 > 288 || cfg->t_ctx > 288 || cfg->t_ctx <= 0)
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "sequence length > 288 tokens is not supported by the attention kernel");
    if (cfg->t_vocab <= 0) MM_FAIL(MMISS_ERR_ARG, "bad vocab size");
    MM_TRY(mmiss_use_device(device));

    mmiss_encoder* e = new (std::nothrow) mmiss_encoder();
    if (!e) MM_FAIL(MMISS_ERR_NOMEM, "out of host memory");
    e->cfg = *cfg;
    if (e->cfg.max_batch_image <= 0) e->cfg.max_batch_image = 256;
    if (e->cfg.max_batch_text <= 0) e->cfg.max_batch_text = 256;
    if (e->cfg.ln_eps <= 0.f) e->cfg.ln_eps = 1e-5f;
    e->device = device;
    e->G = G;
    e->Kp = (int)round_up(3 * cfg->v_patch * cfg->v_patch, 64);
    int rc = MMISS_OK;
    do {
        if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) {
            mmiss_set_error("hipStreamCreate failed");
            rc = MMISS_ERR_HIP;
            break;
        }
        rc = build_tower(e, e->vis, "vision_model", cfg->v_hidden, cfg->v_layers, cfg->v_heads, cfg->v_mlp, G * G + 1,
                         cfg->proj_dim, "post_layernorm", "visual_projection.weight");
        if (rc) break;
        rc = build_tower(e, e->txt, "text_model", cfg->t_hidden, cfg->t_layers, cfg->t_heads, cfg->t_mlp, cfg->t_ctx,
                         cfg->proj_dim, "final_layer_norm", "text_projection.weight");
        if (rc) break;
        const int dv = cfg->v_hidden, PP3 = 3 * cfg->v_patch * cfg->v_patch;
        if ((rc = alloc_zero(e->patch_w, (size_t)dv * e->Kp * 2))) break;
        if ((rc = alloc_zero(e->cls, (size_t)dv * 4))) break;
        if ((rc = alloc_zero(e->pre_g, (size_t)dv * 4))) break;
        if ((rc = alloc_zero(e->pre_b, (size_t)dv * 4))) break;
        if ((rc = alloc_zero(e->tok, (size_t)cfg->t_vocab * cfg->t_hidden * 4))) break;
        Slot s;
        s = Slot(); s.dst = &e->patch_w; s.rows = dv; s.cols = PP3; s.ldd = e->Kp; s.bf16 = true;
        e->slots["vision_model.embeddings.patch_embedding.weight"] = s;
        s = Slot(); s.dst = &e->cls; s.rows = 1; s.cols = dv; s.ldd = dv;
        e->slots["vision_model.embeddings.class_embedding"] = s;
        s = Slot(); s.dst = &e->pre_g; s.rows = 1; s.cols = dv; s.ldd = dv;
        e->slots["vision_model.pre_layrnorm.weight"] = s;  // (sic) the HF key really is "pre_layrnorm"
        s = Slot(); s.dst = &e->pre_b; s.rows = 1; s.cols = dv; s.ldd = dv;
        e->slots["vision_model.pre_layrnorm.bias"] = s;
        s = Slot(); s.dst = &e->tok; s.rows = cfg->t_vocab; s.cols = cfg->t_hidden; s.ldd = cfg->t_hidden;
        e->slots["text_model.embeddings.token_embedding.weight"] = s;
    } while (0);
    if (rc != MMISS_OK) {
        if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
        delete e;
        return rc;
    }
    *out = e;
    return MMISS_OK;
}

extern "C" int mmiss_encoder_destroy(mmiss_encoder* enc) {
    if (!enc) return MMISS_OK;
    (void)hipSetDevice(enc->device);
    (void)hipDeviceSynchronize();
    if (enc->own_stream) (void)hipStreamDestroy(enc->own_stream);
    if (enc->rz_copied) (void)hipEventDestroy(enc->rz_copied);
    if (enc->done_ev) (void)hipEventDestroy(enc->done_ev);
    enc->stager.shutdown();
    if (enc->copy_stream) (void)hipStreamDestroy(enc->copy_stream);
    for (int i = 0; i < 2; ++i) {
        if (enc->ev_copied[i]) (void)hipEventDestroy(enc->ev_copied[i]);
        if (enc->ev_free[i]) (void)hipEventDestroy(enc->ev_free[i]);
    }
    if (enc->rz_host) (void)hipHostFree(enc->rz_host);
    delete enc;
    return MMISS_OK;
}

extern "C" int mmiss_encoder_set_stream(mmiss_encoder* enc, void* hip_stream, int32_t use_own) {
    if (!enc) MM_FAIL(MMISS_ERR_ARG, "null encoder");
    std::lock_guard<std::mutex> lk(enc->mu);
    hipStream_t next = reinterpret_cast<hipStream_t>(hip_stream);
    const bool next_user = use_own == 0;
    if (next_user != enc->has_user_stream || (next_user && next != enc->user_stream)) {
        // the handle's workspaces are shared by consecutive calls: wait for what the last call left on the stream being
        // left — that work only, not whatever else the caller has queued there since (calls on the own stream return drained)
        MM_TRY(mmiss_use_device(enc->device));
        if (enc->async_pending) MM_HIP(hipEventSynchronize(enc->done_ev));
        enc->async_pending = false;
    }
    enc->user_stream = next;
    enc->has_user_stream = next_user;
    return MMISS_OK;
}

extern "C" int mmiss_encoder_set_weight(mmiss_encoder* enc, const char* hf_key, const float* data, int64_t numel,
                                        int* used) {
    if (!enc || !hf_key || !data) MM_FAIL(MMISS_ERR_ARG, "mmiss_encoder_set_weight: null argument");
    std::lock_guard<std::mutex> lk(enc->mu);
    MM_TRY(mmiss_use_device(enc->device));
    auto it = enc->slots.find(hf_key);
    if (it == enc->slots.end()) {
        if (used) *used = 0;
        return MMISS_OK;
    }
    const Slot& s = it->second;
    if (numel != s.rows * s.cols)
        MM_FAIL(MMISS_ERR_ARG, "weight %s: got %lld elements, expected %lld (%lld x %lld)", hf_key, (long long)numel,
                (long long)(s.rows * s.cols), (long long)s.rows, (long long)s.cols);
    hipStream_t st = enc->own_stream;
    const float* src = data;
    if (!mmiss_is_device_ptr(data)) {
        MM_TRY(enc->w_stage.ensure((size_t)numel * 4));
        MM_HIP(hipMemcpyAsync(enc->w_stage.p, data, (size_t)numel * 4, hipMemcpyHostToDevice, st));
        src = enc->w_stage.as<float>();
    }
    if (s.bf16) {
        uint16_t* dst = s.dst->as<uint16_t>() + s.dst_row_off * s.ldd;
        const int grid = (int)((numel + 255) / 256 < 4096 ? (numel + 255) / 256 : 4096);
        hipLaunchKernelGGL(convert_2d_bf16_kernel, dim3(grid), dim3(256), 0, st, src, dst, s.rows, (int)s.cols,
                           (int)s.ldd);
        MM_HIP(hipGetLastError());
    } else {
        // 1-D slots use dst_row_off as an element offset (ldd may be 0 for fused biases)
        float* dst = s.dst->as<float>() + (s.rows == 1 ? s.dst_row_off : s.dst_row_off * s.ldd);
        MM_HIP(hipMemcpyAsync(dst, src, (size_t)numel * 4, hipMemcpyDeviceToDevice, st));
    }
    MM_HIP(hipStreamSynchronize(st));
    enc->seen.insert(hf_key);
    if (used) *used = 1;
    return MMISS_OK;
}

static int build_fp8_weights(mmiss_encoder* enc);

extern "C" int mmiss_encoder_finalize(mmiss_encoder* enc) {
    if (!enc) MM_FAIL(MMISS_ERR_ARG, "null encoder");
    std::lock_guard<std::mutex> lk(enc->mu);
    std::string missing;
    int n_missing = 0;
    for (auto& kv : enc->slots)
        if (!enc->seen.count(kv.first)) {
            if (n_missing < 4) missing += (n_missing ? ", " : "") + kv.first;
            ++n_missing;
        }
    if (n_missing)
        MM_FAIL(MMISS_ERR_STATE, "mmiss_encoder_finalize: %d weight tensors missing (%s%s)", n_missing, missing.c_str(),
                n_missing > 4 ? ", ..." : "");
    // fold LayerNorm1 / LayerNorm2 into the QKV / FC1 weights (ln_mode 2)
    MM_TRY(mmiss_use_device(enc->device));
    for (Tower* tw : {&enc->vis, &enc->txt}) {
        const int d = tw->hidden;
        for (LayerW& L : tw->L) {
            MM_TRY(L.wqkv_f.alloc((size_t)3 * d * d * 2)); MM_TRY(L.cqkv.alloc((size_t)3 * d * 4)); MM_TRY(L.bqkv_f.alloc((size_t)3 * d * 4));
            MM_TRY(L.w1_f.alloc((size_t)tw->mlp * d * 2)); MM_TRY(L.c1.alloc((size_t)tw->mlp * 4)); MM_TRY(L.b1_f.alloc((size_t)tw->mlp * 4));
            hipLaunchKernelGGL(fold_ln_weights_kernel, dim3((3 * d + 3) / 4), dim3(256), 0, enc->own_stream, L.wqkv.as<uint16_t>(),
                               L.ln1g.as<float>(), L.ln1b.as<float>(), L.bqkv.as<float>(), L.wqkv_f.as<uint16_t>(),
                               L.cqkv.as<float>(), L.bqkv_f.as<float>(), 3 * d, d);
            hipLaunchKernelGGL(fold_ln_weights_kernel, dim3((tw->mlp + 3) / 4), dim3(256), 0, enc->own_stream, L.w1.as<uint16_t>(),
                               L.ln2g.as<float>(), L.ln2b.as<float>(), L.b1.as<float>(), L.w1_f.as<uint16_t>(),
                               L.c1.as<float>(), L.b1_f.as<float>(), tw->mlp, d);
        }
    }
    MM_HIP(hipGetLastError());
    MM_HIP(hipStreamSynchronize(enc->own_stream));
    if (enc->fp8_tower[0] || enc->fp8_tower[1]) MM_TRY(build_fp8_weights(enc));
    enc->finalized = true;
    return MMISS_OK;
}

// e4m3 copies of the QKV / FC1 / FC2 weights with per-output-channel scales (from the bf16 weights: 8 -> 3 mantissa bits)
static int build_fp8_weights(mmiss_encoder* enc) {
    hipStream_t st = enc->own_stream;
    for (Tower* tw : {&enc->vis, &enc->txt}) {
        if (tw->fp8_ready || !enc->fp8_tower[tw == &enc->txt ? 1 : 0]) continue;
        const int d = tw->hidden, mlp = tw->mlp;
        auto quant = [&](DevBuf& wb, DevBuf& w8, DevBuf& sc, int N, int K) -> int {
            MM_TRY(w8.alloc((size_t)N * K));
            MM_TRY(sc.alloc((size_t)N * 4));
            hipLaunchKernelGGL(quantize_weights_fp8_kernel, dim3((N + 3) / 4), dim3(256), 0, st, wb.as<uint16_t>(),
                               w8.as<uint8_t>(), sc.as<float>(), N, K);
            MM_HIP(hipGetLastError());
            return MMISS_OK;
        };
        auto quant_fold = [&](DevBuf& wf, DevBuf& w8, DevBuf& sc, DevBuf& c16, int N, int K) -> int {
            MM_TRY(w8.alloc((size_t)N * K));
            MM_TRY(sc.alloc((size_t)N * 4));
            MM_TRY(c16.alloc((size_t)N * 2));
            hipLaunchKernelGGL(quantize_weights_fp8_csum_kernel, dim3((N + 3) / 4), dim3(256), 0, st, wf.as<uint16_t>(), w8.as<uint8_t>(),
                               sc.as<float>(), c16.as<uint16_t>(), N, K);
            MM_HIP(hipGetLastError());
            return MMISS_OK;
        };
        for (LayerW& L : tw->L) {
            MM_TRY(quant(L.wqkv, L.wqkv8, L.sqkv, 3 * d, d));
            MM_TRY(quant(L.w1, L.w1_8, L.s1, mlp, d));
            MM_TRY(quant(L.w2, L.w2_8, L.s2, d, mlp));
            MM_TRY(quant(L.wo, L.wo8, L.so, d, d));
            if (d == 1024 && L.wqkv_f.p && L.w1_f.p) {   // the folded form (run_layers: fold8)
                MM_TRY(quant_fold(L.wqkv_f, L.wqkv8f, L.sqkvf, L.cqkv16, 3 * d, d));
                MM_TRY(quant_fold(L.w1_f, L.w1_8f, L.s1f, L.c1_16, mlp, d));
            }
        }
        tw->fp8_ready = true;
    }
    MM_HIP(hipStreamSynchronize(st));
    return MMISS_OK;
}

// (caller holds enc->mu) switch one tower between the bf16 and the fp8 GEMMs
static int set_tower_fp8(mmiss_encoder* enc, int tower, bool on) {
    if (enc->fp8_tower[tower] == on) return MMISS_OK;
    enc->fp8_tower[tower] = on;
    (tower ? enc->txt : enc->vis).ws_batch = 0;  // (re)allocate the workspaces with / without the fp8 buffers
    if (on && enc->finalized) MM_TRY(build_fp8_weights(enc));
    return MMISS_OK;
}

extern "C" int mmiss_encoder_set_precision(mmiss_encoder* enc, int32_t precision) {
    if (!enc) MM_FAIL(MMISS_ERR_ARG, "null encoder");
    if (precision != MMISS_PREC_BF16 && precision != MMISS_PREC_FP8 && precision != MMISS_PREC_BF16_F32RESID)
        MM_FAIL(MMISS_ERR_ARG, "unknown precision %d", precision);
    std::lock_guard<std::mutex> lk(enc->mu);
    MM_TRY(mmiss_use_device(enc->device));
    MM_HIP(hipStreamSynchronize(enc->stream()));
    enc->precision = precision;
    MM_TRY(set_tower_fp8(enc, 0, precision == MMISS_PREC_FP8));  // the vision tower follows the setting
    MM_TRY(set_tower_fp8(enc, 1, false));                         // the text tower never does by itself (see fp8_tower)
    return MMISS_OK;
}

extern "C" int mmiss_encoder_set_tower_precision(mmiss_encoder* enc, int32_t tower, int32_t precision) {
    if (!enc) MM_FAIL(MMISS_ERR_ARG, "null encoder");
    if (tower != MMISS_TOWER_VISION && tower != MMISS_TOWER_TEXT) MM_FAIL(MMISS_ERR_ARG, "unknown tower %d", tower);
    if (precision != MMISS_PREC_BF16 && precision != MMISS_PREC_FP8)
        MM_FAIL(MMISS_ERR_ARG, "tower precision must be MMISS_PREC_BF16 or MMISS_PREC_FP8, got %d", precision);
    std::lock_guard<std::mutex> lk(enc->mu);
    MM_TRY(mmiss_use_device(enc->device));
    MM_HIP(hipStreamSynchronize(enc->stream()));
    return set_tower_fp8(enc, tower, precision == MMISS_PREC_FP8);
}

// streams / events of the host-input pipeline, created on first use
static int ensure_pipeline(mmiss_encoder* enc) {
    if (!enc->copy_stream) MM_HIP(hipStreamCreateWithFlags(&enc->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (!enc->ev_copied[i]) MM_HIP(hipEventCreateWithFlags(&enc->ev_copied[i], hipEventDisableTiming));
        if (!enc->ev_free[i]) MM_HIP(hipEventCreateWithFlags(&enc->ev_free[i], hipEventDisableTiming));
    }
    return MMISS_OK;
}

// byte range [lo, hi) of the host blob touched by images b0 .. b0+nb-1 (validated again by resize_chunk)
static void rgb_chunk_range(const int64_t* offsets, const int32_t* heights, const int32_t* widths, int b0, int nb,
                            int64_t rgb_bytes, int64_t& lo, int64_t& hi) {
    lo = INT64_MAX; hi = 0;
    for (int i = b0; i < b0 + nb; ++i) {
        const int64_t off = offsets[i], end = off + (int64_t)heights[i] * widths[i] * 3;
        lo = off < lo ? off : lo;
        hi = end > hi ? end : hi;
    }
    if (lo < 0) lo = 0;
    if (hi > rgb_bytes) hi = rgb_bytes;
    if (hi < lo) hi = lo;
}

static int encode_image_impl(mmiss_encoder* enc, const void* pixels, bool src_u8, int32_t B, float* out) {
    if (!enc || !pixels || !out) MM_FAIL(MMISS_ERR_ARG, "mmiss_encode_image: null argument");
    if (B < 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_encode_image: B = %d", B);
    std::lock_guard<std::mutex> lk(enc->mu);
    if (!enc->finalized) MM_FAIL(MMISS_ERR_STATE, "mmiss_encode_image before mmiss_encoder_finalize");
    if (B == 0) return MMISS_OK;
    MM_TRY(mmiss_use_device(enc->device));
    hipStream_t st = enc->stream();
    const int maxb = enc->cfg.max_batch_image, S = enc->cfg.v_image, P = enc->cfg.proj_dim;
    MM_TRY(ensure_tower_ws(enc, enc->vis, maxb, P));
    if (!enc->patches.p) {
        const int64_t Mpp = round_up((int64_t)maxb * enc->G * enc->G, 128) + 192;
        MM_TRY(alloc_zero(enc->patches, (size_t)Mpp * enc->Kp * 2));
    }
    const bool in_dev = mmiss_is_device_ptr(pixels), out_dev = mmiss_is_device_ptr(out);
    const size_t img_bytes = (size_t)3 * S * S * (src_u8 ? 1 : 4);
    // host pixels, several chunks: chunk k+1 crosses PCIe on copy_stream while chunk k is computed (see encode_rgb_impl)
    const bool piped = !in_dev && B > maxb;
    if (piped) {
        MM_TRY(ensure_pipeline(enc));
        MM_TRY(enc->stage2[0].ensure(img_bytes * maxb));
        MM_TRY(enc->stage2[1].ensure(img_bytes * maxb));
        MM_TRY(enc_h2d(enc, enc->stage2[0].p, pixels, img_bytes * maxb, enc->copy_stream));
        MM_HIP(hipEventRecord(enc->ev_copied[0], enc->copy_stream));
    } else if (!in_dev) {
        MM_TRY(enc->pix_stage.ensure(img_bytes * maxb));
    }
    for (int b0 = 0, k = 0; b0 < B; b0 += maxb, ++k) {
        const int nb = (B - b0 < maxb) ? B - b0 : maxb;
        const char* src = reinterpret_cast<const char*>(pixels) + (size_t)b0 * img_bytes;
        if (piped) {
            MM_HIP(hipStreamWaitEvent(st, enc->ev_copied[k & 1], 0));
            src = enc->stage2[k & 1].as<char>();
        } else if (!in_dev) {
            MM_TRY(enc_h2d(enc, enc->pix_stage.p, src, img_bytes * nb, st));
            src = enc->pix_stage.as<char>();
        }
        float* dst = out_dev ? out + (size_t)b0 * P : enc->vis.out_stage.as<float>();
        MM_TRY(encode_image_chunk(enc, src, src_u8, nb, dst, st));
        if (piped) {
            MM_HIP(hipEventRecord(enc->ev_free[k & 1], st));  // (recorded after the whole chunk; im2col is its only reader)
            if (b0 + maxb < B) {
                const int nxt = (k + 1) & 1, b1 = b0 + maxb, n1 = (B - b1 < maxb) ? B - b1 : maxb;
                if (k >= 1) MM_HIP(hipStreamWaitEvent(enc->copy_stream, enc->ev_free[nxt], 0));
                MM_TRY(enc_h2d(enc, enc->stage2[nxt].p, reinterpret_cast<const char*>(pixels) + (size_t)b1 * img_bytes, img_bytes * n1,
                               enc->copy_stream));
                MM_HIP(hipEventRecord(enc->ev_copied[nxt], enc->copy_stream));
            }
        }
        if (!out_dev) {
            MM_HIP(hipMemcpyAsync(out + (size_t)b0 * P, dst, (size_t)nb * P * 4, hipMemcpyDeviceToHost, st));
            MM_HIP(hipStreamSynchronize(st));
        } else if (!in_dev) {
            MM_HIP(hipStreamSynchronize(st));  // the staging buffer is reused by the next chunk
        }
    }
    if (piped) MM_HIP(hipStreamSynchronize(enc->copy_stream));
    return finish_call(enc, st, !out_dev);
}

extern "C" int mmiss_encode_image(mmiss_encoder* enc, const float* pixels, int32_t B, float* out) {
    return encode_image_impl(enc, pixels, false, B, out);
}

extern "C" int mmiss_encode_image_u8(mmiss_encoder* enc, const uint8_t* pixels_u8, int32_t B, float* out) {
    return encode_image_impl(enc, pixels_u8, true, B, out);
}

// raw RGB8 images of any size: resize (shortest edge, bicubic) + centre crop on the GPU, then the uint8 encode path
static int encode_rgb_impl(mmiss_encoder* enc, const uint8_t* rgb, int64_t rgb_bytes, const int64_t* offsets,
                           const int32_t* heights, const int32_t* widths, int32_t B, uint8_t* out_u8, float* out_emb) {
    if (!enc || !rgb || !offsets || !heights || !widths || (!out_u8 && !out_emb))
        MM_FAIL(MMISS_ERR_ARG, "mmiss_*_rgb: null argument");
    if (B < 0 || rgb_bytes < 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_*_rgb: B = %d, rgb_bytes = %lld", B, (long long)rgb_bytes);
    std::lock_guard<std::mutex> lk(enc->mu);
    if (out_emb && !enc->finalized) MM_FAIL(MMISS_ERR_STATE, "mmiss_encode_image_rgb before mmiss_encoder_finalize");
    if (B == 0) return MMISS_OK;
    MM_TRY(mmiss_use_device(enc->device));
    hipStream_t st = enc->stream();
    const int maxb = enc->cfg.max_batch_image, S = enc->cfg.v_image, P = enc->cfg.proj_dim;
    if (out_emb) {
        MM_TRY(ensure_tower_ws(enc, enc->vis, maxb, P));
        if (!enc->patches.p) {
            const int64_t Mpp = round_up((int64_t)maxb * enc->G * enc->G, 128) + 192;
            MM_TRY(alloc_zero(enc->patches, (size_t)Mpp * enc->Kp * 2));
        }
    }
    const bool in_dev = mmiss_is_device_ptr(rgb);
    const bool out_dev = mmiss_is_device_ptr(out_emb ? (const void*)out_emb : (const void*)out_u8);
    const size_t crop_bytes = (size_t)3 * S * S;
    MM_TRY(enc->crop_stage.ensure(crop_bytes * maxb));
    // host blob, several chunks: chunk k+1 crosses PCIe (copy_stream -> stage2[(k+1)&1]) while chunk k is computed
    const bool piped = !in_dev && B > maxb;
    if (piped) {
        MM_TRY(ensure_pipeline(enc));
        int64_t need = 0;
        for (int b0 = 0; b0 < B; b0 += maxb) {
            int64_t lo, hi;
            rgb_chunk_range(offsets, heights, widths, b0, (B - b0 < maxb) ? B - b0 : maxb, rgb_bytes, lo, hi);
            need = hi - lo > need ? hi - lo : need;
        }
        MM_TRY(enc->stage2[0].ensure((size_t)need + 4));
        MM_TRY(enc->stage2[1].ensure((size_t)need + 4));
        int64_t lo, hi;
        rgb_chunk_range(offsets, heights, widths, 0, maxb, rgb_bytes, lo, hi);
        MM_TRY(enc_h2d(enc, enc->stage2[0].p, rgb + lo, (size_t)(hi - lo), enc->copy_stream));
        MM_HIP(hipEventRecord(enc->ev_copied[0], enc->copy_stream));
    }
    for (int b0 = 0, k = 0; b0 < B; b0 += maxb, ++k) {
        const int nb = (B - b0 < maxb) ? B - b0 : maxb;
        uint8_t* crops = (out_u8 && out_dev) ? out_u8 + crop_bytes * b0 : enc->crop_stage.as<uint8_t>();
        if (piped) {
            const int cur = k & 1;
            MM_HIP(hipStreamWaitEvent(st, enc->ev_copied[cur], 0));
            MM_TRY(resize_chunk(enc, rgb, true, rgb_bytes, offsets, heights, widths, b0, nb, crops, st,
                                enc->stage2[cur].as<uint8_t>()));
            MM_HIP(hipEventRecord(enc->ev_free[cur], st));  // the resize has consumed the staged bytes
        } else {
            MM_TRY(resize_chunk(enc, rgb, in_dev, rgb_bytes, offsets, heights, widths, b0, nb, crops, st));
        }
        if (out_u8 && !out_dev) {
            MM_HIP(hipMemcpyAsync(out_u8 + crop_bytes * b0, crops, crop_bytes * nb, hipMemcpyDeviceToHost, st));
        }
        if (out_emb) {
            float* dst = out_dev ? out_emb + (size_t)b0 * P : enc->vis.out_stage.as<float>();
            MM_TRY(encode_image_chunk(enc, crops, true, nb, dst, st));
        }
        if (piped && b0 + maxb < B) {
            // chunk k is enqueued: now move chunk k+1 (a pageable-memory copy blocks this thread, not the GPU)
            const int nxt = (k + 1) & 1, b1 = b0 + maxb;
            int64_t lo, hi;
            rgb_chunk_range(offsets, heights, widths, b1, (B - b1 < maxb) ? B - b1 : maxb, rgb_bytes, lo, hi);
            if (k >= 1) MM_HIP(hipStreamWaitEvent(enc->copy_stream, enc->ev_free[nxt], 0));
            MM_TRY(enc_h2d(enc, enc->stage2[nxt].p, rgb + lo, (size_t)(hi - lo), enc->copy_stream));
            MM_HIP(hipEventRecord(enc->ev_copied[nxt], enc->copy_stream));
        }
        if (out_emb && !out_dev) {
            float* dst = enc->vis.out_stage.as<float>();
            MM_HIP(hipMemcpyAsync(out_emb + (size_t)b0 * P, dst, (size_t)nb * P * 4, hipMemcpyDeviceToHost, st));
        }
        // staging buffers (source bytes, descriptors, crops, output) are reused by the next chunk
        if (b0 + maxb < B || !out_dev || !in_dev) MM_HIP(hipStreamSynchronize(st));
    }
    if (piped) MM_HIP(hipStreamSynchronize(enc->copy_stream));
    return finish_call(enc, st, !out_dev);
}

extern "C" int mmiss_resize_crop_rgb(mmiss_encoder* enc, const uint8_t* rgb, int64_t rgb_bytes, const int64_t* offsets,
                                     const int32_t* heights, const int32_t* widths, int32_t B, uint8_t* out_u8) {
    return encode_rgb_impl(enc, rgb, rgb_bytes, offsets, heights, widths, B, out_u8, nullptr);
}

extern "C" int mmiss_encode_image_rgb(mmiss_encoder* enc, const uint8_t* rgb, int64_t rgb_bytes, const int64_t* offsets,
                                      const int32_t* heights, const int32_t* widths, int32_t B, float* out) {
    return encode_rgb_impl(enc, rgb, rgb_bytes, offsets, heights, widths, B, nullptr, out);
}

extern "C" int mmiss_encode_text(mmiss_encoder* enc, const int32_t* ids, int32_t B, int32_t T, float* out) {
    if (!enc || !ids || !out) MM_FAIL(MMISS_ERR_ARG, "mmiss_encode_text: null argument");
    if (B < 0 || T <= 0 || T > enc->cfg.t_ctx)
        MM_FAIL(MMISS_ERR_ARG, "mmiss_encode_text: B=%d T=%d (context length is %d)", B, T, enc->cfg.t_ctx);
    std::lock_guard<std::mutex> lk(enc->mu);
    if (!enc->finalized) MM_FAIL(MMISS_ERR_STATE, "mmiss_encode_text before mmiss_encoder_finalize");
    if (B == 0) return MMISS_OK;
    MM_TRY(mmiss_use_device(enc->device));
    hipStream_t st = enc->stream();
    const int maxb = enc->cfg.max_batch_text, P = enc->cfg.proj_dim;
    MM_TRY(ensure_tower_ws(enc, enc->txt, maxb, P));
    const bool in_dev = mmiss_is_device_ptr(ids), out_dev = mmiss_is_device_ptr(out);
    if (!in_dev) MM_TRY(enc->ids_stage.ensure((size_t)maxb * enc->cfg.t_ctx * 4));
    for (int b0 = 0; b0 < B; b0 += maxb) {
        const int nb = (B - b0 < maxb) ? B - b0 : maxb;
        const int32_t* src = ids + (size_t)b0 * T;
        if (!in_dev) {
            MM_HIP(hipMemcpyAsync(enc->ids_stage.p, src, (size_t)nb * T * 4, hipMemcpyHostToDevice, st));
            src = enc->ids_stage.as<int32_t>();
        }
        float* dst = out_dev ? out + (size_t)b0 * P : enc->txt.out_stage.as<float>();
        MM_TRY(encode_text_chunk(enc, src, nb, T, dst, st));
        if (!out_dev) {
            MM_HIP(hipMemcpyAsync(out + (size_t)b0 * P, dst, (size_t)nb * P * 4, hipMemcpyDeviceToHost, st));
            MM_HIP(hipStreamSynchronize(st));
        } else if (!in_dev) {
            MM_HIP(hipStreamSynchronize(st));
        }
    }
    return finish_call(enc, st, !out_dev);
}

extern "C" int mmiss_dbg_encoder_set_fuse_ln(mmiss_encoder* enc, int on) {
    if (!enc) MM_FAIL(MMISS_ERR_ARG, "null encoder");
    std::lock_guard<std::mutex> lk(enc->mu);
    if (on == 1) MM_FAIL(MMISS_ERR_UNSUPPORTED, "LayerNorm mode 1 (normalised during operand staging) was removed in round 4: measured 342-371 TF, profiles/gemm_variants_r01.md");
    enc->ln_mode = on < 0 ? -1 : (on > 2 ? 2 : on);  // -1 automatic, 0 separate kernels, 2 folded
    return MMISS_OK;
}

int mmiss_index_build_flags(void);   // api_index.hip: the same question for that translation unit
extern "C" int mmiss_dbg_build_flags(void) {
    int f = mmiss_index_build_flags();
#ifdef MMISS_EXPERIMENTS
    f |= 1;
#endif
#if defined(P256_NO_LATE_WAIT) || defined(P256_SPLIT_STAGE) || defined(P256_STAGE_FIRST) || defined(P256_A_POLICY) || defined(P256_W_POLICY) || defined(P256_PRIO) || defined(MMISS_SCAN_NT) || defined(Q256_STAGE_MID)
    f |= 2;   // built with a timing-experiment macro (tools/*_ab.sh): NOT a product build
#endif
    return f;
}

extern "C" int mmiss_dbg_encoder_record_taps(mmiss_encoder* enc, int on) {
    if (!enc) MM_FAIL(MMISS_ERR_ARG, "null encoder");
    std::lock_guard<std::mutex> lk(enc->mu);
    enc->record_taps = on != 0;
    // force workspace re-allocation with the tap buffer
    if (on) { enc->vis.ws_batch = 0; enc->txt.ws_batch = 0; }
    return MMISS_OK;
}

extern "C" int mmiss_encoder_tap(mmiss_encoder* enc, int tower, int what, float* out, int64_t cap, int64_t* written) {
    if (!enc || !out || cap < 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_encoder_tap: bad argument");
    std::lock_guard<std::mutex> lk(enc->mu);
    MM_TRY(mmiss_use_device(enc->device));
    Tower& tw = tower == 0 ? enc->vis : enc->txt;
    if (tw.last_B <= 0) MM_FAIL(MMISS_ERR_STATE, "mmiss_encoder_tap: no encode call yet on this tower");
    hipStream_t st = enc->stream();
    MM_HIP(hipStreamSynchronize(st));
    const int B = tw.last_B, d = tw.hidden, P = enc->cfg.proj_dim;
    int64_t n = 0;
    const bool out_dev = mmiss_is_device_ptr(out);
    DevBuf tmp;
    if (what >= 0 && what <= tw.layers) {
        if (!tw.taps.p) MM_FAIL(MMISS_ERR_STATE, "taps not recorded: call mmiss_dbg_encoder_record_taps(enc, 1) first");
        n = (int64_t)B * tw.last_T * d;  // rows are packed at the last call's sequence length
        if (n > cap) n = cap;
        const float* src = tw.taps.as<float>() + (size_t)what * tw.tap_stride;
        MM_HIP(hipMemcpy(out, src, (size_t)n * 4, out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    } else if (what == 100) {
        n = (int64_t)B * d;
        if (n > cap) n = cap;
        MM_TRY(tmp.alloc((size_t)n * 4));
        hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(256), dim3(256), 0, st, tw.pooled.as<uint16_t>(), tmp.as<float>(), n);
        MM_HIP(hipStreamSynchronize(st));
        MM_HIP(hipMemcpy(out, tmp.p, (size_t)n * 4, out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    } else if (what == 101) {
        n = (int64_t)B * P;
        if (n > cap) n = cap;
        MM_HIP(hipMemcpy(out, tw.proj_out.p, (size_t)n * 4, out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    } else {
        MM_FAIL(MMISS_ERR_ARG, "mmiss_encoder_tap: unknown tap %d", what);
    }
    if (written) *written = n;
    return MMISS_OK;
}

// ================================================================================================ debug ABI
extern "C" int mmiss_dbg_gemm(int device, void* hip_stream, int epi, int variant, const void* A, const void* W,
                              void* out, const float* bias, const float* aux, int32_t M, int32_t N, int32_t K,
                              int32_t p0, int32_t p1) {
    if (!A || !W || !out) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_gemm: null pointer");
    MM_TRY(mmiss_use_device(device));
    GemmEpi ep{};
    ep.out = out; ep.bias = bias; ep.aux = aux; ep.ldo = N; ep.m_valid = M; ep.p0 = p0; ep.p1 = p1;
    static DevBuf dbg_splitk;  // debug entry only: scratch so that the split-K path can be exercised
    if ((int64_t)M * N <= (1 << 22)) {
        MM_TRY(dbg_splitk.ensure((size_t)8 * M * N * 4));
        ep.splitk_ws = dbg_splitk.as<float>(); ep.splitk_ws_bytes = dbg_splitk.bytes;
    }
    if (variant == 256) return launch_gemm256(reinterpret_cast<hipStream_t>(hip_stream), epi, A, W, ep, M, N, K);
    if (variant > 1000) MM_FAIL(MMISS_ERR_UNSUPPORTED, "GEMM variant %d (ring pipeline / BM x 256 tiles) was removed in round 4: measured slower, profiles/gemm_variants_r01.md", variant);
    return launch_gemm(reinterpret_cast<hipStream_t>(hip_stream), epi, variant, A, W, ep, M, N, K);
}

// the persistent 256 x 256 kernel in isolation (gemm_bf16_p256.h): epi 1 / 2 (bias, bias + QuickGELU) or 7 / 8 (the same
// behind a folded LayerNorm: ln_stats [M][K/64][2], aux = c [N], bias = b' [N]); iters > 0 also times it
extern "C" int mmiss_dbg_gemm_p256(int device, void* hip_stream, int epi, const void* A, const void* W, void* out,
                                   const float* bias, const float* aux, const float* ln_stats, float ln_eps, int32_t M,
                                   int32_t N, int32_t K, int32_t m_valid, int32_t iters, float* ms_per_launch) {
    if (!A || !W || !out) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_gemm_p256: null pointer");
    MM_TRY(mmiss_use_device(device));
    GemmEpi ep{};
    ep.out = out; ep.bias = bias; ep.aux = aux; ep.ldo = N; ep.m_valid = m_valid;
    ep.ln_stats = ln_stats; ep.ln_parts = K / 64; ep.ln_eps = ln_eps;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    if (iters <= 0 || !ms_per_launch) return launch_gemm256p(st, epi, A, W, ep, M, N, K);
    hipEvent_t e0, e1;
    MM_HIP(hipEventCreate(&e0));
    MM_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) MM_TRY(launch_gemm256p(st, epi, A, W, ep, M, N, K));
    MM_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) MM_TRY(launch_gemm256p(st, epi, A, W, ep, M, N, K));
    MM_HIP(hipEventRecord(e1, st));
    MM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MM_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return MMISS_OK;
}

// the residual GEMM on a bf16 stream (out = bf16(f32(out) + A W^T + bias), in place; stats_out [M][N/64][2] optional) in
// isolation: variant 0 = the 160 x 256 tile on the staggered loop (gemm160p_kernel), 128 / 160 / 192 = the 128-column kernel
// with that tile height; iters > 0 also times it (the stream keeps accumulating: only the time means anything then)
extern "C" int mmiss_dbg_gemm_resid16(int device, void* hip_stream, int variant, const void* A, const void* W, void* out,
                                      const float* bias, float* stats_out, int32_t M, int32_t N, int32_t K, int32_t m_valid,
                                      int32_t iters, float* ms_per_launch) {
    if (!A || !W || !out || !bias) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_gemm_resid16: null pointer");
    MM_TRY(mmiss_use_device(device));
    GemmEpi ep{};
    ep.out = out; ep.bias = bias; ep.ldo = N; ep.m_valid = m_valid; ep.stats_out = stats_out;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    auto run = [&]() -> int {
        return variant == 0 ? launch_gemm160p(st, A, W, ep, M, N, K) : launch_gemm_resid16(st, variant, A, W, ep, M, N, K);
    };
    if (iters <= 0 || !ms_per_launch) return run();
    hipEvent_t e0, e1;
    MM_HIP(hipEventCreate(&e0));
    MM_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) MM_TRY(run());
    MM_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) MM_TRY(run());
    MM_HIP(hipEventRecord(e1, st));
    MM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MM_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return MMISS_OK;
}

extern "C" int mmiss_dbg_gemm_time(int device, int epi, int variant, const void* A, const void* W, void* out,
                                   const float* bias, const float* aux, int32_t M, int32_t N, int32_t K, int32_t p0,
                                   int32_t p1, int32_t iters, float* ms_per_launch) {
    if (!A || !W || !out || !ms_per_launch || iters <= 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_gemm_time: bad argument");
    MM_TRY(mmiss_use_device(device));
    GemmEpi ep{};
    ep.out = out; ep.bias = bias; ep.aux = aux; ep.ldo = N; ep.m_valid = M; ep.p0 = p0; ep.p1 = p1;
    hipEvent_t e0, e1;
    MM_HIP(hipEventCreate(&e0));
    MM_HIP(hipEventCreate(&e1));
    auto run = [&]() -> int {
        if (variant == 256) return launch_gemm256(nullptr, epi, A, W, ep, M, N, K);
        if (variant > 1000) MM_FAIL(MMISS_ERR_UNSUPPORTED, "GEMM variant %d was removed in round 4", variant);
        return launch_gemm(nullptr, epi, variant, A, W, ep, M, N, K);
    };
    for (int i = 0; i < 3; ++i) MM_TRY(run());
    MM_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) MM_TRY(run());
    MM_HIP(hipEventRecord(e1, nullptr));
    MM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MM_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = ms / iters;
    return MMISS_OK;
}

extern "C" int mmiss_dbg_layernorm(int device, void* hip_stream, const float* x, const float* gamma, const float* beta,
                                   void* out, int32_t out_bf16, int32_t M, int32_t d, float eps) {
    if (!x || !gamma || !beta || !out) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_layernorm: null pointer");
    MM_TRY(mmiss_use_device(device));
    return launch_layernorm(reinterpret_cast<hipStream_t>(hip_stream), x, gamma, beta, out, out_bf16 != 0, nullptr, M, d,
                            eps);
}

extern "C" int mmiss_dbg_attention(int device, void* hip_stream, const void* qkv, void* ctx, int32_t B, int32_t T,
                                   int32_t H, int32_t causal) {
    if (!qkv || !ctx) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_attention: null pointer");
    MM_TRY(mmiss_use_device(device));
    return launch_attention(reinterpret_cast<hipStream_t>(hip_stream), qkv, ctx, B, T, H, causal != 0);
}

extern "C" int mmiss_dbg_im2col(int device, void* hip_stream, const float* pixels, void* out, int32_t B, int32_t S,
                                int32_t P, int32_t Kp) {
    if (!pixels || !out) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_im2col: null pointer");
    MM_TRY(mmiss_use_device(device));
    return launch_im2col(reinterpret_cast<hipStream_t>(hip_stream), pixels, false, out, B, S, P, Kp);
}


// Experiment (tools/gemm_split_test.py): one GEMM over M rows vs two half-M GEMMs back to back vs the two halves on
// two streams joined by events. ms[0..2] = milliseconds per GEMM-equivalent.
extern "C" int mmiss_dbg_gemm_split_time(int device, int epi, int bm, const void* A, const void* W, void* out,
                                         const float* bias, int32_t M, int32_t N, int32_t K, int32_t iters, float* ms) {
    if (!A || !W || !out || !ms || iters <= 0 || (M % (2 * bm))) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_gemm_split_time: bad argument");
    MM_TRY(mmiss_use_device(device));
    hipStream_t s0, s1;
    MM_HIP(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    MM_HIP(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, join;
    MM_HIP(hipEventCreate(&e0)); MM_HIP(hipEventCreate(&e1));
    MM_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); MM_HIP(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int out_elt = (epi == MMISS_EPI_BIAS_BF16 || epi == MMISS_EPI_BIAS_QGELU_BF16) ? 2 : 4;
    const int Mh = M / 2;
    auto full = [&](hipStream_t s) -> int {
        GemmEpi ep{}; ep.out = out; ep.bias = bias; ep.ldo = N; ep.m_valid = M;
        return launch_gemm(s, epi, bm, A, W, ep, M, N, K);
    };
    auto half = [&](hipStream_t s, int which) -> int {
        GemmEpi ep{}; ep.out = (char*)out + (size_t)which * Mh * N * out_elt; ep.bias = bias; ep.ldo = N; ep.m_valid = Mh;
        return launch_gemm(s, epi, bm, (const char*)A + (size_t)which * Mh * K * 2, W, ep, Mh, N, K);
    };
    for (int mode = 0; mode < 3; ++mode) {
        for (int it = -3; it < iters; ++it) {
            if (it == 0) MM_HIP(hipEventRecord(e0, s0));
            if (mode == 0) { MM_TRY(full(s0)); }
            else if (mode == 1) { MM_TRY(half(s0, 0)); MM_TRY(half(s0, 1)); }
            else {
                MM_HIP(hipEventRecord(fork, s0));
                MM_HIP(hipStreamWaitEvent(s1, fork, 0));
                MM_TRY(half(s0, 0));
                MM_TRY(half(s1, 1));
                MM_HIP(hipEventRecord(join, s1));
                MM_HIP(hipStreamWaitEvent(s0, join, 0));
            }
        }
        MM_HIP(hipEventRecord(e1, s0));
        MM_HIP(hipEventSynchronize(e1));
        float t = 0.f;
        MM_HIP(hipEventElapsedTime(&t, e0, e1));
        ms[mode] = t / iters;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(fork); (void)hipEventDestroy(join);
    (void)hipStreamDestroy(s0); (void)hipStreamDestroy(s1);
    return MMISS_OK;
}

// ------------------------------------------------------------------------------------------------ fp8 kernels in isolation
extern "C" int mmiss_dbg_quantize_weights_fp8(int device, void* hip_stream, const void* w_bf16, void* w8, float* scale,
                                              int32_t N, int32_t K) {
    if (!w_bf16 || !w8 || !scale || N <= 0 || K <= 0 || (K % 4)) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_quantize_weights_fp8: bad argument");
    MM_TRY(mmiss_use_device(device));
    hipLaunchKernelGGL(quantize_weights_fp8_kernel, dim3((N + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream),
                       reinterpret_cast<const uint16_t*>(w_bf16), reinterpret_cast<uint8_t*>(w8), scale, N, K);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// bf16 rows in (the bf16 residual stream): d = 512 / 1024 take the wide kernel of round 4, other d the first form
extern "C" int mmiss_dbg_layernorm16_mxfp8(int device, void* hip_stream, const void* x_bf16, const float* gamma, const float* beta,
                                           void* out8, void* out_scale, int32_t M, int32_t d, float eps) {
    if (!x_bf16 || !gamma || !beta || !out8 || !out_scale) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_layernorm16_mxfp8: null pointer");
    if (M <= 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_layernorm16_mxfp8: M = %d", M);
    MM_TRY(mmiss_use_device(device));
    return launch_layernorm_mxfp8(reinterpret_cast<hipStream_t>(hip_stream), x_bf16, true, gamma, beta,
                                  reinterpret_cast<uint8_t*>(out8), reinterpret_cast<uint8_t*>(out_scale), M, d, eps);
}

// qkv bf16 [B*T, 3*H*64] -> the attention output as MXFP8: ctx8 e4m3 [B*T, H*64] + permuted E8M0 scales [B*T, 16 * ceil(H*64 / 512)]
// (non-causal; T <= 128: the one-pass kernels, 129 <= T <= 288: the long-sequence form)
extern "C" int mmiss_dbg_attention_mx(int device, void* hip_stream, const void* qkv, void* ctx8, void* ctx_scale, int32_t B,
                                      int32_t T, int32_t H) {
    if (!qkv || !ctx8 || !ctx_scale) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_attention_mx: null pointer");
    MM_TRY(mmiss_use_device(device));
    return launch_attention_mx(reinterpret_cast<hipStream_t>(hip_stream), qkv, reinterpret_cast<uint8_t*>(ctx8),
                               reinterpret_cast<uint8_t*>(ctx_scale), mx_scale_row_bytes(H * 64), B, T, H);
}

extern "C" int mmiss_dbg_layernorm_mxfp8(int device, void* hip_stream, const float* x, const float* gamma, const float* beta,
                                         void* out8, void* out_scale, int32_t M, int32_t d, float eps) {
    if (!x || !gamma || !beta || !out8 || !out_scale) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_layernorm_mxfp8: null pointer");
    MM_TRY(mmiss_use_device(device));
    return launch_layernorm_mxfp8(reinterpret_cast<hipStream_t>(hip_stream), x, false, gamma, beta, reinterpret_cast<uint8_t*>(out8),
                                  reinterpret_cast<uint8_t*>(out_scale), M, d, eps);
}

extern "C" int mmiss_dbg_gemm8(int device, void* hip_stream, int epi, int bm, const void* A8, const void* As, const void* W8,
                               const float* wscale, const float* bias, void* out, void* out_scale, int32_t M, int32_t N,
                               int32_t K) {
    MM_TRY(mmiss_use_device(device));
    Gemm8Args g{};
    g.A = reinterpret_cast<const uint8_t*>(A8); g.As = reinterpret_cast<const uint8_t*>(As); g.ld_as = mx_scale_row_bytes(K);
    g.W = reinterpret_cast<const uint8_t*>(W8); g.wscale = wscale; g.bias = bias; g.out = out;
    g.out_scale = reinterpret_cast<uint8_t*>(out_scale); g.ld_os = mx_scale_row_bytes(N);
    g.M = M; g.N = N; g.K = K; g.ldo = N; g.m_valid = M;
    if (bm >= 256) {   // the persistent 256 x 256 kernel (gemm_fp8_p256.h); bm = 256 + v: only the first v rows are valid
        if (bm > 256) g.m_valid = bm - 256 < M ? bm - 256 : M;
        return launch_gemm256p8(reinterpret_cast<hipStream_t>(hip_stream), epi, g);
    }
    return launch_gemm8(reinterpret_cast<hipStream_t>(hip_stream), epi, bm, g);
}

extern "C" int mmiss_dbg_gemm8_xt(int device, void* hip_stream, int epi, int xt, const void* A8, const void* As, const void* W8,
                                  const float* wscale, const float* bias, void* out, void* out_scale, int32_t M, int32_t N, int32_t K,
                                  int32_t m_valid, const void* c16, const float* ln_stats, const void* x16, float ln_eps,
                                  void* q_out, void* q_scale, float* stats_out) {
    MM_TRY(mmiss_use_device(device));
    Gemm8Args g{};
    g.A = reinterpret_cast<const uint8_t*>(A8); g.As = reinterpret_cast<const uint8_t*>(As); g.ld_as = mx_scale_row_bytes(K);
    g.W = reinterpret_cast<const uint8_t*>(W8); g.wscale = wscale; g.bias = bias; g.out = out;
    g.out_scale = reinterpret_cast<uint8_t*>(out_scale); g.ld_os = mx_scale_row_bytes(N);
    g.M = M; g.N = N; g.K = K; g.ldo = N; g.m_valid = m_valid > 0 && m_valid < M ? m_valid : M;
    g.c16 = reinterpret_cast<const uint16_t*>(c16); g.ln_stats = ln_stats; g.x16 = reinterpret_cast<const uint16_t*>(x16); g.ln_eps = ln_eps;
    g.q_out = reinterpret_cast<uint8_t*>(q_out); g.q_scale = reinterpret_cast<uint8_t*>(q_scale); g.ld_qs = mx_scale_row_bytes(N);
    g.stats_out = stats_out;
    return launch_gemm256p8(reinterpret_cast<hipStream_t>(hip_stream), epi, g, xt);
}

extern "C" int mmiss_dbg_quant16_mxfp8_stats(int device, void* hip_stream, const void* x_bf16, void* out8, void* out_scale,
                                             float* stats, int32_t M, int32_t d) {
    if (!x_bf16 || !out8 || !out_scale || !stats || M <= 0 || d != 1024) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_quant16_mxfp8_stats: bad argument (d must be 1024)");
    MM_TRY(mmiss_use_device(device));
    hipLaunchKernelGGL(quant16_mxfp8_stats_1024_kernel, dim3((M + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream),
                       reinterpret_cast<const uint16_t*>(x_bf16), reinterpret_cast<uint8_t*>(out8), reinterpret_cast<uint8_t*>(out_scale),
                       stats, M, mx_scale_row_bytes(d));
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

extern "C" int mmiss_dbg_quantize_weights_fp8_csum(int device, void* hip_stream, const void* w_bf16, void* w8, float* scale, void* c16,
                                                   int32_t N, int32_t K) {
    if (!w_bf16 || !w8 || !scale || !c16 || N <= 0 || K <= 0 || (K % 4)) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_quantize_weights_fp8_csum: bad argument");
    MM_TRY(mmiss_use_device(device));
    hipLaunchKernelGGL(quantize_weights_fp8_csum_kernel, dim3((N + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream),
                       reinterpret_cast<const uint16_t*>(w_bf16), reinterpret_cast<uint8_t*>(w8), scale, reinterpret_cast<uint16_t*>(c16), N, K);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

extern "C" int mmiss_dbg_gemm8_time(int device, int epi, int bm, const void* A8, const void* As, const void* W8,
                                    const float* wscale, const float* bias, void* out, void* out_scale, int32_t M, int32_t N,
                                    int32_t K, int32_t iters, float* ms_per_launch) {
    if (!ms_per_launch || iters <= 0) MM_FAIL(MMISS_ERR_ARG, "mmiss_dbg_gemm8_time: bad argument");
    hipEvent_t e0, e1;
    MM_TRY(mmiss_use_device(device));
    MM_HIP(hipEventCreate(&e0));
    MM_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) MM_TRY(mmiss_dbg_gemm8(device, nullptr, epi, bm, A8, As, W8, wscale, bias, out, out_scale, M, N, K));
    MM_HIP(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) MM_TRY(mmiss_dbg_gemm8(device, nullptr, epi, bm, A8, As, W8, wscale, bias, out, out_scale, M, N, K));
    MM_HIP(hipEventRecord(e1, nullptr));
    MM_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    MM_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = ms / iters;
    return MMISS_OK;
}

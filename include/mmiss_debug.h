/*
 * mmiss_debug.h — kernel-level entry points of libmmiss.so used ONLY by tests/ and bench.py to check
 * and time single kernels in isolation. Not part of the drop-in boundary. Same pointer conventions
 * as mmiss.h, except that every pointer here must be a DEVICE pointer of `device` (the tests allocate
 * them with torch) and calls run on `hip_stream` (NULL = default stream) without synchronising.
 */
#ifndef MMISS_DEBUG_H
#define MMISS_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* epilogues of the bf16 MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (A, W bf16 row-major) */
enum {
    MMISS_EPI_F32 = 0,             /* out f32 [M,N] = acc                                           */
    MMISS_EPI_BIAS_BF16 = 1,       /* out bf16 [M,N] = acc + bias[n]            (K3 QKV)            */
    MMISS_EPI_BIAS_QGELU_BF16 = 2, /* out bf16 [M,N] = quick_gelu(acc + bias)   (K6 FC1)            */
    MMISS_EPI_BIAS_RESID_F32 = 3,  /* out f32 [M,N] += acc + bias[n]            (K5 out-proj, K7 FC2)*/
    MMISS_EPI_PATCH_F32 = 4        /* out f32 row (m/G)*T + 1 + m%G = acc + pos[1 + m%G][n] (K1)    */
};

/* M % 128 == 0, N % 128 == 0, K % 64 == 0. bias f32 [N] (or NULL), aux f32 (pos table for PATCH),
 * p0 = patches per image G, p1 = tokens per image T (PATCH only). variant selects the tile kernel
 * (0 = default). */
int mmiss_dbg_gemm(int device, void* hip_stream, int epi, int variant, const void* A, const void* W,
                   void* out, const float* bias, const float* aux, int32_t M, int32_t N, int32_t K,
                   int32_t p0, int32_t p1);

/* LayerNorm over the last dim: x f32 [M,d] -> out (bf16 if out_bf16 else f32) [M,d]; may be in place for f32 */
int mmiss_dbg_layernorm(int device, void* hip_stream, const float* x, const float* gamma, const float* beta,
                        void* out, int32_t out_bf16, int32_t M, int32_t d, float eps);

/* qkv bf16 [B*T, 3*H*64] -> ctx bf16 [B*T, H*64]; softmax(QK^T/8 (+causal)) V per (b, head) */
int mmiss_dbg_attention(int device, void* hip_stream, const void* qkv, void* ctx, int32_t B, int32_t T,
                        int32_t H, int32_t causal);

/* pixels f32 [B,3,S,S] -> patches bf16 [B*G*G, Kp] (k = c*P*P + ky*P + kx, zero padded to Kp) */
int mmiss_dbg_im2col(int device, void* hip_stream, const float* pixels, void* out, int32_t B, int32_t S,
                     int32_t P, int32_t Kp);

/* HIP-event time of `iters` back-to-back launches of one GEMM (ms per launch) */
int mmiss_dbg_gemm_time(int device, int epi, int variant, const void* A, const void* W, void* out,
                        const float* bias, const float* aux, int32_t M, int32_t N, int32_t K,
                        int32_t p0, int32_t p1, int32_t iters, float* ms_per_launch);

/* the persistent 256 x 256 tile GEMM (csrc/gemm_bf16_p256.h) in isolation: epi 1 / 2 = bias / bias + QuickGELU -> bf16,
 * 7 / 8 = the same behind a LayerNorm folded into W (ln_stats f32 [M][K/64][2] partial (sum, sumsq) per 64 columns,
 * aux = c [N], bias = b' [N]). M % 256 == 0 (rows >= m_valid land in row M - 1), N % 256 == 0, K % 256 == 0.
 * iters > 0 and ms_per_launch != NULL: HIP-event time of `iters` back-to-back launches. */
int mmiss_dbg_gemm_p256(int device, void* hip_stream, int epi, const void* A, const void* W, void* out,
                        const float* bias, const float* aux, const float* ln_stats, float ln_eps, int32_t M, int32_t N,
                        int32_t K, int32_t m_valid, int32_t iters, float* ms_per_launch);

/* the residual GEMM on a bf16 residual stream in isolation: out (bf16 [M,N], IN PLACE) = bf16(f32(out) + A W^T + bias),
 * stats_out (optional) f32 [M][N/64][2] = (sum, sumsq) of the new rows per 64 columns. variant 0 = the 160 x 256 tile on the
 * staggered loop (csrc/gemm_bf16_p160.h: M % 160 == 0, N % 256 == 0, K % 128 == 0), 128 / 160 / 192 = the 128-column kernel of
 * that tile height (M a multiple of it). Rows >= m_valid: untouched, except row M - 1 under variant 0. iters > 0 and
 * ms_per_launch != NULL: HIP-event time of `iters` back-to-back launches (the stream keeps accumulating). */
int mmiss_dbg_gemm_resid16(int device, void* hip_stream, int variant, const void* A, const void* W, void* out,
                           const float* bias, float* stats_out, int32_t M, int32_t N, int32_t K, int32_t m_valid,
                           int32_t iters, float* ms_per_launch);

/* record the residual stream after every layer during encode calls (for mmiss_encoder_tap 0..L) */
struct mmiss_encoder;
int mmiss_dbg_encoder_record_taps(struct mmiss_encoder* enc, int on);
/* how LayerNorm1/2 reach the QKV / FC1 GEMMs: -1 (default) automatic by rows per call, 0 separate LayerNorm kernels,
 * 1 normalised during operand staging, 2 folded algebraically into weights + epilogue (A/B and parity tests) */
int mmiss_dbg_encoder_set_fuse_ln(struct mmiss_encoder* enc, int on);

/* experiment: full GEMM vs two half-M GEMMs (same stream / two streams); ms[3] per GEMM-equivalent */
int mmiss_dbg_gemm_split_time(int device, int epi, int bm, const void* A, const void* W, void* out, const float* bias,
                              int32_t M, int32_t N, int32_t K, int32_t iters, float* ms);

/* fp8 path in isolation (gemm_fp8.h). Scale arrays use the permuted E8M0 layout: 16 * ceil(K / 512) bytes per row, the
 * scale of k-block b = k / 32 at byte (b / 16) * 16 + (b % 4) * 4 + (b / 4) % 4. */
int mmiss_dbg_quantize_weights_fp8(int device, void* hip_stream, const void* w_bf16, void* w8, float* scale, int32_t N,
                                   int32_t K);
int mmiss_dbg_layernorm_mxfp8(int device, void* hip_stream, const float* x, const float* gamma, const float* beta,
                              void* out8, void* out_scale, int32_t M, int32_t d, float eps);
/* the same from bf16 rows (the bf16 residual stream of the large calls) */
int mmiss_dbg_layernorm16_mxfp8(int device, void* hip_stream, const void* x_bf16, const float* gamma, const float* beta,
                                void* out8, void* out_scale, int32_t M, int32_t d, float eps);
/* attention whose output leaves the kernel as MXFP8 (the fp8 out-projection's A operand): qkv bf16 [B*T, 3*H*64] -> ctx8 e4m3
 * [B*T, H*64] + ctx_scale (permuted E8M0, 16 * ceil(H*64 / 512) bytes per row); non-causal, 1 <= T <= 288 (round 6: the one-pass
 * kernels for T <= 128 too) */
int mmiss_dbg_attention_mx(int device, void* hip_stream, const void* qkv, void* ctx8, void* ctx_scale, int32_t B, int32_t T,
                           int32_t H);
/* epi: 0 out bf16 = acc * wscale[n] + bias[n]; 1 out e4m3 + out_scale = mx(quick_gelu(.)); 2 out f32 += . ; bm = 128 | 160 | 192 */
int mmiss_dbg_gemm8(int device, void* hip_stream, int epi, int bm, const void* A8, const void* As, const void* W8,
                    const float* wscale, const float* bias, void* out, void* out_scale, int32_t M, int32_t N, int32_t K);
/* the persistent fp8 GEMM's extensions of round 5 (gemm_fp8_p256.h; M % 256 == 0, rows >= m_valid are padding):
 * xt = 1 (epi 0 / 1, K = 1024): LayerNorm folded in — A8 / As = the RAW bf16 rows x16 [M, K] as MXFP8, W8 the gamma-folded weights,
 *        bias = b', c16 = f16 [N] row sums of the dequantised W8, ln_stats = f32 [M][4][2] (sum, sumsq) per 256-column quarter of
 *        the rows of x16: out = rstd (acc wscale - mean c) + b' (then QuickGELU -> MXFP8 for epi 1);
 * xt = 2 (epi 3, N = 1024): out bf16 += ..., and the new rows also as MXFP8 (q_out e4m3 [M, N], q_scale permuted E8M0) with
 *        stats_out f32 [M][N/256][2] of the tile rows (the rows of a ragged last block get none). */
int mmiss_dbg_gemm8_xt(int device, void* hip_stream, int epi, int xt, const void* A8, const void* As, const void* W8,
                       const float* wscale, const float* bias, void* out, void* out_scale, int32_t M, int32_t N, int32_t K,
                       int32_t m_valid, const void* c16, const float* ln_stats, const void* x16, float ln_eps, void* q_out,
                       void* q_scale, float* stats_out);
/* bf16 rows [M, 1024] -> MXFP8 (raw) + (sum, sumsq) per 256-column quarter, f32 [M][4][2]: the entry of the folded fp8 mode */
int mmiss_dbg_quant16_mxfp8_stats(int device, void* hip_stream, const void* x_bf16, void* out8, void* out_scale, float* stats,
                                  int32_t M, int32_t d);
/* e4m3 codes + per-channel scales of bf16 weights [N, K] and c16 = f16 row sums of the dequantised codes */
int mmiss_dbg_quantize_weights_fp8_csum(int device, void* hip_stream, const void* w_bf16, void* w8, float* scale, void* c16,
                                        int32_t N, int32_t K);
int mmiss_dbg_gemm8_time(int device, int epi, int bm, const void* A8, const void* As, const void* W8, const float* wscale,
                         const float* bias, void* out, void* out_scale, int32_t M, int32_t N, int32_t K, int32_t iters,
                         float* ms_per_launch);

/* bit 0: the library was built with MMISS_EXPERIMENTS (A/B variants kept for timing; the product build rejects them with
 * MMISS_ERR_UNSUPPORTED). bit 1: one of its translation units was built with a timing-experiment macro of tools/(name)_ab.sh
 * (P256_NO_LATE_WAIT, P256_SPLIT_STAGE, P256_STAGE_FIRST, P256_A_POLICY, P256_W_POLICY, P256_PRIO, MMISS_SCAN_NT, Q256_STAGE_MID) — NOT a product build: mmiss_amd._lib.load() refuses it unless
 * MMISS_ALLOW_AB_BUILD=1, so that no test or bench run is attributed to HEAD by accident. The product build returns 0. */
int mmiss_dbg_build_flags(void);

/* process-wide integer tuning knob (A/B experiments from tools/): e.g. "scan_group" = 8 | 16 */
int mmiss_dbg_set_option(const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif

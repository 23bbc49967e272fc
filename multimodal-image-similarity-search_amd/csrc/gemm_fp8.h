// gemm_fp8.h — block-scaled fp8 (OCP e4m3, "MXFP8") MFMA GEMM:  C[M,N] = A8[M,K] * W8[N,K]^T  with fused epilogues.
//
// The fp8 encode path of BASELINE.json configs[4] ("ViT-L/14 fp8 MFMA encode"): the QKV, FC1 and FC2 projections of the
// CLIP towers (HF:modeling_clip.py:293-296,338-350) on v_mfma_scale_f32_16x16x128_f8f6f4 — the only fp8 MFMA form that
// runs above the bf16 rate on gfx950 (2 x bf16 per clock; the non-scaled fp8 MFMA runs AT the bf16 rate,
// MI355X_MICROARCH.md "Matrix cores").
//
// Number format. Activations are MXFP8: e4m3 bytes plus one E8M0 (power-of-two) scale per (row, 32 consecutive k), which
// the instruction applies in hardware. Weights are e4m3 with one f32 scale per OUTPUT CHANNEL (amax / 448), applied in the
// epilogue (their hardware block scales are 1). fp8 keeps 3 mantissa bits at any magnitude, so a power-of-two scale
// loses nothing against an exact one; what the block scale buys is that the producer of an activation only needs the
// maximum over the 32-64 columns it holds itself (no row-wide reduction across workgroups).
//
// Layouts. A8 [M,K] / W8 [N,K] bytes, K contiguous (nn.Linear is [out,in]). A lane's fragment of the 16x16x128 MFMA is 32
// bytes of one row: lane group g = lane>>4 holds k = 16g..16g+15 in registers 0-3 and k = 64+16g..64+16g+15 in registers
// 4-7 (both operands alike); the E8M0 scale of k-block c = k/32 is read from lane group c. Activation scales As [M][16 * ceil(K/512)] bytes: the scale of
// (row, k-block b = k/32) sits at byte (b/16)*16 + (b%4)*4 + (b/4)%4, i.e. a 4x4 transpose inside each group of 16 blocks,
// so that ONE dword per lane = the scales of its lane group for 4 consecutive 128-wide K-tiles (selected by OPSEL).
//
// Accumulation (measured, tools/fp8_probe.py): inside one instruction the 128 products of a row are aligned to the largest
// and summed with ~13 significant bits (448 + 511 * 2^-6 -> 455.875, not 455.984; worst |error| / sum|terms| 1.5e-4),
// between instructions in f32. That is an order of magnitude below the e4m3 rounding of the operands themselves.
//
// Structure = gemm_bf16.h's BM x 128 tile: a 128-element fp8 K-tile is the same 128 bytes per row as a 64-element bf16
// K-tile, so the LDS image, the global_load_lds staging, the XOR swizzle and the per-K-tile LDS traffic are identical;
// each K-tile now carries twice the k and its 4 x JT MFMAs take the cycles of the 8 x JT bf16 ones.
#pragma once
#include "common.h"
#include "gemm_bf16.h"

typedef __attribute__((ext_vector_type(8))) int v8i32;

#define MMISS_EPI8_BIAS_BF16 0        // out bf16 [M,N] = acc * sw[n] + bias[n]                          (QKV)
#define MMISS_EPI8_QGELU_MXFP8 1      // out e4m3 [M,N] + E8M0 scales = mx(quick_gelu(acc * sw[n] + bias[n]))  (FC1)
#define MMISS_EPI8_BIAS_RESID_F32 2   // out f32 [M,N] += acc * sw[n] + bias[n]                          (FC2)
#define MMISS_EPI8_BIAS_RESID_BF16 3  // out bf16 [M,N] = bf16(f32(out) + acc * sw[n] + bias[n])         (FC2, bf16 residual stream)

struct Gemm8Args {
    const uint8_t* A;       // e4m3 [M, K]
    const uint8_t* As;      // E8M0 [M, lds_as] permuted (see header)
    const uint8_t* W;       // e4m3 [N, K]
    const float* wscale;    // [N]
    const float* bias;      // [N]
    void* out;              // bf16 / e4m3 / f32 [M, ldo]
    uint8_t* out_scale;     // QGELU_MXFP8: E8M0 [M, ld_os] permuted for the NEXT GEMM (whose K = this N)
    int M, N, K, ldo, m_valid;
    int ld_as, ld_os;       // bytes per row of As / out_scale
    int m_fast;
    // gemm256p8_kernel only: the last 256-row block holds `ragged` <= 128 valid rows (0: none). They are computed by a
    // register-streamed pass of all workgroups in front of the tile stream; the tiles then cover M - 256 rows.
    int ragged;
    // gemm256p8_kernel<EPI, XT> only (gemm_fp8_p256.h). XT = 1, LayerNorm FOLDED into the BIAS_BF16 / QGELU_MXFP8 GEMMs (K = 1024):
    // A8 / As are the RAW residual rows as MXFP8, W8 the gamma-folded weights, bias = b', c16 = f16 [N] row sums of the
    // dequantised folded weights, ln_stats = [M][4][2] f32 (sum, sumsq) of every 256-column quarter of the bf16 residual row;
    // y = rstd (acc sw - mean c) + b'. x16 = those bf16 rows [M, K] (the ragged pass takes its rows' statistics from them).
    const uint16_t* c16;
    const float* ln_stats;
    const uint16_t* x16;
    float ln_eps;
    // XT = 2, BIAS_RESID_BF16 that ALSO leaves the new bf16 rows as MXFP8 (q_out e4m3 [M, N], q_scale E8M0 [M, ld_qs] permuted)
    // and their statistics stats_out [M][N/256][2] (tile rows only: a ragged block's rows get theirs in the consumer)
    uint8_t* q_out;
    uint8_t* q_scale;
    float* stats_out;
    int ld_qs;
    int stagger;            // EXPERIMENTS builds only (option gemm_p256_stagger): odd workgroups sleep `stagger` x ~4 us before they start
};

// bytes per row of a permuted scale array for K columns
static inline int mx_scale_row_bytes(int K) { return 16 * ((K + 511) / 512); }
// byte offset of k-block b (= k / 32) inside a permuted scale row
__host__ __device__ __forceinline__ int mx_scale_offset(int b) { return (b >> 4) * 16 + (b & 3) * 4 + ((b >> 2) & 3); }

// E8M0 byte e such that |amax| * 2^-(e-127) <= 448 (the largest e4m3), and the multiplier 2^-(e-127)
__device__ __forceinline__ void mx_scale_of(float amax, int& e8, float& inv) {
    const float r = fmaxf(amax, 1e-30f) * (1.0f / 448.0f);
    const uint32_t u = __float_as_uint(r);
    int e = (int)(u >> 23) + ((u & 0x7FFFFFu) != 0);  // ceil(log2(r)) + 127
    e = e < 1 ? 1 : (e > 254 ? 254 : e);
    e8 = e;
    inv = __uint_as_float((uint32_t)(254 - e) << 23);  // 2^(127 - e)
}

// 4 floats -> 4 e4m3 bytes (round to nearest even; callers keep |v| <= 448)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    int p = 0;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (uint32_t)p;
}

template <int OPSEL>
__device__ __forceinline__ f32x4 mma8(v8i32 w, v8i32 a, f32x4 c, int a_scales) {
    // A operand = weight fragment (unit block scales), B operand = activation fragment (scale byte OPSEL of a_scales)
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, c, 0, 0, 0, 0x7F7F7F7F, OPSEL, a_scales);
}

template <int BM, int EPI>
__global__ __launch_bounds__(256, 2) void gemm8_kernel(Gemm8Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN = 128, JT = BM / 32;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, BUF = A_BYTES + W_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbm = g.M / BM, nbn = g.N / BN;
    const int wg = xcd_remap(blockIdx.x, nbm * nbn);
    int bm, bn;
    tile_order(wg, nbm, nbn, g.m_fast, bm, bn);
    const uint8_t* Ab = g.A + (size_t)bm * BM * g.K;
    const uint8_t* Wb = g.W + (size_t)bn * BN * g.K;

    // staging: one wave-instruction = 8 rows x 128 B; slot p of row r holds global chunk p ^ (r & 7) (gemm_bf16.h)
    const int r_in = lane >> 3, p = lane & 7;
    const int src_chunk = (p ^ r_in) * 16;  // bytes
    auto stage = [&](int buf, int kt) {
        char* sA = smem + buf * BUF;
        char* sW = sA + A_BYTES;
        const size_t koff = (size_t)kt * 128 + src_chunk;
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            const int rowblk = wave * (BM / 32) + i;
            glds16(Ab + (size_t)(rowblk * 8 + r_in) * g.K + koff, sA + rowblk * 1024);
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int rowblk = wave * (BN / 32) + i;
            glds16(Wb + (size_t)(rowblk * 8 + r_in) * g.K + koff, sW + rowblk * 1024);
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[4][JT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = g.K / 128;
    const int nq = (nt + 3) >> 2;
    // activation scales of this lane's rows: one dword per 16-row fragment and group of 4 K-tiles
    const uint8_t* sbase = g.As + (size_t)(bm * BM + wm * (BM / 2) + fr) * g.ld_as + fg * 4;
    int sc_cur[JT], sc_nxt[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        sc_cur[j] = *reinterpret_cast<const int*>(sbase + (size_t)j * 16 * g.ld_as);
        sc_nxt[j] = sc_cur[j];
    }

    auto frag = [&](const char* base, int row) -> v8i32 {
        // registers 0-3 = k 16g .. 16g+15, registers 4-7 = k 64+16g .. 64+16g+15 of the 128-wide K-tile (the instruction is
        // two K = 64 halves); hardware k-block c = k 32c .. 32c+31 takes its scale from lane group c (measured:
        // tools/fp8_probe.py — with 32 contiguous k per lane the data was right and the scale association wrong)
        const u32x4 lo = *reinterpret_cast<const u32x4*>(base + row * 128 + ((fg ^ (row & 7)) << 4));
        const u32x4 hi = *reinterpret_cast<const u32x4*>(base + row * 128 + (((4 + fg) ^ (row & 7)) << 4));
        v8i32 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return v;
    };
    auto mma_tile = [&](int buf, auto opsel_tag) {
        constexpr int OPSEL = decltype(opsel_tag)::value;
        const char* sA = smem + buf * BUF;
        const char* sW = sA + A_BYTES;
        v8i32 wf[4], af[JT];
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = frag(sW, wn * 64 + i * 16 + fr);
#pragma unroll
        for (int j = 0; j < JT; ++j) af[j] = frag(sA, wm * (BM / 2) + j * 16 + fr);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < JT; ++j) acc[i][j] = mma8<OPSEL>(wf[i], af[j], acc[i][j], sc_cur[j]);
    };

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) {  // next group's scales fly during this group's four K-tiles
#pragma unroll
            for (int j = 0; j < JT; ++j)
                sc_nxt[j] = *reinterpret_cast<const int*>(sbase + (size_t)j * 16 * g.ld_as + (size_t)(q + 1) * 16);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kt = 4 * q + t;
            if (kt < nt) {
                const int cur = kt & 1;  // (t & 1: four tiles per group)
                if (kt + 1 < nt) stage(cur ^ 1, kt + 1);
                if (t == 0) mma_tile(cur, std::integral_constant<int, 0>{});
                else if (t == 1) mma_tile(cur, std::integral_constant<int, 1>{});
                else if (t == 2) mma_tile(cur, std::integral_constant<int, 2>{});
                else mma_tile(cur, std::integral_constant<int, 3>{});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
#pragma unroll
        for (int j = 0; j < JT; ++j) sc_cur[j] = sc_nxt[j];
    }

    // ---------------------------------------------------------------- epilogue (staging buffers are dead: per-wave patches)
    // acc[i][j][r] = C[m = m0 + j*16 + fr][n = n0 + i*16 + 4*fg + r]
    const int m0 = bm * BM + wm * (BM / 2), n0 = bn * BN + wn * 64;
    char* patch = smem + wave * EPI_PATCH_BYTES;
    const int rrow = lane >> 3, rchunk = lane & 7;
    f32x4 sw[4], bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sw[i] = *reinterpret_cast<const f32x4*>(g.wscale + n0 + i * 16 + 4 * fg);
        bias[i] = *reinterpret_cast<const f32x4*>(g.bias + n0 + i * 16 + 4 * fg);
    }
    f32x4 resid[EPI == MMISS_EPI8_BIAS_RESID_F32 ? JT : 1][2][2];
    if constexpr (EPI == MMISS_EPI8_BIAS_RESID_F32) {  // all residual rows fetched up front (see gemm_bf16.h)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch)
                    resid[j][rh][ch] = *reinterpret_cast<const f32x4*>(
                        reinterpret_cast<const float*>(g.out) + (size_t)(m0 + j * 16 + rh * 8 + rrow) * g.ldo + n0 + ch * 32 + rchunk * 4);
    }
    u32x4 resid16[EPI == MMISS_EPI8_BIAS_RESID_BF16 ? JT : 1][2];  // 8 consecutive columns of 8 rows per 16-row sub-tile
    if constexpr (EPI == MMISS_EPI8_BIAS_RESID_BF16) {
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
                resid16[j][rh] = *reinterpret_cast<const u32x4*>(
                    reinterpret_cast<const uint16_t*>(g.out) + (size_t)(m0 + j * 16 + rh * 8 + rrow) * g.ldo + n0 + rchunk * 8);
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        if constexpr (EPI == MMISS_EPI8_BIAS_BF16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 y = acc[i][j] * sw[i] + bias[i];
                u32x2 pk;
                pk[0] = pack_bf16x2(y[0], y[1]);
                pk[1] = pack_bf16x2(y[2], y[3]);
                *reinterpret_cast<u32x2*>(patch + fr * 144 + i * 32 + fg * 8) = pk;
            }
        } else if constexpr (EPI == MMISS_EPI8_QGELU_MXFP8) {
            // this lane: 16 of the 64 columns of row fr; the row's other 48 sit in lanes fr+16, fr+32, fr+48
            f32x4 y[4];
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                y[i] = acc[i][j] * sw[i] + bias[i];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    y[i][r] = quick_gelu(y[i][r]);
                    amax = fmaxf(amax, fabsf(y[i][r]));
                }
            }
            amax = fmaxf(amax, __shfl_xor(amax, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            int e8;
            float inv;
            mx_scale_of(amax, e8, inv);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<uint32_t*>(patch + fr * 80 + i * 16 + fg * 4) =
                    pack_fp8x4(y[i][0] * inv, y[i][1] * inv, y[i][2] * inv, y[i][3] * inv);
            const int m = m0 + j * 16 + fr;
            if (fg == 0 && m < g.m_valid) {  // the wave's 64 columns = k-blocks n0/32 and n0/32 + 1 of the next GEMM
                uint8_t* so = g.out_scale + (size_t)m * g.ld_os;
                so[mx_scale_offset(n0 >> 5)] = (uint8_t)e8;
                so[mx_scale_offset((n0 >> 5) + 1)] = (uint8_t)e8;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<f32x4*>(patch + fr * 272 + i * 64 + fg * 16) = acc[i][j] * sw[i] + bias[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (EPI == MMISS_EPI8_BIAS_BF16) {
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int row = rh * 8 + rrow, m = m0 + j * 16 + row;
                const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * 144 + rchunk * 16);
                if (m < g.m_valid)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(g.out) + (size_t)m * g.ldo + n0 + rchunk * 8) = v;
            }
        } else if constexpr (EPI == MMISS_EPI8_QGELU_MXFP8) {
            // 16 rows x 64 bytes: 4 lanes x 16 B per row, all 16 rows in one instruction
            const int row = lane >> 2, c16 = lane & 3, m = m0 + j * 16 + row;
            const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * 80 + c16 * 16);
            if (m < g.m_valid)
                *reinterpret_cast<u32x4*>(reinterpret_cast<uint8_t*>(g.out) + (size_t)m * g.ldo + n0 + c16 * 16) = v;
        } else if constexpr (EPI == MMISS_EPI8_BIAS_RESID_BF16) {
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int row = rh * 8 + rrow, m = m0 + j * 16 + row;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + row * 272 + rchunk * 32);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + row * 272 + rchunk * 32 + 16);
                const u32x4 old = resid16[j][rh];
                u32x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = __uint_as_float(old[e] << 16) + (e < 2 ? lo[2 * e] : hi[2 * e - 4]);
                    const float b = __uint_as_float(old[e] & 0xFFFF0000u) + (e < 2 ? lo[2 * e + 1] : hi[2 * e - 3]);
                    pk[e] = pack_bf16x2(a, b);
                }
                if (m < g.m_valid)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(g.out) + (size_t)m * g.ldo + n0 + rchunk * 8) = pk;
            }
        } else {
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int row = rh * 8 + rrow, m = m0 + j * 16 + row;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * 272 + ch * 128 + rchunk * 16);
                    if (m < g.m_valid)
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (size_t)m * g.ldo + n0 + ch * 32 + rchunk * 4) =
                            resid[j][rh][ch] + v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next j
    }
}

template <int BM, int EPI>
static int launch_gemm8_inst(hipStream_t st, const Gemm8Args& g) {
    constexpr int LDS = 2 * (BM + 128) * 128;
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm8_kernel<BM, EPI>), LDS));
    const int nwg = (g.M / BM) * (g.N / 128);
    hipLaunchKernelGGL((gemm8_kernel<BM, EPI>), dim3(nwg), dim3(256), LDS, st, g);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// g.M must be a multiple of bm (128 / 160 / 192), N of 128, K of 128; rows [m_valid, M) of A8 / As must be readable.
static int launch_gemm8(hipStream_t st, int epi, int bm, Gemm8Args g) {
    if (bm == 0) bm = 128;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.M % bm) || (g.N % 128) || (g.K % 128) || !g.A || !g.As || !g.W || !g.wscale ||
        !g.bias || !g.out || g.ld_as < mx_scale_row_bytes(g.K))
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm8: M=%d N=%d K=%d bm=%d ld_as=%d", g.M, g.N, g.K, bm, g.ld_as);
    if (epi == MMISS_EPI8_QGELU_MXFP8 && (!g.out_scale || g.ld_os < mx_scale_row_bytes(g.N)))
        MM_FAIL(MMISS_ERR_ARG, "gemm8: the MXFP8 epilogue needs out_scale with >= %d bytes per row", mx_scale_row_bytes(g.N));
    if (g.m_fast == 0 && g.N / 128 >= 8) g.m_fast = mmiss_option("gemm_band", 5);  // banded tile order, as gemm_bf16.h
    static const char* names[] = {"gemm_fp8_bias", "gemm_fp8_qgelu_mx", "gemm_fp8_bias_resid", "gemm_fp8_bias_resid16"};
    if (epi < 0 || epi > 3) MM_FAIL(MMISS_ERR_ARG, "gemm8: bad epilogue %d", epi);
    const int mv = g.m_valid < g.M ? g.m_valid : g.M;
    const double out_b = epi == MMISS_EPI8_BIAS_BF16 ? 2.0 : (epi == MMISS_EPI8_QGELU_MXFP8 ? 1.0 : (epi == MMISS_EPI8_BIAS_RESID_BF16 ? 4.0 : 8.0));
    MM_PROF(names[epi], st, 2.0 * mv * (double)g.N * g.K, (double)mv * g.K + (double)g.N * g.K + out_b * mv * g.N);
#define GEMM8_CASE(BMV)                                                                                    \
    case BMV:                                                                                              \
        if (epi == MMISS_EPI8_BIAS_BF16) return launch_gemm8_inst<BMV, MMISS_EPI8_BIAS_BF16>(st, g);       \
        if (epi == MMISS_EPI8_QGELU_MXFP8) return launch_gemm8_inst<BMV, MMISS_EPI8_QGELU_MXFP8>(st, g);   \
        if (epi == MMISS_EPI8_BIAS_RESID_BF16) return launch_gemm8_inst<BMV, MMISS_EPI8_BIAS_RESID_BF16>(st, g); \
        return launch_gemm8_inst<BMV, MMISS_EPI8_BIAS_RESID_F32>(st, g);
    switch (bm) {
        GEMM8_CASE(128)
        GEMM8_CASE(160)
        GEMM8_CASE(192)
    }
#undef GEMM8_CASE
    MM_FAIL(MMISS_ERR_ARG, "gemm8: unsupported tile height %d", bm);
}

// ------------------------------------------------------------------------------------------------
// Weight quantisation (once, when the fp8 path is switched on): W bf16 [N,K] -> e4m3 [N,K] + scale[n] = amax_n / 448.
// One wave per output channel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_weights_fp8_kernel(const uint16_t* __restrict__ Wb, uint8_t* __restrict__ W8,
                                                                   float* __restrict__ scale, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint16_t* w = Wb + (size_t)n * K;
    float amax = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(w + k);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf16_bits_to_f32(v[0] & 0xffff)), fabsf(bf16_bits_to_f32(v[0] >> 16))),
                                 fmaxf(fabsf(bf16_bits_to_f32(v[1] & 0xffff)), fabsf(bf16_bits_to_f32(v[1] >> 16)))));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    for (int k = lane * 4; k < K; k += 256) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(w + k);
        // min(): x * (1 / sc) may land one ulp above 448 for the row maximum itself
        const float a = fminf(fmaxf(bf16_bits_to_f32(v[0] & 0xffff) * inv, -448.f), 448.f);
        const float b = fminf(fmaxf(bf16_bits_to_f32(v[0] >> 16) * inv, -448.f), 448.f);
        const float c = fminf(fmaxf(bf16_bits_to_f32(v[1] & 0xffff) * inv, -448.f), 448.f);
        const float d = fminf(fmaxf(bf16_bits_to_f32(v[1] >> 16) * inv, -448.f), 448.f);
        *reinterpret_cast<uint32_t*>(W8 + (size_t)n * K + k) = pack_fp8x4(a, b, c, d);
    }
    if (lane == 0) scale[n] = sc;
}

// The same for weights that carry a folded LayerNorm gamma (W' = bf16(W gamma), fold_ln_weights_kernel): e4m3 codes + per-channel
// scale + c16[n] = f16(sum_k of the DEQUANTISED codes) — the row sum the folded fp8 GEMM's epilogue multiplies the row mean by
// (gemm_fp8_p256.h, XT = 1): of the quantised weights, because those are what the matrix cores multiply.
__global__ __launch_bounds__(256) void quantize_weights_fp8_csum_kernel(const uint16_t* __restrict__ Wb, uint8_t* __restrict__ W8,
                                                                        float* __restrict__ scale, uint16_t* __restrict__ c16, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint16_t* w = Wb + (size_t)n * K;
    float amax = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(w + k);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf16_bits_to_f32(v[0] & 0xffff)), fabsf(bf16_bits_to_f32(v[0] >> 16))),
                                 fmaxf(fabsf(bf16_bits_to_f32(v[1] & 0xffff)), fabsf(bf16_bits_to_f32(v[1] >> 16)))));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    float csum = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const u32x2 v = *reinterpret_cast<const u32x2*>(w + k);
        const float a = fminf(fmaxf(bf16_bits_to_f32(v[0] & 0xffff) * inv, -448.f), 448.f);
        const float b = fminf(fmaxf(bf16_bits_to_f32(v[0] >> 16) * inv, -448.f), 448.f);
        const float c = fminf(fmaxf(bf16_bits_to_f32(v[1] & 0xffff) * inv, -448.f), 448.f);
        const float d = fminf(fmaxf(bf16_bits_to_f32(v[1] >> 16) * inv, -448.f), 448.f);
        const uint32_t q = pack_fp8x4(a, b, c, d);
        *reinterpret_cast<uint32_t*>(W8 + (size_t)n * K + k) = q;
        // the codes' values, exactly (e4m3 -> f32): sums of at most a few thousand multiples of 2^-9 below 2^9 are exact in f32
        const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)q, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)q, true);
        csum += (lo[0] + lo[1]) + (hi[0] + hi[1]);
    }
    csum = wave_sum(csum);
    if (lane == 0) {
        scale[n] = sc;
        const _Float16 h = (_Float16)(csum * sc);
        c16[n] = __builtin_bit_cast(uint16_t, h);
    }
}

// Entry of the folded fp8 mode (hidden 1024): the bf16 residual rows as MXFP8 — RAW, no LayerNorm — and (sum, sumsq) of each
// 256-column quarter of every row, [M][4][2] f32: what the residual GEMMs' XT = 2 epilogue leaves behind from then on.
// One wave per row, lane l holds columns 16 l .. + 15 (two lanes = one 32-column block, 16 lanes = one quarter).
__global__ __launch_bounds__(256) void quant16_mxfp8_stats_1024_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ out,
                                                                       uint8_t* __restrict__ out_scale, float* __restrict__ stats, int M,
                                                                       int ld_os) {
    constexpr int D = 1024;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const int c = lane * 16;
    const u32x4 r0 = *reinterpret_cast<const u32x4*>(x + (size_t)r * D + c), r1 = *reinterpret_cast<const u32x4*>(x + (size_t)r * D + c + 8);
    float v[16];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        v[2 * w] = __uint_as_float(r0[w] << 16);
        v[2 * w + 1] = __uint_as_float(r0[w] & 0xFFFF0000u);
        v[8 + 2 * w] = __uint_as_float(r1[w] << 16);
        v[8 + 2 * w + 1] = __uint_as_float(r1[w] & 0xFFFF0000u);
    }
    float s1 = 0.f, s2 = 0.f, amax = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        s1 += v[e];
        s2 += v[e] * v[e];
        amax = fmaxf(amax, fabsf(v[e]));
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {   // the 16 lanes of a quarter, fixed order
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    amax = fmaxf(amax, __shfl_xor(amax, 1));
    int e8;
    float inv;
    mx_scale_of(amax, e8, inv);
    u32x4 pk;
#pragma unroll
    for (int i = 0; i < 4; ++i) pk[i] = pack_fp8x4(v[4 * i] * inv, v[4 * i + 1] * inv, v[4 * i + 2] * inv, v[4 * i + 3] * inv);
    *reinterpret_cast<u32x4*>(out + (size_t)r * D + c) = pk;
    if ((lane & 1) == 0) out_scale[(size_t)r * ld_os + mx_scale_offset(c >> 5)] = (uint8_t)e8;
    if ((lane & 15) == 0) {
        float* so = stats + ((size_t)r * 4 + (lane >> 4)) * 2;
        so[0] = s1;
        so[1] = s2;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm with MXFP8 output (K2 in front of the fp8 QKV / FC1 GEMMs): same statistics as layernorm_kernel
// (encoder_kernels.h; HF:modeling_clip.py:358-360), the normalised row quantised per 32 columns. One wave per row held in
// registers: lane l holds columns (i*64 + l)*4 .. +3, so 8 consecutive lanes hold one 32-column block.
// ------------------------------------------------------------------------------------------------
template <bool IN_BF16>  // IN_BF16: x is a bf16 residual stream (widened exactly)
__global__ __launch_bounds__(256) void layernorm_mxfp8_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, uint8_t* __restrict__ out,
                                                              uint8_t* __restrict__ out_scale, int M, int d, int ld_os,
                                                              float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if constexpr (IN_BF16) {
            u32x2 w = u32x2{0u, 0u};
            if (c < d) w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(x) + (size_t)r * d + c);
            v[i] = f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u), __uint_as_float(w[1] << 16),
                         __uint_as_float(w[1] & 0xFFFF0000u)};
        } else {
            v[i] = (c < d) ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + (size_t)r * d + c)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[i][e] - mean;
                q += t * t;
            }
        }
    }
    const float var = wave_sum(q) / (float)d;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;  // wave-uniform validity: d % 256 == 0 is not required, d % 32 == 0 is
        f32x4 y = {0.f, 0.f, 0.f, 0.f};
        if (c < d) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * gm[e] + bb[e];
        }
        float amax = fmaxf(fmaxf(fabsf(y[0]), fabsf(y[1])), fmaxf(fabsf(y[2]), fabsf(y[3])));
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        amax = fmaxf(amax, __shfl_xor(amax, 2));
        amax = fmaxf(amax, __shfl_xor(amax, 4));
        int e8;
        float inv;
        mx_scale_of(amax, e8, inv);
        if (c < d) {
            *reinterpret_cast<uint32_t*>(out + (size_t)r * d + c) = pack_fp8x4(y[0] * inv, y[1] * inv, y[2] * inv, y[3] * inv);
            if ((lane & 7) == 0) out_scale[(size_t)r * ld_os + mx_scale_offset(c >> 5)] = (uint8_t)e8;
        }
    }
}

// The same arithmetic for a bf16 residual stream with d = NS * 512 (ViT-L/14: 1024), wide accesses (round 4): a lane holds EIGHT
// consecutive columns per 512-column step (one 16-byte load, one 8-byte store; the first form moves 8 / 4 bytes per lane and
// re-reads gamma and beta — twice the row's own bytes — for every row), a wave walks four rows with gamma and beta in
// registers, all four rows' loads issued before the first is used. A 32-column block = 4 lanes: two xor-shuffles.
// Same statistics, same y, same scales and bytes as layernorm_mxfp8_kernel<true> (the sums run in a different lane order:
// the last bit of mean / rstd may differ).
// D = 768 (round 6: ViT-B/32's fp8 tower, 23 launches per bs-256 encode on the one-lane-four-columns form before): NS = 2 with the
// second step's lanes 32..63 idle (columns 768..1023 do not exist: zeros into the sums, nothing stored) — a quarter of the lane
// slots of a memory-bound kernel; a 32-column block never straddles the boundary (lane 32 = column 768).
template <int NS, int D = NS * 512>
__global__ __launch_bounds__(256) void layernorm16_mxfp8_wide_kernel(const uint16_t* __restrict__ x, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, uint8_t* __restrict__ out,
                                                                     uint8_t* __restrict__ out_scale, int M, int ld_os, float eps) {
    constexpr int RW = 4;
    static_assert(D <= NS * 512 && D > (NS - 1) * 512 && D % 32 == 0, "columns");
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (r0 >= M) return;
    float gm[NS][8], bb[NS][8];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int c = i * 512 + lane * 8;
        const bool live = (i + 1) * 512 <= D || c < D;
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 g0 = live ? *reinterpret_cast<const f32x4*>(gamma + c) : z, g1 = live ? *reinterpret_cast<const f32x4*>(gamma + c + 4) : z;
        const f32x4 b0 = live ? *reinterpret_cast<const f32x4*>(beta + c) : z, b1 = live ? *reinterpret_cast<const f32x4*>(beta + c + 4) : z;
#pragma unroll
        for (int e = 0; e < 4; ++e) { gm[i][e] = g0[e]; gm[i][4 + e] = g1[e]; bb[i][e] = b0[e]; bb[i][4 + e] = b1[e]; }
    }
    u32x4 raw[RW][NS];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int r = r0 + j < M ? r0 + j : M - 1;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const bool live = (i + 1) * 512 <= D || i * 512 + lane * 8 < D;
            raw[j][i] = live ? *reinterpret_cast<const u32x4*>(x + (size_t)r * D + i * 512 + lane * 8) : u32x4{0u, 0u, 0u, 0u};
        }
    }
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int r = r0 + j;
        float v[NS][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                v[i][2 * w] = __uint_as_float(raw[j][i][w] << 16);
                v[i][2 * w + 1] = __uint_as_float(raw[j][i][w] & 0xFFFF0000u);
            }
            s += ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
        }
        const float mean = wave_sum(s) * (1.0f / (float)D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const bool live = (i + 1) * 512 <= D || i * 512 + lane * 8 < D;   // (a column that does not exist adds nothing to the variance)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = v[i][e] - mean;
                q += live ? t * t : 0.f;
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / (float)D) + eps);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            float y[8];
            float amax = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                y[e] = (v[i][e] - mean) * rstd * gm[i][e] + bb[i][e];
                amax = fmaxf(amax, fabsf(y[e]));
            }
            amax = fmaxf(amax, __shfl_xor(amax, 1));
            amax = fmaxf(amax, __shfl_xor(amax, 2));
            int e8;
            float inv;
            mx_scale_of(amax, e8, inv);
            const int c = i * 512 + lane * 8;
            if (r < M && ((i + 1) * 512 <= D || c < D)) {
                u32x2 pk;
                pk[0] = pack_fp8x4(y[0] * inv, y[1] * inv, y[2] * inv, y[3] * inv);
                pk[1] = pack_fp8x4(y[4] * inv, y[5] * inv, y[6] * inv, y[7] * inv);
                *reinterpret_cast<u32x2*>(out + (size_t)r * D + c) = pk;
                if ((lane & 3) == 0) out_scale[(size_t)r * ld_os + mx_scale_offset(c >> 5)] = (uint8_t)e8;
            }
        }
    }
}

// d = 1024 (ViT-L/14): SIXTEEN consecutive columns per lane — two 16-byte loads, ONE 16-byte store per row and lane (the
// 8-column form stores 8 bytes: half-width store instructions), a 32-column block = 2 lanes: one xor-shuffle.
// PACKS (A/B of round 5, option ln_mxfp8_wide = 5): the row's 32 scale bytes leave as eight dwords from eight lanes (four
// lane gathers per row) instead of 32 single-byte stores from 32 lanes
template <int RW, bool PACKS = false>   // RW: rows per wave, all of them loaded before the first is used (4: round 4; 8: A/B of round 5, option ln_mxfp8_wide = 3)
__global__ __launch_bounds__(256) void layernorm16_mxfp8_1024_kernel(const uint16_t* __restrict__ x, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, uint8_t* __restrict__ out,
                                                                     uint8_t* __restrict__ out_scale, int M, int ld_os, float eps) {
    constexpr int D = 1024;
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (r0 >= M) return;
    const int c = lane * 16;
    float gm[16], bb[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c + 4 * i), b = *reinterpret_cast<const f32x4*>(beta + c + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) { gm[4 * i + e] = g[e]; bb[4 * i + e] = b[e]; }
    }
    u32x4 raw[RW][2];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int r = r0 + j < M ? r0 + j : M - 1;
        raw[j][0] = *reinterpret_cast<const u32x4*>(x + (size_t)r * D + c);
        raw[j][1] = *reinterpret_cast<const u32x4*>(x + (size_t)r * D + c + 8);
    }
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int r = r0 + j;
        float v[16];
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                v[8 * h + 2 * w] = __uint_as_float(raw[j][h][w] << 16);
                v[8 * h + 2 * w + 1] = __uint_as_float(raw[j][h][w] & 0xFFFF0000u);
            }
            s += ((v[8 * h] + v[8 * h + 1]) + (v[8 * h + 2] + v[8 * h + 3])) + ((v[8 * h + 4] + v[8 * h + 5]) + (v[8 * h + 6] + v[8 * h + 7]));
        }
        const float mean = wave_sum(s) * (1.0f / (float)D);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float t = v[e] - mean;
            q += t * t;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / (float)D) + eps);
        float y[16];
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            y[e] = (v[e] - mean) * rstd * gm[e] + bb[e];
            amax = fmaxf(amax, fabsf(y[e]));
        }
        amax = fmaxf(amax, __shfl_xor(amax, 1));
        int e8;
        float inv;
        mx_scale_of(amax, e8, inv);
        if (r < M) {
            u32x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) pk[i] = pack_fp8x4(y[4 * i] * inv, y[4 * i + 1] * inv, y[4 * i + 2] * inv, y[4 * i + 3] * inv);
            *reinterpret_cast<u32x4*>(out + (size_t)r * D + c) = pk;
            if constexpr (!PACKS) {
                if ((lane & 1) == 0) out_scale[(size_t)r * ld_os + mx_scale_offset(c >> 5)] = (uint8_t)e8;
            }
        }
        if constexpr (PACKS) {
            // dword t = (group G = t >> 2, k-block c = t & 3) of the permuted row: bytes t' = 0..3 are the scales of column blocks
            // 16 G + 4 t' + c, held by lanes 2 (16 G + 4 t' + c) — gathered by lane t (wave-uniform control flow: every lane shuffles)
            const int t = lane & 7, srcl = 32 * (t >> 2) + 2 * (t & 3);
            uint32_t dw = 0;
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) dw |= (uint32_t)(__shfl(e8, srcl + 8 * tp) & 0xff) << (8 * tp);
            if (r < M && lane < 8) *reinterpret_cast<uint32_t*>(out_scale + (size_t)r * ld_os + 4 * lane) = dw;
        }
    }
}

static int launch_layernorm_mxfp8(hipStream_t st, const void* x, bool x_bf16, const float* gamma, const float* beta, uint8_t* out,
                                  uint8_t* out_scale, int M, int d, float eps) {
    if (d > 1024 || (d % 32)) MM_FAIL(MMISS_ERR_UNSUPPORTED, "layernorm_mxfp8: d=%d (need d <= 1024, d %% 32 == 0)", d);
    MM_PROF(x_bf16 ? "layernorm16_mxfp8" : "layernorm_mxfp8", st, 8.0 * M * d, (double)M * d * (x_bf16 ? 3 : 5));
    if (x_bf16 && (d == 512 || d == 768 || d == 1024) && mmiss_option("ln_mxfp8_wide", 1) != 0) {
        const int grid = (M + 15) / 16;
        if (d == 1024 && mmiss_option("ln_mxfp8_wide", 1) == 1)
            hipLaunchKernelGGL(layernorm16_mxfp8_1024_kernel<4>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(x), gamma,
                               beta, out, out_scale, M, mx_scale_row_bytes(d), eps);
        else if (d == 1024 && mmiss_option("ln_mxfp8_wide", 1) == 5)
            hipLaunchKernelGGL((layernorm16_mxfp8_1024_kernel<4, true>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(x), gamma,
                               beta, out, out_scale, M, mx_scale_row_bytes(d), eps);
        else if (d == 1024 && mmiss_option("ln_mxfp8_wide", 1) == 3)
            hipLaunchKernelGGL(layernorm16_mxfp8_1024_kernel<8>, dim3((M + 31) / 32), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(x), gamma,
                               beta, out, out_scale, M, mx_scale_row_bytes(d), eps);
        else if (d == 1024)   // (option ln_mxfp8_wide = 2: the 8-columns-per-lane form, A/B)
            hipLaunchKernelGGL(layernorm16_mxfp8_wide_kernel<2>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(x), gamma,
                               beta, out, out_scale, M, mx_scale_row_bytes(d), eps);
        else if (d == 768)
            hipLaunchKernelGGL((layernorm16_mxfp8_wide_kernel<2, 768>), dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(x), gamma,
                               beta, out, out_scale, M, mx_scale_row_bytes(d), eps);
        else
            hipLaunchKernelGGL(layernorm16_mxfp8_wide_kernel<1>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint16_t*>(x), gamma,
                               beta, out, out_scale, M, mx_scale_row_bytes(d), eps);
        MM_HIP(hipGetLastError());
        return MMISS_OK;
    }
    if (x_bf16)
        hipLaunchKernelGGL(layernorm_mxfp8_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, st, x, gamma, beta, out, out_scale, M, d,
                           mx_scale_row_bytes(d), eps);
    else
        hipLaunchKernelGGL(layernorm_mxfp8_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, st, x, gamma, beta, out, out_scale, M, d,
                           mx_scale_row_bytes(d), eps);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

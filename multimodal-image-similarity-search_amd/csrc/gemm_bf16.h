// gemm_bf16.h — 16-bit MFMA GEMM:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue), bf16 or f16 inputs.
//
// bf16: K1 (patch projection), K3 (QKV), K5 (out-proj + residual), K6 (FC1 + QuickGELU), K7 (FC2 + residual)
//       and K8's projection of SURVEY.md §2.2 — the nn.Linear / nn.Conv2d arithmetic of
//       HF:modeling_clip.py:148-154,293-296,338-350,674-675.
// f16 : K11's batched score pass (queries x corpus rows) with the GROUPMAX epilogue (retrieval_kernels.h).
//
// Both operands are K-contiguous (nn.Linear stores [out,in]; index rows are [row, dim]), which is exactly the
// per-lane fragment shape of v_mfma_f32_16x16x32_{bf16,f16} (8 consecutive k per lane): no transposes.
// The MFMA "A" operand is the W tile and the "B" operand the A (activation / query) tile: the accumulator
// then holds 4 consecutive n per lane for one m, so epilogue stores are 8-byte (bf16) / 16-byte (f32)
// vectors along the contiguous output dimension.
//
// Structure (guide §5 "minimum 2-phase"): BM x 128 x 64 tile (BM = 128/160/192, chosen per shape so the grid
// fills the 256 CUs x 2 resident blocks evenly), 4 waves (2x2), global_load_lds dwordx4 staging into a
// double-buffered, XOR-swizzled LDS image (swizzle on the SOURCE address and on the read, destination
// linear — guide §5.4 rule 21), one barrier per K-tile, 2 workgroups per CU so one block's staging wait
// overlaps the other's MFMAs.
#pragma once
#include "common.h"

struct GemmEpi {
    void* out;          // f32 or bf16, row stride ldo elements
    const float* bias;  // [N] or null
    const float* aux;   // PATCH: position table [T, N]
    int ldo;            // output row stride (elements)
    int m_valid;        // rows >= m_valid are not stored
    int p0, p1;         // PATCH: p0 = patches per image (G), p1 = tokens per image (T). GROUPMAX: p0 = valid n
    int m_fast;         // block order: 0 = n fastest (blocks sharing an A panel adjacent), 1 = m fastest
    int nt_out;         // bit 0: store the f32 residual rows non-temporally; bit 1: the bf16 outputs too
    // LayerNorm fused into the A operand (ALN kernels: A is the f32 residual stream, normalised while it is staged)
    const float* ln_stats;  // [M][ln_parts][2] partial (sum, sum of squares) of every row
    const float* ln_g;      // gamma [K]
    const float* ln_b;      // beta  [K]
    int ln_parts;
    float ln_eps;
    // BIAS_RESID_F32: if set, partial (sum, sumsq) of the NEW residual rows, [M][N/64][2] — the next LayerNorm's input
    float* stats_out;
    void* xb_out;           // BIAS_RESID_F32: if set, bf16 copy of the new residual rows [M, ldo] (next GEMM's A operand)
    int stats16;            // skinny kernels (gemm_skinny.h): stats_out / ln_stats hold one partial per SIXTEEN columns
    // split-K (launch_gemm only): f32 scratch for the partial products [splits][M][N]; null = never split
    float* splitk_ws;
    size_t splitk_ws_bytes;
    // LNFOLD epilogues of the persistent kernel, K > 768: (mean, rstd) of every row FINISHED, [M][2] (ln_finalize_kernel) —
    // the raw partials of a 256-row tile (K / 64 x 8 bytes per row) no longer fit beside the staging buffers
    const float* ln_final;
};

#define MMISS_EPI_GROUPMAX_F32 5  // internal: out f32 [M, N/16] = max over the lane's 16 n (see decode below)
// internal: LayerNorm folded into the GEMM algebraically. A = bf16(x) (raw residual rows), W' = bf16(W * gamma),
//   LN(x) W^T + b = rstd_m * (x W'^T - mean_m * c_n) + b'_n,   c_n = sum_k W'[n,k],  b'_n = b_n + sum_k beta_k W[n,k]
// (mean_m, rstd_m from ep.ln_stats; c in ep.aux; b' in ep.bias). 7: bf16 out, 8: + QuickGELU.
#define MMISS_EPI_LNFOLD_BF16 7
#define MMISS_EPI_LNFOLD_QGELU_BF16 8
// internal: the residual stream itself kept in bf16 (ep.out = bf16 [M, ldo], updated in place):
//   out = bf16( f32(out) + acc + bias ),  ep.stats_out = partial (sum, sumsq) of the ROUNDED new rows per 64 columns.
// The f32 stream of BIAS_RESID_F32 costs 78.6 MB of read-modify-write + a 19.7 MB bf16 copy per launch at 12800 x 768;
// this form moves 2 x 19.7 MB. Each add rounds the stream to 8 significant bits: 1 - cos vs the fp32 oracle goes from
// ~5e-6 to 3-8e-5 (tolerance 1e-3) — see DESIGN.md "bf16 residual stream".
#define MMISS_EPI_BIAS_RESID_BF16 9

#define GEMM_BN 128
#define GEMM_BK 64

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ float quick_gelu(float x) {
    // x * sigmoid(1.702 x) = x / (1 + 2^(-1.702 log2(e) x))  — HF:activations.py:117-123. One multiply, v_exp_f32 (= 2^x),
    // v_rcp_f32: the two transcendentals are what an epilogue of 128 values per lane costs (8 of its ~28 cycles per value each).
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930157f * x));
}

// XCD-aware, bijective block remap (guide §5 "XCD swizzle must be bijective"): blocks that share an
// XCD (equal bid % 8) get a contiguous run of tiles, so a shared operand panel is fetched into one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

// Logical tile id -> (bm, bn). mode 0: n fastest (blocks sharing an A panel adjacent); 1: m fastest; g >= 2: grouped,
// bands of g m-tiles, m fastest inside a band, so that the ~64 tiles an XCD runs at once cover about g x (64/g)
// tiles and touch g A panels + 64/g W panels instead of ~3 + nbn (less L2 working set per wave of tiles).
__device__ __forceinline__ void tile_order(int wg, int nbm, int nbn, int mode, int& bm, int& bn) {
    if (mode == 0) { bm = wg / nbn; bn = wg - bm * nbn; }
    else if (mode == 1) { bn = wg / nbm; bm = wg - bn * nbm; }
    else {
        const int per = mode * nbn;
        const int grp = wg / per, first = grp * mode;
        const int gsz = min(mode, nbm - first);
        const int in = wg - grp * per;
        bn = in / gsz; bm = first + (in - bn * gsz);
    }
}

// GROUPMAX group g <-> rows: g = ((bn*2 + wn)*4 + fg); member e (0..15) is row
//   bn*128 + wn*64 + 4*fg + (e >> 2)*16 + (e & 3)
__host__ __device__ __forceinline__ int64_t groupmax_row(int64_t g, int e) {
    const int64_t fg = g & 3, wn = (g >> 2) & 1, bn = g >> 3;
    return bn * 128 + wn * 64 + 4 * fg + (e >> 2) * 16 + (e & 3);
}

template <typename IN> struct MfmaIn;
template <> struct MfmaIn<__bf16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct MfmaIn<_Float16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// ------------------------------------------------------------------------------------------------
// Epilogue through LDS: the MFMA accumulator gives a lane 4 consecutive n of ONE row, so direct stores touch every
// 128-byte line in 32- or 64-byte pieces from 2-4 different instructions (measured: 2.3-2.5 TB/s of output for the
// bf16 epilogues, tools/gemm_ksweep.py: the fixed cost of a K=768 GEMM was 40 % of its time). Each wave instead
// transposes 16 rows at a time through a private 16 x 64 LDS patch (row stride padded by 16 B) and then moves
// whole 128-byte row segments: 8 consecutive lanes x 16 B per row, 8 rows per instruction.
// ------------------------------------------------------------------------------------------------
#define EPI_PATCH_BYTES (16 * 272)  // per wave: 16 rows x (64 f32 + 16 B pad); bf16 rows use 144 B of it

// (mean, rstd) of one row from its partial (sum, sum of squares) per 64 columns; the partial sums are contiguous, loaded as
// vectors and all issued before the adds
__device__ __forceinline__ void ln_row_stats(const GemmEpi& ep, int64_t row, float& mean, float& rstd) {
    const f32x4* st = reinterpret_cast<const f32x4*>(ep.ln_stats + (size_t)row * ep.ln_parts * 2);
    const int n4 = ep.ln_parts >> 1;  // hidden % 128 == 0 -> parts is even
    float s1 = 0.f, s2 = 0.f;
    f32x4 buf[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) buf[q] = (q < n4) ? st[q] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) { s1 += buf[q][0] + buf[q][2]; s2 += buf[q][1] + buf[q][3]; }
    const float kd = (float)(ep.ln_parts * 64);
    mean = s1 / kd;
    const float var = fmaxf(s2 / kd - mean * mean, 0.f);
    rstd = 1.0f / sqrtf(var + ep.ln_eps);
}

template <int EPI, int JT>
// stat_part / stat_parts (folded LayerNorm only): the stat_parts waves that share these rows (same wm) share ONE stat_area
// and each finalises 1/stat_parts of the rows' statistics; a workgroup barrier publishes them (all waves call the epilogue).
__device__ __forceinline__ void gemm_epilogue(const GemmEpi& ep, f32x4 (&acc)[4][JT], int m_wave, int n_wave, char* patch,
                                              int lane, char* stat_area = nullptr, int stat_part = 0, int stat_parts = 1,
                                              const float* pre_stats = nullptr,
                                              const f32x4 (*pre_resid)[2][2] = nullptr,
                                              const u32x4 (*pre_resid16)[2] = nullptr) {
    const int fr = lane & 15, fg = lane >> 4;
    const int rrow = lane >> 3, rchunk = lane & 7;  // read-back role: row (of 8) and 16-byte chunk (of 8)
    constexpr bool FOLD = (EPI == MMISS_EPI_LNFOLD_BF16 || EPI == MMISS_EPI_LNFOLD_QGELU_BF16);
    constexpr bool OUT_BF16 = (EPI == MMISS_EPI_BIAS_BF16 || EPI == MMISS_EPI_BIAS_QGELU_BF16 || FOLD);
    f32x4 bias[4], cvec[4];
    if constexpr (OUT_BF16 || EPI == MMISS_EPI_BIAS_RESID_F32 || EPI == MMISS_EPI_BIAS_RESID_BF16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bias[i] = *reinterpret_cast<const f32x4*>(ep.bias + n_wave + i * 16 + 4 * fg);
    }
    float* sstat = nullptr;
    if constexpr (FOLD) {
#pragma unroll
        for (int i = 0; i < 4; ++i) cvec[i] = *reinterpret_cast<const f32x4*>(ep.aux + n_wave + i * 16 + 4 * fg);
        // per-row (mean, rstd) of this wave's JT*16 rows from the partial sums, kept in LDS next to the patch
        sstat = reinterpret_cast<float*>(stat_area);
        const int per = (JT * 16 + stat_parts - 1) / stat_parts;
        const int r_end = (stat_part + 1) * per < JT * 16 ? (stat_part + 1) * per : JT * 16;
        if (pre_stats) {  // computed before the K loop (their load latency hidden behind it): lane -> row stat_part*per + lane
            const int r = stat_part * per + lane;
            if (lane < per && r < r_end) { sstat[2 * r] = pre_stats[0]; sstat[2 * r + 1] = pre_stats[1]; }
        } else {
            for (int r = stat_part * per + lane; r < r_end; r += 64) {
                float mean, rstd;
                ln_row_stats(ep, (int64_t)m_wave + r, mean, rstd);
                sstat[2 * r] = mean;
                sstat[2 * r + 1] = rstd;
            }
        }
        if (stat_parts > 1) {
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    // BIAS_RESID_F32: the residual rows this wave will add to, ALL fetched now (JT x 4 16-byte loads per lane in flight
    // while the first accumulator rows go through the transpose). Loaded where they are used they were JT x 4 dependent
    // round trips per wave — load, s_waitcnt vmcnt(0) (which also waits for the previous store), add, store — and that
    // chain, not bandwidth, was the 15-25 us "fixed cost" of the out-proj / FC2 launches. Rows up to the padded M exist
    // (the caller pads the residual stream), so only the stores are masked by m_valid.
    f32x4 resid[EPI == MMISS_EPI_BIAS_RESID_F32 ? JT : 1][2][2];
    if constexpr (EPI == MMISS_EPI_BIAS_RESID_F32) {
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch)
                    resid[j][rh][ch] = pre_resid ? pre_resid[j][rh][ch]   // (already fetched before the K loop)
                                                 : *reinterpret_cast<const f32x4*>(
                        reinterpret_cast<const float*>(ep.out) + (size_t)(m_wave + j * 16 + rh * 8 + rrow) * ep.ldo + n_wave +
                        ch * 32 + rchunk * 4);
    }
    // BIAS_RESID_BF16: lane (rrow, rchunk) owns 8 consecutive columns of 8 rows per 16-row sub-tile: one 16-byte load /
    // store per row half, whole 128-byte row segments per 8 lanes
    u32x4 resid16[EPI == MMISS_EPI_BIAS_RESID_BF16 ? JT : 1][2];
    if constexpr (EPI == MMISS_EPI_BIAS_RESID_BF16) {
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
                resid16[j][rh] = pre_resid16 ? pre_resid16[j][rh]
                                             : *reinterpret_cast<const u32x4*>(
                    reinterpret_cast<const uint16_t*>(ep.out) + (size_t)(m_wave + j * 16 + rh * 8 + rrow) * ep.ldo + n_wave +
                    rchunk * 8);
    }
#pragma unroll
    for (int j = 0; j < JT; ++j) {
        // ---- transpose-in: lane (fr = row, fg) owns columns i*16 + 4*fg .. +3
        if constexpr (OUT_BF16) {
            float mu = 0.f, rstd = 1.f;
            if constexpr (FOLD) { mu = sstat[2 * (j * 16 + fr)]; rstd = sstat[2 * (j * 16 + fr) + 1]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (FOLD) y[r] = rstd * (acc[i][j][r] - mu * cvec[i][r]) + bias[i][r];
                    else y[r] = acc[i][j][r] + bias[i][r];
                    if constexpr (EPI == MMISS_EPI_BIAS_QGELU_BF16 || EPI == MMISS_EPI_LNFOLD_QGELU_BF16) y[r] = quick_gelu(y[r]);
                }
                u32x2 pk;
                pk[0] = pack_bf16x2(y[0], y[1]);
                pk[1] = pack_bf16x2(y[2], y[3]);
                *reinterpret_cast<u32x2*>(patch + fr * 144 + i * 32 + fg * 8) = pk;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v = acc[i][j];
                if constexpr (EPI == MMISS_EPI_BIAS_RESID_F32 || EPI == MMISS_EPI_BIAS_RESID_BF16) v += bias[i];
                *reinterpret_cast<f32x4*>(patch + fr * 272 + i * 64 + fg * 16) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- row-major out: 8 lanes x 16 B = one 128-byte segment per row, 8 rows per instruction
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            const int row = rh * 8 + rrow;
            const int m = m_wave + j * 16 + row;
            if constexpr (OUT_BF16) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * 144 + rchunk * 16);
                if (m < ep.m_valid) {
                    u32x4* po = reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(ep.out) + (size_t)m * ep.ldo + n_wave + rchunk * 8);
                    if (ep.nt_out & 2) __builtin_nontemporal_store(v, po);
                    else *po = v;
                }
            } else if constexpr (EPI == MMISS_EPI_BIAS_RESID_BF16) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(patch + row * 272 + rchunk * 32);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(patch + row * 272 + rchunk * 32 + 16);
                const u32x4 old = resid16[j][rh];
                float y[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[2 * e] = __uint_as_float(old[e] << 16) + (e < 2 ? lo[2 * e] : hi[2 * e - 4]);
                    y[2 * e + 1] = __uint_as_float(old[e] & 0xFFFF0000u) + (e < 2 ? lo[2 * e + 1] : hi[2 * e - 3]);
                }
                u32x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(y[2 * e], y[2 * e + 1]);
                if (m < ep.m_valid)
                    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(ep.out) + (size_t)m * ep.ldo + n_wave + rchunk * 8) = pk;
                if (ep.stats_out) {  // statistics of what was STORED (the rounded rows are the residual stream from here on)
                    float rs = 0.f, rq = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = __uint_as_float(pk[e] << 16), b = __uint_as_float(pk[e] & 0xFFFF0000u);
                        rs += a + b;
                        rq += a * a + b * b;
                    }
                    rs += __shfl_xor(rs, 1); rq += __shfl_xor(rq, 1);
                    rs += __shfl_xor(rs, 2); rq += __shfl_xor(rq, 2);
                    rs += __shfl_xor(rs, 4); rq += __shfl_xor(rq, 4);
                    if (rchunk == 0 && m < ep.m_valid) {
                        float* so = ep.stats_out + ((size_t)m * (ep.ldo >> 6) + (n_wave >> 6)) * 2;
                        so[0] = rs;
                        so[1] = rq;
                    }
                }
            } else {
                float rs = 0.f, rq = 0.f;  // this lane's share of the row's (sum, sumsq) over the wave's 64 columns
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * 272 + ch * 128 + rchunk * 16);
                    const int n = n_wave + ch * 32 + rchunk * 4;
                    if (m < ep.m_valid) {
                        if constexpr (EPI == MMISS_EPI_F32) {
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (size_t)m * ep.ldo + n) = v;
                        } else if constexpr (EPI == MMISS_EPI_BIAS_RESID_F32) {
                            float* p = reinterpret_cast<float*>(ep.out) + (size_t)m * ep.ldo + n;
                            v = resid[j][rh][ch] + v;
                            // folded-LayerNorm mode: the next kernel reads the bf16 copy, not the f32 stream, so the f32 rows
                            // are stored non-temporally and leave the L2 to the bf16 rows
                            if (ep.xb_out || (ep.nt_out & 1)) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
                            else *reinterpret_cast<f32x4*>(p) = v;
                            if (ep.xb_out) {
                                u32x2 pk;
                                pk[0] = pack_bf16x2(v[0], v[1]);
                                pk[1] = pack_bf16x2(v[2], v[3]);
                                *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(ep.xb_out) + (size_t)m * ep.ldo + n) = pk;
                            }
                            rs += (v[0] + v[1]) + (v[2] + v[3]);
                            rq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                        } else if constexpr (EPI == MMISS_EPI_PATCH_F32) {
                            const int img = m / ep.p0, pt = m - img * ep.p0;
                            const f32x4 pos = *reinterpret_cast<const f32x4*>(ep.aux + (size_t)(1 + pt) * ep.ldo + n);
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + ((size_t)img * ep.p1 + 1 + pt) * ep.ldo + n) = v + pos;
                        }
                    }
                }
                if constexpr (EPI == MMISS_EPI_BIAS_RESID_F32) {
                    if (ep.stats_out) {  // 8 lanes hold one row: fixed-order xor tree, one writer -> deterministic
                        rs += __shfl_xor(rs, 1); rq += __shfl_xor(rq, 1);
                        rs += __shfl_xor(rs, 2); rq += __shfl_xor(rq, 2);
                        rs += __shfl_xor(rs, 4); rq += __shfl_xor(rq, 4);
                        if (rchunk == 0 && m < ep.m_valid) {
                            float* so = ep.stats_out + ((size_t)m * (ep.ldo >> 6) + (n_wave >> 6)) * 2;
                            so[0] = rs;
                            so[1] = rq;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next j
    }
}

// ALN = true: `A` is really the f32 residual stream x [M,K]; the A tile is LayerNorm(x) (HF:modeling_clip.py:358-360,
// fp32 statistics from ep.ln_stats, affine ep.ln_g / ep.ln_b), computed and rounded to bf16 while it is staged
// through registers into the same swizzled LDS image. This removes the separate LayerNorm pass (59 MB per call at
// B=256) and its kernel boundary; the W tile keeps its LDS-DMA path.
// NWN = waves along n (2: BM x 128 tile, 4 waves, two workgroups per CU; 4: BM x 256 tile, 8 waves, one workgroup per CU:
// the A panel is staged once for 256 output columns instead of once per 128).
// S3 = three staging buffers (prefetch distance 2 K-tiles, one workgroup per CU): for grids below the CU count, where a
// CU holds a single workgroup and nothing else hides the load latency of the two-buffer loop (0.55-0.66 us per K-tile
// measured against 0.27 us of MFMA work, tools/gemm_midm_ksweep.py).
// NWM = waves along m (2, or 4 for the 8-wave form of the 128-column tile: two waves per SIMD inside ONE workgroup, for
// grids below the CU count where no second workgroup shares the CU to overlap a wave's LDS-DMA issue and barrier waits).
template <typename IN, int BM, int EPI, bool ALN = false, int NWN = 2, bool S3 = false, int NWM = 2>
__global__ __launch_bounds__(64 * NWN * NWM, (NWN == 2 && NWM == 2 && !S3) ? 2 : 1) void gemm16_kernel(const IN* __restrict__ A,
                                                                             const IN* __restrict__ W, int M, int N, int K,
                                                                             GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename MfmaIn<IN>::frag frag;
    static_assert(NWN == 2 || (NWN == 4 && !ALN), "wide tiles have no fused-LayerNorm staging");
    static_assert(!S3 || (NWN == 2 && !ALN), "the three-buffer loop exists for the plain 128-column tile");
    static_assert(NWM == 2 || (NWM == 4 && NWN == 2 && !ALN && BM % 64 == 0), "8-wave form: 128-column tile, BM % 64 == 0");
    constexpr int BN = 64 * NWN, NWAVES = NWM * NWN;
    constexpr int JT = BM / (16 * NWM);       // 16-row m sub-tiles per wave (wave tile = BM/NWM x 64)
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, BUF = A_BYTES + W_BYTES;
    constexpr int RPT = BM / 32;              // ALN: rows per thread (thread t: 16-byte chunk t&7 of rows (t>>3) + 32 i)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbm = M / BM, nbn = N / BN;
    const int nwg = nbm * nbn;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int bm, bn;
    tile_order(wg, nbm, nbn, ep.m_fast, bm, bn);

    // split-K: gridDim.y workgroups share one output tile, each walks K / gridDim.y of the reduction (row stride stays K)
    // and writes its f32 partial tile to slab blockIdx.y of ep.out (the caller reduces the slabs)
    const int k_len = K / (int)gridDim.y;
    const IN* Ab = A + (size_t)bm * BM * K + (size_t)blockIdx.y * k_len;
    const IN* Wb = W + (size_t)bn * BN * K + (size_t)blockIdx.y * k_len;
    if (gridDim.y > 1) ep.out = reinterpret_cast<float*>(ep.out) + (size_t)blockIdx.y * M * ep.ldo;

    // staging: one wave-instruction = 8 rows x 128 B; lane -> (row r_in, 16-B slot p); slot p of row r
    // holds global chunk p ^ (r & 7)
    const int r_in = lane >> 3, p = lane & 7;
    const int src_chunk = (p ^ r_in) * 8;  // elements
    // LDS-DMA in buffer form (round 3): the per-lane part of a piece's address — row r_in of the piece, swizzled chunk — is
    // ONE 32-bit register computed once; piece, K-tile and tile go into the scalar offset. The flat form spent three 64-bit
    // VALU adds and a register pair per piece.
    const int lane_vo = (r_in * K + src_chunk) * (int)sizeof(IN);
    const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<IN*>(Ab), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<IN*>(Wb), 0, 0x7fffffff, 0x00020000);
    const int row8 = 8 * K * (int)sizeof(IN);
    auto blds16 = [&](const __amdgpu_buffer_rsrc_t& srd, int so, char* dst) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)dst, 16, lane_vo, so, 0, 0);
    };
    auto stage = [&](int buf, int kt) {
        char* sA = smem + buf * BUF;
        char* sW = sA + A_BYTES;
        const int koff = kt * GEMM_BK * (int)sizeof(IN);
        if constexpr (!ALN && NWN == 2 && NWM == 2) {
#pragma unroll
            for (int i = 0; i < BM / 32; ++i) {
                const int rowblk = wave * (BM / 32) + i;
                blds16(srdA, rowblk * row8 + koff, sA + rowblk * 1024);
            }
        } else if constexpr (!ALN) {
#pragma unroll
            for (int i = 0; i < (BM / 8 + NWAVES - 1) / NWAVES; ++i) {
                const int rowblk = wave + i * NWAVES;  // BM/8 row blocks dealt round-robin to the 8 waves
                if (rowblk < BM / 8) blds16(srdA, rowblk * row8 + koff, sA + rowblk * 1024);
            }
        }
#pragma unroll
        for (int i = 0; i < BN / 8 / NWAVES; ++i) {
            const int rowblk = wave * (BN / 8 / NWAVES) + i;
            blds16(srdW, rowblk * row8 + koff, sW + rowblk * 1024);
        }
    };

    const int wm = wave / NWN, wn = wave % NWN;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[4][JT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = k_len / GEMM_BK;

    // ---- folded LayerNorm: this wave's share of its rows' (mean, rstd), fetched now so that the loads fly during the K loop
    constexpr bool FOLD_EPI = (EPI == MMISS_EPI_LNFOLD_BF16 || EPI == MMISS_EPI_LNFOLD_QGELU_BF16);
    float pre_stats[2] = {0.f, 1.f};
    constexpr int STAT_PER = (JT * 16 + NWN - 1) / NWN;  // rows per sharing wave (<= 64 for every tile height)
    if constexpr (FOLD_EPI) {
        static_assert(STAT_PER <= 64, "one row per lane");
        const int r = (wave % NWN) * STAT_PER + lane;
        if (lane < STAT_PER && r < JT * 16)
            ln_row_stats(ep, (int64_t)bm * BM + (wave / NWN) * (BM / NWM) + r, pre_stats[0], pre_stats[1]);
    }

    // ---- residual epilogue: the tile's residual rows are fetched NOW, so that the read half of the epilogue's traffic
    // (39 MB per launch at 12800 x 768) flies under the whole K loop instead of after it. 16 * JT more live VGPRs in the
    // loop: done for tile heights up to 160 rows (223 of the 256 registers two waves per SIMD allow), not for 192.
    constexpr bool PRE_RESID = (EPI == MMISS_EPI_BIAS_RESID_F32) && !ALN && NWN == 2 && NWM == 2 && JT <= 5;
    f32x4 pre_resid[PRE_RESID ? JT : 1][2][2];
    if constexpr (PRE_RESID) {
        if (gridDim.y == 1) {  // (split-K launches use the F32 epilogue, never this one)
            const int rrow = lane >> 3, rchunk = lane & 7;
            const float* xo = reinterpret_cast<const float*>(ep.out) +
                              (size_t)(bm * BM + (wave / NWN) * (BM / NWM) + rrow) * ep.ldo + bn * BN + (wave % NWN) * 64 + rchunk * 4;
#pragma unroll
            for (int j = 0; j < JT; ++j)
#pragma unroll
                for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch)
                        pre_resid[j][rh][ch] = *reinterpret_cast<const f32x4*>(xo + (size_t)(j * 16 + rh * 8) * ep.ldo + ch * 32);
        }
    }

    // (bf16 residual stream: 8 registers per 16-row sub-tile instead of 16 — every tile height prefetches)
    constexpr bool PRE_RESID16 = (EPI == MMISS_EPI_BIAS_RESID_BF16) && !ALN && NWN == 2 && NWM == 2;
    u32x4 pre_resid16[PRE_RESID16 ? JT : 1][2];
    if constexpr (PRE_RESID16) {
        const int rrow = lane >> 3, rchunk = lane & 7;
        const uint16_t* xo = reinterpret_cast<const uint16_t*>(ep.out) +
                             (size_t)(bm * BM + (wave / NWN) * (BM / NWM) + rrow) * ep.ldo + bn * BN + (wave % NWN) * 64 + rchunk * 8;
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
                pre_resid16[j][rh] = *reinterpret_cast<const u32x4*>(xo + (size_t)(j * 16 + rh * 8) * ep.ldo);
    }

    // ---- ALN: per-row (mean, rstd) from the partial sums, register staging of the f32 tile
    const int xc8 = tid & 7, xr0 = tid >> 3;
    float ln_mean[RPT], ln_rstd[RPT];
    f32x4 xr[RPT][2];
    const float* Xb = reinterpret_cast<const float*>(A) + (size_t)bm * BM * K;
    auto x_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const float* src = Xb + (size_t)(xr0 + 32 * i) * K + kt * GEMM_BK + xc8 * 8;
            xr[i][0] = *reinterpret_cast<const f32x4*>(src);
            xr[i][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
    };
    auto x_write = [&](int buf, int kt) {
        char* sA = smem + buf * BUF;
        const int k0 = kt * GEMM_BK + xc8 * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(ep.ln_g + k0), g1 = *reinterpret_cast<const f32x4*>(ep.ln_g + k0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(ep.ln_b + k0), b1 = *reinterpret_cast<const f32x4*>(ep.ln_b + k0 + 4);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row = xr0 + 32 * i;
            float y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = (xr[i][0][e] - ln_mean[i]) * ln_rstd[i] * g0[e] + b0[e];
                y[4 + e] = (xr[i][1][e] - ln_mean[i]) * ln_rstd[i] * g1[e] + b1[e];
            }
            u32x4 pk;
            pk[0] = pack_bf16x2(y[0], y[1]); pk[1] = pack_bf16x2(y[2], y[3]);
            pk[2] = pack_bf16x2(y[4], y[5]); pk[3] = pack_bf16x2(y[6], y[7]);
            *reinterpret_cast<u32x4*>(sA + row * 128 + ((xc8 ^ (row & 7)) << 4)) = pk;
        }
    };
    if constexpr (ALN) {
        float* srow = reinterpret_cast<float*>(smem + 2 * BUF);  // [BM][2] behind the staging buffers
        if (tid < BM) {
            const float* st = ep.ln_stats + (size_t)(bm * BM + tid) * ep.ln_parts * 2;
            float s1 = 0.f, s2 = 0.f;
            for (int pp = 0; pp < ep.ln_parts; ++pp) { s1 += st[2 * pp]; s2 += st[2 * pp + 1]; }
            const float mean = s1 / (float)K;
            const float var = fmaxf(s2 / (float)K - mean * mean, 0.f);
            srow[2 * tid] = mean;
            srow[2 * tid + 1] = 1.0f / sqrtf(var + ep.ln_eps);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            ln_mean[i] = srow[2 * (xr0 + 32 * i)];
            ln_rstd[i] = srow[2 * (xr0 + 32 * i) + 1];
        }
        x_load(0);
    }
    auto mma_tile = [&](int buf) {
        const char* sA = smem + buf * BUF;
        const char* sW = sA + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            frag wf[4], af[JT];
            const int chunk = 4 * s + fg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + i * 16 + fr;
                wf[i] = *reinterpret_cast<const frag*>(sW + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const int row = wm * (BM / NWM) + j * 16 + fr;
                af[j] = *reinterpret_cast<const frag*>(sA + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j) acc[i][j] = MfmaIn<IN>::mma(wf[i], af[j], acc[i][j]);
        }
    };
    if constexpr (S3) {
        // K-tile t lives in buffer t % 3; its loads were issued two steps earlier. A wave issues LPS loads per stage, so
        // "at most LPS outstanding" means stage t has landed while stage t+1 may still fly.
        constexpr int LPS = (BM / 8 + NWAVES - 1) / NWAVES + BN / 8 / NWAVES;  // pieces a wave issues per stage
        static_assert(NWM == 2 || (BM / 8) % NWAVES == 0, "counted waits need the same piece count on every wave");
        stage(0, 0);
        if (nt > 1) stage(1, 1);
        int buf = 0;
        for (int t = 0; t < nt; ++t) {
            if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // every wave's stage-t loads are in LDS; every wave is done reading buffer (t-1) % 3
            if (t + 2 < nt) stage(buf == 0 ? 2 : buf - 1, t + 2);
            mma_tile(buf);
            buf = buf == 2 ? 0 : buf + 1;
        }
        __syncthreads();  // the epilogue reuses the staging buffers as transpose patches
    } else {
    stage(0, 0);
    if constexpr (ALN) x_write(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            stage(cur ^ 1, t + 1);
            if constexpr (ALN) x_load(t + 1);  // f32 rows of the next K-tile fly during this tile's MFMAs
        }
        const char* sA = smem + cur * BUF;
        const char* sW = sA + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            frag wf[4], af[JT];
            const int chunk = 4 * s + fg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + i * 16 + fr;
                wf[i] = *reinterpret_cast<const frag*>(sW + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < JT; ++j) {
                const int row = wm * (BM / NWM) + j * 16 + fr;
                af[j] = *reinterpret_cast<const frag*>(sA + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < JT; ++j) acc[i][j] = MfmaIn<IN>::mma(wf[i], af[j], acc[i][j]);
        }
        if constexpr (ALN) {
            if (t + 1 < nt) x_write(cur ^ 1, t + 1);  // buffer cur^1 was last read before the previous barrier
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    }  // two-buffer loop

    // acc[i][j][reg] = C[m = m_base + j*16][n = n_base + i*16 + reg]
    const int m_base = bm * BM + wm * (BM / NWM) + fr;
    const int n_base = bn * BN + wn * 64 + 4 * fg;

    if constexpr (EPI == MMISS_EPI_GROUPMAX_F32) {
        const int g = (bn * NWN + wn) * 4 + fg;
        float* out = reinterpret_cast<float*>(ep.out);
#pragma unroll
        for (int j = 0; j < JT; ++j) {
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (n_base + i * 16 + r < ep.p0) ? acc[i][j][r] : -INFINITY;
                    mx = fmaxf(mx, v);
                }
            const int m = m_base + j * 16;
            if (m < ep.m_valid) out[(size_t)m * ep.ldo + g] = mx;
        }
        return;
    }

    // all waves are past the loop's last barrier: the staging buffers are dead, each wave takes a private patch
    // folded LayerNorm: the NWN waves of one wm share their rows' statistics (one area per wm, each wave finalises a share)
    gemm_epilogue<EPI, JT>(ep, acc, bm * BM + wm * (BM / NWM), bn * BN + wn * 64, smem + wave * EPI_PATCH_BYTES, lane,
                           smem + NWAVES * EPI_PATCH_BYTES + wm * (JT * 16 * 8), wn, NWN, FOLD_EPI ? pre_stats : nullptr,
                           PRE_RESID ? pre_resid : nullptr, PRE_RESID16 ? pre_resid16 : nullptr);
}

static inline double gemm_flops(int M, int N, int K) { return 2.0 * M * N * K; }

// Tile height for a GEMM over M_rows x N: the 256 CUs hold 2 blocks each (512 slots). A grid of <= 512 tiles runs
// in one round, so fewer, taller tiles win as long as they still fit one round; past ~2 rounds the hardware's
// dynamic dispatch smooths the tail and taller tiles win through operand reuse. Relative cost per row
// (128: 1.00, 160: 0.93, 192: 0.89) fitted to tools/gemm_bench.py on the ViT-B/32 shapes (profiles/gemm_tiles_r01.txt).
static inline int gemm_pick_bm(int64_t M_rows, int N) {
    int best = 128;
    double best_cost = 1e300;
    const int bms[3] = {128, 160, 192};
    const double eff[3] = {1.00, 0.93, 0.89};
    for (int v = 0; v < 3; ++v) {
        const int bm = bms[v];
        const double tiles = (double)((M_rows + bm - 1) / bm) * (N / GEMM_BN);
        const double rounds = tiles <= 1024.0 ? (double)((int64_t)((tiles + 511.0) / 512.0)) : tiles / 512.0 + 0.5;
        const double cost = rounds * bm * eff[v];
        if (cost < best_cost - 1e-9) { best_cost = cost; best = bm; }
    }
    return best;
}

// Tile variant for launch_gemm = the tile height (128 / 160 / 192: BM x 128 tile, 4 waves, 2 workgroups per CU). (The BM x 256
// tiles with 8 waves, variants 2000 + BM of rounds 1-3 — 3-11 % faster as isolated launches, 8 % slower inside the encode —
// were removed in round 4; profiles/gemm_variants_r01.md.)
static inline int gemm_pick_variant(int64_t M_rows, int N) { return gemm_pick_bm(M_rows, N); }

template <typename IN, int BM, int EPI, bool ALN = false, int NWN = 2, bool S3 = false, int NWM = 2>
static int launch_gemm_inst(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K,
                            int splits = 1) {
    constexpr int BN = 64 * NWN;
    constexpr int LDS = (S3 ? 3 : 2) * (BM + BN) * 128 + (ALN ? BM * 8 : 0);
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm16_kernel<IN, BM, EPI, ALN, NWN, S3, NWM>), LDS));
    if (N % BN) MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm: N=%d is not a multiple of the %d-column tile", N, BN);
    const int nwg = (M / BM) * (N / BN);
    GemmEpi e2 = ep;
    // Tile order of the wide GEMMs (>= 8 column tiles): bands of 5 m-tiles, m fastest inside a band. The ~64
    // tiles an XCD runs at once then touch 5 A panels + ~13 W panels (~4 MB, the L2's size) instead of ~3 + all W panels:
    // FETCH_SIZE of the folded FC1 189 -> 136 MB per launch (QKV 117 -> 107 MB) at unchanged time (profiles/r02_traffic.json;
    // 10-tile bands: 150 MB). The narrowest GEMMs (fewer than 8 column tiles) keep n fastest.
    // (ViT-L/14, 32896 rows: also the N = 1024 GEMMs gain — FC2 322 -> 310 us, out-proj 113 -> 109 us; ViT-B/32's N = 768
    // GEMMs, 6 column tiles, do not)
    if (e2.m_fast == 0 && N / BN >= 8) e2.m_fast = mmiss_option("gemm_band", 5);
    const int forced = mmiss_option("gemm_group_m", -1);  // experiment knob (tools/gemm_order_sweep.py)
    if (forced >= 0 && e2.m_fast != 1) e2.m_fast = forced;
    e2.nt_out |= mmiss_option("gemm_nt", 0);
    hipLaunchKernelGGL((gemm16_kernel<IN, BM, EPI, ALN, NWN, S3, NWM>), dim3(nwg, splits), dim3(64 * NWN * NWM), LDS, st,
                       reinterpret_cast<const IN*>(A), reinterpret_cast<const IN*>(W), M, N, K, e2);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// LayerNorm-folded bf16 GEMM (epilogues 7 / 8): A = bf16 residual rows, W = gamma-folded weights, ep.aux = c,
// ep.bias = b', ep.ln_stats / ln_parts / ln_eps = the rows' partial statistics.
static int launch_gemm_fold(hipStream_t st, int epi, int bm, const void* A, const void* W, const GemmEpi& ep, int M, int N,
                            int K) {
    if (bm == 0) bm = 128;
    if (M <= 0 || N <= 0 || K <= 0 || (M % bm) || (N % GEMM_BN) || (K % GEMM_BK) || !ep.ln_stats || !ep.aux || !ep.bias ||
        ep.ln_parts * 64 != K)
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm_fold: M=%d N=%d K=%d bm=%d parts=%d", M, N, K, bm, ep.ln_parts);
    const int mv = ep.m_valid < M ? ep.m_valid : M;
    const double bytes = 2.0 * ((double)mv * K + (double)N * K) + 2.0 * (double)mv * N;
    MM_PROF(epi == MMISS_EPI_LNFOLD_BF16 ? "gemm_bf16_lnfold_bias" : "gemm_bf16_lnfold_qgelu", st, 2.0 * mv * N * K, bytes);
#define GEMM_FOLD_CASE(BMV)                                                                                        \
    case BMV:                                                                                                      \
        return epi == MMISS_EPI_LNFOLD_BF16                                                                        \
                   ? launch_gemm_inst<__bf16, BMV, MMISS_EPI_LNFOLD_BF16>(st, A, W, ep, M, N, K)                   \
                   : launch_gemm_inst<__bf16, BMV, MMISS_EPI_LNFOLD_QGELU_BF16>(st, A, W, ep, M, N, K);
    if (epi != MMISS_EPI_LNFOLD_BF16 && epi != MMISS_EPI_LNFOLD_QGELU_BF16) MM_FAIL(MMISS_ERR_ARG, "gemm_fold: epilogue %d", epi);
    switch (bm) {
        GEMM_FOLD_CASE(128)
        GEMM_FOLD_CASE(160)
        GEMM_FOLD_CASE(192)
    }
#undef GEMM_FOLD_CASE
    MM_FAIL(MMISS_ERR_ARG, "gemm_fold: unsupported tile height %d", bm);
}

// Residual GEMM on a bf16 residual stream (epilogue 9): ep.out = the stream (bf16, read-modify-write), ep.bias set,
// ep.stats_out optional.
static int launch_gemm_resid16(hipStream_t st, int bm, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K) {
    if (bm == 0) bm = 128;
    if (M <= 0 || N <= 0 || K <= 0 || (M % bm) || (N % GEMM_BN) || (K % GEMM_BK) || !ep.out || !ep.bias)
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm_resid16: M=%d N=%d K=%d bm=%d", M, N, K, bm);
    const int mv = ep.m_valid < M ? ep.m_valid : M;
    const double bytes = 2.0 * ((double)mv * K + (double)N * K) + 2.0 * 2.0 * (double)mv * N;
    char pname[48];
    snprintf(pname, sizeof(pname), "gemm_bf16_bias_resid16_k%d", K);
    MM_PROF(pname, st, 2.0 * mv * N * K, bytes);
    switch (bm) {
        case 128: return launch_gemm_inst<__bf16, 128, MMISS_EPI_BIAS_RESID_BF16>(st, A, W, ep, M, N, K);
        case 160: return launch_gemm_inst<__bf16, 160, MMISS_EPI_BIAS_RESID_BF16>(st, A, W, ep, M, N, K);
        case 192: return launch_gemm_inst<__bf16, 192, MMISS_EPI_BIAS_RESID_BF16>(st, A, W, ep, M, N, K);
    }
    MM_FAIL(MMISS_ERR_ARG, "gemm_resid16: unsupported tile height %d", bm);
}


// Two or three staging buffers: a grid of at most 256 workgroups puts one workgroup on a CU whatever its LDS footprint,
// so the deeper pipeline costs no occupancy there and hides the load latency the second workgroup would have hidden.
template <typename IN, int BM, int EPI>
static int launch_gemm_stages(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K,
                              int splits = 1) {
    const int64_t wgs = (int64_t)(M / BM) * (N / GEMM_BN) * splits;
    if (wgs <= 256 && K / splits >= 4 * GEMM_BK && mmiss_option("gemm_s3", 1) != 0) {
        if constexpr (BM == 128) {
            if (mmiss_option("gemm_w8", 1) != 0) return launch_gemm_inst<IN, BM, EPI, false, 2, true, 4>(st, A, W, ep, M, N, K, splits);
        }
        return launch_gemm_inst<IN, BM, EPI, false, 2, true>(st, A, W, ep, M, N, K, splits);
    }
    if constexpr (BM == 128) {
        // the 8-wave form (two buffers) also while the grid is at most one round of 2 workgroups per CU, where many CUs
        // still hold a single workgroup (bs 128: 2.10 -> 2.03 ms, bs 192: 2.60 -> 2.55 ms per encode)
        if (wgs <= mmiss_option("gemm_w8_max_wgs", 512)) return launch_gemm_inst<IN, BM, EPI, false, 2, false, 4>(st, A, W, ep, M, N, K, splits);
    }
    return launch_gemm_inst<IN, BM, EPI>(st, A, W, ep, M, N, K, splits);
}

template <typename IN, int EPI>
static int launch_gemm_bm(hipStream_t st, int bm, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K) {
    switch (bm) {
        case 128: return launch_gemm_stages<IN, 128, EPI>(st, A, W, ep, M, N, K);
        case 160: return launch_gemm_stages<IN, 160, EPI>(st, A, W, ep, M, N, K);
        case 192: return launch_gemm_stages<IN, 192, EPI>(st, A, W, ep, M, N, K);
    }
    MM_FAIL(MMISS_ERR_ARG, "gemm: unsupported tile height %d", bm);
}

// ------------------------------------------------------------------------------------------------
// Split-K for the middle batch sizes. With 150 .. 3000 rows a narrow GEMM (out-proj / FC2: N = 768) has 12-100 tiles
// for 256 CUs and each workgroup walks its whole K behind a barrier per 64-wide step at ~0.9 us per step (one
// workgroup per CU: nothing hides the load latency): FC2 (K = 3072) takes 42 us at ANY batch from 3 to 32 images. So the
// K loop is cut into `splits` slices run by different workgroups (grid.y), each writing an f32 partial tile to a scratch
// slab, and splitk_reduce_kernel sums the slabs in a fixed order (deterministic) and applies the real epilogue.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int Mws, int N,
                                                            GemmEpi ep) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = N >> 2;
    const int m = (int)(idx / n4), n = (int)(idx - (int64_t)m * n4) * 4;
    if (m >= ep.m_valid) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(ws + (size_t)m * N + n);
    for (int sp = 1; sp < splits; ++sp) v += *reinterpret_cast<const f32x4*>(ws + ((size_t)sp * Mws + m) * N + n);
    if constexpr (EPI == MMISS_EPI_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (size_t)m * ep.ldo + n) = v;
    } else if constexpr (EPI == MMISS_EPI_BIAS_BF16 || EPI == MMISS_EPI_BIAS_QGELU_BF16) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(ep.bias + n);
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            y[r] = v[r] + b[r];
            if constexpr (EPI == MMISS_EPI_BIAS_QGELU_BF16) y[r] = quick_gelu(y[r]);
        }
        u32x2 pk;
        pk[0] = pack_bf16x2(y[0], y[1]);
        pk[1] = pack_bf16x2(y[2], y[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(ep.out) + (size_t)m * ep.ldo + n) = pk;
    } else if constexpr (EPI == MMISS_EPI_BIAS_RESID_F32) {
        v += *reinterpret_cast<const f32x4*>(ep.bias + n);
        float* p = reinterpret_cast<float*>(ep.out) + (size_t)m * ep.ldo + n;
        *reinterpret_cast<f32x4*>(p) = *reinterpret_cast<const f32x4*>(p) + v;
    } else {  // MMISS_EPI_PATCH_F32
        const int img = m / ep.p0, pt = m - img * ep.p0;
        const f32x4 pos = *reinterpret_cast<const f32x4*>(ep.aux + (size_t)(1 + pt) * ep.ldo + n);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + ((size_t)img * ep.p1 + 1 + pt) * ep.ldo + n) = v + pos;
    }
}

// Split-K is considered for long-K GEMMs of at most this many tiles (bs-256 sweep, tools/batch_sweep.py: 228 tiles
// gain 7 %, 300 tiles lose 3 % to the reduce traffic).
static inline int gemm_splitk_max_tiles() { return mmiss_option("gemm_splitk_max_tiles", 256); }
static inline bool gemm_splitk_candidate(int64_t tiles, int K) { return K >= 1536 && tiles <= gemm_splitk_max_tiles(); }

// How many K slices for a tiled GEMM of `tiles` workgroups (1 = do not split).
static inline int gemm_splitk_splits(int tiles, int K, int M, int N, const GemmEpi& ep) {
    if (!ep.splitk_ws || ep.stats_out || ep.xb_out || mmiss_option("gemm_splitk", 1) == 0) return 1;
    if (!gemm_splitk_candidate(tiles, K)) return 1;
    int splits = mmiss_option("gemm_splitk_target", 384) / tiles;  // workgroups aimed at (256 CUs x up to 2)
    if (splits < 2) splits = 2;
    if (splits > 8) splits = 8;
    while (splits > 1 && ((K % (splits * GEMM_BK)) != 0 || K / (splits * GEMM_BK) < 6)) --splits;
    if (splits > 1 && (size_t)splits * M * N * 4 > ep.splitk_ws_bytes) return 1;
    return splits;
}

template <int EPI>
static int launch_gemm_splitk(hipStream_t st, int bm, int splits, const void* A, const void* W, const GemmEpi& ep, int M,
                              int N, int K) {
    GemmEpi part{};
    part.out = ep.splitk_ws; part.ldo = N; part.m_valid = M; part.m_fast = ep.m_fast;
    switch (bm) {
        case 128: MM_TRY((launch_gemm_stages<__bf16, 128, MMISS_EPI_F32>(st, A, W, part, M, N, K, splits))); break;
        case 160: MM_TRY((launch_gemm_stages<__bf16, 160, MMISS_EPI_F32>(st, A, W, part, M, N, K, splits))); break;
        default: MM_TRY((launch_gemm_stages<__bf16, 192, MMISS_EPI_F32>(st, A, W, part, M, N, K, splits))); break;
    }
    const int mv = ep.m_valid < M ? ep.m_valid : M;
    const int64_t threads = (int64_t)mv * (N / 4);
    hipLaunchKernelGGL((splitk_reduce_kernel<EPI>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, ep.splitk_ws,
                       splits, M, N, ep);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// bf16 GEMM of the CLIP towers. `bm` = tile height (128/160/192; 0 = 128); M must be a multiple of it.
// gemm_skinny.h (included at the end of this header): the M <= 256 path
static inline bool gemm_skinny_ok(int epi, int mv, int N, int K, const GemmEpi& ep);
static int launch_gemm_skinny(hipStream_t st, int epi, const void* A, const void* W, const GemmEpi& ep, int mv, int N, int K);

static int launch_gemm(hipStream_t st, int epi, int bm, const void* A, const void* W, const GemmEpi& ep, int M, int N,
                       int K) {
    if (M > 0 && N > 0 && K > 0) {
        const int mv0 = ep.m_valid < M ? ep.m_valid : M;
        if (gemm_skinny_ok(epi, mv0, N, K, ep)) {
            static const char* snames[] = {"gemm_skinny_f32", "gemm_skinny_bias", "gemm_skinny_bias_qgelu",
                                           "gemm_skinny_bias_resid", "gemm_skinny_patch"};
            const int oe = (epi == MMISS_EPI_BIAS_BF16 || epi == MMISS_EPI_BIAS_QGELU_BF16) ? 2 : 4;
            MM_PROF(snames[epi], st, gemm_flops(mv0, N, K),
                    2.0 * ((double)mv0 * K + (double)N * K) + (double)oe * mv0 * N * (epi == MMISS_EPI_BIAS_RESID_F32 ? 2 : 1));
            return launch_gemm_skinny(st, epi, A, W, ep, mv0, N, K);
        }
    }
    if (bm == 0) bm = 128;
    if (M <= 0 || N <= 0 || K <= 0 || (M % (bm % 1000)) || (N % GEMM_BN) || (K % GEMM_BK))
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm: M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, bm, GEMM_BN,
                GEMM_BK);
    static const char* names[] = {"gemm_bf16_f32", "gemm_bf16_bias", "gemm_bf16_bias_qgelu",
                                  "gemm_bf16_bias_resid", "gemm_bf16_patch"};
    if (epi < 0 || epi > 4) MM_FAIL(MMISS_ERR_ARG, "gemm: bad epilogue %d", epi);
    const int out_elt = (epi == MMISS_EPI_BIAS_BF16 || epi == MMISS_EPI_BIAS_QGELU_BF16) ? 2 : 4;
    const int mv = ep.m_valid < M ? ep.m_valid : M;
    const double bytes = 2.0 * ((double)mv * K + (double)N * K) +
                         (double)out_elt * mv * N * (epi == MMISS_EPI_BIAS_RESID_F32 ? 2 : 1);
    if (bm < 1000) {
        const int splits = gemm_splitk_splits((M / bm) * (N / GEMM_BN), K, M, N, ep);
        if (splits > 1) {
            static const char* knames[] = {"gemm_splitk_f32", "gemm_splitk_bias", "gemm_splitk_bias_qgelu",
                                           "gemm_splitk_bias_resid", "gemm_splitk_patch"};
            MM_PROF(knames[epi], st, gemm_flops(mv, N, K), bytes + 8.0 * splits * (double)M * N);
            switch (epi) {
                case MMISS_EPI_F32: return launch_gemm_splitk<MMISS_EPI_F32>(st, bm, splits, A, W, ep, M, N, K);
                case MMISS_EPI_BIAS_BF16: return launch_gemm_splitk<MMISS_EPI_BIAS_BF16>(st, bm, splits, A, W, ep, M, N, K);
                case MMISS_EPI_BIAS_QGELU_BF16: return launch_gemm_splitk<MMISS_EPI_BIAS_QGELU_BF16>(st, bm, splits, A, W, ep, M, N, K);
                case MMISS_EPI_BIAS_RESID_F32: return launch_gemm_splitk<MMISS_EPI_BIAS_RESID_F32>(st, bm, splits, A, W, ep, M, N, K);
                default: return launch_gemm_splitk<MMISS_EPI_PATCH_F32>(st, bm, splits, A, W, ep, M, N, K);
            }
        }
    }
    // the residual epilogue serves two shapes of very different arithmetic intensity (out-proj K = hidden, FC2 K = mlp):
    // they are timed as separate classes so that each can be held against its own roof
    char pname[48];
    if (epi == MMISS_EPI_BIAS_RESID_F32) snprintf(pname, sizeof(pname), "%s_k%d", names[epi], K);
    else snprintf(pname, sizeof(pname), "%s", names[epi]);
    MM_PROF(pname, st, gemm_flops(mv, N, K), bytes + (ep.xb_out ? 2.0 * mv * N : 0.0));
    switch (epi) {
        case MMISS_EPI_F32: return launch_gemm_bm<__bf16, MMISS_EPI_F32>(st, bm, A, W, ep, M, N, K);
        case MMISS_EPI_BIAS_BF16: return launch_gemm_bm<__bf16, MMISS_EPI_BIAS_BF16>(st, bm, A, W, ep, M, N, K);
        case MMISS_EPI_BIAS_QGELU_BF16: return launch_gemm_bm<__bf16, MMISS_EPI_BIAS_QGELU_BF16>(st, bm, A, W, ep, M, N, K);
        case MMISS_EPI_BIAS_RESID_F32: return launch_gemm_bm<__bf16, MMISS_EPI_BIAS_RESID_F32>(st, bm, A, W, ep, M, N, K);
        default: return launch_gemm_bm<__bf16, MMISS_EPI_PATCH_F32>(st, bm, A, W, ep, M, N, K);
    }
}

#include "gemm_skinny.h"

// Skinny bf16 GEMM for M <= 128 rows (<= 320 when N <= 1024): C[M,N] = A[M,K] · W[N,K]^T with the encoder's epilogues.
//
// Why: the reference serves ONE image or ONE query string per request (backend/app/utils.py:76-77,88-97), i.e.
// M = 50 or <= 77 token rows per GEMM, and the pruned last layer / the projection head run on M = batch rows. The
// tiled kernel (gemm_bf16.h) then launches M/128 x N/128 = 6..24 workgroups on 256 CUs and each walks its whole K
// behind a barrier per 64-wide step: 5-36 us per GEMM, pure latency. Here the weight matrix is the streamed operand,
// exactly like the index rows of the retrieval scan:
//   * one workgroup per (16 output features, group of MT m-tiles): N/16 x ceil(M/16/MT) workgroups, 4 waves (8 when
//     K >= 2048), each wave owns an equal slice of K;
//   * a wave loads its W fragment (16 n x 32 k, the MFMA A operand) and the MT activation fragments (16 m x 32 k,
//     B operands) straight from global/L2 into registers - no LDS, no barrier in the loop - and issues MT
//     v_mfma_f32_16x16x32_bf16; the loads of up to 8 k-steps are in flight before the first MFMA of a trip;
//   * the K-slices are summed through LDS in a fixed order (deterministic), then wave w finishes m-tiles
//     w, w+4, ...: bias / QuickGELU / residual add / patch-position epilogue, 4 consecutive n per lane.
// Rows >= M are clamped to row M-1 on load and never stored, so any [M,K] buffer is safe.
// Measured against the tiled kernel (tools/gemm_skinny_bench.py, ViT-B/32 shapes): M=50: QKV 5.4 vs 10.1 us, out-proj
// 3.7 vs 12.6 us, FC1 5.5 vs 10.9 us, FC2 7.5 vs 33 us; M=256: out-proj 6.4 vs 12.6 us, FC2 15.8 vs 33 us, but QKV / FC1
// 12.3-12.8 vs 10-10.9 us - hence the dispatch rule in gemm_skinny_ok.
#pragma once
#include "gemm_bf16.h"

template <int EPI, int MT, int NW, int U>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W,
                                                          int M, int N, int K, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kper = K / NW, k0 = wave * kper;
    const __bf16* wp = W + (size_t)(n0 + fr) * K + k0 + fg * 8;
    const __bf16* ab = A + k0 + fg * 8;
    const int m_base = blockIdx.y * (MT * 16);  // grid.y walks groups of MT m-tiles
    uint32_t aoff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = m_base + t * 16 + fr;
        aoff[t] = (uint32_t)(m < M ? m : M - 1) * (uint32_t)K;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // folded LayerNorm: (mean, rstd) of the row this lane finishes first (m-tile t = wave), summed from its K/16 partial sums
    // right behind the first trip's operand loads, so that these loads fly under the K loop's own load latency
    constexpr bool FOLD = (EPI == MMISS_EPI_LNFOLD_BF16 || EPI == MMISS_EPI_LNFOLD_QGELU_BF16);
    auto row_mean_rstd = [&](int m, float& mean, float& rstd) {
        const f32x4* st = reinterpret_cast<const f32x4*>(ep.ln_stats + (size_t)m * ep.ln_parts * 2);
        float s1 = 0.f, s2 = 0.f;
        const int n4 = ep.ln_parts >> 1;
        // all of a row's partial sums in flight at once (K = 768: 24 loads, ONE round trip; in steps of 8 they were three
        // dependent round trips, ~3 us of a 9 us kernel: rocprofv3 of one request, round 3). Same summation order.
        for (int q0 = 0; q0 < n4; q0 += 24) {
            f32x4 buf[24];
#pragma unroll
            for (int j = 0; j < 24; ++j) buf[j] = (q0 + j < n4) ? st[q0 + j] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 24; ++j) { s1 += buf[j][0] + buf[j][2]; s2 += buf[j][1] + buf[j][3]; }
        }
        const float kd = (float)(ep.ln_parts * 16);
        mean = s1 / kd;
        rstd = 1.0f / sqrtf(fmaxf(s2 / kd - mean * mean, 0.f) + ep.ln_eps);
    };
    float pre_mean = 0.f, pre_rstd = 1.f;

    // Everything the epilogue reads from memory is requested NOW, so that it flies under the weight stream: bias / c of this
    // lane's 4 output features, and (residual epilogues) the old residual values of the FIRST m-tile this wave finishes. Loaded
    // where they are used they were one more dependent round trip (~1 us of a 9 us kernel) behind the K loop and the reduction.
    const int n_pre = n0 + fg * 4;
    constexpr bool HAS_BIAS = (EPI != MMISS_EPI_F32 && EPI != MMISS_EPI_PATCH_F32);
    f32x4 pre_bias = f32x4{0.f, 0.f, 0.f, 0.f}, pre_cv = f32x4{0.f, 0.f, 0.f, 0.f}, pre_res = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (HAS_BIAS) pre_bias = *reinterpret_cast<const f32x4*>(ep.bias + n_pre);
    if constexpr (FOLD) pre_cv = *reinterpret_cast<const f32x4*>(ep.aux + n_pre);
    if constexpr (EPI == MMISS_EPI_BIAS_RESID_F32) {
        const int m = m_base + wave * 16 + fr;
        if (wave < MT && m < M) pre_res = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(ep.out) + (size_t)m * ep.ldo + n_pre);
    }

    // U k-steps per trip: every load of the trip (U weight fragments from HBM/MALL, U*MT activation fragments from L2)
    // is issued before its first MFMA - the loop is bound by memory latency, so what matters is loads in flight.
    // The launcher picks NW so that a wave's whole K-slice is one trip where registers allow (U*(1+MT)*4 VGPRs).
    for (int k = 0; k < kper; k += 32 * U) {
        bf16x8 w[U], a[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k + 32 * u < kper) {
                w[u] = *reinterpret_cast<const bf16x8*>(wp + k + 32 * u);
#pragma unroll
                for (int t = 0; t < MT; ++t) a[u][t] = *reinterpret_cast<const bf16x8*>(ab + aoff[t] + k + 32 * u);
            }
        }
        if constexpr (FOLD) {  // behind the first trip's operand loads: both latencies run together
            if (k == 0) {
                const int m = m_base + wave * 16 + fr;
                if (wave < MT && m < M) row_mean_rstd(m, pre_mean, pre_rstd);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k + 32 * u < kper) {
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u], a[u][t], acc[t], 0, 0, 0);
            }
        }
    }

    // ---- sum the NW K-slices in a fixed order
    f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int t = 0; t < MT; ++t) red[(wave * MT + t) * 64 + lane] = acc[t];
    __syncthreads();
    const int n = n0 + fg * 4;  // this lane's 4 consecutive output features
    for (int t = wave; t < MT; t += NW) {
        f32x4 v = red[(0 * MT + t) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[(w * MT + t) * 64 + lane];
        const int m = m_base + t * 16 + fr;
        if constexpr (EPI == MMISS_EPI_BIAS_RESID_F32) {
            if (ep.stats_out) {
                // partial (sum, sumsq) of the NEW residual row over this workgroup's 16 columns: the four lanes fr + 16 fg of
                // a row, fixed xor order, one writer -> deterministic. [M][N/16][2]: the next folded skinny GEMM sums them.
                float rs = 0.f, rq = 0.f;
                if (m < M) {
                    const f32x4 oldv = (t == wave) ? pre_res
                                                   : *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(ep.out) + (size_t)m * ep.ldo + n);
                    const f32x4 nv = oldv + v + pre_bias;
                    rs = (nv[0] + nv[1]) + (nv[2] + nv[3]);
                    rq = (nv[0] * nv[0] + nv[1] * nv[1]) + (nv[2] * nv[2] + nv[3] * nv[3]);
                }
                rs += __shfl_xor(rs, 16); rq += __shfl_xor(rq, 16);
                rs += __shfl_xor(rs, 32); rq += __shfl_xor(rq, 32);
                if (fg == 0 && m < M) {
                    float* so = ep.stats_out + ((size_t)m * (N >> 4) + (n0 >> 4)) * 2;
                    so[0] = rs;
                    so[1] = rq;
                }
            }
        }
        if (m >= M) continue;
        if constexpr (EPI == MMISS_EPI_F32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (size_t)m * ep.ldo + n) = v;
        } else if constexpr (EPI == MMISS_EPI_BIAS_BF16 || EPI == MMISS_EPI_BIAS_QGELU_BF16) {
            const f32x4 b = pre_bias;
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
        } else if constexpr (EPI == MMISS_EPI_LNFOLD_BF16 || EPI == MMISS_EPI_LNFOLD_QGELU_BF16) {
            // LayerNorm folded into the weights (gemm_bf16.h, epilogues 7 / 8): A = bf16(x), W = bf16(W * gamma),
            // y = rstd_m * (acc - mean_m * c_n) + b'_n; (mean, rstd) of row m from the 16-column partial sums the producing
            // skinny residual GEMM (or skinny_row_stats16_kernel) left in ep.ln_stats [M][ln_parts][2], ln_parts = K / 16
            float mean = pre_mean, rstd = pre_rstd;
            if (t != wave) row_mean_rstd(m, mean, rstd);  // (more m-tiles than waves: the later ones pay the loads here)
            const f32x4 b = pre_bias;
            const f32x4 cv = pre_cv;
            float y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[r] = rstd * (v[r] - mean * cv[r]) + b[r];
                if constexpr (EPI == MMISS_EPI_LNFOLD_QGELU_BF16) y[r] = quick_gelu(y[r]);
            }
            u32x2 pk;
            pk[0] = pack_bf16x2(y[0], y[1]);
            pk[1] = pack_bf16x2(y[2], y[3]);
            *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(ep.out) + (size_t)m * ep.ldo + n) = pk;
        } else if constexpr (EPI == MMISS_EPI_BIAS_RESID_F32) {
            v += pre_bias;
            float* p = reinterpret_cast<float*>(ep.out) + (size_t)m * ep.ldo + n;
            v = ((t == wave) ? pre_res : *reinterpret_cast<const f32x4*>(p)) + v;
            *reinterpret_cast<f32x4*>(p) = v;
            if (ep.xb_out) {  // bf16 copy of the new residual rows: the A operand of the next folded GEMM
                u32x2 pk;
                pk[0] = pack_bf16x2(v[0], v[1]);
                pk[1] = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(ep.xb_out) + (size_t)m * ep.ldo + n) = pk;
            }
        } else {  // MMISS_EPI_PATCH_F32
            const int img = m / ep.p0, pt = m - img * ep.p0;
            const f32x4 pos = *reinterpret_cast<const f32x4*>(ep.aux + (size_t)(1 + pt) * ep.ldo + n);
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + ((size_t)img * ep.p1 + 1 + pt) * ep.ldo + n) = v + pos;
        }
    }
}

// Whether launch_gemm should take the skinny path for `mv` valid rows.
// Whether launch_gemm should take the skinny path for `mv` valid rows: up to 128 rows always, up to 320 rows when the
// output is narrow (N <= 1024: out-proj / FC2 / projection head, where the tiled kernel has only 12-24 workgroups;
// ViT-L/14 at one image = 257 rows: out-proj 9.5 vs 15.1 us, FC2 28 vs 48 us, but QKV / FC1 18-23 vs 12-13 us).
// Above that the tiled kernel takes over, with split-K for the long-K GEMM (whole encode at 400 rows: 0.88 vs 0.94 ms).
static inline bool gemm_skinny_ok(int epi, int mv, int N, int K, const GemmEpi& ep) {
    if (mmiss_option("gemm_skinny", 1) == 0) return false;
    const int forced = mmiss_option("gemm_skinny_max_m", 0);
    const int max_m = forced > 0 ? forced : (N <= 1024 ? 320 : 128);
    const bool fold = epi == MMISS_EPI_LNFOLD_BF16 || epi == MMISS_EPI_LNFOLD_QGELU_BF16;  // (launch_gemm_skinny_fold only)
    if ((ep.stats_out || ep.xb_out || fold) && !ep.stats16) return false;  // the tiled kernels' 64-column statistics layout
    return mv >= 1 && mv <= max_m && (N % 16) == 0 && (K % 128) == 0 && ((epi >= 0 && epi <= 4) || fold);
}

template <int EPI, int MT, int NW>
static int launch_gemm_skinny_nw(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int mv, int N, int K) {
    constexpr int U = MT <= 1 ? 8 : MT <= 2 ? 8 : MT <= 4 ? 6 : MT <= 8 ? 3 : 2;
    constexpr int LDS = NW * MT * 64 * 16;
    static_assert(LDS <= 64 * 1024, "reduction buffer must fit the default dynamic LDS limit");
    const int groups = ((mv + 15) / 16 + MT - 1) / MT;
    hipLaunchKernelGGL((gemm_skinny_kernel<EPI, MT, NW, U>), dim3(N / 16, groups), dim3(NW * 64), LDS, st,
                       reinterpret_cast<const __bf16*>(A), reinterpret_cast<const __bf16*>(W), mv, N, K, ep);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// waves per workgroup = K-slices: long K (FC2: 3072) is cut 8 ways so that a wave walks 12 k-steps (two trips), not 24
template <int EPI, int MT>
static int launch_gemm_skinny_inst(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int mv, int N, int K) {
    if constexpr (MT <= 8) {
        if (K >= 2048 && (K % 256) == 0) return launch_gemm_skinny_nw<EPI, MT, 8>(st, A, W, ep, mv, N, K);
    }
    return launch_gemm_skinny_nw<EPI, MT, 4>(st, A, W, ep, mv, N, K);
}

template <int EPI>
static int launch_gemm_skinny_mt(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int mv, int N, int K) {
    const int tiles = (mv + 15) / 16;
    // m-tiles per workgroup (MT): grid = N/16 x ceil(tiles/MT). A workgroup moves (1 + MT) x 16 x K operand elements
    // through its CU's L1 - the bound of this kernel - and the grid runs in max(1, workgroups/256) rounds, so pick the MT
    // with the smallest (1 + MT) x rounds (tools/gemm_skinny_bench.py: out-proj M=50: MT=1 3.7 us vs MT=4 5.7 us;
    // FC1 M=50: MT=4 5.5 us vs MT=1 7.5 us).
    int per = mmiss_option("gemm_skinny_mt", 0);
    if (per != 1 && per != 2 && per != 4 && per != 8 && per != 16) {
        double best = 1e30;
        for (int mt = 1; mt <= 16; mt *= 2) {
            const double wgs = (double)(N / 16) * ((tiles + mt - 1) / mt);
            // A workgroup is latency, not bandwidth: ~2.4 us of fixed cost + ~0.67 us per 16-row operand tile it moves
            // (out-proj at 50 rows: MT = 1 3.7 us, MT = 4 5.7 us), and a partly filled round costs a whole one at every K —
            // round 3, rocprofv3 of one request: QKV as 288 workgroups (MT = 2) 8.9 us = two rounds of 4.4; as 144
            // workgroups (MT = 4) one round. (Round 2 counted fractional rounds for short K.)
            const double rounds = (double)(int64_t)((wgs + 255.0) / 256.0);
            const double cost = (K >= 2048 ? (1.0 + mt) : (3.6 + 1.0 + mt)) * rounds;
            if (cost < best - 1e-9) { best = cost; per = mt; }
            if (mt >= tiles) break;
        }
    }
    switch (per) {
        case 1: return launch_gemm_skinny_inst<EPI, 1>(st, A, W, ep, mv, N, K);
        case 2: return launch_gemm_skinny_inst<EPI, 2>(st, A, W, ep, mv, N, K);
        case 4: return launch_gemm_skinny_inst<EPI, 4>(st, A, W, ep, mv, N, K);
        case 8: return launch_gemm_skinny_inst<EPI, 8>(st, A, W, ep, mv, N, K);
        default: return launch_gemm_skinny_inst<EPI, 16>(st, A, W, ep, mv, N, K);
    }
}

static int launch_gemm_skinny(hipStream_t st, int epi, const void* A, const void* W, const GemmEpi& ep, int mv, int N, int K) {
    switch (epi) {
        case MMISS_EPI_F32: return launch_gemm_skinny_mt<MMISS_EPI_F32>(st, A, W, ep, mv, N, K);
        case MMISS_EPI_BIAS_BF16: return launch_gemm_skinny_mt<MMISS_EPI_BIAS_BF16>(st, A, W, ep, mv, N, K);
        case MMISS_EPI_BIAS_QGELU_BF16: return launch_gemm_skinny_mt<MMISS_EPI_BIAS_QGELU_BF16>(st, A, W, ep, mv, N, K);
        case MMISS_EPI_BIAS_RESID_F32: return launch_gemm_skinny_mt<MMISS_EPI_BIAS_RESID_F32>(st, A, W, ep, mv, N, K);
        case MMISS_EPI_LNFOLD_BF16: return launch_gemm_skinny_mt<MMISS_EPI_LNFOLD_BF16>(st, A, W, ep, mv, N, K);
        case MMISS_EPI_LNFOLD_QGELU_BF16: return launch_gemm_skinny_mt<MMISS_EPI_LNFOLD_QGELU_BF16>(st, A, W, ep, mv, N, K);
        default: return launch_gemm_skinny_mt<MMISS_EPI_PATCH_F32>(st, A, W, ep, mv, N, K);
    }
}

// Folded-LayerNorm GEMM on the skinny kernel (one request at a time: the LayerNorm launches disappear): A = bf16(x) rows,
// W = gamma-folded weights, ep.aux = c, ep.bias = b', ep.ln_stats = [M][K/16][2] partial sums, ep.ln_parts = K / 16.
static int launch_gemm_skinny_fold(hipStream_t st, int epi, const void* A, const void* W, const GemmEpi& ep, int mv, int N, int K) {
    if (!ep.stats16 || !ep.ln_stats || !ep.aux || !ep.bias || ep.ln_parts * 16 != K || (ep.ln_parts & 1) ||
        !gemm_skinny_ok(epi, mv, N, K, ep))
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm_skinny_fold: M=%d N=%d K=%d parts=%d", mv, N, K, ep.ln_parts);
    MM_PROF(epi == MMISS_EPI_LNFOLD_BF16 ? "gemm_skinny_lnfold_bias" : "gemm_skinny_lnfold_qgelu", st, gemm_flops(mv, N, K),
            2.0 * ((double)mv * K + (double)N * K) + 2.0 * mv * N);
    return launch_gemm_skinny(st, epi, A, W, ep, mv, N, K);
}

// (sum, sumsq) of every 16-column slice of the residual rows + their bf16 copy: the entry point of the skinny folded mode
// (afterwards the skinny residual GEMMs' epilogues keep both up to date). One wave per row.
static __global__ __launch_bounds__(256) void skinny_row_stats16_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                                uint16_t* __restrict__ xb, int M, int d) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    for (int c = lane * 4; c < d; c += 256) {   // 4 lanes share a 16-column slice
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * d + c);
        u32x2 pk;
        pk[0] = pack_bf16x2(v[0], v[1]);
        pk[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(xb + (size_t)r * d + c) = pk;
        float s = (v[0] + v[1]) + (v[2] + v[3]);
        float q = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        s += __shfl_xor(s, 1); q += __shfl_xor(q, 1);
        s += __shfl_xor(s, 2); q += __shfl_xor(q, 2);
        if ((lane & 3) == 0) {
            float* o = stats + ((size_t)r * (d >> 4) + (c >> 4)) * 2;
            o[0] = s;
            o[1] = q;
        }
    }
}

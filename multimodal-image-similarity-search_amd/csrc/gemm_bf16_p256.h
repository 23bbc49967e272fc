// gemm_bf16_p256.h — PERSISTENT form of the 256x256x64 phase-pipelined bf16 GEMM (gemm_bf16_256.h) for the wide GEMMs
// of the towers (QKV, FC1: N >= 1536), same contract C[M,N] = A[M,K] * W[N,K]^T + fused epilogue.
//
// Why: as one workgroup per tile the 256x256 kernel pays, per tile, a pipeline fill from cold, a drain,
// and an epilogue that goes through the staging LDS — and at 12 800 rows every workgroup is in its epilogue at the same
// time, so nothing hides the stores (QKV 55 us for 2 rounds of a 20 us K loop; FC1 could not use the tile at all).
// Here the grid is one workgroup per CU (256) and the workgroup OWNS its tile schedule:
//   * its tiles form ONE continuous K-tile stream (the structure of gemm256_strip_kernel): the loads of the next tile's
//     first two K-tiles are issued by the ordinary t+1 / t+2 prefetch of the current tile's last two K-tiles;
//   * the epilogue runs from REGISTERS while those loads fly: LayerNorm fold / bias / QuickGELU, bf16 pack, two
//     v_permlane16_swap per 16x32 block so that every lane owns 8 consecutive columns, and 16-byte global stores that
//     are never waited for (64-byte row segments, two adjacent instructions per 128-byte line). No LDS traffic, so the
//     staging buffers stay live across the tile boundary;
//   * everything the epilogue needs from memory arrives by LDS-DMA as well — the tile's 256 bias / c values (one 4-byte
//     piece per wave) and, in the folded-LayerNorm mode, the raw (sum, sumsq) partials of its 256 rows (K/256 16-byte
//     pieces per wave), finalised to (mean, rstd) by 256 threads in a wait-free phase of K-tile 2 — because an ordinary
//     load beside LDS-DMA makes hipcc drain the whole pipeline (guide §5, trap (b));
//   * the stores and those pieces count in vmcnt (in issue order, guide: `s_waitcnt vmcnt(N)`), so the four counted waits
//     that follow an epilogue allow P256_EX more operations in flight; from the fifth wait on every such operation is
//     older than the slot being waited for and the plain count applies again;
//   * tile order: XCD x (blockIdx % 8) works on 32 logically consecutive tiles per round, 8 row blocks x 4 column tiles
//     in the banded order of tile_order() (12 operand panels per XCD and round instead of ~3 + all of W); the tiles of
//     the last, partial round are dealt round-robin over the XCDs.
#pragma once
#include "gemm_bf16_256.h"

#define P256_BC (8 * G256_SLOT)       // 2 x 2 KB: bias [256] f32 + c [256] f32 of the current / next tile
#define P256_TABLE (P256_BC + 4096)   // 256 x (mean, rstd)
#define P256_RAW (P256_TABLE + 2048)  // 256 rows x K/64 x (sum, sumsq) f32, as they lie in memory

template <int EPI, int XP>  // XP = 16-byte statistic pieces per wave and tile (K / 256 in the folded modes, else 0)
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W, int M,
                                                          int N, int K, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef bf16x8 frag;
    constexpr bool FOLD = (EPI == MMISS_EPI_LNFOLD_BF16 || EPI == MMISS_EPI_LNFOLD_QGELU_BF16);
    constexpr bool GELU = (EPI == MMISS_EPI_BIAS_QGELU_BF16 || EPI == MMISS_EPI_LNFOLD_QGELU_BF16);
    static_assert(FOLD == (XP > 0), "statistic pieces exist exactly in the folded modes");
    constexpr int EX = 16 + 1 + XP;  // vector-memory operations of a wave between two tiles' K streams
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nbm = M >> 8, nbn = N >> 8;
    const int nt = K / GEMM_BK;

    // ---- this workgroup's tile list: ordinal i -> logical tile -> (bm, bn)
    const int T = nbm * nbn, G = gridDim.x;
    const bool xs = (G & 7) == 0 && T >= G;
    const int full = xs ? T / G : 0, rem = T - full * G;
    const int bx = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const int mine = xs ? full + ((bj * 8 + bx) < rem ? 1 : 0) : 1;
    auto tile_of = [&](int i, int& bm, int& bn) {
        int L;
        if (xs) L = (i < full) ? i * G + bx * (G >> 3) + bj : full * G + bj * 8 + bx;
        else L = xcd_remap(blockIdx.x, T);
        tile_order(L, nbm, nbn, ep.m_fast, bm, bn);
    };

    const int r_in = lane >> 3, p = lane & 7;
    const int src_chunk = (p ^ r_in) * 8;
    const int a_row0 = (wave >> 2) * 128 + (wave & 3) * 16 + r_in;
    const int w_row0 = (wave >> 1) * 64 + (wave & 1) * 16 + r_in;
    const size_t a_lane = (size_t)a_row0 * K + src_chunk, w_lane = (size_t)w_row0 * K + src_chunk;
    // stage slot `which` (0 = A m0, 1 = A m1, 2 = W n0, 3 = W n1) of buffer b from K-tile kt of the tile at (Ab, Wb)
    auto stage = [&](int which, int b, const __bf16* Ab, const __bf16* Wb, int kt, bool live) {
        if (!live) return;
        char* dst = smem + (b * 4 + which) * G256_SLOT + wave * 2048;
        if (which < 2) {
            const __bf16* src = Ab + a_lane + (size_t)(which * 64) * K + (size_t)kt * GEMM_BK;
            glds16(src, dst);
            glds16(src + (size_t)8 * K, dst + 1024);
        } else {
            const __bf16* src = Wb + w_lane + (size_t)((which - 2) * 32) * K + (size_t)kt * GEMM_BK;
            glds16(src, dst);
            glds16(src + (size_t)8 * K, dst + 1024);
        }
    };
    // what the epilogue of tile (bm, bn) reads from memory, by LDS-DMA: 1 + XP pieces per wave
    auto stage_x = [&](int bm, int bn, int par) {
        const float* src = ((FOLD && wave >= 4) ? ep.aux : ep.bias) + bn * 256 + (wave & 3) * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + P256_BC + par * 2048 + wave * 256), 4, 0, 0);
        if constexpr (FOLD) {
            const char* st = reinterpret_cast<const char*>(ep.ln_stats) + (size_t)bm * 256 * (size_t)(XP * 32) + lane * 16;
#pragma unroll
            for (int q = 0; q < XP; ++q) glds16(st + (wave * XP + q) * 1024, smem + P256_RAW + (wave * XP + q) * 1024);
        }
    };

    frag am[4][2];
    frag wq[2][2][2];
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int a_off[4][2], w_off[2][2];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const int row = wm * 64 + mf * 16 + fr;
#pragma unroll
        for (int s = 0; s < 2; ++s) a_off[mf][s] = row * 128 + (((4 * s + fg) ^ (row & 7)) << 4);
    }
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
        const int row = wn * 32 + nf * 16 + fr;
#pragma unroll
        for (int s = 0; s < 2; ++s) w_off[nf][s] = row * 128 + (((4 * s + fg) ^ (row & 7)) << 4);
    }

#define P256_READ_A(b, mq)                                                                          \
    {                                                                                               \
        const char* sl = smem + ((b) * 4 + (mq)) * G256_SLOT;                                       \
        _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) {                                          \
            am[mf][0] = *reinterpret_cast<const frag*>(sl + a_off[mf][0]);                          \
            am[mf][1] = *reinterpret_cast<const frag*>(sl + a_off[mf][1]);                          \
        }                                                                                           \
    }
#define P256_READ_W(b, nq)                                                                          \
    {                                                                                               \
        const char* sl = smem + ((b) * 4 + 2 + (nq)) * G256_SLOT;                                   \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) {                                          \
            wq[nq][nf][0] = *reinterpret_cast<const frag*>(sl + w_off[nf][0]);                      \
            wq[nq][nf][1] = *reinterpret_cast<const frag*>(sl + w_off[nf][1]);                      \
        }                                                                                           \
    }
#define P256_MMA(mq, nq)                                                                            \
    {                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                              \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                               \
            _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)                                        \
                _Pragma("unroll") for (int mf = 0; mf < 4; ++mf)                                    \
                    acc[(nq) * 2 + nf][(mq) * 4 + mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(    \
                        wq[nq][nf][s], am[mf][s], acc[(nq) * 2 + nf][(mq) * 4 + mf], 0, 0, 0);      \
        __builtin_amdgcn_s_setprio(0);                                                              \
    }
// counted wait: 10 = the five slot loads (2 pieces each) that stay in flight; `post` = the stores and pieces of the
// previous tile's epilogue are younger than the slot waited for; !live2 = stream tail (fewer loads were issued)
#define P256_WAIT(live2, post)                                                           \
    {                                                                                    \
        if (!(live2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   \
        else if (post) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 + EX) : "memory");    \
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                           \
    }
#define P256_BARRIER()                         \
    {                                          \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    }

    // ---- stream state: the tile being computed (ordinal ti) and the tiles the positions t+1 / t+2 lie in
    int ti = 0, kt = 0, cbm, cbn;
    tile_of(0, cbm, cbn);
    const __bf16 *A0 = A + (size_t)cbm * 256 * K, *W0 = W + (size_t)cbn * 256 * K;  // tile of position t
    const __bf16 *A1 = A0, *W1 = W0, *A2 = A0, *W2 = W0;                            // tiles of positions t+1, t+2
    int o1 = 0, k1 = 1, o2 = 0, k2 = 2;                                             // (nt >= 4: both still in tile 0)
    const int dump_row = M - 1;  // rows >= m_valid are stored to the last pad row (the store COUNT must not depend on data)

    stage_x(cbm, cbn, 0);
    stage(0, 0, A0, W0, 0, true); stage(2, 0, A0, W0, 0, true); stage(3, 0, A0, W0, 0, true); stage(1, 0, A0, W0, 0, true);
    stage(0, 1, A0, W0, 1, true); stage(2, 1, A0, W0, 1, true); stage(3, 1, A0, W0, 1, true);
    P256_WAIT(true, false);
    P256_BARRIER();

    const int Tk = mine * nt;
    for (int t = 0; t < Tk; ++t) {
        const int b = t & 1;
        const bool live2 = (t + 2 < Tk), live1 = (t + 1 < Tk);
        const bool post0 = ti > 0 && kt == 0, post1 = ti > 0 && kt == 1;
        // phase 0: quadrant (m0, n0)
        P256_READ_A(b, 0);
        P256_READ_W(b, 0);
        stage(1, b ^ 1, A1, W1, k1, live1);
        P256_MMA(0, 0);
        P256_WAIT(live2, post0 || post1);
        P256_BARRIER();
        // phase 1: quadrant (m0, n1)
        P256_READ_W(b, 1);
        stage(0, b, A2, W2, k2, live2);
        P256_MMA(0, 1);
        P256_WAIT(live2, post0);
        P256_BARRIER();
        // phase 2: quadrant (m1, n1); no wait here — the place for the (mean, rstd) table of this tile's rows
        P256_READ_A(b, 1);
        stage(2, b, A2, W2, k2, live2);
        if constexpr (FOLD) {
            if (kt == 2 && tid < 256) {
                const f32x4* st = reinterpret_cast<const f32x4*>(smem + P256_RAW + tid * (XP * 32));
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int q = 0; q < XP * 2; ++q) { const f32x4 v = st[q]; s1 += v[0] + v[2]; s2 += v[1] + v[3]; }
                const float kd = (float)K;
                const float mean = s1 / kd;
                const float var = fmaxf(s2 / kd - mean * mean, 0.f);
                float* tb = reinterpret_cast<float*>(smem + P256_TABLE);
                tb[2 * tid] = mean;
                tb[2 * tid + 1] = 1.0f / sqrtf(var + ep.ln_eps);
            }
        }
        P256_MMA(1, 1);
        P256_BARRIER();
        // phase 3: quadrant (m1, n0) — operands already in registers
        stage(3, b, A2, W2, k2, live2);
        P256_MMA(1, 0);
        P256_WAIT(live2, post0);
        P256_BARRIER();
        // advance the prefetch positions
        if (++k1 == nt) {
            k1 = 0; ++o1;
            if (o1 < mine) { int bm, bn; tile_of(o1, bm, bn); A1 = A + (size_t)bm * 256 * K; W1 = W + (size_t)bn * 256 * K; }
        }
        if (++k2 == nt) {
            k2 = 0; ++o2;
            if (o2 < mine) { int bm, bn; tile_of(o2, bm, bn); A2 = A + (size_t)bm * 256 * K; W2 = W + (size_t)bn * 256 * K; }
        }
        if (++kt == nt) {
            // ---- tile (cbm, cbn) is complete: epilogue from registers while the next tile's first K-tiles are in flight
            const char* bc = smem + P256_BC + (ti & 1) * 2048;
            f32x4 bias[4], cvec[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bias[i] = *reinterpret_cast<const f32x4*>(bc + (wn * 64 + i * 16 + 4 * fg) * 4);
                if constexpr (FOLD) cvec[i] = *reinterpret_cast<const f32x4*>(bc + 1024 + (wn * 64 + i * 16 + 4 * fg) * 4);
            }
            const float* tb = reinterpret_cast<const float*>(smem + P256_TABLE);
            uint16_t* outp = reinterpret_cast<uint16_t*>(ep.out);
            const int colb = cbn * 256 + wn * 64 + (fg >> 1) * 8 + (fg & 1) * 16;  // + pair * 32
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rl = wm * 128 + j * 16 + fr;
                const int m = cbm * 256 + rl;
                float mu = 0.f, rs = 1.f;
                if constexpr (FOLD) { mu = tb[2 * rl]; rs = tb[2 * rl + 1]; }
                uint32_t pk[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float y[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (FOLD) y[r] = rs * (acc[i][j][r] - mu * cvec[i][r]) + bias[i][r];
                        else y[r] = acc[i][j][r] + bias[i][r];
                        if constexpr (GELU) y[r] = quick_gelu(y[r]);
                    }
                    pk[i][0] = pack_bf16x2(y[0], y[1]);
                    pk[i][1] = pack_bf16x2(y[2], y[3]);
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                uint16_t* orow = outp + (size_t)(m < ep.m_valid ? m : dump_row) * ep.ldo + colb;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    // lanes of the odd 16-lane rows take the even rows' values of the second 16-column block and give
                    // their values of the first: every lane ends up with 8 consecutive columns
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
                    u32x4 v;
                    v[0] = s0[0]; v[1] = s1[0]; v[2] = s0[1]; v[3] = s1[1];
                    *reinterpret_cast<u32x4*>(orow + pr * 32) = v;
                }
            }
            kt = 0;
            ++ti;
            if (ti < mine) {
                tile_of(ti, cbm, cbn);
                stage_x(cbm, cbn, ti & 1);
            }
        }
    }
    (void)A0; (void)W0;
}
#undef P256_READ_A
#undef P256_READ_W
#undef P256_MMA
#undef P256_WAIT
#undef P256_BARRIER

template <int EPI, int XP>
static int launch_gemm256p_inst(hipStream_t st, const void* A, const void* W, const GemmEpi& ep_in, int M, int N, int K) {
    GemmEpi ep = ep_in;
    if (ep.m_fast == 0) ep.m_fast = mmiss_option("gemm_p256_band", 8);
    const int lds = P256_RAW + XP * 8192;
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm256p_kernel<EPI, XP>), lds));
    const int T = (M / 256) * (N / 256);
    const int grid = T >= 256 ? 256 : T;
    hipLaunchKernelGGL((gemm256p_kernel<EPI, XP>), dim3(grid), dim3(512), lds, st, reinterpret_cast<const __bf16*>(A),
                       reinterpret_cast<const __bf16*>(W), M, N, K, ep);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// can this GEMM run on the persistent kernel? (bf16 output epilogues; the folded forms need K = 512 or 768: the raw
// statistics of a tile must fit beside the staging buffers)
static inline bool gemm256p_ok(int epi, int M, int N, int K) {
    if (M <= 0 || (M % 256) || N <= 0 || (N % 256) || K < 256 || (K % 256)) return false;
    if (epi == MMISS_EPI_LNFOLD_BF16 || epi == MMISS_EPI_LNFOLD_QGELU_BF16) return K == 512 || K == 768;
    return epi == MMISS_EPI_BIAS_BF16 || epi == MMISS_EPI_BIAS_QGELU_BF16;
}

// M = rows padded to 256 (the output and, in the folded modes, ep.ln_stats must hold M rows); rows >= ep.m_valid are
// computed but land in row M - 1.
static int launch_gemm256p(hipStream_t st, int epi, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K) {
    if (!gemm256p_ok(epi, M, N, K)) MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm256p: epi=%d M=%d N=%d K=%d", epi, M, N, K);
    const bool fold = epi == MMISS_EPI_LNFOLD_BF16 || epi == MMISS_EPI_LNFOLD_QGELU_BF16;
    if (!ep.out || !ep.bias || (fold && (!ep.ln_stats || !ep.aux || ep.ln_parts * 64 != K)))
        MM_FAIL(MMISS_ERR_ARG, "gemm256p: missing operand (fold=%d parts=%d)", (int)fold, ep.ln_parts);
    if (ep.m_valid < M && ep.m_valid > M - 1) MM_FAIL(MMISS_ERR_ARG, "gemm256p: no pad row");
    static const char* names[] = {"", "gemm_bf16_bias", "gemm_bf16_bias_qgelu"};
    const int mv = ep.m_valid < M ? ep.m_valid : M;
    const double bytes = 2.0 * ((double)mv * K + (double)N * K) + 2.0 * (double)mv * N;
    MM_PROF(fold ? (epi == MMISS_EPI_LNFOLD_BF16 ? "gemm_bf16_lnfold_bias" : "gemm_bf16_lnfold_qgelu") : names[epi], st,
            gemm_flops(mv, N, K), bytes);
    switch (epi) {
        case MMISS_EPI_BIAS_BF16: return launch_gemm256p_inst<MMISS_EPI_BIAS_BF16, 0>(st, A, W, ep, M, N, K);
        case MMISS_EPI_BIAS_QGELU_BF16: return launch_gemm256p_inst<MMISS_EPI_BIAS_QGELU_BF16, 0>(st, A, W, ep, M, N, K);
        case MMISS_EPI_LNFOLD_BF16:
            return K == 768 ? launch_gemm256p_inst<MMISS_EPI_LNFOLD_BF16, 3>(st, A, W, ep, M, N, K)
                            : launch_gemm256p_inst<MMISS_EPI_LNFOLD_BF16, 2>(st, A, W, ep, M, N, K);
        default:
            return K == 768 ? launch_gemm256p_inst<MMISS_EPI_LNFOLD_QGELU_BF16, 3>(st, A, W, ep, M, N, K)
                            : launch_gemm256p_inst<MMISS_EPI_LNFOLD_QGELU_BF16, 2>(st, A, W, ep, M, N, K);
    }
}

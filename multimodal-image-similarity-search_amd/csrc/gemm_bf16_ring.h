// gemm_bf16_ring.h — ring-pipelined variant of the BM x 128 GEMM of gemm_bf16.h (same tiles, same 2 workgroups per
// CU, same epilogue), built to lift the main loop of the 128-row structure:
//   gemm16_kernel stages one 64-deep K-tile ahead and drains it (vmcnt(0) + barrier) every K-tile: its loop runs at
//   ~47 % MFMA utilisation (tools/gemm_ksweep.py slope), the 256^2 kernel with counted waits at ~69 %.
// Here the K dimension is cut into 32-deep stages kept in a 4-slot LDS ring (same LDS bytes as the 2 x 64-deep
// buffers): three stages are in flight, each stage is retired by a COUNTED s_waitcnt vmcnt(2 stages) right before
// the raw s_barrier that publishes it, the slot freed by that barrier is refilled at once, and the MFMA fragments
// are double-buffered in registers so the ds_reads of stage s+1 run under the 20 MFMAs of stage s.
//
// LDS image of a stage: rows of 64 B (32 bf16), 16 rows per 1-KiB LDS-DMA piece (lane -> row lane>>2, 16-byte slot
// lane&3). Rows r, r+4, r+8, r+12 share a 256-byte bank row, so the slot is swizzled by the row's quad:
// slot = chunk ^ F[(r>>2)&3], F = {0,2,3,1} — conflict-free for all four ds_read_b128 lane groups (swizzle on the
// source address and on the read, LDS destination linear: guide §5.4 rule 21).
#pragma once
#include "gemm_bf16.h"

#define GR_BK 32
#define GR_NS 4

// F = {0,2,3,1}
__device__ __forceinline__ int gr_f(int q) { return q == 0 ? 0 : (q == 1 ? 2 : (q == 2 ? 3 : 1)); }

__device__ __forceinline__ void gr_wait(int n) {
    switch (n) {
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <typename IN, int BM, int EPI>
__global__ __launch_bounds__(256, 2) void gemm16r_kernel(const IN* __restrict__ A, const IN* __restrict__ W, int M,
                                                         int N, int K, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename MfmaIn<IN>::frag frag;
    constexpr int JT = BM / 32;
    constexpr int A_BYTES = BM * 64, STAGE = (BM + GEMM_BN) * 64;
    constexpr int PA = BM / 16, P = PA + GEMM_BN / 16;  // LDS-DMA pieces per stage: A rows, then W rows
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbm = M / BM, nbn = N / GEMM_BN;
    const int nwg = nbm * nbn;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int bm, bn;
    tile_order(wg, nbm, nbn, ep.m_fast, bm, bn);
    const IN* Ab = A + (size_t)bm * BM * K;
    const IN* Wb = W + (size_t)bn * GEMM_BN * K;
    const int nst = K / GR_BK;

    // pieces of this wave: wave, wave+4, ... ; cnt = how many (the vmcnt unit of this wave)
    const int cnt = (P - wave + 3) / 4;
    const int prow = lane >> 2, pslot = lane & 3;
    const int src_chunk = (pslot ^ gr_f((prow >> 2) & 3)) * 8;  // elements
    auto stage = [&](int st) {
        char* base = smem + (st & (GR_NS - 1)) * STAGE;
        const size_t koff = (size_t)st * GR_BK + src_chunk;
#pragma unroll
        for (int i = 0; i < (P + 3) / 4; ++i) {
            const int piece = wave + 4 * i;
            if (piece < P) {
                if (piece < PA)
                    glds16(Ab + (size_t)(piece * 16 + prow) * K + koff, base + piece * 1024);
                else
                    glds16(Wb + (size_t)((piece - PA) * 16 + prow) * K + koff, base + A_BYTES + (piece - PA) * 1024);
            }
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    // fragment offsets inside a stage (row*64 + swizzled slot*16); the quad of a fragment row is (fr>>2)&3 because
    // every sub-tile base is a multiple of 16 rows
    const int fslot = (fg ^ gr_f((fr >> 2) & 3)) * 16;
    int a_off[JT], w_off[4];
#pragma unroll
    for (int j = 0; j < JT; ++j) a_off[j] = (wm * (BM / 2) + j * 16 + fr) * 64 + fslot;
#pragma unroll
    for (int i = 0; i < 4; ++i) w_off[i] = A_BYTES + (wn * 64 + i * 16 + fr) * 64 + fslot;

    f32x4 acc[4][JT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    frag wfX[4], afX[JT], wfY[4], afY[JT];

#define GR_READ(WF, AF, st)                                                                         \
    {                                                                                               \
        const char* sb = smem + ((st) & (GR_NS - 1)) * STAGE;                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) WF[i] = *reinterpret_cast<const frag*>(sb + w_off[i]); \
        _Pragma("unroll") for (int j = 0; j < JT; ++j) AF[j] = *reinterpret_cast<const frag*>(sb + a_off[j]); \
    }
#define GR_MMA(WF, AF)                                                                              \
    {                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                               \
            _Pragma("unroll") for (int j = 0; j < JT; ++j) acc[i][j] = MfmaIn<IN>::mma(WF[i], AF[j], acc[i][j]); \
        __builtin_amdgcn_s_setprio(0);                                                              \
    }
    // one pipeline step: stage s is in registers (CUR); retire stage s+1, publish it, refill the slot stage s held,
    // start reading s+1 into NXT, run the MFMAs of stage s
#define GR_STEP(s, CURW, CURA, NXTW, NXTA)                                                          \
    {                                                                                               \
        if ((s) + 1 < nst) {                                                                        \
            const int ahead = nst - 2 - (s);                                                        \
            gr_wait(cnt * (ahead > 2 ? 2 : ahead));                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave's reads of stage s are done */ \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            __builtin_amdgcn_s_barrier();                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            if ((s) + GR_NS < nst) stage((s) + GR_NS);                                              \
            GR_READ(NXTW, NXTA, (s) + 1);                                                           \
        }                                                                                           \
        GR_MMA(CURW, CURA);                                                                         \
    }

    // ---- prologue: stages 0..3 in flight, stage 0 retired and read
#pragma unroll
    for (int st = 0; st < GR_NS; ++st)
        if (st < nst) stage(st);
    {
        const int ahead = nst - 1;  // stages 1..3 may still be in flight
        gr_wait(cnt * (ahead > 3 ? 3 : ahead));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    GR_READ(wfX, afX, 0);

    int s = 0;
    for (; s + 1 < nst; s += 2) {
        GR_STEP(s, wfX, afX, wfY, afY);
        GR_STEP(s + 1, wfY, afY, wfX, afX);
    }
    if (s < nst) GR_STEP(s, wfX, afX, wfY, afY);
#undef GR_READ
#undef GR_MMA
#undef GR_STEP

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // every wave is done with the ring before it is reused by the epilogue patches
    gemm_epilogue<EPI, JT>(ep, acc, bm * BM + wm * (BM / 2), bn * GEMM_BN + wn * 64, smem + wave * EPI_PATCH_BYTES, lane);
}

template <typename IN, int BM, int EPI>
static int launch_gemm_ring_inst(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K) {
    constexpr int LDS = GR_NS * (BM + GEMM_BN) * 64;
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm16r_kernel<IN, BM, EPI>), LDS));
    const int nwg = (M / BM) * (N / GEMM_BN);
    hipLaunchKernelGGL((gemm16r_kernel<IN, BM, EPI>), dim3(nwg), dim3(256), LDS, st, reinterpret_cast<const IN*>(A),
                       reinterpret_cast<const IN*>(W), M, N, K, ep);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

template <int EPI>
static int launch_gemm_ring_bm(hipStream_t st, int bm, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K) {
    switch (bm) {
        case 128: return launch_gemm_ring_inst<__bf16, 128, EPI>(st, A, W, ep, M, N, K);
        case 160: return launch_gemm_ring_inst<__bf16, 160, EPI>(st, A, W, ep, M, N, K);
        case 192: return launch_gemm_ring_inst<__bf16, 192, EPI>(st, A, W, ep, M, N, K);
    }
    MM_FAIL(MMISS_ERR_ARG, "gemm ring: unsupported tile height %d", bm);
}

// same contract as launch_gemm (gemm_bf16.h)
static int launch_gemm_ring(hipStream_t st, int epi, int bm, const void* A, const void* W, const GemmEpi& ep, int M, int N,
                            int K) {
    if (bm == 0) bm = 128;
    if (M <= 0 || N <= 0 || K <= 0 || (M % bm) || (N % GEMM_BN) || (K % GR_BK))
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm ring: M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, bm, GEMM_BN, GR_BK);
    static const char* names[] = {"gemm_bf16_f32", "gemm_bf16_bias", "gemm_bf16_bias_qgelu", "gemm_bf16_bias_resid",
                                  "gemm_bf16_patch"};
    if (epi < 0 || epi > 4) MM_FAIL(MMISS_ERR_ARG, "gemm ring: bad epilogue %d", epi);
    const int out_elt = (epi == MMISS_EPI_BIAS_BF16 || epi == MMISS_EPI_BIAS_QGELU_BF16) ? 2 : 4;
    const int mv = ep.m_valid < M ? ep.m_valid : M;
    const double bytes = 2.0 * ((double)mv * K + (double)N * K) +
                         (double)out_elt * mv * N * (epi == MMISS_EPI_BIAS_RESID_F32 ? 2 : 1);
    MM_PROF(names[epi], st, gemm_flops(mv, N, K), bytes);
    switch (epi) {
        case MMISS_EPI_F32: return launch_gemm_ring_bm<MMISS_EPI_F32>(st, bm, A, W, ep, M, N, K);
        case MMISS_EPI_BIAS_BF16: return launch_gemm_ring_bm<MMISS_EPI_BIAS_BF16>(st, bm, A, W, ep, M, N, K);
        case MMISS_EPI_BIAS_QGELU_BF16: return launch_gemm_ring_bm<MMISS_EPI_BIAS_QGELU_BF16>(st, bm, A, W, ep, M, N, K);
        case MMISS_EPI_BIAS_RESID_F32: return launch_gemm_ring_bm<MMISS_EPI_BIAS_RESID_F32>(st, bm, A, W, ep, M, N, K);
        default: return launch_gemm_ring_bm<MMISS_EPI_PATCH_F32>(st, bm, A, W, ep, M, N, K);
    }
}

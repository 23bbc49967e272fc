// gemm_bf16_256.h — 256x256x64 tile, 8-wave, phase-pipelined variant of the 16-bit MFMA GEMM of gemm_bf16.h
// (same contract: C[M,N] = A[M,K] * W[N,K]^T + fused epilogue), for the wide GEMMs of the towers (QKV, FC1).
//
// Why a second structure: the 128-row kernel stages 32-40 KB per 32 MFMAs and drains its loads at one barrier per
// K-tile; its main loop tops out near 40 % of the MFMA peak (profiles/r01_gemm_pmc_summary.csv: 43 % of the wave
// cycles are issue stalls, 30 % waits). Here (guide §5 "256² 8-phase template" ideas, own schedule):
//   * 8 waves = 2 (M) x 4 (N), each owning 128 x 64 of the output -> 64 MFMAs per wave per K-tile for 8 LDS-DMA
//     loads (1:8 instead of 1:4) and 24 ds_read_b128 (instead of 32 per 64);
//   * a K-tile is consumed in 4 phases, one 64x32 quadrant of the wave's output each, in the order
//     (m0,n0) (m0,n1) (m1,n1) (m1,n0); the LDS image of a K-tile is 4 slots of 16 KB (the m0 / m1 halves of every
//     wave's activation rows, the n0 / n1 halves of every wave's weight rows), and a slot is dead as soon as its
//     quadrant phase has read it: its NEXT-BUT-ONE K-tile is staged into it two phases later;
//   * so 5 slot loads (10 global_load_lds per wave) are always in flight across the phase barriers, retired by a
//     COUNTED s_waitcnt vmcnt(10) — never 0 in steady state — and raw s_barrier (a __syncthreads() would drain them);
//   * reads of a slot happen one phase after the barrier that follows the wait retiring it (RAW), restaging at least
//     one barrier after its last read (WAR).
#pragma once
#include "gemm_bf16.h"

#define G256_SLOT 16384
#define G256_LDS (8 * G256_SLOT)

template <typename IN, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const IN* __restrict__ A, const IN* __restrict__ W, int M,
                                                         int N, int K, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename MfmaIn<IN>::frag frag;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nbm = M >> 8, nbn = N >> 8;
    const int nwg = nbm * nbn;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int bm, bn;
    tile_order(wg, nbm, nbn, ep.m_fast, bm, bn);
    const IN* Ab = A + (size_t)bm * 256 * K;
    const IN* Wb = W + (size_t)bn * 256 * K;
    const int nt = K / GEMM_BK;

    // ---- staging: slot `which` (0 = A m0, 1 = A m1, 2 = W n0, 3 = W n1) of buffer b with K-tile kt.
    // A wave stages slot rows 16*wave .. 16*wave+15 (two 8-row LDS-DMA pieces). Slot row r of an A slot is the
    // activation row (r>>6)*128 + mq*64 + (r&63) of the tile; of a W slot the weight row (r>>5)*64 + nq*32 + (r&31).
    const int r_in = lane >> 3, p = lane & 7;
    const int src_chunk = (p ^ r_in) * 8;  // XOR swizzle on the source; the LDS image stays lane-linear
    const int a_row0 = (wave >> 2) * 128 + (wave & 3) * 16 + r_in;  // + mq*64 + i*8
    const int w_row0 = (wave >> 1) * 64 + (wave & 1) * 16 + r_in;   // + nq*32 + i*8
    auto stage = [&](int which, int b, int kt) {
        if (kt >= nt) return;
        char* dst = smem + (b * 4 + which) * G256_SLOT + wave * 2048;
        const size_t koff = (size_t)kt * GEMM_BK + src_chunk;
        if (which < 2) {
            const IN* src = Ab + (size_t)(a_row0 + which * 64) * K + koff;
            glds16(src, dst);
            glds16(src + (size_t)8 * K, dst + 1024);
        } else {
            const IN* src = Wb + (size_t)(w_row0 + (which - 2) * 32) * K + koff;
            glds16(src, dst);
            glds16(src + (size_t)8 * K, dst + 1024);
        }
    };

    frag am[4][2];     // activation fragments of the current m-half: [m sub-tile][k step]
    frag wq[2][2][2];  // weight fragments of both n-halves:          [n half][n sub-tile][k step]
    f32x4 acc[4][8];   // [i = nq*2 + nf][j = mq*4 + mf]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets inside a slot (row*128 + swizzled 16-byte chunk), k step 0; step 1 = chunk + 4
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

#define G256_READ_A(b, mq)                                                                          \
    {                                                                                               \
        const char* sl = smem + ((b) * 4 + (mq)) * G256_SLOT;                                       \
        _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) {                                          \
            am[mf][0] = *reinterpret_cast<const frag*>(sl + a_off[mf][0]);                          \
            am[mf][1] = *reinterpret_cast<const frag*>(sl + a_off[mf][1]);                          \
        }                                                                                           \
    }
#define G256_READ_W(b, nq)                                                                          \
    {                                                                                               \
        const char* sl = smem + ((b) * 4 + 2 + (nq)) * G256_SLOT;                                   \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) {                                          \
            wq[nq][nf][0] = *reinterpret_cast<const frag*>(sl + w_off[nf][0]);                      \
            wq[nq][nf][1] = *reinterpret_cast<const frag*>(sl + w_off[nf][1]);                      \
        }                                                                                           \
    }
#define G256_MMA(mq, nq)                                                                            \
    {                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                              \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                               \
            _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)                                        \
                _Pragma("unroll") for (int mf = 0; mf < 4; ++mf)                                    \
                    acc[(nq) * 2 + nf][(mq) * 4 + mf] =                                             \
                        MfmaIn<IN>::mma(wq[nq][nf][s], am[mf][s], acc[(nq) * 2 + nf][(mq) * 4 + mf]); \
        __builtin_amdgcn_s_setprio(0);                                                              \
    }
#define G256_WAIT(full)                                                  \
    {                                                                    \
        if (full) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            \
    }
#define G256_BARRIER()                         \
    {                                          \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    }

    // ---- prologue: K-tile 0 completely, K-tile 1 except its m1 slot (staged by phase 0 of K-tile 0)
    stage(0, 0, 0); stage(2, 0, 0); stage(3, 0, 0); stage(1, 0, 0);
    stage(0, 1, 1); stage(2, 1, 1); stage(3, 1, 1);
    G256_WAIT(nt >= 2);  // retires A m0 / W n0 of K-tile 0 (the 5 younger slot loads may still fly)
    G256_BARRIER();

    for (int t = 0; t < nt; ++t) {
        const int b = t & 1;
        const bool full = (t + 2 < nt);  // the counted wait assumes the loads of K-tile t+2 were issued
        // phase 0: quadrant (m0, n0)
        G256_READ_A(b, 0);
        G256_READ_W(b, 0);
        stage(1, b ^ 1, t + 1);
        G256_MMA(0, 0);
        G256_WAIT(full);
        G256_BARRIER();
        // phase 1: quadrant (m0, n1)
        G256_READ_W(b, 1);
        stage(0, b, t + 2);
        G256_MMA(0, 1);
        G256_WAIT(full);
        G256_BARRIER();
        // phase 2: quadrant (m1, n1)
        G256_READ_A(b, 1);
        stage(2, b, t + 2);
        G256_MMA(1, 1);
        G256_BARRIER();
        // phase 3: quadrant (m1, n0) — operands already in registers
        stage(3, b, t + 2);
        G256_MMA(1, 0);
        G256_WAIT(full);
        G256_BARRIER();
    }

    if constexpr (EPI == MMISS_EPI_GROUPMAX_F32) {
        // same group numbering as the 128-row kernel: g = (64-column block index) * 4 + fg (groupmax_row decodes it)
        const int g = (bn * 4 + wn) * 4 + fg;
        const int n0 = bn * 256 + wn * 64 + 4 * fg;
        float* out = reinterpret_cast<float*>(ep.out);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (n0 + i * 16 + r < ep.p0) ? acc[i][j][r] : -INFINITY;
                    mx = fmaxf(mx, v);
                }
            const int m = bm * 256 + wm * 128 + j * 16 + fr;
            if (m < ep.m_valid) out[(size_t)m * ep.ldo + g] = mx;
        }
        return;
    } else {
        // (folded LayerNorm: the 4 waves of one m-half share the (mean, rstd) table of its 128 rows, behind the patches)
        gemm_epilogue<EPI, 8>(ep, acc, bm * 256 + wm * 128, bn * 256 + wn * 64, smem + wave * EPI_PATCH_BYTES, lane,
                              smem + 8 * EPI_PATCH_BYTES + wm * (128 * 8), wn, 4);
    }
}

// ------------------------------------------------------------------------------------------------
// Strip variant for the retrieval score GEMM (GROUPMAX epilogue only): with K = 512 a 256x256 tile is just 8 K-tiles,
// and with one workgroup per CU every tile paid its own pipeline fill (7 slot loads from cold), its drain (the last
// two K-tiles wait on vmcnt(0)) and a workgroup launch - about a quarter of the 18 us per tile. Here a workgroup walks
// `strip` consecutive 256-row N-tiles of the index for one 256-query M-tile as ONE continuous K-tile stream: the
// staging of the next N-tile's first K-tiles is issued by the ordinary t+1 / t+2 prefetch of the last two K-tiles of
// the current one, the group-max epilogue (registers and global stores only, no LDS) runs while those loads fly, and
// the accumulators are simply zeroed. Sibling workgroups (the other M-tiles of the same strip) are adjacent in launch
// order, so the index rows still come from HBM once and from L2 for the rest.
// FILTER = true (the second pass of the threshold-filtered selection, api_index.hip): instead of writing every group maximum,
// a lane APPENDS (maximum, group) to its query's candidate list when the maximum reaches the query's threshold tau[q] — the
// k'-th best group maximum of a sample of the index, a lower bound of the final k'-th best, so every group of the true
// top-k' passes. Appends are rare (~k' / sample fraction per query over the whole pass): the score matrix never touches
// HBM. Tiles [bn_begin, N/256) only (the sample owns the tiles before).
struct StripFilter {
    const float* tau;     // threshold of query m at tau[m * tau_stride] (-inf: keep everything)
    int tau_stride;
    int32_t* cnt;         // [M] entries appended so far (may run past cap: the excess is dropped and the query re-done)
    float* buf_s;         // [M][cap]
    int32_t* buf_g;       // [M][cap] group ids
    int cap;
    int bn_begin;
};

template <typename IN, bool FILTER = false>
__global__ __launch_bounds__(512, 2) void gemm256_strip_kernel(const IN* __restrict__ A, const IN* __restrict__ W, int M,
                                                               int N, int K, int strip, GemmEpi ep, StripFilter flt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename MfmaIn<IN>::frag frag;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nbm = M >> 8, nbn = N >> 8;
    const int bn_begin = FILTER ? flt.bn_begin : 0;
    const int nstrips = (nbn - bn_begin + strip - 1) / strip;
    const int wg = xcd_remap(blockIdx.x, nbm * nstrips);
    const int sidx = wg / nbm, bm = wg - sidx * nbm;   // m fastest: the M-tiles of one strip run side by side
    const int bn0 = bn_begin + sidx * strip;
    const int ntiles = (nbn - bn0 < strip) ? nbn - bn0 : strip;
    const IN* Ab = A + (size_t)bm * 256 * K;
    const IN* Wb = W + (size_t)bn0 * 256 * K;
    const int nt = K / GEMM_BK;
    const int T = ntiles * nt;  // K-tiles of the whole strip

    const int r_in = lane >> 3, p = lane & 7;
    const int src_chunk = (p ^ r_in) * 8;
    const int a_row0 = (wave >> 2) * 128 + (wave & 3) * 16 + r_in;
    const int w_row0 = (wave >> 1) * 64 + (wave & 1) * 16 + r_in;
    // stage slot `which` of buffer b with stream position (tile, kt); `live` = the position exists
    auto stage = [&](int which, int b, int tile, int kt, bool live) {
        if (!live) return;
        char* dst = smem + (b * 4 + which) * G256_SLOT + wave * 2048;
        const size_t koff = (size_t)kt * GEMM_BK + src_chunk;
        if (which < 2) {
            const IN* src = Ab + (size_t)(a_row0 + which * 64) * K + koff;
            glds16(src, dst);
            glds16(src + (size_t)8 * K, dst + 1024);
        } else {
            const IN* src = Wb + ((size_t)tile * 256 + w_row0 + (which - 2) * 32) * K + koff;
            glds16(src, dst);
            glds16(src + (size_t)8 * K, dst + 1024);
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

    float tau_r[8];  // FILTER: thresholds of this lane's 8 queries
    if constexpr (FILTER) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = bm * 256 + wm * 128 + j * 16 + fr;
            tau_r[j] = (m < ep.m_valid) ? flt.tau[(size_t)m * flt.tau_stride] : INFINITY;
        }
    }
    // stream positions of t+1 and t+2, advanced once per K-tile (no division in the loop)
    int kt1 = (nt > 1) ? 1 : 0, tile1 = (nt > 1) ? 0 : 1;
    int kt2 = (nt > 2) ? 2 : (2 % nt), tile2 = 2 / nt;
    // prologue: stream position 0 completely, position 1 except its m1 slot
    stage(0, 0, 0, 0, true); stage(2, 0, 0, 0, true); stage(3, 0, 0, 0, true); stage(1, 0, 0, 0, true);
    stage(0, 1, tile1, kt1, T > 1); stage(2, 1, tile1, kt1, T > 1); stage(3, 1, tile1, kt1, T > 1);
    G256_WAIT(T >= 2);
    G256_BARRIER();

    int kt = 0, tile = 0;
    for (int t = 0; t < T; ++t) {
        const int b = t & 1;
        const bool full = (t + 2 < T);
        const bool live1 = (t + 1 < T);
        // phase 0: quadrant (m0, n0)
        G256_READ_A(b, 0);
        G256_READ_W(b, 0);
        stage(1, b ^ 1, tile1, kt1, live1);
        G256_MMA(0, 0);
        G256_WAIT(full);
        G256_BARRIER();
        // phase 1: quadrant (m0, n1)
        G256_READ_W(b, 1);
        stage(0, b, tile2, kt2, full);
        G256_MMA(0, 1);
        G256_WAIT(full);
        G256_BARRIER();
        // phase 2: quadrant (m1, n1)
        G256_READ_A(b, 1);
        stage(2, b, tile2, kt2, full);
        G256_MMA(1, 1);
        G256_BARRIER();
        // phase 3: quadrant (m1, n0)
        stage(3, b, tile2, kt2, full);
        G256_MMA(1, 0);
        G256_WAIT(full);
        G256_BARRIER();
        if (++kt1 == nt) { kt1 = 0; ++tile1; }
        if (++kt2 == nt) { kt2 = 0; ++tile2; }
        if (++kt == nt) {
            // ---- this N-tile is complete: group maxima out (same numbering as gemm256_kernel), accumulators reset
            const int bn = bn0 + tile;
            const int g = (bn * 4 + wn) * 4 + fg;
            const int n0 = bn * 256 + wn * 64 + 4 * fg;
            float* out = reinterpret_cast<float*>(ep.out);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = (n0 + i * 16 + r < ep.p0) ? acc[i][j][r] : -INFINITY;
                        mx = fmaxf(mx, v);
                        acc[i][j][r] = 0.f;
                    }
                const int m = bm * 256 + wm * 128 + j * 16 + fr;
                if constexpr (FILTER) {
                    if (mx >= tau_r[j] && mx > -INFINITY) {  // (tau_r = +inf for pad queries; -inf maxima = all-pad groups)
                        const int pos = atomicAdd(flt.cnt + m, 1);
                        if (pos < flt.cap) {
                            flt.buf_s[(size_t)m * flt.cap + pos] = mx;
                            flt.buf_g[(size_t)m * flt.cap + pos] = g;
                        }
                    }
                } else {
                    if (m < ep.m_valid) out[(size_t)m * ep.ldo + g] = mx;
                }
            }
            kt = 0;
            ++tile;
        }
    }
}
#undef G256_READ_A
#undef G256_READ_W
#undef G256_MMA
#undef G256_WAIT
#undef G256_BARRIER

template <typename IN>
static int launch_gemm256_strip(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K,
                                int strip, const StripFilter* flt = nullptr) {
    if (M <= 0 || N <= 0 || K <= 0 || (M % 256) || (N % 256) || (K % GEMM_BK) || strip < 1)
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm256_strip: M=%d N=%d K=%d strip=%d", M, N, K, strip);
    const int nbn = N / 256;
    if (flt) {
        if (flt->bn_begin < 0 || flt->bn_begin >= nbn || !flt->tau || !flt->cnt || !flt->buf_s || !flt->buf_g || flt->cap <= 0)
            MM_FAIL(MMISS_ERR_ARG, "gemm256_strip: bad filter");
        MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm256_strip_kernel<IN, true>), G256_LDS));
        const int nwg = (M / 256) * ((nbn - flt->bn_begin + strip - 1) / strip);
        hipLaunchKernelGGL((gemm256_strip_kernel<IN, true>), dim3(nwg), dim3(512), G256_LDS, st, reinterpret_cast<const IN*>(A),
                           reinterpret_cast<const IN*>(W), M, N, K, strip, ep, *flt);
    } else {
        MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm256_strip_kernel<IN, false>), G256_LDS));
        const int nwg = (M / 256) * ((nbn + strip - 1) / strip);
        hipLaunchKernelGGL((gemm256_strip_kernel<IN, false>), dim3(nwg), dim3(512), G256_LDS, st, reinterpret_cast<const IN*>(A),
                           reinterpret_cast<const IN*>(W), M, N, K, strip, ep, StripFilter{});
    }
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

template <typename IN, int EPI>
static int launch_gemm256_inst(hipStream_t st, const void* A, const void* W, const GemmEpi& ep_in, int M, int N, int K) {
    GemmEpi ep = ep_in;
    if (ep.m_fast == 0) ep.m_fast = mmiss_option("gemm_band_256", 0);  // m-band tile order (tile_order), 0 = n fastest
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm256_kernel<IN, EPI>), G256_LDS));
    const int nwg = (M / 256) * (N / 256);
    hipLaunchKernelGGL((gemm256_kernel<IN, EPI>), dim3(nwg), dim3(512), G256_LDS, st, reinterpret_cast<const IN*>(A),
                       reinterpret_cast<const IN*>(W), M, N, K, ep);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// bf16 GEMM on the 256x256 tile: M, N multiples of 256, K a multiple of 64; epilogues F32 / BIAS_BF16 /
// BIAS_QGELU_BF16 / BIAS_RESID_F32 and the folded-LayerNorm pair LNFOLD_BF16 / LNFOLD_QGELU_BF16 (ep.ln_stats, ep.aux set).
static int launch_gemm256(hipStream_t st, int epi, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || (M % 256) || (N % 256) || (K % GEMM_BK))
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm256: M=%d N=%d K=%d must be multiples of 256/256/%d", M, N, K, GEMM_BK);
    const bool fold = epi == MMISS_EPI_LNFOLD_BF16 || epi == MMISS_EPI_LNFOLD_QGELU_BF16;
    static const char* names[] = {"gemm_bf16_f32", "gemm_bf16_bias", "gemm_bf16_bias_qgelu", "gemm_bf16_bias_resid"};
    if (!fold && (epi < 0 || epi > 3)) MM_FAIL(MMISS_ERR_ARG, "gemm256: bad epilogue %d", epi);
    if (fold && (!ep.ln_stats || !ep.aux || !ep.bias || ep.ln_parts * 64 != K))
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm256 fold: K=%d parts=%d", K, ep.ln_parts);
    const int out_elt = (epi == MMISS_EPI_BIAS_BF16 || epi == MMISS_EPI_BIAS_QGELU_BF16 || fold) ? 2 : 4;
    const int mv = ep.m_valid < M ? ep.m_valid : M;
    const double bytes = 2.0 * ((double)mv * K + (double)N * K) +
                         (double)out_elt * mv * N * (epi == MMISS_EPI_BIAS_RESID_F32 ? 2 : 1);
    MM_PROF(fold ? (epi == MMISS_EPI_LNFOLD_BF16 ? "gemm_bf16_lnfold_bias" : "gemm_bf16_lnfold_qgelu") : names[epi], st,
            gemm_flops(mv, N, K), bytes);
    switch (epi) {
        case MMISS_EPI_F32: return launch_gemm256_inst<__bf16, MMISS_EPI_F32>(st, A, W, ep, M, N, K);
        case MMISS_EPI_BIAS_BF16: return launch_gemm256_inst<__bf16, MMISS_EPI_BIAS_BF16>(st, A, W, ep, M, N, K);
        case MMISS_EPI_BIAS_QGELU_BF16: return launch_gemm256_inst<__bf16, MMISS_EPI_BIAS_QGELU_BF16>(st, A, W, ep, M, N, K);
        case MMISS_EPI_LNFOLD_BF16: return launch_gemm256_inst<__bf16, MMISS_EPI_LNFOLD_BF16>(st, A, W, ep, M, N, K);
        case MMISS_EPI_LNFOLD_QGELU_BF16: return launch_gemm256_inst<__bf16, MMISS_EPI_LNFOLD_QGELU_BF16>(st, A, W, ep, M, N, K);
        default: return launch_gemm256_inst<__bf16, MMISS_EPI_BIAS_RESID_F32>(st, A, W, ep, M, N, K);
    }
}

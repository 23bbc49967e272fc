// attention_stream.h — K4 for the ViT-L/14 regime (257 keys = 16 x 16 + 1, non-causal, >= 256 (item, head) pairs): one persistent
// workgroup of 16 waves per CU walks its (item, head) pairs with the NEXT pair's K/V rows arriving by LDS-DMA while the
// current pair is computed.
//
// Why (round 4's ablation of attention_long_kernel, profiles/attention_l14_r04.txt): its two phases ADD — 21 us of K/V
// staging with nothing computing + 62 us of query tiles with nothing loading — because the two workgroups of a CU start
// together and stay in step; and its 17 query tiles (257 = 16 x 16 + 1) fall on 8 waves as 3 + 7 x 2: the workgroup lives
// three tile times for 2.1 of work. Prefetching the next pair through registers cost more occupancy than it hid.
// Here:
//   * two K/V images (2 x 72 KB) in one workgroup's LDS; the rows of pair n+1 are fetched by `buffer_load ... lds`
//     (no registers, 4-5 pieces of 1 KB per wave) right after the barrier that ends pair n-1, and are waited for — with the
//     query fragments of pair n+1, ordinary loads issued behind them — at the end of pair n's key loop, a whole compute phase
//     later, BEFORE the output stores (so that no wait ever sees a young store). One barrier per pair.
//   * both images are XOR-swizzled 128-byte rows (a piece of LDS-DMA is 1 KB of consecutive LDS bytes = 8 whole rows, so
//     padding is not an option: the permutation is applied to the SOURCE chunk each lane fetches): K as before
//     (chunk ^ (row & 7)); V by chunk ^ (((row >> 1) & 3) << 1): the transposing read is served 32 lanes = 8 rows x 32 B at
//     a time, and rows of equal parity (equal bank base) then lie in four different 32-byte windows — all 64 banks once.
//   * 16 waves take the 16 full query tiles, one each; the 17th tile — ONE query, the last patch — is split by KEYS: waves
//     0..8 run one key-pair step each for it, leave (offset, denominator, 64 outputs) in LDS, and wave 15 merges the nine
//     partial softmaxes behind the barrier. A pair costs every wave 9-10 steps instead of 18 or 27.
// The arithmetic of the 16 full tiles is attention_long_kernel's, instruction for instruction: their output bits are equal.
// The last query's sums are associated differently (nine partial sums merged): equal within the test's tolerance.
#pragma once
#include "common.h"
#include "gemm_fp8.h"

typedef float ats_f32x2 __attribute__((ext_vector_type(2)));

#define ATS_T 257            // keys = queries: 16 full tiles + the last patch
#define ATS_TP 288           // rows of an image (the PV product of the odd last key tile reads 32)
#define ATS_IMG (ATS_TP * 256)                 // one pair: K [TP][128 B] at 0, V [TP][128 B] behind it
#define ATS_PART_STRIDE 80   // bytes per (wave, lane group) partial: 16 outputs + offset + denominator (+ 2 pad)
#define ATS_PSZ (9 * 4 * ATS_PART_STRIDE)
#define ATS_PART (2 * ATS_IMG)                 // 2 areas of partial softmaxes of the last query
#define ATS_QTAIL (ATS_PART + 2 * ATS_PSZ)     // 2 x 256 B: the last query's row (twice), by one 4-byte piece
#define ATS_LDS (ATS_QTAIL + 512)

__device__ __forceinline__ float ats_max_over_lane_groups(float v) {   // max over lanes l, l ^ 16, l ^ 32, l ^ 48
    uint32_t u = __float_as_uint(v);
    auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = mm_max2(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
    u = __float_as_uint(v);
    auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return mm_max2(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
}

// one 1 KB (16 B per lane) / 256 B (4 B per lane) piece global -> LDS. Inline asm on purpose: behind the builtin hipcc puts an
// s_waitcnt vmcnt in front of the first transposing LDS read that follows (it cannot tell the two images apart) — a drain
// of the prefetch at the head of every pair. The waits are ours: vmcnt(0) before the barrier that hands an image over.
// m0 is a reserved register: hipcc ignores it in a clobber list (and says so), so it is NOT listed. What makes writing it safe
// is that nothing else in this kernel may depend on it — tests/test_abi_cpu.py disassembles the kernel and holds every
// instruction that touches m0 to be one of these asm statements' own (s_mov_b32 m0 / s_nop 0 / buffer_load ... lds).
__device__ __forceinline__ void ats_dma16(u32x4 srd, uint32_t lds, int vo) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(vo), "s"(srd) : "memory");
}
__device__ __forceinline__ void ats_dma4(u32x4 srd, uint32_t lds, int vo) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds), "v"(vo), "s"(srd) : "memory");
}

static inline bool attention_stream_ok(int B, int T, int H, bool causal) {
    // enough (item, head) pairs for every CU to stream a few: below that attention_long_kernel's query-tile splits fill the chip better
    // (option attention_stream_min_pairs, default 256: measured, profiles/attention_stream_r05.txt)
    return !causal && T == ATS_T && (int64_t)B * H >= mmiss_option("attention_stream_min_pairs", 256) && H > 0 &&
           (int64_t)T * 6 * H * 64 < (1ll << 31);
}

template <bool MXOUT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void attention_stream_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ ctx, int H,
                                                                int nitems, uint8_t* __restrict__ ctx8, uint8_t* __restrict__ ctxs,
                                                                int ld_s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = ATS_T, TP = ATS_TP, IMG = ATS_IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dmodel = H * 64, ld2 = 6 * dmodel;  // bytes per token row of qkv
    const int fr = lane & 15, fg = lane >> 4;
    const int tq = fr >> 2, tp = fr & 3;
    const float c_exp = 0.125f * 1.4426950408889634f;
    const float thr_raw = 8.0f / c_exp;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    // per-lane LDS offsets inside an image
    const int kb0 = fr * 128 + ((fg ^ (fr & 7)) << 4);
    const int kb1 = fr * 128 + (((4 + fg) ^ (fr & 7)) << 4);
    int vb[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vb[dt] = TP * 128 + (4 * fg + tq) * 128 + ((dt ^ (((4 * fg + tq) >> 1) & 3)) << 5) + tp * 8;
    // per-lane source offsets of a piece (8 rows x 128 B): LDS position (row = lane >> 3, chunk position = lane & 7) holds
    // the row's chunk cp ^ (row & 7) (K) / cp ^ (((row >> 1) & 3) << 1) (V)
    const int prow = lane >> 3, cp = lane & 7;
    const int klane = prow * ld2 + ((cp ^ prow) << 4) + 2 * dmodel;
    const int vlane = prow * ld2 + ((cp ^ (((prow >> 1) & 3) << 1)) << 4) + 4 * dmodel;
    constexpr int NP = (T + 7) >> 3;              // 33 pieces per operand hold a valid row
    u32x4 ones_raw = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    asm volatile("" : "+v"(ones_raw));
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_raw);

    // both images start as zeros: rows >= T of V must be finite (their probabilities are exact zeros); the out-of-range rows
    // of piece 32 are written as zeros (or not at all) by the buffer form
    for (int i = tid; i < ATS_LDS / 16; i += 1024) *reinterpret_cast<u32x4*>(smem + i * 16) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();

    // K/V rows of pair (b, h) into image `slot` (every wave 4-5 pieces), the last query's row by wave 15, and this wave's
    // query fragments (ordinary loads, issued LAST)
    auto stage = [&](int b, int h, int slot, u32x4 (&raw)[2]) {
        const uint16_t* base = qkv + (size_t)b * T * 3 * dmodel;
        const uint64_t a = (uint64_t)base;
        const u32x4 srd = {(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, (uint32_t)(T * ld2), 0x00020000u};
        const uint32_t img = lds0 + slot * IMG;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int j = wave + 16 * i;
            if (j < 2 * NP) {
                const bool isv = j >= NP;
                const int p = isv ? j - NP : j;
                ats_dma16(srd, img + (isv ? TP * 128 : 0) + p * 1024, (isv ? vlane : klane) + p * 8 * ld2 + h * 128);
            }
        }
        if (wave == 15) ats_dma4(srd, lds0 + ATS_QTAIL + slot * 256, 256 * ld2 + h * 128 + (lane & 31) * 4);
        const uint16_t* qrow = base + (size_t)(wave * 16 + fr) * 3 * dmodel + h * 64 + fg * 8;
        raw[0] = *reinterpret_cast<const u32x4*>(qrow);
        raw[1] = *reinterpret_cast<const u32x4*>(qrow + 32);
    };

    struct State {
        float m;
        f32x4 lacc;
        f32x4 oacc[4];
    };
    auto reset = [&](State& s) {
        s.m = -INFINITY;
        s.lacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) s.oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // one step = two key tiles (the 32 keys of one PV MFMA) at byte offsets koff (K) / voff (V) of the image; the odd last
    // tile (ODD: key 256 alone is valid) is a step of its own, its second half exact zeros
    auto step = [&](State& s, const bf16x8 (&qf)[2], const char* img, int koff, int voff, auto odd_tag) {
        constexpr bool ODD = decltype(odd_tag)::value;
        auto score_tile = [&](int o) -> f32x4 {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(img + kb0 + o);
            const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(img + kb1 + o);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[1], a, 0, 0, 0);
            return a;
        };
        f32x4 p0 = score_tile(koff), p1;
        if constexpr (!ODD) p1 = score_tile(koff + 2048);
        // the V fragments of this step are requested here: they land while the vector unit does the softmax
        bf16x8 vf[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(img + vb[dt] + voff));
            const bf16x4 v1 =
                __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(img + vb[dt] + voff + 2048));
            vf[dt][0] = v0[0]; vf[dt][1] = v0[1]; vf[dt][2] = v0[2]; vf[dt][3] = v0[3];
            vf[dt][4] = v1[0]; vf[dt][5] = v1[1]; vf[dt][6] = v1[2]; vf[dt][7] = v1[3];
        }
        float lm;
        if constexpr (ODD) {
            mm_mfma_settle("+v"(p0));
            p0[0] = fg == 0 ? p0[0] : -INFINITY;
            p0[1] = p0[2] = p0[3] = -INFINITY;
            lm = p0[0];
        } else {
            mm_mfma_settle("+v"(p0), "+v"(p1));
            lm = mm_max3(mm_max3(mm_max3(p0[0], p0[1], p0[2]), p0[3], p1[0]), p1[1], mm_max2(p1[2], p1[3]));
        }
        // lm is lane-local here: some lane's maximum passes the threshold exactly when the query's does, so the exchange over
        // the four lanes of a query is only needed in the (rare) step that raises an offset
        if (__any(lm > s.m + thr_raw)) {
            lm = ats_max_over_lane_groups(lm);
            const float mn = (lm > s.m + thr_raw) ? lm : s.m;
            const float alpha = __builtin_amdgcn_exp2f((s.m - mn) * c_exp);
            s.m = mn;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s.oacc[dt][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) s.lacc[r] *= alpha;
        }
        const float mc = s.m * c_exp;
        u32x4 praw;
        if constexpr (ODD) {
            praw[0] = pack_bf16x2(__builtin_amdgcn_exp2f(__builtin_fmaf(p0[0], c_exp, -mc)), 0.f);
            praw[1] = praw[2] = praw[3] = 0u;
        } else {
            const ats_f32x2 c2 = {c_exp, c_exp}, nmc2 = {-mc, -mc};
#pragma unroll
            for (int r = 0; r < 4; r += 2) {   // (packed fma: same rounding as the scalar form)
                const ats_f32x2 a0 = __builtin_elementwise_fma(ats_f32x2{p0[r], p0[r + 1]}, c2, nmc2);
                const ats_f32x2 a1 = __builtin_elementwise_fma(ats_f32x2{p1[r], p1[r + 1]}, c2, nmc2);
                p0[r] = __builtin_amdgcn_exp2f(a0[0]); p0[r + 1] = __builtin_amdgcn_exp2f(a0[1]);
                p1[r] = __builtin_amdgcn_exp2f(a1[0]); p1[r + 1] = __builtin_amdgcn_exp2f(a1[1]);
            }
            praw[0] = pack_bf16x2(p0[0], p0[1]);
            praw[1] = pack_bf16x2(p0[2], p0[3]);
            praw[2] = pack_bf16x2(p1[0], p1[1]);
            praw[3] = pack_bf16x2(p1[2], p1[3]);
        }
        const bf16x8 pf = __builtin_bit_cast(bf16x8, praw);
        s.lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, s.lacc, 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) s.oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt], pf, s.oacc[dt], 0, 0, 0);
    };
    // the 16 queries of a tile leave as bf16 rows or as MXFP8 (attention_long_kernel's epilogue)
    auto emit = [&](const f32x4 (&oacc)[4], float inv, int b, int h, int q) {
        if constexpr (MXOUT) {
            const size_t row = (size_t)b * T + (q < T ? q : 0);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                // (the block maximum of the UNNORMALISED outputs times 1 / l, and one multiplier 2^-e / l per value: the same bits as
                // normalising first — scaling by a power of two commutes with the rounding — for half the multiplies)
                float amax = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) amax = fmaxf(amax, fabsf(oacc[2 * blk + i][r]));
                amax = ats_max_over_lane_groups(amax) * inv;
                int e8;
                float sinv;
                mx_scale_of(amax, e8, sinv);
                sinv *= inv;
                if (q < T) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        *reinterpret_cast<uint32_t*>(ctx8 + row * dmodel + h * 64 + (2 * blk + i) * 16 + 4 * fg) =
                            pack_fp8x4(oacc[2 * blk + i][0] * sinv, oacc[2 * blk + i][1] * sinv, oacc[2 * blk + i][2] * sinv, oacc[2 * blk + i][3] * sinv);
                    if (fg == 0) ctxs[row * ld_s + mx_scale_offset(2 * h + blk)] = (uint8_t)e8;
                }
            }
        } else if (q < T) {
            uint16_t* orow = ctx + ((size_t)b * T + q) * dmodel + h * 64 + 4 * fg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                u32x2 pk;
                pk[0] = pack_bf16x2(oacc[dt][0] * inv, oacc[dt][1] * inv);
                pk[1] = pack_bf16x2(oacc[dt][2] * inv, oacc[dt][3] * inv);
                *reinterpret_cast<u32x2*>(orow + dt * 16) = pk;
            }
        }
    };
    // the last query of pair (b, h): nine partial softmaxes over disjoint key ranges (area `par`), merged by one wave
    auto merge_tail = [&](int b, int h, int par) {
        const char* pa = smem + ATS_PART + par * ATS_PSZ + fg * ATS_PART_STRIDE;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 9; ++w) M = fmaxf(M, *reinterpret_cast<const float*>(pa + w * 4 * ATS_PART_STRIDE + 64));
        f32x4 o[4];
        float l = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int w = 0; w < 9; ++w) {
            const char* pw = pa + w * 4 * ATS_PART_STRIDE;
            const float a = __builtin_amdgcn_exp2f((*reinterpret_cast<const float*>(pw + 64) - M) * c_exp);
            l = __builtin_fmaf(*reinterpret_cast<const float*>(pw + 68), a, l);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(pw + dt * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] = __builtin_fmaf(v[r], a, o[dt][r]);
            }
        }
        emit(o, 1.0f / l, b, h, 256 + fr);   // (only lane fr = 0 of each group holds a query < T)
    };

    int item = blockIdx.x;
    if (item >= nitems) return;   // (whole workgroup: uniform)
    int b = item / H, h = item - b * H;
    u32x4 qn[2];
    stage(b, h, 0, qn);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qn[0]), "+v"(qn[1])::"memory");
    bf16x8 qf[2];
    qf[0] = __builtin_bit_cast(bf16x8, qn[0]);
    qf[1] = __builtin_bit_cast(bf16x8, qn[1]);
    __syncthreads();
    int pb = -1, ph = 0;
#pragma unroll 1
    for (int n = 0; item < nitems; ++n) {
        const int slot = n & 1;
        const int next = item + gridDim.x;
        const int nb = next / H, nh = next - nb * H;
        const char* img = smem + slot * IMG;
        if (next < nitems) stage(nb, nh, slot ^ 1, qn);
        if (wave == 15 && pb >= 0) merge_tail(pb, ph, slot ^ 1);
        State s;
        if (wave < 9) {
            // every column of this tile is the last query (its row lies twice in the 256-byte area): lane group fr = 0 counts
            bf16x8 qt[2];
            qt[0] = *reinterpret_cast<const bf16x8*>(smem + ATS_QTAIL + slot * 256 + fg * 16);
            qt[1] = *reinterpret_cast<const bf16x8*>(smem + ATS_QTAIL + slot * 256 + 64 + fg * 16);
            reset(s);
            if (wave < 8) step(s, qt, img, wave * 4096, wave * 4096, std::false_type{});
            else step(s, qt, img, 8 * 4096, 8 * 4096, std::true_type{});
            if (fr == 0) {
                char* pw = smem + ATS_PART + slot * ATS_PSZ + (wave * 4 + fg) * ATS_PART_STRIDE;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(pw + dt * 16) = s.oacc[dt];
                *reinterpret_cast<float*>(pw + 64) = s.m;
                *reinterpret_cast<float*>(pw + 68) = s.lacc[0];
            }
        }
        reset(s);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) step(s, qf, img, ks * 4096, ks * 4096, std::false_type{});
        step(s, qf, img, 8 * 4096, 8 * 4096, std::true_type{});
        // the next pair's rows and query fragments were requested a whole key loop ago; the previous pair's stores are older
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(qn[0]), "+v"(qn[1])::"memory");
        emit(s.oacc, 1.0f / s.lacc[0], b, h, wave * 16 + fr);
        qf[0] = __builtin_bit_cast(bf16x8, qn[0]);
        qf[1] = __builtin_bit_cast(bf16x8, qn[1]);
        __syncthreads();
        pb = b; ph = h;
        b = nb; h = nh;
        item = next;
        if (item >= nitems && wave == 15) merge_tail(pb, ph, slot);
    }
}

template <bool MXOUT>
static int launch_attention_stream(hipStream_t st, const void* qkv, void* ctx, uint8_t* ctx8, uint8_t* ctxs, int ld_s, int B, int H) {
    const int nitems = B * H;
    // every workgroup the same number of pairs where the count allows (2048 pairs: 256 x 8)
    const int per = (nitems + 255) / 256;
    const int grid = (nitems + per - 1) / per;
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&attention_stream_kernel<MXOUT>), ATS_LDS));
    hipLaunchKernelGGL((attention_stream_kernel<MXOUT>), dim3(grid), dim3(1024), ATS_LDS, st, (const uint16_t*)qkv, (uint16_t*)ctx, H,
                       nitems, ctx8, ctxs, ld_s);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// gemm_fp8_p256.h — PERSISTENT 256x256x128 block-scaled fp8 (MXFP8) GEMM: the contract of gemm_fp8.h
// (C[M,N] = A8[M,K] * W8[N,K]^T with the BIAS_BF16 / QGELU_MXFP8 / BIAS_RESID_BF16 epilogues, the same bytes out) on the
// loop of gemm_bf16_p256.h. BASELINE configs[4] (ViT-L/14 fp8 encode, the reference's model: backend/app/utils.py:16-17):
// QKV, FC1, FC2 and the out-projection of the vision tower at batch 128 (33 024 padded rows).
//
// Why: gemm8_kernel is the BM x 128 tile with two barriers and a drained pipeline per K-tile — 1.5-1.6 PF,
// 30-32 % of the 5 PF fp8 roof, the structure the guide caps at ~36 %. A 128-element fp8 K-tile is the same 128 bytes per
// row as a 64-element bf16 one: the LDS image, the buffer-form LDS-DMA staging, the XOR swizzle, the fragment reads (two
// ds_read_b128 per 16-row sub-tile: k = 16 g .. + 15 and 64 + 16 g .. + 15 of lane group g — exactly the two chunks the bf16
// fragments take) and the phase structure of gemm256p_kernel carry over unchanged; a phase's 16 bf16 MFMAs of 4 passes
// become 8 v_mfma_scale_f32_16x16x128_f8f6f4 of 8 passes: the same matrix-pipe time per K-tile for twice the flops.
//   * one workgroup per CU walks its tiles as ONE K-tile stream (prefetch t+1 / t+2 runs across tile boundaries, the
//     epilogue out of the accumulators while the next tile's first K-tiles fly), wave halves one barrier apart;
//   * what is new is the third operand, the activations' E8M0 block scales (one byte per row and 32 k, laid out per row as
//     16 bytes per 512 k: dword c = k-block c of four consecutive K-tiles, gemm_fp8.h). They arrive by LDS-DMA as well (an
//     ordinary load beside LDS-DMA drains the pipeline): ONE 4-byte piece per wave at the head of every K-tile PAIR — pair p
//     of a tile brings the scale dwords of slot rows of A m(p & 1) for the 512-k group AFTER the one pair p lies in (for the
//     tile's last group: group 0 of the workgroup's next tile) into a two-parity ring of 2 x 4 KB, so the piece count per
//     pair is constant and every counted wait is an immediate: the four waits behind a piece allow one more operation in
//     flight, the fifth retires it, a barrier later (and a whole K-tile before the group starts) every wave may read it.
//     A lane reads its scale as ONE byte (ds_read_u8 at ring + row * 16 + 4 g + K-tile-in-group): OPSEL stays 0.
//   * epilogue inputs — bias[256] (waves 0-3) and the weights' per-channel scales (waves 4-7) of the next tile — one 4-byte
//     piece per wave, as in the bf16 kernel; the bf16 residual rows of BIAS_RESID_BF16 by ordinary loads at the head of the
//     epilogue (the fragment registers are dead there), transposed to the accumulator layout through the wave's 2 KB patch.
//   * a RAGGED last row block (ViT-L/14 at batch 128: 32 896 = 128 x 256 + 128 rows) is not a tile: 128 row blocks x N / 256
//     column tiles is a whole number of rounds of 256 CUs for every GEMM of the tower, 129 row blocks leave 4-16 tiles
//     for a last round that costs 0.8 of a tile time on 8-32 CUs with the rest of the chip idle (QKV 6.8 tile times for
//     6.05 of work, the N = 1024 GEMMs 2.8 for 2.02). The block's <= 128 valid rows are computed in front of the tile
//     stream by ALL workgroups instead: units of 16 rows x 64 (or 32) columns, one or two per workgroup, a unit's K-tiles
//     dealt over the eight waves (operands straight from global memory into registers, every load in flight at once — a
//     GEMV-shaped job), the eight partial sums added in wave order through LDS, the epilogue by wave 0. ~4 us instead of
//     13-44. (The partial sums make these rows differ from gemm8_kernel's in the last bits of the f32 sums; the tile rows
//     stay bit-identical.)
//   * K % 256 == 0 (round 6; was K % 512). A tile is K / 256 K-tile PAIRS; the scale ring works in 512-k groups of two pairs,
//     one piece per pair. K = 768 — every GEMM input of ViT-B/32, the metric's own model — is three pairs: a full group and a
//     HALF group (K-tiles 4, 5 = bytes 0, 1 of the row's second 16 scale bytes). Template parameter ODD = such a tile. The
//     half group's one pair must bring BOTH slots of the next tile's group 0, so an ODD kernel issues two pieces at the head
//     of EVERY pair — both slots of the group that follows the pair's own; the second pair of a full group repeats the first
//     one's pieces (the same bytes to the same place: 512 B per wave and tile) — which keeps the piece count per pair a
//     constant (every counted wait an immediate: the four waits behind the pieces allow two more operations in flight, the
//     fifth retires both) AND the loop's control flow the K % 512 kernel's: a third form of the pair for the half group alone
//     cost the register allocator 390 spilled registers. The K-tile counter that picks a lane's scale byte restarts at every
//     tile. Everything else — the ring's parity per group, the epilogues, the ragged pass — is the K % 512 kernel's; the
//     ODD = 0 instantiations are unchanged.
// K >= 512, K % 256 == 0, M, N % 256 == 0.
#pragma once
#include <type_traits>
#include "gemm_fp8.h"
#include "gemm_bf16_256.h"

#define Q256_BC (8 * G256_SLOT)        // 2 x 2 KB: bias [256] f32 + wscale [256] f32 of the current / next tile
#define Q256_TILES (Q256_BC + 4096)    // this workgroup's tile list: 64 x packed (bm, bn, half)
#define Q256_RING (Q256_TILES + 256)   // 2 x 4 KB: scale dwords [parity][A slot mq][slot row 128][k-block 4]
#define Q256_PATCH (Q256_RING + 8192)  // 8 waves x 2 KB: the epilogue's transposes (XT = 1: a wave's 1 KB of raw row statistics lands here during the K stream)
#define Q256_LDS (Q256_PATCH + 16384)
#define Q256_C16 Q256_LDS              // XT = 1: 2 x 512 B: c [256] f16 of the current / next tile
#define Q256_TABLE (Q256_C16 + 1024)   // XT = 1: 256 x (mean, rstd) of the tile's rows
#define Q256_LDS_FOLD (Q256_TABLE + 2048)

// max over lanes l, l ^ 16, l ^ 32, l ^ 48 (the four lane groups that hold one row's columns) by two swaps in the vector unit
// (v_permlane32_swap, v_permlane16_swap): __shfl_xor's ds_bpermute takes its address from the lane id, a register the
// compiler computes once and then keeps — or spills — across the whole K stream
__device__ __forceinline__ float q256_max_over_lane_groups(float v) {
    uint32_t u = __float_as_uint(v);
    auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
    u = __float_as_uint(v);
    auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
}

// XT: 0 plain; 1 = LayerNorm folded in (BIAS_BF16 / QGELU_MXFP8; Gemm8Args); 2 = BIAS_RESID_BF16 + MXFP8 copy + row statistics
__device__ __forceinline__ float q256_sum_over_lane_groups(float v) {   // sum over lanes l, l ^ 16, l ^ 32, l ^ 48: (l + l^32) + (l^16 + l^48)
    uint32_t u = __float_as_uint(v);
    auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
    u = __float_as_uint(v);
    auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
}
template <int CTRL>
__device__ __forceinline__ float q256_dpp(float v) {   // the value of the lane CTRL names: 0xB1 / 0x4E = lane ^ 1 / ^ 2 (quad permutes), 0x141 = 7 - lane within 8
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ f32x4 q256_halves_to_f32x4(u32x2 h) {   // four f16 -> f32
    // (scalar bit casts of the two words: a bit_cast of the vector's ELEMENTS to a two-half vector came out of hipcc as the first
    // word twice)
    const uint32_t w0 = h[0], w1 = h[1];
    return f32x4{(float)__builtin_bit_cast(_Float16, (uint16_t)(w0 & 0xffffu)), (float)__builtin_bit_cast(_Float16, (uint16_t)(w0 >> 16)),
                 (float)__builtin_bit_cast(_Float16, (uint16_t)(w1 & 0xffffu)), (float)__builtin_bit_cast(_Float16, (uint16_t)(w1 >> 16))};
}

template <int EPI, int XT = 0, int ODD = 0>
__global__ __launch_bounds__(512, 2) void gemm256p8_kernel(Gemm8Args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(EPI == MMISS_EPI8_BIAS_BF16 || EPI == MMISS_EPI8_QGELU_MXFP8 || EPI == MMISS_EPI8_BIAS_RESID_BF16, "epilogue");
    static_assert(XT == 0 || (XT == 1 && EPI != MMISS_EPI8_BIAS_RESID_BF16) || (XT == 2 && EPI == MMISS_EPI8_BIAS_RESID_BF16), "extension");
    static_assert(ODD == 0 || XT == 0, "the extensions exist for K = 1024 only");
    constexpr bool FOLD = XT == 1, MXQ = XT == 2;
    // vector-memory operations of a wave between two tiles' K streams: the epilogue's stores (16; MXQ: 5 x 8 + the statistics) + the
    // pieces of stage_x (bias / wscale; FOLD: + c + the raw statistics)
    constexpr int EX = (MXQ ? 41 : 16) + (FOLD ? 3 : 1);
    // An (unused) value of the accumulator register class makes the compiler select the MFMAs in their AGPR form: the 128
    // accumulators live in a[0:127], the 128 arch VGPRs are left to the 8-register operand tuples of the K = 128 MFMA, the
    // addresses and the epilogue. With everything in one file of 256 registers the allocator could not keep sixteen 8-tuples
    // beside 32 scattered 4-tuples (40-50 spilled registers) — and a spill is a vector-memory operation the counted waits do
    // not know about.
    { int agpr_form_; asm volatile("" : "=a"(agpr_form_)); }
#define Q256_PIN "+a"
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int M = g.M, N = g.N, K = g.K;
    const int nbm = (M >> 8) - (g.ragged > 0 ? 1 : 0), nbn = N >> 8;   // (a ragged last row block is not a tile)
    const int nt = K >> 7;          // K-tiles per tile, even (ODD = 0: a multiple of 4)
    const int npair = nt >> 1;      // ODD = npair & 1
    const int ngrp = (nt + 3) >> 2; // 512-k scale groups per tile (ODD: the last one is a half group of one pair)
    __builtin_assume(npair >= 2);   // (K >= 512: no zero-trip copies of the K loop, whose accumulator joins cost registers)
#ifdef MMISS_EXPERIMENTS
    // timing experiment (profiles/gemm_fp8_p256_r05.txt): do the workgroups' epilogues cost more because all 256 run them at once?
    if (g.stagger > 0 && (blockIdx.x & 1))
        for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
#endif

    // ---- the ragged last row block, in front of everything (no LDS-DMA is in flight yet: ordinary loads, ordinary waits)
    if (g.ragged > 0) {
        const int lane = tid & 63;
        const int fr = lane & 15, fg = lane >> 4;
        const int row0 = M - 256;                                   // first row of the block
        const int rgroups = (g.ragged + 15) >> 4;                   // 16-row groups with a valid row
        // columns per unit: the narrowest of 32 / 64 / 128 that leaves at most one unit per workgroup (the MXFP8 epilogue scales
        // a row per 64 columns: at least 64); ViT-L/14: N = 1024 -> 32 (256 units), 3072 -> 128 (192), 4096 -> 128 (256)
        int cw = EPI == MMISS_EPI8_QGELU_MXFP8 ? 64 : 32;
        while (cw < 128 && rgroups * (N / cw) > (int)gridDim.x) cw <<= 1;
        const int ncol = cw >> 4;                                   // 16-column MFMA tiles per unit (2, 4 or 8)
        const int units = rgroups * (N / cw);
        f32x4* red = reinterpret_cast<f32x4*>(smem);                // [wave][column tile][lane]: 8 x 8 x 64 x 16 B = 64 KB
        // NC = 16-column tiles per unit, a COMPILE-TIME count: with a run-time bound every load of the unrolled loops sat
        // behind its own branch and its own s_waitcnt vmcnt(0) — twenty dependent memory round trips per unit (guide, trap (c))
        auto ragged_units = [&](auto nc_tag) {
            constexpr int NC = decltype(nc_tag)::value;
            constexpr int NE = NC < 4 ? NC : 4;                     // column tiles per epilogue wave
            for (int u = blockIdx.x; u < units; u += gridDim.x) {
                const int rg = u % rgroups, cg = u / rgroups;
                const int m = row0 + rg * 16 + fr;                  // this lane's activation row (pad rows of the block are readable)
                const int n0 = cg * cw;
                f32x4 racc[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) racc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                const uint8_t* ap = g.A + (size_t)m * K + fg * 16;
                const uint8_t* sp = g.As + (size_t)m * g.ld_as + fg * 4;
                const uint8_t* wp = g.W + (size_t)(n0 + fr) * K + fg * 16;
                for (int kt = wave; kt < nt; kt += 8) {              // this wave's K-tiles of the unit
                    const u32x4 alo = *reinterpret_cast<const u32x4*>(ap + kt * 128);
                    const u32x4 ahi = *reinterpret_cast<const u32x4*>(ap + kt * 128 + 64);
                    const int scw = *reinterpret_cast<const int*>(sp + (kt >> 2) * 16);     // four K-tiles' scales of k-block fg
                    u32x4 wlo[NC], whi[NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        wlo[c] = *reinterpret_cast<const u32x4*>(wp + (size_t)c * 16 * K + kt * 128);
                        whi[c] = *reinterpret_cast<const u32x4*>(wp + (size_t)c * 16 * K + kt * 128 + 64);
                    }
                    const v8i32 af = __builtin_bit_cast(v8i32, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
                    const int scb = (int)((unsigned)scw >> (8 * (kt & 3))) & 0xff;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const v8i32 wf = __builtin_bit_cast(v8i32, __builtin_shufflevector(wlo[c], whi[c], 0, 1, 2, 3, 4, 5, 6, 7));
                        racc[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf, af, racc[c], 0, 0, 0, 0x7F7F7F7F, 0, scb);
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) red[(wave * 8 + c) * 64 + lane] = racc[c];
                // the epilogue's operands fly while the partial sums settle: wave e < NC / 4 (wave 0 alone for 32 columns) takes
                // the unit's 16-column tiles 4 e .. 4 e + 3
                const int c0 = wave * 4;
                const bool epi_wave = c0 < NC;
                const bool live = m < g.m_valid;
                f32x4 wsv[NE], bv[NE];
                [[maybe_unused]] f32x4 cv[NE];
                u32x2 oldv[NE];
                if constexpr (FOLD) {
                    // (mean, rstd) of the unit's 16 rows from the bf16 residual rows themselves (the producing GEMM's statistics
                    // cover tile rows only): wave w sums rows 2 w and 2 w + 1, 16 values per lane, and leaves them behind `red`
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const int row = row0 + rg * 16 + 2 * wave + rr;
                        const u32x4 h0 = *reinterpret_cast<const u32x4*>(g.x16 + (size_t)row * K + lane * 16);
                        const u32x4 h1 = *reinterpret_cast<const u32x4*>(g.x16 + (size_t)row * K + lane * 16 + 8);
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a = __uint_as_float(h0[e] << 16), b = __uint_as_float(h0[e] & 0xFFFF0000u);
                            const float c = __uint_as_float(h1[e] << 16), d = __uint_as_float(h1[e] & 0xFFFF0000u);
                            s1 += (a + b) + (c + d);
                            s2 += (a * a + b * b) + (c * c + d * d);
                        }
                        s1 = wave_sum(s1);
                        s2 = wave_sum(s2);
                        const float mean = s1 * (1.0f / 1024.0f);
                        const float var = fmaxf(s2 * (1.0f / 1024.0f) - mean * mean, 0.f);
                        if (lane == 0)
                            *reinterpret_cast<u32x2*>(smem + 65536 + (2 * wave + rr) * 8) =
                                u32x2{__float_as_uint(mean), __float_as_uint(1.0f / sqrtf(var + g.ln_eps))};
                    }
                }
                if (epi_wave) {
#pragma unroll
                    for (int c = 0; c < NE; ++c) {
                        const int n = n0 + (c0 + c) * 16 + 4 * fg;
                        wsv[c] = *reinterpret_cast<const f32x4*>(g.wscale + n);
                        bv[c] = *reinterpret_cast<const f32x4*>(g.bias + n);
                        if constexpr (FOLD) cv[c] = q256_halves_to_f32x4(*reinterpret_cast<const u32x2*>(g.c16 + n));
                        if constexpr (EPI == MMISS_EPI8_BIAS_RESID_BF16)
                            oldv[c] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(g.out) + (size_t)(live ? m : M - 1) * g.ldo + n);
                    }
                }
                __syncthreads();
                if (epi_wave) {
                    // y[c][r] = C[m = row fr of the group][n = n0 + (c0 + c) * 16 + 4 * fg + r]: the eight partial sums in wave order
                    f32x4 y[NE];
                    float amax = 0.f;
#pragma unroll
                    for (int c = 0; c < NE; ++c) {
                        f32x4 sum = red[(c0 + c) * 64 + lane];
#pragma unroll
                        for (int w = 1; w < 8; ++w) sum = sum + red[(w * 8 + c0 + c) * 64 + lane];
                        if constexpr (FOLD) {
                            const float mean = *reinterpret_cast<const float*>(smem + 65536 + fr * 8);
                            const float rstd = *reinterpret_cast<const float*>(smem + 65536 + fr * 8 + 4);
                            y[c] = (sum * wsv[c] - mean * cv[c]) * rstd + bv[c];
                        } else {
                            y[c] = sum * wsv[c] + bv[c];
                        }
                        if constexpr (EPI == MMISS_EPI8_QGELU_MXFP8) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                y[c][r] = quick_gelu(y[c][r]);
                                amax = fmaxf(amax, fabsf(y[c][r]));
                            }
                        }
                    }
                    const int nb = n0 + c0 * 16;   // first column of this wave's (up to) 64
                    if constexpr (EPI == MMISS_EPI8_QGELU_MXFP8) {      // (cw >= 64: four whole column tiles)
                        amax = q256_max_over_lane_groups(amax);
                        int e8;
                        float inv;
                        mx_scale_of(amax, e8, inv);
                        if (live) {
#pragma unroll
                            for (int c = 0; c < NE; ++c)
                                *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(g.out) + (size_t)m * g.ldo + nb + c * 16 + 4 * fg) =
                                    pack_fp8x4(y[c][0] * inv, y[c][1] * inv, y[c][2] * inv, y[c][3] * inv);
                            if (fg < 2) g.out_scale[(size_t)m * g.ld_os + mx_scale_offset((nb >> 5) + fg)] = (uint8_t)e8;
                        }
                    } else if (live) {
                        uint16_t* orow = reinterpret_cast<uint16_t*>(g.out) + (size_t)m * g.ldo + nb + 4 * fg;
#pragma unroll
                        for (int c = 0; c < NE; ++c) {
                            if constexpr (EPI == MMISS_EPI8_BIAS_RESID_BF16)
                                y[c] = f32x4{__uint_as_float(oldv[c][0] << 16), __uint_as_float(oldv[c][0] & 0xFFFF0000u),
                                             __uint_as_float(oldv[c][1] << 16), __uint_as_float(oldv[c][1] & 0xFFFF0000u)} + y[c];
                            u32x2 pk;
                            pk[0] = pack_bf16x2(y[c][0], y[c][1]);
                            pk[1] = pack_bf16x2(y[c][2], y[c][3]);
                            *reinterpret_cast<u32x2*>(orow + c * 16) = pk;
                            if constexpr (MXQ)
                                y[c] = f32x4{__uint_as_float(pk[0] << 16), __uint_as_float(pk[0] & 0xFFFF0000u), __uint_as_float(pk[1] << 16),
                                             __uint_as_float(pk[1] & 0xFFFF0000u)};
                        }
                    }
                    if constexpr (MXQ) {
                        // the MXFP8 copy of the new bf16 rows: NE / 2 whole 32-column blocks per epilogue wave (every lane takes part
                        // in the lane-group maxima; only live rows store). The rows' statistics are the consumer's business.
#pragma unroll
                        for (int b = 0; b < NE / 2; ++b) {
                            float amax = 0.f;
#pragma unroll
                            for (int c = 2 * b; c < 2 * b + 2; ++c)
#pragma unroll
                                for (int r = 0; r < 4; ++r) amax = fmaxf(amax, fabsf(y[c][r]));
                            amax = q256_max_over_lane_groups(amax);
                            int e8;
                            float inv;
                            mx_scale_of(amax, e8, inv);
                            if (live) {
#pragma unroll
                                for (int c = 2 * b; c < 2 * b + 2; ++c)
                                    *reinterpret_cast<uint32_t*>(g.q_out + (size_t)m * N + nb + c * 16 + 4 * fg) =
                                        pack_fp8x4(y[c][0] * inv, y[c][1] * inv, y[c][2] * inv, y[c][3] * inv);
                                if (fg == 0) g.q_scale[(size_t)m * g.ld_qs + mx_scale_offset((nb >> 5) + b)] = (uint8_t)e8;
                            }
                        }
                    }
                }
                __syncthreads();   // `red` is rewritten by the next unit, then handed to the tile stream
            }
        };
        if (ncol == 8) ragged_units(std::integral_constant<int, 8>{});
        else if (ncol == 4) ragged_units(std::integral_constant<int, 4>{});
        else if constexpr (EPI != MMISS_EPI8_QGELU_MXFP8) ragged_units(std::integral_constant<int, 2>{});
        // (no wait for the stores: they are older than every operation of the tile stream, so they retire first and the counted
        // waits never see them; the barrier above is what hands `red` to the staging slots)
    }

    // ---- tile list (gemm256p_kernel: rounds of G tiles of a banded global order; a last round of at most G/2 tiles in halves)
    const int T = nbm * nbn, G = gridDim.x;
    const int pb = xcd_remap(blockIdx.x, G);
    const int full_rounds = T / G, rem = T - full_rounds * G;
    const bool halves = full_rounds >= 1 && rem > 0 && 2 * rem <= G;
    const int nfull = full_rounds + ((!halves && pb < rem) ? 1 : 0);
    const int nhalf = (halves && pb < 2 * rem) ? 1 : 0;
    const int mine = nfull + nhalf;
    if (tid < mine && tid < 64) {
        const bool hf = tid >= nfull;
        const int L = hf ? full_rounds * G + (pb >> 1) : tid * G + pb;
        int bm_, bn_;
        tile_order(L, nbm, nbn, g.m_fast, bm_, bn_);
        reinterpret_cast<unsigned*>(smem + Q256_TILES)[tid] = (unsigned)bm_ | ((unsigned)bn_ << 16) | (hf ? (1u << 30) : 0u) |
                                                             ((hf && (pb & 1)) ? (1u << 31) : 0u);
    }
    __syncthreads();
    auto tile_of = [&](int i, int& bm, int& bn, int& half) {  // half: 0 whole tile, 1 / 2 = its m0 / m1 rows only
        const unsigned pk = (unsigned)__builtin_amdgcn_readfirstlane(((const __attribute__((address_space(3))) int*)(smem + Q256_TILES))[i]);
        bm = pk & 0xffff;
        bn = (pk >> 16) & 0x3fff;
        half = (pk >> 30) ? 1 + (int)(pk >> 31) : 0;
    };

    // ---- LDS-DMA sources, buffer form. Slot row r of an A slot is the activation row (r>>6)*128 + mq*64 + (r&63) of the tile,
    // of a W slot the weight row (r>>5)*64 + nq*32 + (r&31); a wave stages slot rows 16*wave .. +15 as two 8-row pieces of
    // 128-byte rows; XOR swizzle of the 16-byte chunk on the source side.
    // Everything the K loop derives from the lane id is recomputed (a few dozen vector instructions) behind every epilogue
    // from the hardware's lane count, not kept: held across the epilogue these ~10 registers were spilled and reloaded —
    // scratch loads behind an s_waitcnt vmcnt(0), i.e. a drain of the staging pipeline per tile.
    auto lane_now = [&]() -> int {   // (the mask comes out of an asm statement: the count is not hoisted out of the tile loop and kept)
        unsigned ones = ~0u;
        asm volatile("" : "+s"(ones));
        return (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
    };
    int a_vo, w_vo, s_vo;
    uint32_t ab[2], wb[2], sc_lane;
    auto lane_consts = [&]() {
        const int l = lane_now();
        const int r_in = l >> 3, p = l & 7;
        const int src_chunk = (p ^ r_in) * 16;
        a_vo = ((wave >> 2) * 128 + (wave & 3) * 16 + r_in) * K + src_chunk;
        w_vo = ((wave >> 1) * 64 + (wave & 1) * 16 + r_in) * K + src_chunk;
        // scale piece: lane = (slot row 16 wave + lane / 4, k-block lane & 3) of one A slot; the slot's rows in the scalar offset
        s_vo = ((wave >> 2) * 128 + (wave & 3) * 16 + (l >> 2)) * g.ld_as + (l & 3) * 4;
        const int fr = l & 15, fg = l >> 4;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            ab[s] = (wm * 64 + fr) * 128 + (((4 * s + fg) ^ (fr & 7)) << 4);
            wb[s] = (8 * G256_SLOT / 2) + (wn * 32 + fr) * 128 + (((4 * s + fg) ^ (fr & 7)) << 4);
        }
        // scale bytes: ring + parity * 4096 + mq * 2048 + (slot row wm*64 + mf*16 + fr) * 16 + fg * 4 + K-tile in group
        sc_lane = Q256_RING + (wm * 64 + fr) * 16 + fg * 4;
    };
    lane_consts();
    const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(g.A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(g.W), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdS = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(g.As), 0, 0x7fffffff, 0x00020000);
    const int row8 = 8 * K;  // bytes between a slot's two 8-row pieces
    const int stage_dst = wave * 2048;
#define Q256_SLOT(which, b) (((which) * 2 + (b)) * G256_SLOT)
#define Q256_BLDS(srd, vo, so, dst) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(dst), 16, vo, so, 0, 0)
#define Q256_BLDS4(srd, vo, so, dst) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(dst), 4, vo, so, 0, 0)
// LIVE = false: the slot is not read (the A m1 slot of a half tile): its pieces shrink to 4 bytes per lane, the COUNT stays
#define Q256_STAGE(which, b, oA, oW, LIVE)                                                                   \
    {                                                                                                       \
        char* dst_ = smem + Q256_SLOT(which, b) + stage_dst;                                                \
        if constexpr ((which) < 2) {                                                                        \
            const int so_ = (oA);                                                                           \
            if (LIVE) {                                                                                     \
                Q256_BLDS(srdA, a_vo, so_, dst_);                                                           \
                Q256_BLDS(srdA, a_vo, so_ + row8, dst_ + 1024);                                             \
            } else {                                                                                        \
                Q256_BLDS4(srdA, a_vo, so_, dst_);                                                          \
                Q256_BLDS4(srdA, a_vo, so_ + row8, dst_ + 1024);                                            \
            }                                                                                               \
        } else {                                                                                            \
            const int so_ = (oW) + ((which) & 1) * 4 * row8;                                                \
            Q256_BLDS(srdW, w_vo, so_, dst_);                                                               \
            Q256_BLDS(srdW, w_vo, so_ + row8, dst_ + 1024);                                                 \
        }                                                                                                   \
    }
    // bias / weight scales of tile (., bn): one 4-byte piece per wave
    const __amdgpu_buffer_rsrc_t srdX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wave >= 4 ? g.wscale : g.bias), 0, 0x7fffffff, 0x00020000);
    auto stage_x = [&](int bm, int bn, int par) {
        const int l = lane_now();        // (recomputed: no 64-bit lane address kept)
        Q256_BLDS4(srdX, l * 4, (bn * 256 + (wave & 3) * 64) * 4, smem + Q256_BC + par * 2048 + wave * 256);
        if constexpr (FOLD) {
            // c [256] f16 = 512 B: two 4-byte pieces (waves of equal parity bring the same bytes); the tile rows' raw statistics:
            // 256 rows x 32 B, rows 32 wave .. + 31 into this wave's OWN patch (free until its next epilogue)
            const __amdgpu_buffer_rsrc_t srdC = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(g.c16), 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t srdT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.ln_stats), 0, 0x7fffffff, 0x00020000);
            Q256_BLDS4(srdC, l * 4, (bn * 256 + (wave & 1) * 128) * 2, smem + Q256_C16 + par * 512 + (wave & 1) * 256);
            Q256_BLDS(srdT, l * 16, (bm * 256 + wave * 32) * 32, smem + Q256_PATCH + wave * 2048);
        }
    };

    // ---- fragment reads: one base per operand and 64-k half + immediates (slot, 16-row sub-tile)
    typedef const __attribute__((address_space(3))) uint8_t* lds_u8;
    v8i32 am[4];
    v8i32 wq[2][2];
    int sc[4];
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define Q256_READ_A(b, mq)                                                                                   \
    _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) {                                                      \
        const u32x4 lo_ = *reinterpret_cast<const u32x4*>(smem + ab[0] + Q256_SLOT(mq, b) + mf * 2048);     \
        const u32x4 hi_ = *reinterpret_cast<const u32x4*>(smem + ab[1] + Q256_SLOT(mq, b) + mf * 2048);     \
        am[mf][0] = lo_[0]; am[mf][1] = lo_[1]; am[mf][2] = lo_[2]; am[mf][3] = lo_[3];                     \
        am[mf][4] = hi_[0]; am[mf][5] = hi_[1]; am[mf][6] = hi_[2]; am[mf][7] = hi_[3];                     \
        sc[mf] = (int)((lds_u8)smem)[sc_lane + sc_off + (mq) * 2048 + mf * 256];               \
    }
#define Q256_READ_W(b, nq)                                                                                   \
    _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) {                                                      \
        const u32x4 lo_ = *reinterpret_cast<const u32x4*>(smem + wb[0] + Q256_SLOT(nq, b) + nf * 2048);     \
        const u32x4 hi_ = *reinterpret_cast<const u32x4*>(smem + wb[1] + Q256_SLOT(nq, b) + nf * 2048);     \
        wq[nq][nf][0] = lo_[0]; wq[nq][nf][1] = lo_[1]; wq[nq][nf][2] = lo_[2]; wq[nq][nf][3] = lo_[3];     \
        wq[nq][nf][4] = hi_[0]; wq[nq][nf][5] = hi_[1]; wq[nq][nf][6] = hi_[2]; wq[nq][nf][7] = hi_[3];     \
    }
// 8 block-scaled MFMAs: weights = A operand (unit block scales), activations = B operand (scale byte 0 of sc[mf])
// STG = the phase's two LDS-DMA pieces when they are issued from the MIDDLE of the MFMA part (Q256_STAGE_MID: after the first
// four MFMAs — 24 of an MFMA's 32 cycles are free issue slots of the wave, and the partner half's read part, the longer side of
// every barrier-to-barrier segment, loses its most expensive instructions), empty otherwise
#define Q256_MMA(mq, nq, STG)                                                                                \
    {                                                                                                       \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) {                                                  \
            _Pragma("unroll") for (int mf = 0; mf < 4; ++mf)                                                \
                acc[(nq) * 2 + nf][(mq) * 4 + mf] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(       \
                    wq[nq][nf], am[mf], acc[(nq) * 2 + nf][(mq) * 4 + mf], 0, 0, 0, 0x7F7F7F7F, 0, sc[mf]); \
            if (nf == 0) {                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                          \
                STG;                                                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                          \
            }                                                                                               \
        }                                                                                                   \
        /* pinned: without a use at this point the optimiser sinks the MFMAs (pure functions of registers) behind the */ \
        /* pair's sixteen barriers — every fragment of two K-tiles alive at once, 200+ spilled registers */       \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)                                                    \
            _Pragma("unroll") for (int mf = 0; mf < 4; ++mf)                                                \
                asm volatile("" : Q256_PIN(acc[(nq) * 2 + nf][(mq) * 4 + mf]));                                \
        __builtin_amdgcn_s_setprio(0);                                                                      \
    }
#ifdef Q256_STAGE_MID
#define Q256_STG_READ(...)
#define Q256_STG_MMA(...) Q256_STAGE(__VA_ARGS__)
#define Q256_WADJ 2
#else
#define Q256_STG_READ(...) Q256_STAGE(__VA_ARGS__)
#define Q256_STG_MMA(...)
#define Q256_WADJ 0
#endif
// counted wait: 10 younger slot pieces stay in flight; POST: + the previous tile's epilogue operations; XS: + the pair's scale piece
// (ODD: two scale pieces per pair)
#define Q256_WAIT(POST, XS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 - Q256_WADJ + ((POST) ? EX : 0) + ((XS) ? 1 + ODD : 0)) : "memory")
#define Q256_BARRIER()                       \
    {                                        \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    }
#define Q256_LATE_READS_DONE() \
    if (wm == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// One K-tile out of buffer B (gemm256p_kernel's P256_KTILE): four phases [reads + one slot's staging + wait] B [8 MFMAs] B.
// P0 / P1 / P3: POST form of the three waits; X0 / X1 / X3: the pair's scale piece is younger than the slot the wait retires.
#define Q256_KTILE(B, P0, P1, P3, X0, X1, X3, HALF)                                                          \
    {                                                                                                       \
        Q256_READ_A(B, 0);                                                                                  \
        Q256_READ_W(B, 0);                                                                                  \
        Q256_STG_READ(1, (B) ^ 1, oA1 + mA1, oW1, mA1 != 0); /* A m1 of K-tile t+1 */                       \
        Q256_LATE_READS_DONE();                                                                             \
        Q256_WAIT(P0, X0);                /* retires W n1 of this K-tile */                                 \
        Q256_BARRIER();                                                                                     \
        Q256_MMA(0, 0, Q256_STG_MMA(1, (B) ^ 1, oA1 + mA1, oW1, mA1 != 0));                                 \
        Q256_BARRIER();                                                                                     \
        Q256_READ_W(B, 1);                                                                                  \
        Q256_STG_READ(0, B, oA2, oW2, true);       /* A m0 of K-tile t+2 */                                 \
        Q256_LATE_READS_DONE();                                                                             \
        Q256_WAIT(P1, X1);                /* retires A m1 of this K-tile */                                 \
        Q256_BARRIER();                                                                                     \
        Q256_MMA(0, 1, Q256_STG_MMA(0, B, oA2, oW2, true));                                                 \
        Q256_BARRIER();                                                                                     \
        if constexpr (!(HALF)) { Q256_READ_A(B, 1); }                                                       \
        Q256_STG_READ(2, B, oA2, oW2, true);       /* W n0 of K-tile t+2; nothing new is read in the next phase: no wait */ \
        Q256_LATE_READS_DONE();                                                                             \
        Q256_BARRIER();                                                                                     \
        if constexpr (!(HALF)) { Q256_MMA(1, 1, Q256_STG_MMA(2, B, oA2, oW2, true)); }                      \
        else { Q256_STG_MMA(2, B, oA2, oW2, true); }                                                        \
        Q256_BARRIER();                                                                                     \
        Q256_STG_READ(3, B, oA2, oW2, true);       /* W n1 of K-tile t+2 */                                 \
        Q256_WAIT(P3, X3);                /* retires A m0 / W n0 of the next K-tile */                      \
        Q256_BARRIER();                                                                                     \
        if constexpr (!(HALF)) { Q256_MMA(1, 0, Q256_STG_MMA(3, B, oA2, oW2, true)); }                      \
        else { Q256_STG_MMA(3, B, oA2, oW2, true); }                                                        \
        Q256_BARRIER();                                                                                     \
        Q256_ADVANCE();                                                                                     \
    }
// position t+1 becomes the old t+2; t+2 moves on one K-tile (into the next tile, or wraps in the last one); the scale byte
// offset moves on one K-tile: + 1 inside a group, other parity at a group's end
#define Q256_ADVANCE()                                                                                       \
    oA1 = oA2; oW1 = oW2; mA1 = mA2;                                                                        \
    if (++k2 == nt) {                                                                                       \
        k2 = 0;                                                                                             \
        if (o2 + 1 < mine) ++o2;                                                                            \
        int bm_, bn_, hf_;                                                                                  \
        tile_of(o2, bm_, bn_, hf_);                                                                         \
        oA2 = bm_ * 256 * K + (hf_ == 2 ? 8 * row8 : 0);                                                    \
        oW2 = bn_ * 256 * K;                                                                                \
        mA2 = hf_ ? 0 : 8 * row8;                                                                           \
    } else {                                                                                                \
        oA2 += 128; oW2 += 128;                                                                             \
    }                                                                                                       \
    ++kt_cur;                                                                                               \
    if constexpr (ODD) { if (kt_cur == nt) kt_cur = 0; } /* (a half group ends the tile: the count restarts with the next) */ \
    sc_off = (kt_cur & 3) == 0 ? ((sc_off & 4096) ^ 4096) : sc_off + 1;
// the scale piece at the head of a pair: A slot `pc_mq` of scale group `pc_g` of tile position `pc_o` into the ring parity of
// that group; then the cursor moves on (the tile's last group is followed by group 0 of the workgroup's next tile, whose
// rows come from the tile list; past the last tile it re-reads the last one)
#define Q256_SCALE_PIECE()                                                                                   \
    {                                                                                                       \
        Q256_BLDS4(srdS, s_vo, pc_so, smem + Q256_RING + pc_par + pc_mq * 2048 + wave * 256);               \
        if (pc_mq == 0) { pc_mq = 1; pc_so += 64 * g.ld_as; }                                               \
        else {                                                                                              \
            pc_mq = 0; pc_par ^= 4096;                                                                      \
            if (++pc_g == ngrp) {                                                                           \
                pc_g = 0;                                                                                   \
                if (pc_o + 1 < mine) ++pc_o;                                                                \
                int bm_, bn_, hf_;                                                                          \
                tile_of(pc_o, bm_, bn_, hf_);                                                               \
                pc_so = (bm_ * 256 + (hf_ == 2 ? 64 : 0)) * g.ld_as;                                        \
            } else {                                                                                        \
                pc_so += 16 - 64 * g.ld_as;                                                                 \
            }                                                                                               \
        }                                                                                                   \
    }

// ODD: both slots of the cursor's group at the head of every pair; the cursor moves on behind the second pair of a full group and
// behind the half group's only pair (kp = this pair's index in its tile)
#define Q256_SCALE_PIECE2(kp)                                                                                \
    {                                                                                                       \
        Q256_BLDS4(srdS, s_vo, pc_so, smem + Q256_RING + pc_par + wave * 256);                              \
        Q256_BLDS4(srdS, s_vo, pc_so + 64 * g.ld_as, smem + Q256_RING + pc_par + 2048 + wave * 256);        \
        if (((kp) & 1) || (kp) == npair - 1) {                                                              \
            pc_par ^= 4096;                                                                                 \
            if (++pc_g == ngrp) {                                                                           \
                pc_g = 0;                                                                                   \
                if (pc_o + 1 < mine) ++pc_o;                                                                \
                int bm_, bn_, hf_;                                                                          \
                tile_of(pc_o, bm_, bn_, hf_);                                                               \
                pc_so = (bm_ * 256 + (hf_ == 2 ? 64 : 0)) * g.ld_as;                                        \
            } else {                                                                                        \
                pc_so += 16;                                                                                \
            }                                                                                               \
        }                                                                                                   \
    }
#define Q256_PAIR_PIECES(kp) \
    if constexpr (ODD) { Q256_SCALE_PIECE2(kp); } else { Q256_SCALE_PIECE(); }

    // ---- stream state
    int cbm, cbn, chalf;
    tile_of(0, cbm, cbn, chalf);
    int oA1 = cbm * 256 * K + (chalf == 2 ? 8 * row8 : 0), oW1 = cbn * 256 * K;  // K-tile 0 of tile 0, then position t+1
    int oA2 = oA1 + 128, oW2 = oW1 + 128;
    int mA1 = chalf ? 0 : 8 * row8, mA2 = mA1;   // where a position's A m1 slot comes from (a half tile has none: m0 again)
    int o2 = 0, k2 = 1;
    int kt_cur = 0;                              // K-tile of the tile being computed (only its low two bits matter)
    int sc_off = 0;                              // ring parity * 4096 + K-tile in its scale group
    const int dump_row = M - 1;                  // rows >= m_valid are stored to the last pad row (the store COUNT must not depend on data)
    // scale-piece cursor: group 0 of tile 0 goes out in the prologue (both slots), the loop starts with group 1 (or the next tile's group 0)
    int pc_o = 0, pc_g = 0, pc_mq = 0, pc_par = 0;
    int pc_so = (cbm * 256 + (chalf == 2 ? 64 : 0)) * g.ld_as;

    // prologue: the first group's scale dwords, the first tile's bias / weight scales, K-tile 0 completely, K-tile 1 except its
    // A m1 slot (staged by phase 0 of K-tile 0)
    Q256_SCALE_PIECE();
    Q256_SCALE_PIECE();
    stage_x(cbm, cbn, 0);
    {
        char* d0 = smem + stage_dst;
        Q256_BLDS(srdA, a_vo, oA1, d0 + Q256_SLOT(0, 0)); Q256_BLDS(srdA, a_vo, oA1 + row8, d0 + Q256_SLOT(0, 0) + 1024);
        Q256_BLDS(srdW, w_vo, oW1, d0 + Q256_SLOT(2, 0)); Q256_BLDS(srdW, w_vo, oW1 + row8, d0 + Q256_SLOT(2, 0) + 1024);
        Q256_BLDS(srdW, w_vo, oW1 + 4 * row8, d0 + Q256_SLOT(3, 0)); Q256_BLDS(srdW, w_vo, oW1 + 5 * row8, d0 + Q256_SLOT(3, 0) + 1024);
        if (mA1 != 0) { Q256_BLDS(srdA, a_vo, oA1 + mA1, d0 + Q256_SLOT(1, 0)); Q256_BLDS(srdA, a_vo, oA1 + mA1 + row8, d0 + Q256_SLOT(1, 0) + 1024); }
        else { Q256_BLDS4(srdA, a_vo, oA1, d0 + Q256_SLOT(1, 0)); Q256_BLDS4(srdA, a_vo, oA1 + row8, d0 + Q256_SLOT(1, 0) + 1024); }
        Q256_BLDS(srdA, a_vo, oA2, d0 + Q256_SLOT(0, 1)); Q256_BLDS(srdA, a_vo, oA2 + row8, d0 + Q256_SLOT(0, 1) + 1024);
        Q256_BLDS(srdW, w_vo, oW2, d0 + Q256_SLOT(2, 1)); Q256_BLDS(srdW, w_vo, oW2 + row8, d0 + Q256_SLOT(2, 1) + 1024);
        Q256_BLDS(srdW, w_vo, oW2 + 4 * row8, d0 + Q256_SLOT(3, 1)); Q256_BLDS(srdW, w_vo, oW2 + 5 * row8, d0 + Q256_SLOT(3, 1) + 1024);
    }
    // position t+1 = K-tile 1 (its A m1 slot is still to come), t+2 = K-tile 2
    oA1 = oA2; oW1 = oW2;
    oA2 += 128; oW2 += 128;
    k2 = 2;
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // all but the five youngest slot loads: A m0 / W n0 of K-tile 0, the pieces in front of them
    Q256_BARRIER();
    if (wm == 1) Q256_BARRIER();  // the lower half runs one barrier behind from here on

    // ---- epilogue of the tile (cbm, cbn) out of the accumulators: JN = 8 (whole tile) or 4 (half tile: 64 rows per wave, hrow = 0 / 64)
    // 16-row sub-tiles per wave. acc[i][j][r] = C[m = wm*128 + hrow + j*16 + fr][n = wn*64 + i*16 + 4*fg + r].
    auto epilogue = [&](auto jn_c, int hrow, int par) {
        constexpr int JN = decltype(jn_c)::value;
        // the lane id from the hardware, here: nothing the epilogue derives from it is kept live (or spilled) through the K loop
        const int lane_e = lane_now();
        const int fr = lane_e & 15, fg = lane_e >> 4;
        // bias and weight scale of this lane's four columns of column block i: read from LDS where they are used (32 registers
        // held across the eight sub-tiles were the difference between fitting and spilling)
        const char* bc = smem + Q256_BC + par * 2048 + (wn * 64 + 4 * fg) * 4;
#define Q256_BIAS(i) (*reinterpret_cast<const f32x4*>(bc + (i) * 64))
#define Q256_SW(i) (*reinterpret_cast<const f32x4*>(bc + 1024 + (i) * 64))
        char* patch = smem + Q256_PATCH + wave * 2048;
        const int rrow = lane_e >> 3, rchunk = lane_e & 7;
        const int m_base = cbm * 256 + wm * 128 + hrow;
        const int n_base = cbn * 256 + wn * 64;
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, 0x7fffffff, 0x00020000);
        // FOLD: (mean, rstd) of the tile's 256 rows. This wave's patch holds the raw (sum, sumsq) quarters of rows 32 wave .. + 31
        // (stage_x, a whole K stream ago); lanes 0..31 finish one row each into the table; a barrier hands the table to everybody
        // (the lower wave half was aligned with the upper one in front of the epilogue: all eight waves are here).
        const float* tab = reinterpret_cast<const float*>(smem + Q256_TABLE) + (wm * 128 + hrow + fr) * 2;   // + j * 32
        const char* cc = smem + Q256_C16 + par * 512 + (wn * 64 + 4 * fg) * 2;                                // + i * 32
#define Q256_C(i) q256_halves_to_f32x4(*reinterpret_cast<const u32x2*>(cc + (i) * 32))
        [[maybe_unused]] f32x4 cf[4];   // FOLD: c of this lane's 16 columns, converted once (per (i, j) it was 128 conversions + 32 LDS reads a lane)
        if constexpr (FOLD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) cf[i] = Q256_C(i);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(patch + (lane_e & 31) * 32);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(patch + (lane_e & 31) * 32 + 16);
            const float s1 = (q0[0] + q0[2]) + (q1[0] + q1[2]), s2 = (q0[1] + q0[3]) + (q1[1] + q1[3]);
            const float mean = s1 * (1.0f / 1024.0f);
            const float var = fmaxf(s2 * (1.0f / 1024.0f) - mean * mean, 0.f);
            if (lane_e < 32)
                *reinterpret_cast<u32x2*>(smem + Q256_TABLE + (wave * 32 + lane_e) * 8) =
                    u32x2{__float_as_uint(mean), __float_as_uint(1.0f / sqrtf(var + g.ln_eps))};
            Q256_BARRIER();
        }
        if constexpr (EPI == MMISS_EPI8_BIAS_BF16 || EPI == MMISS_EPI8_BIAS_RESID_BF16) {
            // 16 rows x 64 bf16 columns through the patch: 8-byte slot s of row r at slot s ^ 2 (r & 7); whole 128-byte rows per store
            const int wr_off = fr * 128, wr_sw = 2 * (fr & 7);
            [[maybe_unused]] const __amdgpu_buffer_rsrc_t srdQO = __builtin_amdgcn_make_buffer_rsrc(MXQ ? g.q_out : nullptr, 0, 0x7fffffff, 0x00020000);
            [[maybe_unused]] const __amdgpu_buffer_rsrc_t srdQS = __builtin_amdgcn_make_buffer_rsrc(MXQ ? g.q_scale : nullptr, 0, 0x7fffffff, 0x00020000);
            // the residual rows, four 16-row sub-tiles (64 rows x 128 bytes per wave, 32 registers) fetched at a time
            u32x4 old[EPI == MMISS_EPI8_BIAS_RESID_BF16 ? 4 : 1][2];
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                if constexpr (EPI == MMISS_EPI8_BIAS_RESID_BF16) {
                    if ((j & 3) == 0) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int rh = 0; rh < 2; ++rh) {
                                const int m = m_base + (j + jj) * 16 + rh * 8 + rrow;
                                const int vo = ((m < g.m_valid ? m : dump_row) * g.ldo + n_base + rchunk * 8) * 2;
                                old[jj][rh] = __builtin_amdgcn_raw_buffer_load_b128(srdO, vo, 0, 0);
                            }
                    }
                    // old rows -> accumulator layout: written as they were loaded (row rh*8 + rrow, chunk rchunk), read per (fr, i)
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh) {
                        const int row = rh * 8 + rrow;
                        *reinterpret_cast<u32x4*>(patch + row * 128 + (((2 * rchunk) ^ (2 * (row & 7))) << 3)) = old[j & 3][rh];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    // (a lane reads the old values of column block i from the very 8 bytes it then overwrites with the new ones: no
                    // second barrier, and one block's old values in registers at a time)
                }
                float rm_j = 0.f, rstd_j = 1.f;   // FOLD: y = acc (sw rstd) + (c (-mean rstd) + b')
                if constexpr (FOLD) { rstd_j = tab[j * 32 + 1]; rm_j = -tab[j * 32] * rstd_j; }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 y;
                    if constexpr (FOLD) y = acc[i][j] * (Q256_SW(i) * rstd_j) + (cf[i] * rm_j + Q256_BIAS(i));
                    else y = acc[i][j] * Q256_SW(i) + Q256_BIAS(i);
                    if constexpr (EPI == MMISS_EPI8_BIAS_RESID_BF16) {
                        const u32x2 o2_ = *reinterpret_cast<const u32x2*>(patch + wr_off + (((i * 4 + fg) ^ wr_sw) << 3));
                        y = f32x4{__uint_as_float(o2_[0] << 16), __uint_as_float(o2_[0] & 0xFFFF0000u), __uint_as_float(o2_[1] << 16),
                                  __uint_as_float(o2_[1] & 0xFFFF0000u)} + y;
                    }
                    u32x2 pk;
                    pk[0] = pack_bf16x2(y[0], y[1]);
                    pk[1] = pack_bf16x2(y[2], y[3]);
                    *reinterpret_cast<u32x2*>(patch + wr_off + (((i * 4 + fg) ^ wr_sw) << 3)) = pk;
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                [[maybe_unused]] float rs1[2], rs2[2];
                [[maybe_unused]] int re8[2];
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    const int row = rh * 8 + rrow;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * 128 + (((2 * rchunk) ^ (2 * (row & 7))) << 3));
                    const int m = m_base + j * 16 + row;
                    const int mrow = m < g.m_valid ? m : dump_row;
                    __builtin_amdgcn_raw_buffer_store_b128(v, srdO, (mrow * g.ldo + n_base + rchunk * 8) * 2, 0, 0);
                    if constexpr (MXQ) {
                        // the MXFP8 copy and the statistics, in the ROW layout the bf16 rows leave in: this lane holds 8 consecutive
                        // columns of the row, 4 neighbouring lanes a 32-column block, 8 a row's 64 columns — two quad permutes for
                        // the block maximum, those and a half-row mirror for the sums (in the accumulator layout it was 200
                        // vector instructions and 13 LDS operations per sub-tile)
                        float x[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { x[2 * e] = __uint_as_float(v[e] << 16); x[2 * e + 1] = __uint_as_float(v[e] & 0xFFFF0000u); }
                        float s1 = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                        float s2 = ((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) + ((x[4] * x[4] + x[5] * x[5]) + (x[6] * x[6] + x[7] * x[7]));
                        float am = fmaxf(fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3]))),
                                         fmaxf(fmaxf(fabsf(x[4]), fabsf(x[5])), fmaxf(fabsf(x[6]), fabsf(x[7]))));
                        am = fmaxf(am, q256_dpp<0xB1>(am));
                        am = fmaxf(am, q256_dpp<0x4E>(am));
                        s1 += q256_dpp<0xB1>(s1); s2 += q256_dpp<0xB1>(s2);
                        s1 += q256_dpp<0x4E>(s1); s2 += q256_dpp<0x4E>(s2);
                        s1 += q256_dpp<0x141>(s1); s2 += q256_dpp<0x141>(s2);
                        float inv;
                        mx_scale_of(am, re8[rh], inv);
                        u32x2 q;
                        q[0] = pack_fp8x4(x[0] * inv, x[1] * inv, x[2] * inv, x[3] * inv);
                        q[1] = pack_fp8x4(x[4] * inv, x[5] * inv, x[6] * inv, x[7] * inv);
                        __builtin_amdgcn_raw_buffer_store_b64(q, srdQO, mrow * g.N + n_base + rchunk * 8, 0, 0);
                        rs1[rh] = s1;
                        rs2[rh] = s2;
                    }
                }
                if constexpr (MXQ) {
                    // the scale bytes of the sub-tile's 16 rows x 2 blocks: lanes with rchunk 0 / 4 hold those of row rrow, 1 / 5 those of
                    // row 8 + rrow (every lane runs the ONE store; 32 are active)
                    const int row = ((rchunk & 1) ? 8 : 0) + rrow;
                    const int m = m_base + j * 16 + row;
                    const int vo = (m < g.m_valid ? m : dump_row) * g.ld_qs + mx_scale_offset((n_base >> 5) + (rchunk >> 2));
                    if ((rchunk & 2) == 0) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)((rchunk & 1) ? re8[1] : re8[0]), srdQS, vo, 0, 0);
                    // the rows' sums over the wave's 64 columns, parked in four of the sub-tile's (consumed) accumulator registers until
                    // the end of the epilogue: arch VGPRs held across the sub-tiles were spilled ones
                    acc[0][j][0] = rs1[0]; acc[0][j][1] = rs2[0]; acc[0][j][2] = rs1[1]; acc[0][j][3] = rs2[1];
                }
                __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next j
                __builtin_amdgcn_sched_barrier(0);   // (one sub-tile's accumulators in VGPRs at a time)
            }
            if constexpr (MXQ) {
                // row statistics of the tile: every wave leaves its rows' 64-column sums in its own patch (row j 16 + fr of its
                // half at 8 bytes), a barrier, then wave w adds the four column blocks of tile rows 32 w .. + 31 in block order
                // and stores them: stats_out[m][bn] — ONE store per wave.
                {
                    float k[4] = {0.f, 0.f, 0.f, 0.f};   // lane (rrow, rchunk) writes sub-tile j = rchunk's rows rrow and 8 + rrow
#pragma unroll
                    for (int q = 0; q < JN; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            k[e] = rchunk == q ? acc[0][q][e] : k[e];
                            acc[0][q][e] = 0.f;
                        }
                    if (rchunk < JN) {
                        *reinterpret_cast<u32x2*>(patch + (rchunk * 16 + rrow) * 8) = u32x2{__float_as_uint(k[0]), __float_as_uint(k[1])};
                        *reinterpret_cast<u32x2*>(patch + (rchunk * 16 + 8 + rrow) * 8) = u32x2{__float_as_uint(k[2]), __float_as_uint(k[3])};
                    }
                }
                Q256_BARRIER();
                const int R = wave * 32 + (lane_e & 31);          // tile row
                const char* src = smem + Q256_PATCH + (R >> 7) * 4 * 2048 + (R & 127) * 8;
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const u32x2 v = *reinterpret_cast<const u32x2*>(src + b * 2048);
                    t1 += __uint_as_float(v[0]);
                    t2 += __uint_as_float(v[1]);
                }
                const int m = cbm * 256 + R;
                const __amdgpu_buffer_rsrc_t srdST = __builtin_amdgcn_make_buffer_rsrc(g.stats_out, 0, 0x7fffffff, 0x00020000);
                if (lane_e < 32)
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(t1), __float_as_uint(t2)}, srdST,
                                                          (((m < g.m_valid ? m : dump_row) * nbn + cbn) * 2) * 4, 0, 0);
            }
        } else {
            // QuickGELU -> MXFP8: ONE scale per row and the wave's 64 columns (= k-blocks n/32 and n/32 + 1 of the next GEMM, the
            // same byte twice: gemm8_kernel's format), 16 rows x 64 bytes through the patch (80-byte rows), one store per j
            const __amdgpu_buffer_rsrc_t srdQ = __builtin_amdgcn_make_buffer_rsrc(g.out_scale, 0, 0x7fffffff, 0x00020000);
            const int sc_col = mx_scale_offset((n_base >> 5) + (fg & 1));   // lane groups 0 / 2 write the first byte, 1 / 3 the second
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                f32x4 y[4];
                float amax = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (FOLD) {
                        const float rstd_j = tab[j * 32 + 1], rm_j = -tab[j * 32] * rstd_j;
                        y[i] = acc[i][j] * (Q256_SW(i) * rstd_j) + (cf[i] * rm_j + Q256_BIAS(i));
                    } else {
                        y[i] = acc[i][j] * Q256_SW(i) + Q256_BIAS(i);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        y[i][r] = quick_gelu(y[i][r]);
                        amax = fmaxf(amax, fabsf(y[i][r]));
                    }
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                amax = q256_max_over_lane_groups(amax);
                int e8;
                float inv;
                mx_scale_of(amax, e8, inv);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<uint32_t*>(patch + fr * 80 + i * 16 + fg * 4) =
                        pack_fp8x4(y[i][0] * inv, y[i][1] * inv, y[i][2] * inv, y[i][3] * inv);
                {   // the row's scale byte: every lane stores (no data-dependent instruction count)
                    const int m = m_base + j * 16 + fr;
                    const int vo = (m < g.m_valid ? m : dump_row) * g.ld_os + sc_col;
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)e8, srdQ, vo, 0, 0);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                {
                    const int row = lane_e >> 2, c16 = lane_e & 3;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * 80 + c16 * 16);
                    const int m = m_base + j * 16 + row;
                    const int vo = (m < g.m_valid ? m : dump_row) * g.ldo + n_base + c16 * 16;
                    __builtin_amdgcn_raw_buffer_store_b128(v, srdO, vo, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- whole tiles. Pair kp of a tile: scale piece, K-tile 2 kp out of buffer 0, K-tile 2 kp + 1 out of buffer 1.
    for (int ti = 0; ti < nfull; ++ti) {
        for (int kp = 0; kp < npair; ++kp) {
            Q256_PAIR_PIECES(kp);
            if (kp == 0 && ti > 0) {
                Q256_KTILE(0, true, true, true, true, true, true, false);
                Q256_KTILE(1, true, false, false, true, false, false, false);
            } else {
                Q256_KTILE(0, false, false, false, true, true, true, false);
                Q256_KTILE(1, false, false, false, true, false, false, false);
            }
        }
        // The tile is complete. The upper half waits one barrier for the lower half's last MFMAs, both run the epilogue side
        // by side, then the lower half falls one barrier behind again.
        if (wm == 0) Q256_BARRIER();
        epilogue(std::integral_constant<int, 8>{}, 0, ti & 1);
        lane_consts();
        if (ti + 1 < mine) {
            tile_of(ti + 1, cbm, cbn, chalf);
            stage_x(cbm, cbn, (ti + 1) & 1);
            if (wm == 1) Q256_BARRIER();
        }
    }
    // ---- the half tile of the last round, if this workgroup has one (always behind at least one whole tile: POST waits)
    if (nhalf) {
        for (int kp = 0; kp < npair; ++kp) {
            Q256_PAIR_PIECES(kp);
            if (kp == 0) {
                Q256_KTILE(0, true, true, true, true, true, true, true);
                Q256_KTILE(1, true, false, false, true, false, false, true);
            } else {
                Q256_KTILE(0, false, false, false, true, true, true, true);
                Q256_KTILE(1, false, false, false, true, false, false, true);
            }
        }
        if (wm == 0) Q256_BARRIER();
        epilogue(std::integral_constant<int, 4>{}, chalf == 2 ? 64 : 0, nfull & 1);
    }
    // the unconditional staging of the last K-tiles is still in flight: LDS must not be handed on with DMA writes pending
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef Q256_SLOT
#undef Q256_PIN
#undef Q256_BIAS
#undef Q256_SW
#undef Q256_C
#undef Q256_BLDS
#undef Q256_BLDS4
#undef Q256_STAGE
#undef Q256_READ_A
#undef Q256_READ_W
#undef Q256_MMA
#undef Q256_WAIT
#undef Q256_BARRIER
#undef Q256_LATE_READS_DONE
#undef Q256_KTILE
#undef Q256_STG_READ
#undef Q256_STG_MMA
#undef Q256_WADJ
#undef Q256_ADVANCE
#undef Q256_SCALE_PIECE
#undef Q256_SCALE_PIECE2
#undef Q256_PAIR_PIECES

// can this GEMM run on the persistent fp8 kernel?
static inline bool gemm256p8_ok(int epi, int M, int N, int K) {
    if (M <= 0 || (M % 256) || N <= 0 || (N % 256) || K < 512 || (K % 256)) return false;
    if ((int64_t)(M / 256) * (N / 256) > 48 * 256 || M / 256 > 0xffff) return false;  // (tile table: 64 entries per workgroup)
    if ((int64_t)M * N * 2 >= (1LL << 31) || (int64_t)M * K >= (1LL << 31) || (int64_t)N * K >= (1LL << 31)) return false;  // (32-bit buffer offsets)
    return epi == MMISS_EPI8_BIAS_BF16 || epi == MMISS_EPI8_QGELU_MXFP8 || epi == MMISS_EPI8_BIAS_RESID_BF16;
}

template <int EPI, int XT, int ODD = 0>
static int launch_gemm256p8_inst(hipStream_t st, const Gemm8Args& g) {
    const int T = (g.M / 256 - (g.ragged > 0 ? 1 : 0)) * (g.N / 256);
    const int grid = T >= 256 ? 256 : T;
    constexpr int LDS = XT == 1 ? Q256_LDS_FOLD : Q256_LDS;
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm256p8_kernel<EPI, XT, ODD>), LDS));
    hipLaunchKernelGGL((gemm256p8_kernel<EPI, XT, ODD>), dim3(grid), dim3(512), LDS, st, g);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// the extensions of round 5 (Gemm8Args): xt = 1 LayerNorm folded into BIAS_BF16 / QGELU_MXFP8 (K = 1024), xt = 2 BIAS_RESID_BF16 that
// also leaves the new rows as MXFP8 with their statistics (N = 1024; no half tiles: the tile count must not end in a short round)
static inline bool gemm256p8_xt_ok(int epi, int xt, const Gemm8Args& g) {
    if (xt == 0) return true;
    if (g.K % 512) return false;   // (the extensions: whole 512-k groups only)
    if (xt == 1)
        return (epi == MMISS_EPI8_BIAS_BF16 || epi == MMISS_EPI8_QGELU_MXFP8) && g.K == 1024 && g.c16 && g.ln_stats && g.x16;
    if (xt != 2 || epi != MMISS_EPI8_BIAS_RESID_BF16 || g.N != 1024 || !g.q_out || !g.q_scale || !g.stats_out ||
        g.ld_qs < mx_scale_row_bytes(g.N) || (int64_t)g.M * g.ld_qs >= (1LL << 31))
        return false;
    const int T = (g.M / 256 - (g.ragged > 0 ? 1 : 0)) * (g.N / 256), G = T >= 256 ? 256 : T;
    const int rem = T % G;
    return !(T / G >= 1 && rem > 0 && 2 * rem <= G);
}

// g.M = rows padded to 256 (A8, As, out and out_scale must hold g.M rows); rows >= g.m_valid are computed and land in row M - 1.
static int launch_gemm256p8(hipStream_t st, int epi, Gemm8Args g, int xt = 0) {
    if (!gemm256p8_ok(epi, g.M, g.N, g.K) || !g.A || !g.As || !g.W || !g.wscale || !g.bias || !g.out || g.ld_as < mx_scale_row_bytes(g.K) ||
        (int64_t)g.M * g.ld_as >= (1LL << 31))
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm256p8: epi=%d M=%d N=%d K=%d ld_as=%d", epi, g.M, g.N, g.K, g.ld_as);
    if (epi == MMISS_EPI8_QGELU_MXFP8 && (!g.out_scale || g.ld_os < mx_scale_row_bytes(g.N)))
        MM_FAIL(MMISS_ERR_ARG, "gemm256p8: the MXFP8 epilogue needs out_scale with >= %d bytes per row", mx_scale_row_bytes(g.N));
    if (g.m_fast == 0) g.m_fast = mmiss_option("gemm_p256_band", 8);  // row blocks per band of the global tile order
    // a last row block with at most 128 valid rows goes through the register-streamed pass (option gemm_p256_ragged = 0: a tile)
    g.ragged = 0;
    if (g.M >= 512 && g.m_valid > g.M - 256 && g.m_valid <= g.M - 128 && mmiss_option("gemm_p256_ragged", 1) != 0)
        g.ragged = g.m_valid - (g.M - 256);
#ifdef MMISS_EXPERIMENTS
    g.stagger = mmiss_option("gemm_p256_stagger", 0);
#endif
    if (!gemm256p8_xt_ok(epi, xt, g)) MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm256p8: extension %d of epilogue %d at M=%d N=%d K=%d", xt, epi, g.M, g.N, g.K);
    static const char* names[] = {"gemm_fp8_bias_p256", "gemm_fp8_qgelu_mx_p256", "", "gemm_fp8_bias_resid16_p256"};
    static const char* names_x[] = {"gemm_fp8_lnfold_bias_p256", "gemm_fp8_lnfold_qgelu_mx_p256", "", "gemm_fp8_bias_resid16_mxq_p256"};
    const int mv = g.m_valid < g.M ? g.m_valid : g.M;
    const double out_b = epi == MMISS_EPI8_BIAS_BF16 ? 2.0 : (epi == MMISS_EPI8_QGELU_MXFP8 ? 1.0 : (xt == 2 ? 5.0 : 4.0));
    MM_PROF(xt ? names_x[epi] : names[epi], st, 2.0 * mv * (double)g.N * g.K, (double)mv * g.K + (double)g.N * g.K + out_b * mv * g.N);
    if (xt == 1) {
        if (epi == MMISS_EPI8_BIAS_BF16) return launch_gemm256p8_inst<MMISS_EPI8_BIAS_BF16, 1>(st, g);
        return launch_gemm256p8_inst<MMISS_EPI8_QGELU_MXFP8, 1>(st, g);
    }
    if (xt == 2) return launch_gemm256p8_inst<MMISS_EPI8_BIAS_RESID_BF16, 2>(st, g);
    if (g.K % 512) {   // an odd number of K-tile pairs per tile (K = 768: ViT-B/32)
        if (epi == MMISS_EPI8_BIAS_BF16) return launch_gemm256p8_inst<MMISS_EPI8_BIAS_BF16, 0, 1>(st, g);
        if (epi == MMISS_EPI8_QGELU_MXFP8) return launch_gemm256p8_inst<MMISS_EPI8_QGELU_MXFP8, 0, 1>(st, g);
        return launch_gemm256p8_inst<MMISS_EPI8_BIAS_RESID_BF16, 0, 1>(st, g);
    }
    if (epi == MMISS_EPI8_BIAS_BF16) return launch_gemm256p8_inst<MMISS_EPI8_BIAS_BF16, 0>(st, g);
    if (epi == MMISS_EPI8_QGELU_MXFP8) return launch_gemm256p8_inst<MMISS_EPI8_QGELU_MXFP8, 0>(st, g);
    return launch_gemm256p8_inst<MMISS_EPI8_BIAS_RESID_BF16, 0>(st, g);
}

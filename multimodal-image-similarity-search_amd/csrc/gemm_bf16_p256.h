// gemm_bf16_p256.h — PERSISTENT 256x256x64 bf16 GEMM for the wide GEMMs of the towers (QKV, FC1: N >= 1536), same
// contract as gemm_bf16.h: C[M,N] = A[M,K] * W[N,K]^T + fused epilogue (bf16 out; bias / bias + QuickGELU, optionally behind
// a LayerNorm folded into W).
//
// Why (measurements of round 3 in profiles/gemm_p256_r03.txt): the one-tile-per-workgroup 256x256 kernel
// (gemm_bf16_256.h) pays per tile a pipeline fill from cold, a drain and an epilogue through the staging LDS, all of it with
// every workgroup of the chip in the same state at the same time; and its K loop spends its time NEXT TO the matrix pipe,
// not in it (PMC: matrix cores busy 45 % of a wave's lifetime; ablations: MFMAs, fragment reads, staging and barriers add
// up instead of overlapping). Here the grid is one workgroup per CU and the workgroup owns its tile schedule:
//   * its tiles form ONE continuous K-tile stream: the loads of the next tile's first two K-tiles are issued by the
//     ordinary t+1 / t+2 prefetch of the current tile's last two K-tiles, the epilogue runs while they fly;
//   * two barriers per phase — [fragment reads + one slot's staging + counted wait] B [16 MFMAs] B — and the waves of the
//     lower tile half (wm = 1: waves 4-7, the SIMD partners of waves 0-3) run ONE barrier behind, so that on every SIMD one
//     wave issues MFMAs while its partner reads fragments and stages (guide: the 256^2 8-phase template's staggered wave
//     groups; MICROARCH "two waves per SIMD", item 9);
//   * the loop is unrolled over the two staging buffers: every LDS address is base register + immediate, every LDS-DMA
//     source a scalar base + a per-lane 32-bit offset computed once, staging is unconditional (past the end of the stream
//     it re-reads the last tile into free slots) and the waits are plain immediates — the first version of this loop spent
//     0.3 us per K-tile on scalar bookkeeping alone (tools/gemm_p256_ablate.py);
//   * everything the epilogue needs from memory arrives by LDS-DMA as well — the tile's 256 bias / c values (one 4-byte
//     piece per wave) and, in the folded-LayerNorm mode, the raw (sum, sumsq) partials of its 256 rows (K/256 16-byte
//     pieces per wave), finalised to (mean, rstd) by 256 threads in a wait-free phase of K-tile 2 — because an ordinary
//     load beside LDS-DMA makes hipcc drain the whole pipeline (guide §5, trap (b));
//   * the epilogue's stores and those pieces count in vmcnt (in issue order), so the four counted waits that follow an
//     epilogue allow EX more operations in flight; from the fifth wait on every such operation is older than the slot
//     being waited for and the plain count applies again;
//   * tile order: rounds of 256 tiles of a banded global order, 32 logically consecutive tiles (8 row blocks x 4 column
//     tiles) per XCD; a last round of at most 128 tiles is dealt in half tiles so that every CU takes part in it.
#pragma once
#include <type_traits>
#include "gemm_bf16_256.h"

#define P256_BC (8 * G256_SLOT)       // 2 x 2 KB: bias [256] f32 + c [256] f32 of the current / next tile
#define P256_TABLE (P256_BC + 4096)   // 256 x (mean, rstd)
#define P256_TILES (P256_TABLE + 2048)  // this workgroup's tile list: 64 x (bm | bn << 16), decoded once before the K stream
#define P256_RAW (P256_TILES + 256)    // 256 rows x K/64 x (sum, sumsq) f32, as they lie in memory; the epilogue's patches
#define P256_FTAB (P256_RAW + 16384)   // XP = 1 (finished statistics): 2 x 256 x (mean, rstd) of the current / next tile, behind the patches

// eight e4m3 codes (one lane's 8 consecutive k of an fp8 index row) -> the f16 fragment of the MFMA, exactly (e4m3 fits f16)
__device__ __forceinline__ f16x8 p256_widen_f8(u32x2 w) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f16x8 r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], false);
        const f2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[i], true);
        const h2 l2 = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(lo[0], lo[1]));
        const h2 g2 = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(hi[0], hi[1]));
        r[4 * i + 0] = l2[0]; r[4 * i + 1] = l2[1]; r[4 * i + 2] = g2[0]; r[4 * i + 3] = g2[1];
    }
    return r;
}

// DBG (timing experiments only, EXPERIMENTS builds, tools/gemm_p256_ablate.py; results are wrong by construction), a
// compile-time mask: 1 = no MFMAs, 2 = no staging inside the loop, 4 = no fragment reads, 8 = every tile loads tile (0, 0)
// (operands L2-hot), 16 = no epilogue, 32 = epilogue without its stores, 64 = no barriers, 128 = the loop on
// v_mfma_f32_32x32x16_bf16 (8 instructions of 8 passes per phase instead of 16 of 4; the same fragment reads, the same LDS
// layout; the epilogue reads the accumulators as if they had the 16 x 16 layout)
// STYLE: how the epilogue gets from "a lane holds 4 consecutive columns of one row" to wide stores: 0 = a 16 x 64 transpose
// per wave through a private 2 KB LDS patch, whole 128-byte rows per store instruction; 1 = two v_permlane16_swap per
// 16 x 32 block, 8 consecutive columns per lane, 64-byte row segments (two adjacent instructions per line).
// XP = statistic pieces per wave and tile in the folded modes: K / 256 16-byte pieces of RAW partials (K = 512, 768: finalised
// in the kernel), or 1 = ONE 4-byte piece of FINISHED (mean, rstd) pairs from ep.ln_final (any K; ln_finalize_kernel ran first);
// 0 otherwise
template <int EPI, int XP, int STYLE = 0, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W, int M,
                                                          int N, int K, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __bf16 IN_T;
    typedef bf16x8 frag;
    constexpr bool FOLD = (EPI == MMISS_EPI_LNFOLD_BF16 || EPI == MMISS_EPI_LNFOLD_QGELU_BF16);
    constexpr bool GELU = (EPI == MMISS_EPI_BIAS_QGELU_BF16 || EPI == MMISS_EPI_LNFOLD_QGELU_BF16);
    static_assert(FOLD == (XP > 0), "statistic pieces exist exactly in the folded modes");
    constexpr bool FINAL = FOLD && XP == 1;
    constexpr int EX = 16 + 1 + XP;  // vector-memory operations of a wave between two tiles' K streams
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nbm = M >> 8, nbn = N >> 8;
    const int nt = K / GEMM_BK;  // even (K % 128 == 0)

    // ---- this workgroup's tile list. Tiles are dealt in rounds of G (256): in round i workgroup pb (its position among the
    // workgroups of its XCD first, xcd_remap) takes tile i * G + pb of one banded global order (bands of 8 row blocks x all
    // column tiles, m fastest inside a band: an XCD's 32 tiles of a round are 8 row blocks x 4 column tiles). A LAST round of
    // at most G/2 tiles is dealt in HALVES — the m0 rows (wm * 128 + 0..63) and the m1 rows (+64) of a tile go to neighbouring
    // workgroups: FC1 at 12 800 rows has 600 tiles = 2 rounds + 88, i.e. 3 tile times for 2.34 tiles of work per CU; as
    // 176 half tiles the tail costs ~0.65 of a tile time (a half tile runs phases 0 and 1 of every K-tile with MFMAs and
    // keeps the other two for their staging, waits and barriers only).
    // (Per-XCD row-block ranges, with A panels meant to stay in the L2 across an XCD's rounds, measured the same time and the
    // same fabric reads — FETCH_SIZE x 2 = 120 MB per FC1 launch: one round streams 4.6 MB of distinct operand bytes through
    // the XCD's 4 MB L2 — and cannot balance a tail; profiles/gemm_p256_r03.txt.)
    const int T = nbm * nbn, G = gridDim.x;
    const int pb = xcd_remap(blockIdx.x, G);
    const int full_rounds = T / G, rem = T - full_rounds * G;
    const bool halves = full_rounds >= 1 && rem > 0 && 2 * rem <= G;
    const int nfull = full_rounds + ((!halves && pb < rem) ? 1 : 0);       // whole tiles of this workgroup
    const int nhalf = (halves && pb < 2 * rem) ? 1 : 0;                    // + at most one half tile, always last
    const int mine = nfull + nhalf;
    // table entry: bm | bn << 16 | half tile << 30 | which half << 31
    if (tid < mine && tid < 64) {
        const bool hf = tid >= nfull;
        const int L = hf ? full_rounds * G + (pb >> 1) : tid * G + pb;
        int bm_, bn_;
        tile_order(L, nbm, nbn, ep.m_fast, bm_, bn_);
        reinterpret_cast<unsigned*>(smem + P256_TILES)[tid] = (unsigned)bm_ | ((unsigned)bn_ << 16) | (hf ? (1u << 30) : 0u) |
                                                             ((hf && (pb & 1)) ? (1u << 31) : 0u);
    }
    __syncthreads();
    // (decoded ONCE, in parallel, into LDS: the integer divisions are VALU code on this target, and inlined at the places of
    // the unrolled K loop that move on to a next tile they pushed the kernel over its 256 registers: 233 spilled, 6 x slower)
    auto tile_of = [&](int i, int& bm, int& bn, int& half) {  // half: 0 whole tile, 1 / 2 = its m0 / m1 rows only
        // (an LDS-address-space load: through a generic pointer this was a flat_load + s_waitcnt vmcnt(0) — a drain of the
        // whole staging pipeline at every tile switch)
        const unsigned pk = (unsigned)__builtin_amdgcn_readfirstlane(((const __attribute__((address_space(3))) int*)(smem + P256_TILES))[i]);
        bm = pk & 0xffff;
        bn = (pk >> 16) & 0x3fff;
        half = (pk >> 30) ? 1 + (int)(pk >> 31) : 0;
    };

    // ---- LDS-DMA sources, buffer form (one descriptor per operand): per-lane byte offset of the lane's row / chunk, computed
    // once (ONE VGPR per operand), everything else — tile, K-tile, m / n half, 8-row piece — in the scalar offset.
    // Slot row r of an A slot is the activation row (r>>6)*128 + mq*64 + (r&63) of the tile, of a W slot the weight row
    // (r>>5)*64 + nq*32 + (r&31); a wave stages slot rows 16*wave .. 16*wave+15; XOR swizzle on the source chunk.
    // LDS slot order: A m0 (buffer 0, 1), A m1 (0, 1), W n0 (0, 1), W n1 (0, 1), so that ONE base register per operand and
    // k step reaches every slot of the operand with a 16-bit immediate.
    const int r_in = lane >> 3, p = lane & 7;
    const int src_chunk = (p ^ r_in) * 8;
    const int a_vo = (((wave >> 2) * 128 + (wave & 3) * 16 + r_in) * K + src_chunk) * 2;
    const int w_vo = (((wave >> 1) * 64 + (wave & 1) * 16 + r_in) * K + src_chunk) * 2;
    const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(W), 0, 0x7fffffff, 0x00020000);
    const int row8 = 8 * K * 2;  // bytes between a slot's two 8-row pieces
    const int w_row8 = row8;
    constexpr bool W8 = false;   // (the retrieval kernel below also takes fp8 index rows as its W operand)
    const int stage_dst = wave * 2048;
#define P256_SLOT(which, b) (((which) * 2 + (b)) * G256_SLOT)
// cache policy of the operand loads (the builtin's aux: 1 sc0, 2 nt, 16 sc1), per operand; A/B knobs of
// tools/p256_policy_ab.sh (-DP256_A_POLICY=.. -DP256_W_POLICY=..), 0 = default in product builds
#ifdef P256_A_POLICY
#define P256_AUX_srdA P256_A_POLICY
#else
#define P256_AUX_srdA 0
#endif
#ifdef P256_W_POLICY
#define P256_AUX_srdW P256_W_POLICY
#else
#define P256_AUX_srdW 0
#endif
#define P256_BLDS(srd, vo, so, dst) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(dst), 16, vo, so, 0, P256_AUX_##srd)
#define P256_BLDS4(srd, vo, so, dst) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(dst), 4, vo, so, 0, P256_AUX_##srd)
// LIVE = false: the slot is not read (the A m1 slot of a half tile): its two pieces shrink to 4 bytes per lane — the K
// stream is bound by LDS-DMA bytes per CU (64 KB per K-tile at ~40 GB/s), the operation COUNT must stay what the waits assume
// PCS: which of the slot's two 8-row pieces (bit 0 / bit 1)
#define P256_STAGE_P(which, b, oA, oW, LIVE, PCS)                                                            \
    if constexpr (!((DBG & 2) != 0)) {                                                                      \
        char* dst_ = smem + P256_SLOT(which, b) + stage_dst;                                                \
        if constexpr ((which) < 2) {                                                                        \
            const int so_ = (oA);                                                                           \
            if (LIVE) {                                                                                     \
                if constexpr (((PCS) & 1) != 0) P256_BLDS(srdA, a_vo, so_, dst_);                           \
                if constexpr (((PCS) & 2) != 0) P256_BLDS(srdA, a_vo, so_ + row8, dst_ + 1024);             \
            } else {                                                                                        \
                if constexpr (((PCS) & 1) != 0) P256_BLDS4(srdA, a_vo, so_, dst_);                          \
                if constexpr (((PCS) & 2) != 0) P256_BLDS4(srdA, a_vo, so_ + row8, dst_ + 1024);            \
            }                                                                                               \
        } else if constexpr (W8) {                                                                          \
            /* fp8 index rows: a slot is 128 rows x 64 bytes = ONE 16-row piece per wave (the counted waits know: P256_WAIT) */ \
            const int so_ = (oW) + ((which) & 1) * 4 * w_row8;                                              \
            char* dst8_ = smem + P256_SLOT(which, b) + wave * 1024;                                         \
            if constexpr (((PCS) & 1) != 0) P256_BLDS(srdW, w_vo, so_, dst8_);                              \
        } else {                                                                                            \
            const int so_ = (oW) + ((which) & 1) * 4 * w_row8;                                              \
            if constexpr (((PCS) & 1) != 0) P256_BLDS(srdW, w_vo, so_, dst_);                               \
            if constexpr (((PCS) & 2) != 0) P256_BLDS(srdW, w_vo, so_ + w_row8, dst_ + 1024);               \
        }                                                                                                   \
    }
// Both pieces of a slot are issued in the read part of the phase. P256_SPLIT_STAGE issues the second one from the middle of
// the MFMA part instead (where a piece is said to cost ~60 cycles of issue against 100-185 beside fragment reads): measured
// 3-6 % SLOWER (4096^3: 116-121 vs 111-113 us on one box, tools/p256_split_ab.sh) — the read part is not what bounds the loop.
#ifndef P256_SPLIT_STAGE
#define P256_STAGE(which, b, oA, oW, LIVE) P256_STAGE_P(which, b, oA, oW, LIVE, 3)
#define P256_STAGE_MID(which, b, oA, oW, LIVE)
#define P256_INFLIGHT 10
#else
#define P256_STAGE(which, b, oA, oW, LIVE) P256_STAGE_P(which, b, oA, oW, LIVE, 1)
#define P256_STAGE_MID(which, b, oA, oW, LIVE) P256_STAGE_P(which, b, oA, oW, LIVE, 2)
#define P256_INFLIGHT 9
#endif
    // what the epilogue of tile (bm, bn) reads from memory, by LDS-DMA: 1 + XP pieces per wave
    auto stage_x = [&](int bm, int bn, int par) {
        const float* src = ((FOLD && wave >= 4) ? ep.aux : ep.bias) + bn * 256 + (wave & 3) * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem + P256_BC + par * 2048 + wave * 256), 4, 0, 0);
        if constexpr (FINAL) {
            // the tile's 256 x (mean, rstd) = 512 floats, 64 per wave; two tables (as for bias / c): a wave may be here while
            // another is still in the previous tile's epilogue, reading that tile's table
            const float* fs = ep.ln_final + (size_t)bm * 512 + wave * 64 + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)fs,
                                             (__attribute__((address_space(3))) void*)(smem + P256_FTAB + par * 2048 + wave * 256), 4, 0, 0);
        } else if constexpr (FOLD) {
            const char* st = reinterpret_cast<const char*>(ep.ln_stats) + (size_t)bm * 256 * (size_t)(XP * 32) + lane * 16;
#pragma unroll
            for (int q = 0; q < XP; ++q) glds16(st + (wave * XP + q) * 1024, smem + P256_RAW + (wave * XP + q) * 1024);
        }
    };

    // ---- fragment reads: one base per operand and k step + immediate (slot, 16-row sub-tile)
    uint32_t ab[2], wb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        ab[s] = (wm * 64 + fr) * 128 + (((4 * s + fg) ^ (fr & 7)) << 4);
        wb[s] = P256_SLOT(2, 0) + (wn * 32 + fr) * 128 + (((4 * s + fg) ^ (fr & 7)) << 4);
    }
    // (DBG & 128: lane l feeds row l & 31, k = 16 ks + 8 (l >> 5) .. + 7 of k step ks = 0 .. 3; one base per k step)
    uint32_t ab32[4], wb32[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        ab32[s] = (wm * 64 + (lane & 31)) * 128 + (((2 * s + (lane >> 5)) ^ (lane & 7)) << 4);
        wb32[s] = P256_SLOT(2, 0) + (wn * 32 + (lane & 31)) * 128 + (((2 * s + (lane >> 5)) ^ (lane & 7)) << 4);
    }
    frag am[4][2];
    frag wq[2][2][2];
    f32x4 acc[4][8];
    f32x16 acc32[2][4];   // (DBG & 128) [n quarter][m quarter x 2 + 32-row block]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
    auto acc_get = [&](int i, int j, int r) -> float {
        if constexpr ((DBG & 128) != 0) return acc32[i >> 1][j >> 1][((i & 1) * 2 + (j & 1)) * 4 + r];
        else return acc[i][j][r];
    };
    auto acc_zero = [&](int i, int j) {
        if constexpr ((DBG & 128) != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc32[i >> 1][j >> 1][((i & 1) * 2 + (j & 1)) * 4 + r] = 0.f;
        } else acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define P256_READ_A(b, mq)                                                                                   \
    if constexpr ((DBG & 128) != 0) {                                                                       \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                    \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                \
                am[mb * 2 + (ks >> 1)][ks & 1] = *reinterpret_cast<const frag*>(smem + ab32[ks] + P256_SLOT(mq, b) + mb * 4096); \
    } else if constexpr (!((DBG & 4) != 0)) {                                                               \
        _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) {                                                  \
            am[mf][0] = *reinterpret_cast<const frag*>(smem + ab[0] + P256_SLOT(mq, b) + mf * 2048);        \
            am[mf][1] = *reinterpret_cast<const frag*>(smem + ab[1] + P256_SLOT(mq, b) + mf * 2048);        \
        }                                                                                                   \
    }
#define P256_READ_W(b, nq)                                                                                   \
    if constexpr (W8) {                                                                                     \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) {                                                  \
            wq[nq][nf][0] = __builtin_bit_cast(frag, p256_widen_f8(*reinterpret_cast<const u32x2*>(smem + wb[0] + P256_SLOT(nq, b) + nf * 1024))); \
            wq[nq][nf][1] = __builtin_bit_cast(frag, p256_widen_f8(*reinterpret_cast<const u32x2*>(smem + wb[1] + P256_SLOT(nq, b) + nf * 1024))); \
        }                                                                                                   \
    } else if constexpr ((DBG & 128) != 0) {                                                                \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                    \
            wq[nq][ks >> 1][ks & 1] = *reinterpret_cast<const frag*>(smem + wb32[ks] + P256_SLOT(nq, b));   \
    } else if constexpr (!((DBG & 4) != 0)) {                                                               \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) {                                                  \
            wq[nq][nf][0] = *reinterpret_cast<const frag*>(smem + wb[0] + P256_SLOT(nq, b) + nf * 2048);    \
            wq[nq][nf][1] = *reinterpret_cast<const frag*>(smem + wb[1] + P256_SLOT(nq, b) + nf * 2048);    \
        }                                                                                                   \
    }
#define P256_MMA_HALF(mq, nq, s)                                                                             \
    if constexpr ((DBG & 128) != 0) {                                                                       \
        if constexpr (std::is_same<IN_T, __bf16>::value) {                                                  \
            _Pragma("unroll") for (int kq = 0; kq < 2; ++kq)                                                \
                _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                            \
                    acc32[nq][(mq) * 2 + mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                     \
                        wq[nq][s][kq], am[mb * 2 + (s)][kq], acc32[nq][(mq) * 2 + mb], 0, 0, 0);            \
        }                                                                                                   \
    } else {                                                                                                \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)                                                    \
            _Pragma("unroll") for (int mf = 0; mf < 4; ++mf)                                                \
                acc[(nq) * 2 + nf][(mq) * 4 + mf] = MfmaIn<IN_T>::mma(                                      \
                    wq[nq][nf][s], am[mf][s], acc[(nq) * 2 + nf][(mq) * 4 + mf]);                           \
    }
// wave priority inside / outside the MFMA part (A/B knob of tools/p256_prio_ab.sh: -DP256_PRIO=mma*4+rest; product 1 / 0)
#ifdef P256_PRIO
#define P256_PRIO_MMA ((P256_PRIO) / 4)
#define P256_PRIO_REST ((P256_PRIO) % 4)
#else
#define P256_PRIO_MMA 1
#define P256_PRIO_REST 0
#endif
// 16 MFMAs with MID (a slot's second LDS-DMA piece, or nothing) issued behind the first eight
#define P256_MMA(mq, nq, MID)                                                                                \
    if constexpr ((DBG & 1) != 0) {                                                                         \
        _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) asm volatile("" ::"v"(am[mf][0]), "v"(am[mf][1]));  \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) asm volatile("" ::"v"(wq[nq][nf][0]), "v"(wq[nq][nf][1])); \
        _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)                                                    \
            _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) asm volatile("" : "+v"(acc[(nq) * 2 + nf][(mq) * 4 + mf])); \
        MID;                                                                                                \
    } else {                                                                                                \
        __builtin_amdgcn_s_setprio(P256_PRIO_MMA);                                                          \
        P256_MMA_HALF(mq, nq, 0)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        MID;                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        P256_MMA_HALF(mq, nq, 1)                                                                            \
        __builtin_amdgcn_s_setprio(P256_PRIO_REST);                                                         \
    }
// counted wait: P256_INFLIGHT = the pieces of the younger slot loads that stay in flight (four slots and a half: 9); POST = the stores and pieces of the
// previous tile's epilogue are younger than the slot waited for
// WHICH = which of the K-tile's three waits (0, 1: behind the A stagings of phases 0 / 1; 3: behind W n1's in phase 3). With two
// operations per slot and wave they all leave 10 younger operations in flight (6 + 2w, 6 + 2w, 4 + 3w for w = 2 operations
// per W slot); an fp8 W slot is ONE operation per wave (w = 1): 8, 8, 7.
#define P256_WAIT(POST, WHICH)                                                                                 \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((W8 ? ((WHICH) == 3 ? P256_INFLIGHT - 3 : P256_INFLIGHT - 2) : P256_INFLIGHT) + ((POST) ? EX : 0)) : "memory")
#define P256_BARRIER()                                                \
    {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                            \
        if constexpr (!((DBG & 64) != 0)) __builtin_amdgcn_s_barrier(); \
        __builtin_amdgcn_sched_barrier(0);                            \
    }
// The slots the LATE half (wm = 1) reads in a phase are restaged by the early half right after the next barrier: its reads
// must have completed before that barrier (the early half's own reads complete a whole barrier earlier, with its MFMAs).
#ifdef P256_NO_LATE_WAIT   // (A/B experiment only: without it the restage-after-read order holds by timing, not by construction)
#define P256_LATE_READS_DONE()
#else
#define P256_LATE_READS_DONE() \
    if (wm == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
// One K-tile out of buffer B: four phases [reads + one slot's staging + wait] B [16 MFMAs] B. The wait that retires a slot
// sits in the read part of the phase BEFORE the one that reads it (both halves' waits and a barrier precede both halves'
// reads); a slot is restaged in the phase AFTER its last read. P0 / P1 / P3 = post-epilogue form of the three waits; FIN =
// this is K-tile 2 of a tile (the raw statistics landed behind the waits of K-tile 1).
// P256_STAGE_FIRST (A/B experiment, tools/p256_stagefirst_ab.sh): a phase's two LDS-DMA pieces issued BEFORE its fragment
// reads instead of behind them (they touch different slots: either order is correct)
#ifdef P256_STAGE_FIRST
#define P256_STAGE_A(...) P256_STAGE(__VA_ARGS__)
#define P256_STAGE_B(...)
#else
#define P256_STAGE_A(...)
#define P256_STAGE_B(...) P256_STAGE(__VA_ARGS__)
#endif
#define P256_KTILE(B, P0, P1, P3, FIN, HALF)                                                                       \
    {                                                                                                       \
        P256_STAGE_A(1, (B) ^ 1, oA1 + mA1, oW1, mA1 != 0);                                                 \
        P256_READ_A(B, 0);                                                                                  \
        P256_READ_W(B, 0);                                                                                  \
        P256_STAGE_B(1, (B) ^ 1, oA1 + mA1, oW1, mA1 != 0); /* A m1 of K-tile t+1 */                                          \
        P256_LATE_READS_DONE();                                                                             \
        P256_WAIT(P0, 0);                 /* retires W n1 of this K-tile */                                 \
        P256_BARRIER();                                                                                     \
        P256_MMA(0, 0, P256_STAGE_MID(1, (B) ^ 1, oA1 + mA1, oW1, mA1 != 0));                               \
        P256_BARRIER();                                                                                     \
        P256_STAGE_A(0, B, oA2, oW2, true);                                                                 \
        P256_READ_W(B, 1);                                                                                  \
        P256_STAGE_B(0, B, oA2, oW2, true);       /* A m0 of K-tile t+2 */                                          \
        P256_LATE_READS_DONE();                                                                             \
        P256_WAIT(P1, 1);                 /* retires A m1 of this K-tile */                                 \
        P256_BARRIER();                                                                                     \
        P256_MMA(0, 1, P256_STAGE_MID(0, B, oA2, oW2, true));                                               \
        P256_BARRIER();                                                                                     \
        P256_STAGE_A(2, B, oA2, oW2, true);                                                                 \
        if constexpr (!(HALF)) { P256_READ_A(B, 1); }                                                       \
        P256_STAGE_B(2, B, oA2, oW2, true);       /* W n0 of K-tile t+2; nothing new is read in the next phase: no wait */ \
        P256_LATE_READS_DONE();                                                                             \
        P256_BARRIER();                                                                                     \
        if constexpr (!(HALF)) { P256_MMA(1, 1, P256_STAGE_MID(2, B, oA2, oW2, true)); }                    \
        else { P256_STAGE_MID(2, B, oA2, oW2, true); }                                                      \
        P256_BARRIER();                                                                                     \
        P256_STAGE(3, B, oA2, oW2, true);       /* W n1 of K-tile t+2 */                                          \
        P256_FIN_HOOK(FIN);                                                                                 \
        P256_WAIT(P3, 3);                 /* retires A m0 / W n0 of the next K-tile */                      \
        P256_BARRIER();                                                                                     \
        if constexpr (!(HALF)) { P256_MMA(1, 0, P256_STAGE_MID(3, B, oA2, oW2, true)); }                    \
        else { P256_STAGE_MID(3, B, oA2, oW2, true); }                                                      \
        P256_BARRIER();                                                                                     \
        P256_ADVANCE();                                                                                     \
    }
// (encoder GEMM) K-tile 2 of a tile makes the (mean, rstd) table of the tile's rows in its last, read-free phase
#define P256_FIN_HOOK(FIN)                                                                                   \
    if constexpr (FOLD && !FINAL && (FIN)) {                                                                \
        if (kp == 1 && tid < 256) finalize_stats();                                                         \
    }
// position t+1 becomes the old t+2; t+2 moves on one K-tile (into the next tile, or wraps in the last one)
#define P256_ADVANCE()                                                                                       \
    oA1 = oA2; oW1 = oW2; mA1 = mA2;                                                                        \
    if (++k2 == nt) {                                                                                       \
        k2 = 0;                                                                                             \
        if (o2 + 1 < mine) ++o2;                                                                            \
        int bm_, bn_, hf_;                                                                                  \
        tile_of(o2, bm_, bn_, hf_);                                                                         \
        oA2 = bm_ * 256 * K * 2 + (hf_ == 2 ? 8 * row8 : 0);                                                \
        oW2 = bn_ * 256 * K * 2;                                                                            \
        mA2 = hf_ ? 0 : 8 * row8;                                                                           \
        if constexpr ((DBG & 8) != 0) { oA2 = 0; oW2 = 0; }                                                 \
    } else {                                                                                                \
        oA2 += GEMM_BK * 2; oW2 += GEMM_BK * 2;                                                             \
    }

    auto finalize_stats = [&]() {  // thread t: (mean, rstd) of the tile's row t from its K/64 partial (sum, sumsq)
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
    };

    // ---- stream state
    int cbm, cbn, chalf;
    tile_of(0, cbm, cbn, chalf);
    int oA1 = cbm * 256 * K * 2 + (chalf == 2 ? 8 * row8 : 0), oW1 = cbn * 256 * K * 2;  // K-tile 0 of tile 0, then position t+1
    if constexpr ((DBG & 8) != 0) { oA1 = 0; oW1 = 0; }
    int oA2 = oA1 + GEMM_BK * 2, oW2 = oW1 + GEMM_BK * 2;
    int mA1 = chalf ? 0 : 8 * row8, mA2 = mA1;   // where a position's A m1 slot comes from (a half tile has none: m0 again)
    int o2 = 0, k2 = 1;
    const int dump_row = M - 1;  // rows >= m_valid are stored to the last pad row (the store COUNT must not depend on data)

    // prologue: K-tile 0 completely, K-tile 1 except its A m1 slot (staged by phase 0 of K-tile 0)
    stage_x(cbm, cbn, 0);
    {
        char* d0 = smem + stage_dst;
        P256_BLDS(srdA, a_vo, oA1, d0 + P256_SLOT(0, 0)); P256_BLDS(srdA, a_vo, oA1 + row8, d0 + P256_SLOT(0, 0) + 1024);
        P256_BLDS(srdW, w_vo, oW1, d0 + P256_SLOT(2, 0)); P256_BLDS(srdW, w_vo, oW1 + row8, d0 + P256_SLOT(2, 0) + 1024);
        P256_BLDS(srdW, w_vo, oW1 + 4 * row8, d0 + P256_SLOT(3, 0)); P256_BLDS(srdW, w_vo, oW1 + 5 * row8, d0 + P256_SLOT(3, 0) + 1024);
        P256_BLDS(srdA, a_vo, oA1 + mA1, d0 + P256_SLOT(1, 0)); P256_BLDS(srdA, a_vo, oA1 + mA1 + row8, d0 + P256_SLOT(1, 0) + 1024);
        P256_BLDS(srdA, a_vo, oA2, d0 + P256_SLOT(0, 1)); P256_BLDS(srdA, a_vo, oA2 + row8, d0 + P256_SLOT(0, 1) + 1024);
        P256_BLDS(srdW, w_vo, oW2, d0 + P256_SLOT(2, 1)); P256_BLDS(srdW, w_vo, oW2 + row8, d0 + P256_SLOT(2, 1) + 1024);
        P256_BLDS(srdW, w_vo, oW2 + 4 * row8, d0 + P256_SLOT(3, 1)); P256_BLDS(srdW, w_vo, oW2 + 5 * row8, d0 + P256_SLOT(3, 1) + 1024);
    }
    // position t+1 = K-tile 1 (its A m1 slot is still to come), t+2 = K-tile 2
    oA1 = oA2; oW1 = oW2;
    oA2 += GEMM_BK * 2; oW2 += GEMM_BK * 2;
    k2 = 2;
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // all but the five youngest slot loads: A m0 / W n0 of K-tile 0 (and the epilogue pieces) have landed
    P256_BARRIER();
    if (wm == 1) P256_BARRIER();  // the lower half runs one barrier behind from here on

    // ---- epilogue of the tile (cbm, cbn) out of the accumulators, JN = 8 (whole tile) or 4 (half tile: 64 rows per wave,
    // hrow = 0 / 64 for its m0 / m1 rows) 16-row sub-tiles per wave. Runs while the next tile's first K-tiles are in flight.
    auto epilogue = [&](auto jn_c, int hrow, int par) {
        constexpr int JN = decltype(jn_c)::value;
        if constexpr (!((DBG & 16) != 0)) {
        int lane_e = lane;  // (opaque here: what the epilogue derives from the lane id is not kept live through the K loop)
        asm volatile("" : "+v"(lane_e));
        const int fr = lane_e & 15, fg = lane_e >> 4;
        const char* bc = smem + P256_BC + par * 2048;
        f32x4 bias[4], cvec[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bias[i] = *reinterpret_cast<const f32x4*>(bc + (wn * 64 + i * 16 + 4 * fg) * 4);
            if constexpr (FOLD) cvec[i] = *reinterpret_cast<const f32x4*>(bc + 1024 + (wn * 64 + i * 16 + 4 * fg) * 4);
        }
        float mu[JN], rs[JN];
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            mu[j] = 0.f; rs[j] = 1.f;
            if constexpr (FOLD) {
                const float* tb = reinterpret_cast<const float*>(smem + (FINAL ? P256_FTAB + par * 2048 : P256_TABLE)) +
                                  2 * (wm * 128 + hrow + j * 16 + fr);
                mu[j] = tb[0]; rs[j] = tb[1];
            }
        }
        uint16_t* outp = reinterpret_cast<uint16_t*>(ep.out);
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc(ep.out, 0, 0x7fffffff, 0x00020000);
        const int st_pol = ep.nt_out;  // cache policy of the output stores (launch_gemm256p_inst)
        if constexpr (STYLE == 0) {
            // Each wave transposes 16 rows x 64 columns at a time through a private 2 KB patch (its own share of the raw
            // statistics area, dead since K-tile 2 and refilled by this wave only after these stores). 8-byte slot s of
            // row r lies at slot s ^ 2 (r & 7).
            char* patch = smem + P256_RAW + wave * (XP >= 2 ? XP * 1024 : 2048);  // = the bytes stage_x() of THIS wave refills
            const int rrow = lane_e >> 3, rchunk = lane_e & 7;
            const int wr_off = fr * 128, wr_sw = 2 * (fr & 7);
            const int col = cbn * 256 + wn * 64 + rchunk * 8;
#pragma unroll
            for (int j = 0; j < JN; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float y[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (FOLD) y[r] = rs[j] * (acc_get(i, j, r) - mu[j] * cvec[i][r]) + bias[i][r];
                        else y[r] = acc_get(i, j, r) + bias[i][r];
                        if constexpr (GELU) y[r] = quick_gelu(y[r]);
                    }
                    u32x2 pk;
                    pk[0] = pack_bf16x2(y[0], y[1]);
                    pk[1] = pack_bf16x2(y[2], y[3]);
                    *reinterpret_cast<u32x2*>(patch + wr_off + (((i * 4 + fg) ^ wr_sw) << 3)) = pk;
                    acc_zero(i, j);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    const int row = rh * 8 + rrow;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(patch + row * 128 + (((2 * rchunk) ^ (2 * (row & 7))) << 3));
                    const int m = cbm * 256 + wm * 128 + hrow + j * 16 + row;
                    const int vo = ((m < ep.m_valid ? m : dump_row) * ep.ldo + col) * 2;
                    if constexpr ((DBG & 32) != 0) asm volatile("" ::"v"(v));
                    else if (st_pol == 16) __builtin_amdgcn_raw_buffer_store_b128(v, srdO, vo, 0, 16);   // sc1: written through, dropped from L2
                    else if (st_pol == 2) __builtin_amdgcn_raw_buffer_store_b128(v, srdO, vo, 0, 2);      // nt
                    else __builtin_amdgcn_raw_buffer_store_b128(v, srdO, vo, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next j
            }
        } else {
            const int colb = cbn * 256 + wn * 64 + (fg >> 1) * 8 + (fg & 1) * 16;  // + pair * 32
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                uint32_t pk[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float y[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (FOLD) y[r] = rs[j] * (acc_get(i, j, r) - mu[j] * cvec[i][r]) + bias[i][r];
                        else y[r] = acc_get(i, j, r) + bias[i][r];
                        if constexpr (GELU) y[r] = quick_gelu(y[r]);
                    }
                    pk[i][0] = pack_bf16x2(y[0], y[1]);
                    pk[i][1] = pack_bf16x2(y[2], y[3]);
                    acc_zero(i, j);
                }
                const int m = cbm * 256 + wm * 128 + hrow + j * 16 + fr;
                uint16_t* orow = outp + (size_t)(m < ep.m_valid ? m : dump_row) * ep.ldo + colb;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    // lanes of the odd 16-lane rows take the even rows' values of the second 16-column block and give
                    // their values of the first: every lane ends up with 8 consecutive columns
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
                    u32x4 v;
                    v[0] = s0[0]; v[1] = s1[0]; v[2] = s0[1]; v[3] = s1[1];
                    if constexpr ((DBG & 32) != 0) asm volatile("" ::"v"(v));
                    else *reinterpret_cast<u32x4*>(orow + pr * 32) = v;
                }
            }
        }
        }  // (DBG & 16)
    };

    const int npair = nt >> 1;
    // ---- whole tiles
    for (int ti = 0; ti < nfull; ++ti) {
        for (int kp = 0; kp < npair; ++kp) {
            if (kp == 0 && ti > 0) {
                P256_KTILE(0, true, true, true, false, false);
                P256_KTILE(1, true, false, false, false, false);
            } else {
                P256_KTILE(0, false, false, false, true, false);
                P256_KTILE(1, false, false, false, false, false);
            }
        }
        // The tile is complete. The upper half waits one barrier for the lower half's last MFMAs, both run the epilogue side
        // by side, then the lower half falls one barrier behind again.
        if (wm == 0) P256_BARRIER();
        epilogue(std::integral_constant<int, 8>{}, 0, ti & 1);
        if (ti + 1 < mine) {
            tile_of(ti + 1, cbm, cbn, chalf);
            stage_x(cbm, cbn, (ti + 1) & 1);
            if (wm == 1) P256_BARRIER();
        }
    }
    // ---- the half tile of the last round, if this workgroup has one (always behind at least one whole tile: POST waits).
    // Out of the loop above so that the upper accumulators are plainly dead here (inside it the compiler kept them alive
    // through the half tile's K loop and spilled).
    if (nhalf) {
        for (int kp = 0; kp < npair; ++kp) {
            if (kp == 0) {
                P256_KTILE(0, true, true, true, false, true);
                P256_KTILE(1, true, false, false, false, true);
            } else {
                P256_KTILE(0, false, false, false, true, true);
                P256_KTILE(1, false, false, false, false, true);
            }
        }
        if (wm == 0) P256_BARRIER();
        epilogue(std::integral_constant<int, 4>{}, chalf == 2 ? 64 : 0, nfull & 1);
    }
    if constexpr ((DBG & 16) != 0) {  // (ablation without epilogue: keep the accumulators alive)
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] += acc_get(i, j, r);
        if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) reinterpret_cast<float*>(ep.out)[tid] = sum[0];
    }
    // the unconditional staging of the last K-tiles is still in flight: LDS must not be handed on with DMA writes pending
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef P256_FIN_HOOK
#undef P256_ADVANCE
#define P256_FIN_HOOK(FIN)
#define P256_ADVANCE()                                                                                       \
    oA1 = oA2; oW1 = oW2;                                                                                   \
    if (++k2 == nt) {                                                                                       \
        k2 = 0;                                                                                             \
        if (o2 + 1 < mine) ++o2;                                                                            \
        oA2 = 0; oW2 = o2 * 256 * K * WELT;                                                                 \
    } else {                                                                                                \
        oA2 += GEMM_BK * (int)sizeof(IN); oW2 += GEMM_BK * WELT;                                            \
    }

// ------------------------------------------------------------------------------------------------
// The retrieval score GEMM (GROUPMAX epilogue, gemm256_strip_kernel's contract: a workgroup walks `strip` consecutive
// 256-row tiles of the index for one 256-query tile as ONE K-tile stream) on the loop of the persistent kernel above:
// staggered wave halves, buffer-form LDS-DMA with scalar offsets, immediates for every LDS address. The epilogue is
// registers only (16-row group maxima; dense stores or, FILTER, rare threshold-passing appends), so the counted waits never
// have to allow for it. K = D of the index (a multiple of 128).
// ------------------------------------------------------------------------------------------------
// FILTER: 0 = dense group maxima, 1 = (group maximum, group) appended where the maximum reaches the query's tau, 2 = ROW
// threshold pass of the widen pass (api_index.hip sweep_queries): every index row whose score reaches tau_q is appended to
// the query's list flt.buf_g[m][..] (row ids in any order; flt.cnt[m] may run past cap: the excess is dropped and the
// query goes to the exhaustive pass).
// W8 (round 4): the index rows are fp8 (MMISS_F8: e4m3 codes of 128 x) — half the staged bytes per row. A W slot is then
// 128 rows x 64 bytes: one 16-row LDS-DMA piece per wave (lane = row * 4 + 16-byte unit, unit XOR-swizzled by (row >> 2) & 3
// so that the 8-byte fragment reads of rows r, r + 4, r + 8, r + 12 — one bank window — fall on four different chunks), a
// fragment is one ds_read_b64 widened exactly to f16 in registers (p256_widen_f8), the MFMA stays the f16 one with the f16
// queries; scores come out times 128 (the codes' fixed scale) and are scaled back where they leave the registers.
template <typename IN, int FILTER, bool W8 = false>
__global__ __launch_bounds__(512, 2) void gemm256s_kernel(const IN* __restrict__ A, const void* __restrict__ Wv, int M, int N, int K,
                                                          int strip, GemmEpi ep, StripFilter flt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef IN IN_T;
    typedef typename MfmaIn<IN>::frag frag;
    static_assert(!W8 || std::is_same<IN, _Float16>::value, "fp8 rows are widened to f16");
    constexpr int WELT = W8 ? 1 : (int)sizeof(IN);
    constexpr float WSCALE = W8 ? 1.0f / 128.0f : 1.0f;
    constexpr int DBG = 0;
    uint32_t ab32[4], wb32[4]; f32x16 acc32[2][4];   // (named by the loop macros' DBG & 128 branches only: never touched here)
    constexpr bool FOLD = false;
    // fp8 rows carry one inverse norm each (ep.aux, f8_row_inv_kernel): the 64 of a wave's column quarter arrive by a 4-byte
    // LDS-DMA piece per wave into the wave's private 256 bytes behind the staging slots. The piece is issued at the head of
    // EVERY K-tile pair (the current index tile's values again: 256 bytes per wave), so that the operation counts of the
    // waits are the same in every pair — the four waits behind a piece allow for one more operation in flight (the POST
    // form of the waits, EX = 1), the fifth retires it. (Issued once per tile between the K streams, with plain and POST
    // pairs side by side in the loop as in gemm256p_kernel, this kernel spilled 270 registers.)
    constexpr int EX = W8 ? 1 : 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nbm = M >> 8, nbn = N >> 8;
    const int nt = K / GEMM_BK;
    const int bn_begin = FILTER != 0 ? flt.bn_begin : 0;
    const int nstrips = (nbn - bn_begin + strip - 1) / strip;
    const int wg = xcd_remap(blockIdx.x, nbm * nstrips);
    const int sidx = wg / nbm, bm = wg - sidx * nbm;   // m fastest: the M-tiles of one strip run side by side
    const int bn0 = bn_begin + sidx * strip;
    const int mine = (nbn - bn0 < strip) ? nbn - bn0 : strip;
    const IN* Ab = A + (size_t)bm * 256 * K;
    const char* Wb = reinterpret_cast<const char*>(Wv) + (size_t)bn0 * 256 * K * WELT;  // offsets inside a strip fit 32 bits (strip <= 32 tiles of <= 1 MB)

    const int r_in = lane >> 3, p = lane & 7;
    const int src_chunk = (p ^ r_in) * 8;
    const int a_vo = (((wave >> 2) * 128 + (wave & 3) * 16 + r_in) * K + src_chunk) * (int)sizeof(IN);
    // W slot row r = weight row (r >> 5) * 64 + nq * 32 + (r & 31); a wave stages slot rows 16 * wave .. + 15: as two 8-row
    // pieces of 128-byte rows (f16), or as ONE piece of sixteen 64-byte rows (fp8: lane = 4 * row + unit)
    const int r8 = lane >> 2;   // fp8: the lane's row of the piece
    const int w_vo = W8 ? (((wave >> 1) * 64 + (wave & 1) * 16 + r8) * K + (((lane & 3) ^ ((r8 >> 2) & 3)) << 4))
                        : (((wave >> 1) * 64 + (wave & 1) * 16 + r_in) * K + src_chunk) * (int)sizeof(IN);
    const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<IN*>(Ab), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wb), 0, 0x7fffffff, 0x00020000);
    const int row8 = 8 * K * (int)sizeof(IN);
    const int w_row8 = 8 * K * WELT;
    const int stage_dst = wave * 2048;
    const int mA1 = 8 * row8;
    uint32_t ab[2], wb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        ab[s] = (wm * 64 + fr) * 128 + (((4 * s + fg) ^ (fr & 7)) << 4);
        wb[s] = W8 ? P256_SLOT(2, 0) + (wn * 32 + fr) * 64 + (((4 * s + fg) ^ (2 * ((fr >> 2) & 3))) << 3)
                   : P256_SLOT(2, 0) + (wn * 32 + fr) * 128 + (((4 * s + fg) ^ (fr & 7)) << 4);
    }
    frag am[4][2];
    frag wq[2][2][2];
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float tau_r[8];  // FILTER: thresholds of this lane's 8 queries, complete before the stream starts
#pragma unroll
    for (int j = 0; j < 8; ++j) tau_r[j] = INFINITY;
    if constexpr (FILTER != 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = bm * 256 + wm * 128 + j * 16 + fr;
            if (m < ep.m_valid) tau_r[j] = flt.tau[(size_t)m * flt.tau_stride];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(tau_r[j]));
    }

    int oA1 = 0, oW1 = 0, oA2 = GEMM_BK * (int)sizeof(IN), oW2 = GEMM_BK * WELT;
    int o2 = 0, k2 = 1;
    auto stage_inv = [&](int ti_) {   // the inverse norms of index tile bn0 + ti_, this wave's 64 rows (both halves load their copy)
        if constexpr (W8) {
            const float* src = ep.aux + (size_t)(bn0 + ti_) * 256 + wn * 64 + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(smem + G256_LDS + wave * 256), 4, 0, 0);
        }
    };
    {
        char* d0 = smem + stage_dst;
        // (the same operations per slot and wave as in the loop, in the same order: the counted waits depend on it)
#define P256_PRO_W(which, b, oW)                                                                                      \
        if constexpr (W8) {                                                                                          \
            P256_BLDS(srdW, w_vo, (oW) + ((which) & 1) * 4 * w_row8, smem + P256_SLOT(which, b) + wave * 1024);      \
        } else {                                                                                                     \
            P256_BLDS(srdW, w_vo, (oW) + ((which) & 1) * 4 * w_row8, d0 + P256_SLOT(which, b));                      \
            P256_BLDS(srdW, w_vo, (oW) + ((which) & 1) * 4 * w_row8 + w_row8, d0 + P256_SLOT(which, b) + 1024);      \
        }
        P256_BLDS(srdA, a_vo, oA1, d0 + P256_SLOT(0, 0)); P256_BLDS(srdA, a_vo, oA1 + row8, d0 + P256_SLOT(0, 0) + 1024);
        P256_PRO_W(2, 0, oW1)
        P256_PRO_W(3, 0, oW1)
        P256_BLDS(srdA, a_vo, oA1 + mA1, d0 + P256_SLOT(1, 0)); P256_BLDS(srdA, a_vo, oA1 + mA1 + row8, d0 + P256_SLOT(1, 0) + 1024);
        P256_BLDS(srdA, a_vo, oA2, d0 + P256_SLOT(0, 1)); P256_BLDS(srdA, a_vo, oA2 + row8, d0 + P256_SLOT(0, 1) + 1024);
        P256_PRO_W(2, 1, oW2)
        P256_PRO_W(3, 1, oW2)
#undef P256_PRO_W
    }
    oA1 = oA2; oW1 = oW2;
    oA2 += GEMM_BK * (int)sizeof(IN); oW2 += GEMM_BK * WELT;
    k2 = 2;
    // all but the five youngest slot loads (4 + 3w operations: 10, or 7 with one operation per fp8 W slot)
    if constexpr (W8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    P256_BARRIER();
    if (wm == 1) P256_BARRIER();

    const int npair = nt >> 1;
    float* out = reinterpret_cast<float*>(ep.out);
    for (int ti = 0; ti < mine; ++ti) {
        for (int kp = 0; kp < npair; ++kp) {
            stage_inv(ti);   // (W8; younger than the slots the next four waits retire, older than what the fifth retires)
            P256_KTILE(0, W8, W8, W8, false, false);
            P256_KTILE(1, W8, false, false, false, false);
        }
        // ---- this index tile is complete: group maxima out (the group numbering of gemm256_kernel), accumulators reset.
        // A few hundred cycles of VALU: it runs without leaving the staggered rhythm (each half does it in front of its next
        // read part, the other half is in its MFMAs), and only the index's last tile pays for masking the pad rows.
        const int bn = bn0 + ti;
        const int g = (bn * 4 + wn) * 4 + fg;
        const int n0 = bn * 256 + wn * 64 + 4 * fg;
        const bool whole = (bn + 1) * 256 <= ep.p0;
        asm volatile("s_nop 11" ::: "memory");   // the asm maxima below read the accumulators of the tile's last MFMAs (mm_mfma_settle)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (W8) {
            // scores of fp8 rows: x inv[row] (this wave's own piece: the one issued at the head of the tile's first pair landed
            // behind that pair's fifth wait, later ones rewrite the same bytes)
            const char* ivp = smem + G256_LDS + wave * 256 + fg * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 iv = *reinterpret_cast<const f32x4*>(ivp + i * 64);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = acc[i][j] * iv;
            }
        }
        // Appending to a query's list is an atomic add WITH return (the slot) followed by the stores into that slot. Round 6: the
        // slots of a tile's eight 16-query blocks are requested FIRST — one atomic per (lane, block) that has something to append,
        // for all of the lane's passing rows at once — and used after the last request: one exposed round trip per tile and wave.
        // Requested where they were used (rounds 3-5) they were up to eight sequential round trips of ~0.4 us per tile with the
        // other waves waiting at the next barrier: the widen pass of the bench step took 78 us with 66 rows per query passing,
        // 117 us with 147, against 43 us with 10 (tools/sweep_probe.py, profiles/sweep_r06.txt).
        if constexpr (FILTER == 2) {
            // ... and a lane keeps WHICH of its 16 rows pass as a bit mask: the slow path is 16 compares into the mask, one popcount, one
            // atomic, and later a loop over the set bits (one or two) — as 16 + 16 branches around a counter and a store per block it
            // was ~2500 vector instructions per wave and tile on an index where most blocks have a passing row, two waves per SIMD.
            uint32_t pmask[8];
            int pos[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = bm * 256 + wm * 128 + j * 16 + fr;
                float mx = mm_max3(acc[0][j][0], acc[0][j][1], acc[0][j][2]);   // (mm_max*: no canonicalising v_max per operand)
                mx = mm_max3(mx, acc[0][j][3], acc[1][j][0]);
#pragma unroll
                for (int i = 1; i < 4; ++i) {
                    mx = mm_max3(mx, acc[i][j][1], acc[i][j][2]);
                    mx = i < 3 ? mm_max3(mx, acc[i][j][3], acc[i + 1][j][0]) : mm_max2(mx, acc[i][j][3]);
                }
                mx *= WSCALE;   // (fp8 rows: the codes are 128 x the stored values; a power of two, exact)
                uint32_t pm = 0;
                if (mx >= tau_r[j]) {   // rare: some row of this lane's 16 reaches the query's threshold (tau_r = +inf for pad queries)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            pm |= (acc[i][j][r] * WSCALE >= tau_r[j] && (whole || n0 + i * 16 + r < ep.p0)) ? (1u << (i * 4 + r)) : 0u;
                }
                pmask[j] = pm;
                pos[j] = 0;
                if (pm) pos[j] = atomicAdd(flt.cnt + m, __popc(pm));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t pm = pmask[j];
                if (pm) {
                    const int m = bm * 256 + wm * 128 + j * 16 + fr;
                    int p = pos[j];
                    while (pm) {
                        const int b = __ffs(pm) - 1;
                        pm &= pm - 1;
                        if (p < flt.cap) flt.buf_g[(size_t)m * flt.cap + p] = n0 + (b >> 2) * 16 + (b & 3);
                        ++p;
                    }
                }
            }
            continue;
        }
        // (FILTER = 1: a block's maximum and its slot wait in two of the block's own — consumed, zeroed — accumulator registers
        // until the second loop: sixteen more live registers were four spilled ones)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float mx = -INFINITY;
            if (whole) {
                // 16 values: 8 three-way maxima without the canonicalising v_max per operand (mm_max3, common.h)
                mx = mm_max3(acc[0][j][0], acc[0][j][1], acc[0][j][2]);
                mx = mm_max3(mx, acc[0][j][3], acc[1][j][0]);
#pragma unroll
                for (int i = 1; i < 4; ++i) {
                    mx = mm_max3(mx, acc[i][j][1], acc[i][j][2]);
                    mx = i < 3 ? mm_max3(mx, acc[i][j][3], acc[i + 1][j][0]) : mm_max2(mx, acc[i][j][3]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = (n0 + i * 16 + r < ep.p0) ? acc[i][j][r] : -INFINITY;
                        mx = fmaxf(mx, v);
                        acc[i][j][r] = 0.f;
                    }
            }
            const int m = bm * 256 + wm * 128 + j * 16 + fr;
            mx *= WSCALE;
            if constexpr (FILTER == 1) {
                int pos_ = -1;
                if (mx >= tau_r[j] && mx > -INFINITY)   // (tau_r = +inf for pad queries; -inf maxima = all-pad groups)
                    pos_ = atomicAdd(flt.cnt + m, 1);
                acc[0][j][0] = mx;
                acc[0][j][1] = __int_as_float(pos_);
            } else {
                if (m < ep.m_valid) out[(size_t)m * ep.ldo + g] = mx;
            }
        }
        if constexpr (FILTER == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = bm * 256 + wm * 128 + j * 16 + fr;
                const int pos_ = __float_as_int(acc[0][j][1]);
                if (pos_ >= 0 && pos_ < flt.cap) {
                    flt.buf_s[(size_t)m * flt.cap + pos_] = acc[0][j][0];
                    flt.buf_g[(size_t)m * flt.cap + pos_] = g;
                }
                acc[0][j][0] = 0.f;
                acc[0][j][1] = 0.f;
            }
        }
    }
    if (wm == 0) P256_BARRIER();   // (the upper half's last barrier: the lower half is still one behind)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#undef P256_STAGE
#undef P256_STAGE_A
#undef P256_STAGE_B
#undef P256_STAGE_P
#undef P256_STAGE_MID
#undef P256_INFLIGHT
#undef P256_MMA_HALF
#undef P256_SLOT
#undef P256_BLDS
#undef P256_AUX_srdA
#undef P256_AUX_srdW
#undef P256_BLDS4
#undef P256_READ_A
#undef P256_READ_W
#undef P256_MMA
#undef P256_WAIT
#undef P256_BARRIER
#undef P256_KTILE
#undef P256_FIN_HOOK
#undef P256_ADVANCE
#undef P256_LATE_READS_DONE

template <int EPI, int XP, int STYLE, int DBG>
static int launch_gemm256p_kern(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K) {
    const int lds = XP == 1 ? P256_FTAB + 4096 : P256_RAW + (XP >= 2 ? XP * 8192 : 16384);  // (the epilogue's transpose patches live in the raw-statistics area)
    const int T = (M / 256) * (N / 256);
    const int grid = T >= 256 ? 256 : T;
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm256p_kernel<EPI, XP, STYLE, DBG>), lds));
    hipLaunchKernelGGL((gemm256p_kernel<EPI, XP, STYLE, DBG>), dim3(grid), dim3(512), lds, st, reinterpret_cast<const __bf16*>(A),
                       reinterpret_cast<const __bf16*>(W), M, N, K, ep);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

template <int EPI, int XP>
static int launch_gemm256p_inst(hipStream_t st, const void* A, const void* W, const GemmEpi& ep_in, int M, int N, int K) {
    GemmEpi ep = ep_in;
    if (ep.m_fast == 0) ep.m_fast = mmiss_option("gemm_p256_band", 8);  // row blocks per band of the global tile order
    // Cache policy of the output stores: 0 plain, 2 nt, 16 sc1 (written through, not kept in the XCD's L2). Measured equal in
    // time and in fabric reads (FETCH_SIZE x 2 = 119-126 MB per FC1 launch either way): one round of an XCD's 32 tiles streams
    // 4.6 MB of distinct operand bytes through its 4 MB L2, so nothing survives to the next round whatever the stores do.
    ep.nt_out = mmiss_option("gemm_p256_store", 0);
#ifdef MMISS_EXPERIMENTS
    if constexpr (EPI == MMISS_EPI_LNFOLD_QGELU_BF16 && XP == 3) {
        const int dbg = mmiss_option("gemm_p256_dbg", 0);  // timing ablations (wrong results by construction)
#define P256_DBG_CASE(D) if (dbg == D) return launch_gemm256p_kern<EPI, XP, 0, D>(st, A, W, ep, M, N, K);
        P256_DBG_CASE(1) P256_DBG_CASE(2) P256_DBG_CASE(4) P256_DBG_CASE(8) P256_DBG_CASE(5) P256_DBG_CASE(6) P256_DBG_CASE(7)
        P256_DBG_CASE(16) P256_DBG_CASE(32) P256_DBG_CASE(64) P256_DBG_CASE(23) P256_DBG_CASE(39) P256_DBG_CASE(87)
        P256_DBG_CASE(128) P256_DBG_CASE(136) P256_DBG_CASE(144) P256_DBG_CASE(132)
#undef P256_DBG_CASE
        if (dbg) MM_FAIL(MMISS_ERR_ARG, "gemm_p256_dbg = %d is not compiled", dbg);
    }
    if (mmiss_option("gemm_p256_style", 0) == 1) return launch_gemm256p_kern<EPI, XP, 1, 0>(st, A, W, ep, M, N, K);
#endif
    return launch_gemm256p_kern<EPI, XP, 0, 0>(st, A, W, ep, M, N, K);
}

// can this GEMM run on the persistent kernel? (bf16 output epilogues; the folded forms need K = 512 or 768: the raw
// statistics of a tile must fit beside the staging buffers)
static inline bool gemm256p_ok(int epi, int M, int N, int K, bool final_stats = false) {
    if (M <= 0 || (M % 256) || N <= 0 || (N % 256) || K < 256 || (K % 256)) return false;
    if ((int64_t)(M / 256) * (N / 256) > 48 * 256 || M / 256 > 0xffff) return false;  // (tile table: 64 entries per workgroup)
    if ((int64_t)M * N * 2 >= (1LL << 31) || (int64_t)M * K * 2 >= (1LL << 31) || (int64_t)N * K * 2 >= (1LL << 31)) return false;  // (32-bit buffer offsets)
    if (epi == MMISS_EPI_LNFOLD_BF16 || epi == MMISS_EPI_LNFOLD_QGELU_BF16) return final_stats || K == 512 || K == 768;
    return epi == MMISS_EPI_BIAS_BF16 || epi == MMISS_EPI_BIAS_QGELU_BF16;
}

// M = rows padded to 256 (the output and, in the folded modes, ep.ln_stats must hold M rows); rows >= ep.m_valid are
// computed but land in row M - 1.
static int launch_gemm256p(hipStream_t st, int epi, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K) {
    const bool fold = epi == MMISS_EPI_LNFOLD_BF16 || epi == MMISS_EPI_LNFOLD_QGELU_BF16;
    const bool fin = fold && ep.ln_final != nullptr;
    if (!gemm256p_ok(epi, M, N, K, fin)) MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm256p: epi=%d M=%d N=%d K=%d", epi, M, N, K);
    if (!ep.out || !ep.bias || (fold && (!ep.aux || (!fin && (!ep.ln_stats || ep.ln_parts * 64 != K)))))
        MM_FAIL(MMISS_ERR_ARG, "gemm256p: missing operand (fold=%d parts=%d)", (int)fold, ep.ln_parts);
    static const char* names[] = {"", "gemm_bf16_bias_p256", "gemm_bf16_bias_qgelu_p256"};
    const int mv = ep.m_valid < M ? ep.m_valid : M;
    const double bytes = 2.0 * ((double)mv * K + (double)N * K) + 2.0 * (double)mv * N;
    MM_PROF(fold ? (epi == MMISS_EPI_LNFOLD_BF16 ? "gemm_bf16_lnfold_bias_p256" : "gemm_bf16_lnfold_qgelu_p256") : names[epi], st,
            gemm_flops(mv, N, K), bytes);
    switch (epi) {
        case MMISS_EPI_BIAS_BF16: return launch_gemm256p_inst<MMISS_EPI_BIAS_BF16, 0>(st, A, W, ep, M, N, K);
        case MMISS_EPI_BIAS_QGELU_BF16: return launch_gemm256p_inst<MMISS_EPI_BIAS_QGELU_BF16, 0>(st, A, W, ep, M, N, K);
        case MMISS_EPI_LNFOLD_BF16:
            if (fin) return launch_gemm256p_inst<MMISS_EPI_LNFOLD_BF16, 1>(st, A, W, ep, M, N, K);
            return K == 768 ? launch_gemm256p_inst<MMISS_EPI_LNFOLD_BF16, 3>(st, A, W, ep, M, N, K)
                            : launch_gemm256p_inst<MMISS_EPI_LNFOLD_BF16, 2>(st, A, W, ep, M, N, K);
        default:
            if (fin) return launch_gemm256p_inst<MMISS_EPI_LNFOLD_QGELU_BF16, 1>(st, A, W, ep, M, N, K);
            return K == 768 ? launch_gemm256p_inst<MMISS_EPI_LNFOLD_QGELU_BF16, 3>(st, A, W, ep, M, N, K)
                            : launch_gemm256p_inst<MMISS_EPI_LNFOLD_QGELU_BF16, 2>(st, A, W, ep, M, N, K);
    }
}

// the strip score GEMM on the staggered loop (same arguments as launch_gemm256_strip; K % 128 == 0). W8: W = fp8 rows (1 B / elt)
template <typename IN, bool W8 = false>
static int launch_gemm256s(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K, int strip,
                           const StripFilter* flt = nullptr) {
    if (M <= 0 || N <= 0 || K < 256 || (M % 256) || (N % 256) || (K % 128) || strip < 1 || strip > 32 ||
        (int64_t)strip * 256 * K * (int64_t)sizeof(IN) >= (1LL << 31) || (int64_t)256 * K * (int64_t)sizeof(IN) >= (1LL << 31))
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm256s: M=%d N=%d K=%d strip=%d", M, N, K, strip);
    const int nbn = N / 256;
    if (W8 && !ep.aux) MM_FAIL(MMISS_ERR_ARG, "gemm256s: fp8 rows need their inverse norms (ep.aux)");
    const int lds = G256_LDS + (W8 ? 2048 : 0);   // + 8 waves x 64 inverse norms
    if (flt) {
        const bool rows_mode = flt->buf_s == nullptr;   // (the row threshold pass keeps no scores)
        if (flt->bn_begin < 0 || flt->bn_begin >= nbn || !flt->tau || !flt->cnt || !flt->buf_g || flt->cap <= 0)
            MM_FAIL(MMISS_ERR_ARG, "gemm256s: bad filter");
        const int nwg = (M / 256) * ((nbn - flt->bn_begin + strip - 1) / strip);
        if (rows_mode) {
            MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm256s_kernel<IN, 2, W8>), lds));
            hipLaunchKernelGGL((gemm256s_kernel<IN, 2, W8>), dim3(nwg), dim3(512), lds, st, reinterpret_cast<const IN*>(A),
                               W, M, N, K, strip, ep, *flt);
        } else {
            MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm256s_kernel<IN, 1, W8>), lds));
            hipLaunchKernelGGL((gemm256s_kernel<IN, 1, W8>), dim3(nwg), dim3(512), lds, st, reinterpret_cast<const IN*>(A),
                               W, M, N, K, strip, ep, *flt);
        }
    } else {
        MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm256s_kernel<IN, 0, W8>), lds));
        const int nwg = (M / 256) * ((nbn + strip - 1) / strip);
        hipLaunchKernelGGL((gemm256s_kernel<IN, 0, W8>), dim3(nwg), dim3(512), lds, st, reinterpret_cast<const IN*>(A),
                           W, M, N, K, strip, ep, StripFilter{});
    }
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// gemm_bf16_p160.h — 160 x 256 x 64 bf16 GEMM tile on a staggered two-barrier loop (the discipline of gemm_bf16_p256.h), for
// the NARROW GEMMs of a ViT-B/32 layer on the bf16 residual stream (MMISS_EPI_BIAS_RESID_BF16: out = bf16(f32(out) + A W^T +
// bias) in place, + the row statistics of the new rows): out-projection 12800 x 768 x 768 and FC2 12800 x 768 x 3072.
//
// Why this tile: 12 800 rows x 768 columns are 150 tiles of 256 x 256 — 0.59 of the chip — and 300 of 128 x 256; as 160 x 256
// they are 80 x 3 = 240 workgroups = ONE round on 256 CUs. The 160 x 128 tile of gemm_bf16.h (the same 240-wide grid, two
// workgroups per CU) runs the older loop — one barrier per K-tile behind an s_waitcnt vmcnt(0) — at 0.86-0.96 PF on FC2;
// stream-K over 256 x 256 tiles lost the operand sharing between neighbouring workgroups and was slower
// (profiles/gemm_p256_r03.txt, section 9).
//   * 8 waves as 2 (m) x 4 (n): a wave owns 80 rows x 64 columns = 5 x 4 accumulator blocks of 16 x 16.
//   * TWO phases per K-tile, one per K STEP of 32 columns — [5 A + 4 W fragment reads, staging] B [20 MFMAs] B, twice — with the
//     two wave halves one barrier apart (one half's read part beside the other's MFMAs). Measured on FC2, us per K-tile net of
//     the launch's ~12 us of fill, prefetch and epilogue (tools/gemm_p160_ablate.py; the 80 MFMAs alone take 0.58, the 52 KB
//     of LDS-DMA alone 0.62-0.67, the 36 fragment reads alone 0.29-0.35): the four quadrant phases of the 256 x 256 tile
//     (12 / 12 / 8 / 8 MFMAs; 80 rows do not halve evenly) 1.3; two staggered phases split by n half (14 + 4 reads) 1.06; these
//     two (9 + 9 reads) 0.96; an unstaggered form with two fragment sets in registers (a wave reads the next phase's fragments,
//     then issues this phase's MFMAs) 1.03. Whatever the order, MFMAs + fragment reads take their SUM (0.88 without staging):
//     on this machine a 1 KB fragment landing in the register file costs the matrix pipe what an MFMA's 1 KB of results
//     costs it, and an 80 x 64 wave tile needs 9 fragments per 20 MFMAs. FC2 ends at 0.97-0.99 PF instead of 0.90.
//   * THREE staging buffers of 52 KB (A 160 rows, W n0 / n1 128 rows each), K-tile t+2 staged while t is computed: a K-tile is
//     ~0.9 us here, two of them in flight (104 KB per CU) keep the prefetch distance of the 256 x 256 loop in time.
//   * 20 A pieces of 8 rows for 8 waves: every wave issues THREE LDS-DMA operations for the A slot (waves 4-7: two pieces and a
//     4-byte-per-lane filler into a dump area) and two per W slot — 7 per K-tile on every wave, so one counted wait serves all.
//   * one tile per workgroup: the bf16 rows it adds to (its own output rows) and its bias are fetched BEFORE the K loop; the
//     epilogue's transpose patches lie over staging buffer 0.
// Fragment layout and swizzle: gemm_bf16_p256.h.
#pragma once
#include "gemm_bf16_256.h"

#define G160_A_BYTES 20480                  // 160 rows x 128 B
#define G160_W_BYTES 16384                  // 128 rows x 128 B
#define G160_BUF (G160_A_BYTES + 2 * G160_W_BYTES)   // one K-tile: A | W n0 | W n1
#define G160_DUMP (3 * G160_BUF)            // 8 waves x 256 B: where the 4-byte filler pieces land
#define G160_LDS (G160_DUMP + 2048)         // 161 792 B of the CU's 163 840

// VARIANT (timing experiments, EXPERIMENTS builds, tools/gemm_p160_ablate.py; results are wrong by construction): a mask —
// 1 = no MFMAs, 2 = no staging, 4 = no fragment reads. 0 = the kernel.
// EPI: MMISS_EPI_BIAS_RESID_BF16 (the residual GEMMs) or MMISS_EPI_PATCH_F32 (the patch-embedding GEMM: f32 rows scattered to
// item * tokens + 1 + patch with the position row added, gemm_bf16.h gemm_epilogue's contract; no bias, no residual).
// KT: the K-tile count as a compile-time tag (0 = read K at run time). It exists so that the shapes of one encode are
// DISTINCT SYMBOLS in rocprofv3's kernel trace and PMC passes — out-projection (K = 768: <9,0,12>) and FC2 (K = 3072: <9,0,48>)
// were one symbol with a 24-75 us "class average" in round 3 — and gives the K loop a constant trip count.
template <int EPI, int VARIANT, int KT = 0>
__global__ __launch_bounds__(512, 2) void gemm160p_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W, int M,
                                                          int N, int K, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef bf16x8 frag;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int nbm = M / 160, nbn = N >> 8;
    const int nt = KT > 0 ? KT : K / GEMM_BK;  // even (K % 128 == 0)
    const int wg = xcd_remap(blockIdx.x, nbm * nbn);
    int bm, bn;
    tile_order(wg, nbm, nbn, 0, bm, bn);   // n fastest: the column tiles of a row block run side by side on one XCD
    const __bf16* Ab = A + (size_t)bm * 160 * K;
    const __bf16* Wb = W + (size_t)bn * 256 * K;

    // ---- LDS-DMA sources, buffer form: one per-lane offset (row r_in of an 8-row piece, swizzled 16-byte chunk), everything
    // else scalar. A slot row r = tile row r; piece q = rows 8q .. 8q+7 at q * 1024. W slot row r = weight row
    // (r >> 5) * 64 + nq * 32 + (r & 31) (the 32 rows of n half nq of each of the four wave columns).
    const int r_in = lane >> 3, p = lane & 7;
    const int lane_vo = (r_in * K + ((p ^ r_in) * 8)) * 2;
    const __amdgpu_buffer_rsrc_t srdA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Ab), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t srdW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Wb), 0, 0x7fffffff, 0x00020000);
    const int row8 = 8 * K * 2;                 // bytes between two consecutive 8-row pieces
    const int a_so = wave * row8;               // A pieces w, w + 8 and (waves 0-3) w + 16
    const int a_dst = wave * 1024;
    const int w_so = ((wave >> 1) * 64 + (wave & 1) * 16) * K * 2;   // W slot rows 16w, 16w + 8; the n1 slot is 32 weight rows on
    const int w_dst = wave * 2048;
    char* const dump = smem + G160_DUMP + wave * 256;
#define G160_BLDS(srd, so, dst) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(dst), 16, lane_vo, so, 0, 0)
#define G160_FILL(srd, so) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(dump), 4, lane_vo, so, 0, 0)
// the A slot of buffer b2 <- K-tile at byte offset ko: THREE operations per wave
#define G160_STAGE_A(b2, ko)                                                                                 \
    if constexpr (!((VARIANT & 2) != 0)) {                                                                  \
        char* sl_ = smem + (b2) * G160_BUF + a_dst;                                                         \
        G160_BLDS(srdA, a_so + (ko), sl_);                                                                  \
        G160_BLDS(srdA, a_so + 8 * row8 + (ko), sl_ + 8 * 1024);                                            \
        if (wave < 4) { G160_BLDS(srdA, a_so + 16 * row8 + (ko), sl_ + 16 * 1024); }                        \
        else { G160_FILL(srdA, a_so + (ko)); }                                                              \
    }
// the W slot of n half nq: TWO operations per wave
#define G160_STAGE_W(b2, nq, ko)                                                                             \
    if constexpr (!((VARIANT & 2) != 0)) {                                                                  \
        char* sl_ = smem + (b2) * G160_BUF + G160_A_BYTES + (nq) * G160_W_BYTES + w_dst;                    \
        const int so_ = w_so + (nq) * 4 * row8 + (ko);                                                      \
        G160_BLDS(srdW, so_, sl_);                                                                          \
        G160_BLDS(srdW, so_ + row8, sl_ + 1024);                                                            \
    }

    // ---- fragment reads: one base per operand and k step + immediate (buffer, 16-row block)
    uint32_t ab[2], wb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t sw = (((4 * s + fg) ^ (fr & 7)) << 4);
        ab[s] = (wm * 80 + fr) * 128 + sw;
        wb[s] = G160_A_BYTES + (wn * 32 + fr) * 128 + sw;
    }
    frag am[5];
    frag wq[4];
    f32x4 acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

// the fragments of k step s (32 of the K-tile's 64 columns): five A row blocks, four W column blocks (n0: 0, 1; n1: 2, 3)
#define G160_READ(b, s)                                                                                      \
    if constexpr (!((VARIANT & 4) != 0)) {                                                                  \
        _Pragma("unroll") for (int mf = 0; mf < 5; ++mf)                                                    \
            am[mf] = *reinterpret_cast<const frag*>(smem + ab[s] + (b) * G160_BUF + mf * 2048);             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
            wq[i] = *reinterpret_cast<const frag*>(smem + wb[s] + (b) * G160_BUF + (i >> 1) * G160_W_BYTES + (i & 1) * 2048); \
    }
// the 20 MFMAs of one phase: all five row blocks against all four column blocks, one k step
#define G160_MMA()                                                                                           \
    if constexpr ((VARIANT & 1) != 0) {                                                                     \
        _Pragma("unroll") for (int mf = 0; mf < 5; ++mf) asm volatile("" ::"v"(am[mf]));                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(wq[i]));                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
            _Pragma("unroll") for (int mf = 0; mf < 5; ++mf) asm volatile("" : "+v"(acc[i][mf]));           \
    } else {                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
            _Pragma("unroll") for (int mf = 0; mf < 5; ++mf)                                                \
                acc[i][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[i], am[mf], acc[i][mf], 0, 0, 0);   \
        __builtin_amdgcn_s_setprio(0);                                                                      \
    }
#define G160_BARRIER()                            \
    {                                             \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    }
// The half that runs one barrier behind reads a buffer for the last time in the interval right before the other half restages
// it: its reads must have completed before the barrier between the two.
#define G160_LATE_READS_DONE() \
    if (wm == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// K-tile t out of buffer B while K-tile t+2 (byte offset ko2) is staged into buffer B2 = (B + 2) % 3, which held K-tile t-1.
// The two phases are the two K STEPS of the tile (32 columns each): every read part pulls 9 fragments (5 A + 4 W) and every
// MFMA part is the full 5 x 4 block. A wave issues 7 LDS-DMA operations per K-tile (A A A in phase 0, W0 W0 W1 W1 in phase
// 1). All three slots of a K-tile are read from its phase 0 on: the one wait, in phase 1, retires K-tile t+1 completely —
// younger than it are exactly K-tile t+2's seven: vmcnt(7). Both halves' waits and a barrier precede both halves' reads of
// what they retire.
#define G160_KTILE(B, B2)                                                                                    \
    {                                                                                                       \
        G160_READ(B, 0);                                                                                    \
        G160_STAGE_A(B2, ko2);                                                                              \
        G160_LATE_READS_DONE();                                                                             \
        G160_BARRIER();                                                                                     \
        G160_MMA();                                                                                         \
        G160_BARRIER();                                                                                     \
        G160_READ(B, 1);                                                                                    \
        G160_STAGE_W(B2, 0, ko2);                                                                           \
        G160_STAGE_W(B2, 1, ko2);                                                                           \
        G160_LATE_READS_DONE();                                                                             \
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                                                    \
        G160_BARRIER();                                                                                     \
        G160_MMA();                                                                                         \
        G160_BARRIER();                                                                                     \
        if (++k2 == nt) { k2 = 0; ko2 = 0; } else { ko2 += GEMM_BK * 2; }                                   \
    }

    // ---- the bf16 rows this wave will add to (its own output rows: nobody else touches them), in the accumulator layout —
    // 8 bytes per lane and 16 x 16 block, 40 registers — and its bias values: requested NOW, in front of the prologue's LDS-DMA,
    // so that the epilogue of this one-tile workgroup starts without a memory round trip
    constexpr bool RESID = (EPI == MMISS_EPI_BIAS_RESID_BF16);
    static_assert(RESID || EPI == MMISS_EPI_PATCH_F32, "gemm160p_kernel: epilogue");
    u32x2 resid[5][4];
    f32x4 bias[4];
    if constexpr (RESID) {
        const uint16_t* ob = reinterpret_cast<const uint16_t*>(ep.out) + (size_t)(bm * 160 + wm * 80 + fr) * ep.ldo + bn * 256 + wn * 64 + 4 * fg;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) resid[j][i] = *reinterpret_cast<const u32x2*>(ob + (size_t)j * 16 * ep.ldo + i * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) bias[i] = *reinterpret_cast<const f32x4*>(ep.bias + bn * 256 + wn * 64 + i * 16 + 4 * fg);
    }

    // prologue: K-tiles 0 and 1 into buffers 0 and 1 (14 operations per wave)
    int ko2 = 0, k2 = 2;
    G160_STAGE_A(0, 0);
    G160_STAGE_W(0, 0, 0);
    G160_STAGE_W(0, 1, 0);
    G160_STAGE_A(1, GEMM_BK * 2);
    G160_STAGE_W(1, 0, GEMM_BK * 2);
    G160_STAGE_W(1, 1, GEMM_BK * 2);
    ko2 = (nt > 2) ? 2 * GEMM_BK * 2 : 0;   // (K >= 256: nt >= 4)
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");  // K-tile 0 has landed (younger: K-tile 1's seven)
    // ... and with it the 24 ordinary loads above, which are older: their round trip ran beside the prologue's. Pinned here
    // so that the compiler's own wait for them sits in front of the K loop, not inside it.
    if constexpr (RESID) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(resid[j][i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(bias[i]));
    }
    G160_BARRIER();
    if (wm == 1) G160_BARRIER();  // the lower half runs one barrier behind from here on

#pragma unroll 1   // (nt may be a compile-time constant, KT: the loop stays a loop — one copy of the K-tile triple)
    for (int t = 0; t < nt; t += 3) {
        G160_KTILE(0, 2);
        if (t + 1 < nt) G160_KTILE(1, 0);
        if (t + 2 < nt) G160_KTILE(2, 1);
    }
    if (wm == 0) G160_BARRIER();   // (the upper half's last barrier: the lower half is still one behind)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wrap-around staging of the last K-tiles has landed ...
    G160_BARRIER();                // ... on every wave, and every wave has read its last fragments: the buffers are free

    if constexpr (EPI == MMISS_EPI_PATCH_F32) {
        // f32 rows: 16 rows x 32 columns at a time through the wave's 2 KB patch (16-byte chunks XOR-swizzled by the row), read back
        // as whole 128-byte row segments; patch row m of the GEMM is token 1 + m % p0 of item m / p0, the position row is added
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int efr = lane_e & 15, efg = lane_e >> 4;
        const int rrow = lane_e >> 3, rchunk = lane_e & 7;
        char* patch = smem + wave * 2048;   // (over staging buffer 0)
        float* outp = reinterpret_cast<float*>(ep.out);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
                    *reinterpret_cast<f32x4*>(patch + efr * 128 + (((ii * 4 + efg) ^ (efr & 7)) << 4)) = acc[2 * h + ii][j];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int rh = 0; rh < 2; ++rh) {
                    const int row = rh * 8 + rrow;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * 128 + ((rchunk ^ (row & 7)) << 4));
                    const int m = bm * 160 + wm * 80 + j * 16 + row;
                    const int n = bn * 256 + wn * 64 + h * 32 + rchunk * 4;
                    if (m < ep.m_valid) {
                        const int img = m / ep.p0, pt = m - img * ep.p0;
                        const f32x4 pos = *reinterpret_cast<const f32x4*>(ep.aux + (size_t)(1 + pt) * ep.ldo + n);
                        *reinterpret_cast<f32x4*>(outp + ((size_t)img * ep.p1 + 1 + pt) * ep.ldo + n) = v + pos;
                    }
                }
                __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next half
            }
        }
    }
    // ---- epilogue of MMISS_EPI_BIAS_RESID_BF16 (gemm_bf16.h gemm_epilogue: same arithmetic, same statistics, bit for bit),
    // per wave: 5 row blocks x 64 columns through the wave's 2 KB transpose patch, whole 128-byte row segments per store
    if constexpr (RESID) {
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int efr = lane_e & 15, efg = lane_e >> 4;
        const int rrow = lane_e >> 3, rchunk = lane_e & 7;
        const __amdgpu_buffer_rsrc_t srdO = __builtin_amdgcn_make_buffer_rsrc(ep.out, 0, 0x7fffffff, 0x00020000);
        char* patch = smem + wave * 2048;   // (over staging buffer 0)
        const int wr_off = efr * 128, wr_sw = 2 * (efr & 7);
        const int col = bn * 256 + wn * 64 + rchunk * 8;
        const int dump_row = M - 1;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = acc[i][j] + bias[i];
                float y[4];
                y[0] = __uint_as_float(resid[j][i][0] << 16) + v[0];
                y[1] = __uint_as_float(resid[j][i][0] & 0xFFFF0000u) + v[1];
                y[2] = __uint_as_float(resid[j][i][1] << 16) + v[2];
                y[3] = __uint_as_float(resid[j][i][1] & 0xFFFF0000u) + v[3];
                u32x2 pk;
                pk[0] = pack_bf16x2(y[0], y[1]);
                pk[1] = pack_bf16x2(y[2], y[3]);
                *reinterpret_cast<u32x2*>(patch + wr_off + (((i * 4 + efg) ^ wr_sw) << 3)) = pk;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int row = rh * 8 + rrow;
                const u32x4 pk = *reinterpret_cast<const u32x4*>(patch + row * 128 + (((2 * rchunk) ^ (2 * (row & 7))) << 3));
                const int m = bm * 160 + wm * 80 + j * 16 + row;
                const int vo = ((m < ep.m_valid ? m : dump_row) * ep.ldo + col) * 2;
                __builtin_amdgcn_raw_buffer_store_b128(pk, srdO, vo, 0, 0);
                if (ep.stats_out) {  // of what was STORED (the rounded rows are the residual stream from here on)
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
                        float* so = ep.stats_out + ((size_t)m * (ep.ldo >> 6) + ((bn * 256 + wn * 64) >> 6)) * 2;
                        so[0] = rs;
                        so[1] = rq;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next j
        }
    }
}
#undef G160_BLDS
#undef G160_FILL
#undef G160_STAGE_A
#undef G160_STAGE_W
#undef G160_READ
#undef G160_MMA
#undef G160_BARRIER
#undef G160_LATE_READS_DONE
#undef G160_KTILE

// can the residual GEMM on the bf16 stream run on this tile? (the tile must divide the padded rows / the columns; 32-bit offsets)
static inline bool gemm160p_ok(int M, int N, int K) {
    if (M <= 0 || (M % 160) || N <= 0 || (N % 256) || K < 256 || (K % 128)) return false;
    if ((int64_t)M * N * 2 >= (1LL << 31) || (int64_t)160 * K * 2 >= (1LL << 31) || (int64_t)256 * K * 2 >= (1LL << 31)) return false;
    return true;
}

// out (bf16 [M, ldo], in place) = bf16(f32(out) + A W^T + bias), ep.stats_out optional; M padded to 160 (rows >= ep.m_valid land
// in row M - 1)
static int launch_gemm160p(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K) {
    if (!gemm160p_ok(M, N, K)) MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm160p: M=%d N=%d K=%d", M, N, K);
    if (!ep.out || !ep.bias || ep.ldo < N || (ep.ldo % 64)) MM_FAIL(MMISS_ERR_ARG, "gemm160p: missing operand");
    const int mv = ep.m_valid < M ? ep.m_valid : M;
    const double bytes = 2.0 * ((double)mv * K + (double)N * K) + 2.0 * 2.0 * (double)mv * N;
    char pname[48];
    snprintf(pname, sizeof(pname), "gemm_bf16_bias_resid16_p160_k%d", K);
    MM_PROF(pname, st, 2.0 * mv * N * K, bytes);
#ifdef MMISS_EXPERIMENTS
    const int dbg = mmiss_option("gemm_p160_dbg", 0);
#define G160_DBG_CASE(D)                                                                                                        \
    if (dbg == D) {                                                                                                             \
        MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm160p_kernel<MMISS_EPI_BIAS_RESID_BF16, D>), G160_LDS));                             \
        hipLaunchKernelGGL((gemm160p_kernel<MMISS_EPI_BIAS_RESID_BF16, D>), dim3((M / 160) * (N / 256)), dim3(512), G160_LDS, st,                            \
                           reinterpret_cast<const __bf16*>(A), reinterpret_cast<const __bf16*>(W), M, N, K, ep);                \
        MM_HIP(hipGetLastError());                                                                                              \
        return MMISS_OK;                                                                                                        \
    }
    G160_DBG_CASE(1) G160_DBG_CASE(2) G160_DBG_CASE(3) G160_DBG_CASE(4) G160_DBG_CASE(5) G160_DBG_CASE(6) G160_DBG_CASE(7)
#undef G160_DBG_CASE
#endif
#define G160_KT_CASE(KT_)                                                                                                        \
    if (K == (KT_) * GEMM_BK) {                                                                                                  \
        MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm160p_kernel<MMISS_EPI_BIAS_RESID_BF16, 0, KT_>), G160_LDS)); \
        hipLaunchKernelGGL((gemm160p_kernel<MMISS_EPI_BIAS_RESID_BF16, 0, KT_>), dim3((M / 160) * (N / 256)), dim3(512), G160_LDS, \
                           st, reinterpret_cast<const __bf16*>(A), reinterpret_cast<const __bf16*>(W), M, N, K, ep);            \
        MM_HIP(hipGetLastError());                                                                                              \
        return MMISS_OK;                                                                                                        \
    }
    // the towers' shapes: ViT-B/32 out-projection / FC2 (768, 3072), its text tower (512, 2048), ViT-L/14 (1024, 4096)
    G160_KT_CASE(12) G160_KT_CASE(48) G160_KT_CASE(8) G160_KT_CASE(32) G160_KT_CASE(16) G160_KT_CASE(64)
#undef G160_KT_CASE
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm160p_kernel<MMISS_EPI_BIAS_RESID_BF16, 0>), G160_LDS));
    hipLaunchKernelGGL((gemm160p_kernel<MMISS_EPI_BIAS_RESID_BF16, 0>), dim3((M / 160) * (N / 256)), dim3(512), G160_LDS, st, reinterpret_cast<const __bf16*>(A),
                       reinterpret_cast<const __bf16*>(W), M, N, K, ep);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// The patch-embedding GEMM (MMISS_EPI_PATCH_F32 of gemm_bf16.h: ep.out f32 [items * p1, ldo], ep.aux = position table, ep.p0 = patches
// per item, ep.p1 = tokens per item) on the same tile: 12 544 patch rows x 768 x 3072 at bs 256 = 79 x 3 tiles.
static int launch_gemm160p_patch(hipStream_t st, const void* A, const void* W, const GemmEpi& ep, int M, int N, int K) {
    if (!gemm160p_ok(M, N, K)) MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm160p_patch: M=%d N=%d K=%d", M, N, K);
    if (!ep.out || !ep.aux || ep.p0 <= 0 || ep.p1 <= 0 || ep.ldo < N) MM_FAIL(MMISS_ERR_ARG, "gemm160p_patch: missing operand");
    const int mv = ep.m_valid < M ? ep.m_valid : M;
    MM_PROF("gemm_bf16_patch_p160", st, 2.0 * mv * N * K, 2.0 * ((double)mv * K + (double)N * K) + 4.0 * (double)mv * N);
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&gemm160p_kernel<MMISS_EPI_PATCH_F32, 0>), G160_LDS));
    hipLaunchKernelGGL((gemm160p_kernel<MMISS_EPI_PATCH_F32, 0>), dim3((M / 160) * (N / 256)), dim3(512), G160_LDS, st,
                       reinterpret_cast<const __bf16*>(A), reinterpret_cast<const __bf16*>(W), M, N, K, ep);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

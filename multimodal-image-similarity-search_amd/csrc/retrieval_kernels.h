// retrieval_kernels.h — flat cosine index kernels (K10, K11, K12, X1 of SURVEY.md §2.2).
//
// What they replace: chromadb's cosine-space collection as the reference uses it
// (backend/app/utils.py:127-130 create, backend/app/main.py:735-740 add, :761-765 query) — rows are
// L2-normalised when added, a query returns the k rows of smallest cosine distance 1 - <q^, c^>.
//
// Exactness contract (DESIGN.md "retrieval parity"): selection runs in two stages.
//   stage 1 (scan, HBM- or MFMA-bound): approximate scores on the matrix cores (f16 x f16 products are
//     exact, fp32 accumulation in hardware order) with a fused running top-k' per wave, k' > k;
//   stage 2 (rerank, negligible): the k' survivors are re-scored in a CANONICAL order in fp64
//     (lane l sums d = l, l+64, ... sequentially; then a fixed xor-butterfly 32,16,..,1) and sorted by
//     (distance asc, label asc). oracle/retrieval_oracle.c restates exactly that arithmetic, so ids AND
//     distances are bit-identical between the two, on any shard layout.
#pragma once
#include "common.h"
#include "gemm_bf16.h"
#include <math.h>
#include <type_traits>

#define SCAN_NEG_INF (-INFINITY)
#define SCAN_ROW_NONE 0x7fffffff

// ------------------------------------------------------------------------------------------------
// canonical fp64 reductions (mirrored by oracle/retrieval_oracle.c: canon_dot / canon_sumsq)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_butterfly_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o);
    return v;
}

// fp8 storage (OCP e4m3) with the FIXED scale 2^7: a unit-norm row has |x_i| <= 1, so 128 x_i fits e4m3 (largest value 448) and
// a component of typical size 1/sqrt(D) keeps its three mantissa bits; the stored VALUE is decode(byte) / 128, exactly
// representable in f16 and f32 (oracle/retrieval_oracle.py normalize_rows(.., "f8") restates the same rounding).
#define MMISS_F8_SCALE 128.0f
struct F8 {
    uint8_t b;
    F8() = default;
    __device__ explicit F8(float y) { b = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(y * MMISS_F8_SCALE, 0.0f, 0, false) & 0xff); }
    __device__ explicit operator float() const { return __builtin_amdgcn_cvt_f32_fp8((int)b, 0) * (1.0f / MMISS_F8_SCALE); }
};
// four consecutive codes of a row -> their values
__device__ __forceinline__ void f8x4_values(uint32_t w, float (&v)[4]) {
    v[0] = __builtin_amdgcn_cvt_f32_fp8((int)w, 0) * (1.0f / MMISS_F8_SCALE);
    v[1] = __builtin_amdgcn_cvt_f32_fp8((int)w, 1) * (1.0f / MMISS_F8_SCALE);
    v[2] = __builtin_amdgcn_cvt_f32_fp8((int)w, 2) * (1.0f / MMISS_F8_SCALE);
    v[3] = __builtin_amdgcn_cvt_f32_fp8((int)w, 3) * (1.0f / MMISS_F8_SCALE);
}

template <typename T> __device__ __forceinline__ double widen(T v) { return (double)(float)v; }

// ------------------------------------------------------------------------------------------------
// K10: row normalisation at add time.  y = (float)((double)x / sqrt(canon_sumsq(x))), then the
// storage rounding (f16: RNE from that float). One wave per row. D % 64 == 0.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                             int64_t n, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float* x = src + r * D;
    double acc = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double v = (double)x[d];
        acc = acc + v * v;
    }
    const double nrm = sqrt(wave_butterfly_sum(acc));
    for (int d = lane; d < D; d += 64) {
        float y = (float)((double)x[d] / nrm);
        asm volatile("" : "+v"(y));  // keep the float rounding: storage = RNE_f16(RNE_f32(x / nrm)), two roundings
        dst[r * D + d] = (T)y;
    }
}

// MMISS_F8 rows: the e4m3 rounding moves a row's norm by up to ~3 %, so the codes alone would give 1 - |stored| cos instead of
// a cosine distance (round 4). Every fp8 row therefore carries ONE float: inv[r] = float(1 / canonical norm of the values its
// codes stand for). The row the index represents is values * inv (unit norm up to that one rounding); every score — the
// scan's, the score GEMM's, the canonical re-rank's — is the dot product with the codes' values TIMES inv[r]. One wave per
// row; recomputed from the codes wherever rows are (re)written, so the save file does not hold it.
__global__ __launch_bounds__(256) void f8_row_inv_kernel(const uint8_t* __restrict__ codes, int64_t n, int D, float* __restrict__ inv) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const uint8_t* c = codes + r * D;
    double acc = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double v = (double)(__builtin_amdgcn_cvt_f32_fp8((int)c[d], 0) * (1.0f / MMISS_F8_SCALE));
        acc = acc + v * v;
    }
    const double nrm = sqrt(wave_butterfly_sum(acc));
    if (lane == 0) inv[r] = (float)(1.0 / nrm);
}

// Query preparation: qn = canonical normalisation (f32, used by the rerank), qs = qn in the storage
// dtype for the scan's MFMA operand, rows [Q, Qpad) zero-filled.
// eps_q (optional): the exactness guard's per-query bound on |approximate - canonical| score (api_index.hip, "exactness
// contract"): eps_fixed (accumulation + slack terms, the same for every query) + cnorm * |qs - qn|_2 — what rounding the
// scan operand to T moves a score by is <(qs - qn), c> <= |qs - qn|_2 |c|_2 (Cauchy-Schwarz; cnorm >= |c|_2 of any stored row),
// with the ACTUAL rounding error of this query instead of the worst case 2^-11 |q|.
template <typename T>
__global__ __launch_bounds__(256) void prep_queries_kernel(const float* __restrict__ q, float* __restrict__ qn,
                                                           T* __restrict__ qs, int Q, int Qpad, int D,
                                                           float* __restrict__ eps_q, double eps_fixed, double cnorm) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= Qpad) return;
    if (r >= Q) {
        for (int d = lane; d < D; d += 64) qs[(size_t)r * D + d] = (T)0.0f;
        return;
    }
    const float* x = q + (size_t)r * D;
    double acc = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double v = (double)x[d];
        acc = acc + v * v;
    }
    const double nrm = sqrt(wave_butterfly_sum(acc));
    double dq = 0.0;
    for (int d = lane; d < D; d += 64) {
        float y = (float)((double)x[d] / nrm);
        asm volatile("" : "+v"(y));
        const T ys = (T)y;
        qn[(size_t)r * D + d] = y;
        qs[(size_t)r * D + d] = ys;
        const double e = (double)(float)ys - (double)y;
        dq = dq + e * e;
    }
    if (eps_q) {
        dq = wave_butterfly_sum(dq);
        // rounded UP: a bound. A NaN / inf query (norm 0 or inf) gets an infinite bound: never "proven", harmless.
        const double e = (eps_fixed + cnorm * sqrt(dq)) * 1.0001;
        float ef = (float)e;
        if ((double)ef < e) ef = nextafterf(ef, INFINITY);
        if (lane == 0) eps_q[r] = (e == e) ? ef : INFINITY;
    }
}

// ------------------------------------------------------------------------------------------------
// running top-k' list of one wave for one query, kept in LDS: append when a score beats the list's
// threshold, compact (64-lane bitonic sort by (score desc, row asc)) when it fills up.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cand_better(float sa, int ra, float sb, int rb) {
    return sa > sb || (sa == sb && ra < rb);
}

// lane ^ J exchange of one dword. J = 1, 2, 4, 8 stay inside a row of 16 lanes: DPP moves in the VALU (quad_perm for 1 and
// 2; lane ^ 4 = half-mirror then quad reverse, lane ^ 8 = row mirror then half-mirror) instead of ds_bpermute round trips
// through the LDS crossbar — 18 of the 21 stages of the 64-lane sort; 16 and 32 keep the bpermute.
template <int J>
__device__ __forceinline__ int lane_xor(int v) {
    if constexpr (J == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (J == 4)
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, false), 0x1B, 0xF, 0xF, false);
    else if constexpr (J == 8)
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, false), 0x141, 0xF, 0xF, false);
    else return __shfl_xor(v, J);
}

// sorts the 64 (s, r) pairs held one per lane: best first (bitonic network, 21 compare-exchange stages)
template <int K, int J>
__device__ __forceinline__ void wave_sort64_stage(float& s, int& r, int lane) {
    const float os = __int_as_float(lane_xor<J>(__float_as_int(s)));
    const int orr = lane_xor<J>(r);
    const bool lower = (lane & J) == 0;
    const bool up = (lane & K) == 0;  // at K == 64 every lane is "up": final order best-first
    const bool keep_best = (lower == up);
    const bool take = keep_best ? cand_better(os, orr, s, r) : cand_better(s, r, os, orr);
    if (take) { s = os; r = orr; }
}
__device__ __forceinline__ void wave_sort64(float& s, int& r, int lane) {
    wave_sort64_stage<2, 1>(s, r, lane);
    wave_sort64_stage<4, 2>(s, r, lane);  wave_sort64_stage<4, 1>(s, r, lane);
    wave_sort64_stage<8, 4>(s, r, lane);  wave_sort64_stage<8, 2>(s, r, lane);  wave_sort64_stage<8, 1>(s, r, lane);
    wave_sort64_stage<16, 8>(s, r, lane); wave_sort64_stage<16, 4>(s, r, lane); wave_sort64_stage<16, 2>(s, r, lane);
    wave_sort64_stage<16, 1>(s, r, lane);
    wave_sort64_stage<32, 16>(s, r, lane); wave_sort64_stage<32, 8>(s, r, lane); wave_sort64_stage<32, 4>(s, r, lane);
    wave_sort64_stage<32, 2>(s, r, lane);  wave_sort64_stage<32, 1>(s, r, lane);
    wave_sort64_stage<64, 32>(s, r, lane); wave_sort64_stage<64, 16>(s, r, lane); wave_sort64_stage<64, 8>(s, r, lane);
    wave_sort64_stage<64, 4>(s, r, lane);  wave_sort64_stage<64, 2>(s, r, lane);  wave_sort64_stage<64, 1>(s, r, lane);
}

// ------------------------------------------------------------------------------------------------
// K11 stage 1: scan + fused top-k'.
//
// grid = (slabs, query tiles); block = 4 waves. A block owns NQ = 16*NQT queries (their MFMA B
// fragments staged once into LDS, rows padded by 16 B so the 16 query rows of a fragment read hit 16
// different bank groups) and a contiguous slab of 16-row tiles; wave w takes tiles w, w+4, ...
// A fragments (corpus rows) go global -> VGPR directly: each byte of the corpus is used by exactly one
// wave once (guide "GEMV / M <= 16" row: no LDS round trip for a streamed, unshared operand), 16-byte
// loads, the K loop unrolled so 4-8 loads per lane are in flight.
//   f16 rows: v_mfma_f32_16x16x32_f16 — lane (r = lane&15, g = lane>>4) feeds row r, k = 32s + 8g ..+7
//   f32 rows: v_mfma_f32_16x16x4_f32 x4 per 16-byte load — lane feeds k = 16s + 4g + t to MFMA t
//             (both operands use the same k permutation, so the dot product is unchanged)
// Accumulator: D[row = 4g + reg][col = lane&15] -> a lane always owns the same query, so its threshold
// lives in a register.
// ------------------------------------------------------------------------------------------------
template <typename T> struct ScanTraits;
// ELT = bytes per stored element, QELT = bytes per element of the query operand (f16 for f16 AND fp8 rows: the codes widen
// exactly to f16 in registers), SCORE_SCALE = what the accumulated dot product of the raw operands is multiplied by
template <> struct ScanTraits<_Float16> { static constexpr int ELT = 2, QELT = 2; static constexpr float SCORE_SCALE = 1.0f; };
template <> struct ScanTraits<float> { static constexpr int ELT = 4, QELT = 4; static constexpr float SCORE_SCALE = 1.0f; };
template <> struct ScanTraits<F8> { static constexpr int ELT = 1, QELT = 2; static constexpr float SCORE_SCALE = 1.0f / MMISS_F8_SCALE; };

struct ScanArgs {
    const void* rows;     // [N, D] storage dtype
    int64_t N;
    int D;
    const void* qs;       // [Qpad, D] storage dtype, Qpad % 16 == 0
    int Q;
    const float* cur_s;   // [Q] paging cursor (score, row) or null: only entries strictly after it pass
    const int32_t* cur_r;
    int kp;               // list length k'
    int tiles_per_block;  // 16-row tiles per slab
    float* out_s;         // [gridDim.x][Q][kp]
    int32_t* out_r;
    // threshold mode (scan_topk_kernel<.., THR = true>, the widen pass): no lists — every row whose approximate score reaches
    // thr[q] is appended to glist[q][..] (row ids, any order; gcnt[q] counts them and may run past gcap: the excess is
    // dropped and the query goes to the exhaustive pass)
    const float* thr;     // [Q]
    int32_t* gcnt;        // [Q]
    int32_t* glist;       // [Q][gcap]
    int gcap;
    const float* inv;     // fp8 rows: [>= N rounded up to 16] inverse norm of every row's values (f8_row_inv_kernel); scores are x inv
};

template <int NQT, int CAP>
struct ScanLds {
    // byte offsets inside dynamic LDS
    int qstride;  // bytes per query row
    int off_cs, off_cr, off_cnt, off_tau, total;
    __host__ __device__ ScanLds(int D, int elt) {
        qstride = D * elt + 16;
        const int NQ = 16 * NQT;
        off_cs = (int)round_up16(NQ * qstride);
        off_cr = off_cs + 4 * NQ * CAP * 4;
        off_cnt = off_cr + 4 * NQ * CAP * 4;
        off_tau = off_cnt + 4 * NQ * 4;
        total = off_tau + 4 * NQ * 4;
    }
    static __host__ __device__ int round_up16(int x) { return (x + 15) & ~15; }
};

// A stored row is read ONCE per query block by one wave: a stream. MMISS_SCAN_NT marks the loads non-temporal so that a
// scan does not push the towers' weights out of the 256 MB Infinity Cache between two requests (tools/scan_nt_ab.sh).
__device__ __forceinline__ u32x4 scan_row_load(const u32x4* p) {
#ifdef MMISS_SCAN_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

template <typename T, int NQT, int CAP, int GS = 8, bool THR = false>
__global__ __launch_bounds__(256) void scan_topk_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ELT = ScanTraits<T>::ELT, QELT = ScanTraits<T>::QELT;
    constexpr int NQ = 16 * NQT;
    const ScanLds<NQT, CAP> L(a.D, QELT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int qbase = blockIdx.y * NQ;
    const int D = a.D;

    char* sQ = smem;
    float* cs = reinterpret_cast<float*>(smem + L.off_cs) + wave * NQ * CAP;
    int* cr = reinterpret_cast<int*>(smem + L.off_cr) + wave * NQ * CAP;
    int* cnt = reinterpret_cast<int*>(smem + L.off_cnt) + wave * NQ;
    float* tau = reinterpret_cast<float*>(smem + L.off_tau) + wave * NQ;

    // stage the query block (16-byte chunks), init list state
    {
        const int chunks_per_row = D * QELT / 16;
        const char* qsrc = reinterpret_cast<const char*>(a.qs) + (size_t)qbase * D * QELT;
        for (int i = tid; i < NQ * chunks_per_row; i += 256) {
            const int qr = i / chunks_per_row, c = i - qr * chunks_per_row;
            *reinterpret_cast<u32x4*>(sQ + qr * L.qstride + c * 16) =
                *reinterpret_cast<const u32x4*>(qsrc + ((size_t)qr * chunks_per_row + c) * 16);
        }
        if constexpr (!THR)
            for (int i = lane; i < NQ; i += 64) { cnt[i] = 0; tau[i] = SCAN_NEG_INF; }
    }
    __syncthreads();

    float tau_r[NQT], cur_s[NQT];
    int cur_r[NQT];
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
        tau_r[qt] = SCAN_NEG_INF;
        const int q = qbase + qt * 16 + fr;
        cur_s[qt] = (a.cur_s && q < a.Q) ? a.cur_s[q] : INFINITY;
        cur_r[qt] = (a.cur_r && q < a.Q) ? a.cur_r[q] : -1;
        if constexpr (THR) tau_r[qt] = q < a.Q ? a.thr[q] : INFINITY;
    }

    // compaction of every list of this wave that is longer than kp (wave-uniform control flow)
    auto compact_all = [&]() {
        for (int ql = 0; ql < NQ; ++ql) {
            const int c = cnt[ql];
            if (c > a.kp) {
                float s = SCAN_NEG_INF;
                int r = SCAN_ROW_NONE;
                if (lane < c && lane < CAP) { s = cs[ql * CAP + lane]; r = cr[ql * CAP + lane]; }
                wave_sort64(s, r, lane);
                if (lane < a.kp) { cs[ql * CAP + lane] = s; cr[ql * CAP + lane] = r; }
                const float t = __shfl(s, a.kp - 1);
                if (lane == 0) { cnt[ql] = a.kp; tau[ql] = t; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) tau_r[qt] = tau[qt * 16 + fr];
    };

    const int64_t ntiles_total = (a.N + 15) >> 4;
    const int64_t tile0 = (int64_t)blockIdx.x * a.tiles_per_block;
    int64_t tile_end = tile0 + a.tiles_per_block;
    if (tile_end > ntiles_total) tile_end = ntiles_total;

    for (int64_t tile = tile0 + wave; tile < tile_end; tile += 4) {
        const int64_t row0 = tile << 4;
        int64_t rload = row0 + fr;
        if (rload >= a.N) rload = a.N - 1;  // clamp: loads stay in bounds, the result is masked below
        const char* rp = reinterpret_cast<const char*>(a.rows) + (size_t)rload * D * ELT + fg * 16;
        const char* qp = sQ + fr * L.qstride + fg * 16;
        f32x4 acc[NQT];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) acc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // fp8 rows: what each of this lane's four rows (row0 + 4 fg + reg) is scaled by — issued with the tile's row loads
        f32x4 rinv = {ScanTraits<T>::SCORE_SCALE, ScanTraits<T>::SCORE_SCALE, ScanTraits<T>::SCORE_SCALE, ScanTraits<T>::SCORE_SCALE};
        if constexpr (ELT == 1) rinv = *reinterpret_cast<const f32x4*>(a.inv + row0 + 4 * fg) * ScanTraits<T>::SCORE_SCALE;
        // f16 / f32 rows: 64 bytes of every row per step (4 lanes x 16 B: 32 f16 or 16 f32 elements of k); steps are issued
        // in groups of 8 (then 4) with all the group's global loads ahead of its MFMAs, so 8 (4) 1-KiB wave-loads are in
        // flight per wave
        auto steps = [&](auto nsteps_tag, int byte0) {
            constexpr int NS = decltype(nsteps_tag)::value;
            u32x4 araw[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) araw[s] = scan_row_load(reinterpret_cast<const u32x4*>(rp + byte0 + s * 64));
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt) {
                    const u32x4 braw = *reinterpret_cast<const u32x4*>(qp + qt * 16 * L.qstride + byte0 + s * 64);
                    if constexpr (ELT == 2) {
                        acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, araw[s]),
                                                                         __builtin_bit_cast(f16x8, braw), acc[qt], 0, 0, 0);
                    } else {
                        const f32x4 af = __builtin_bit_cast(f32x4, araw[s]), bf = __builtin_bit_cast(f32x4, braw);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t], bf[t], acc[qt], 0, 0, 0);
                    }
                }
            }
        };
        // fp8 rows: a lane loads SIXTEEN codes (16 bytes: whole 64-byte row segments per four lanes, as the f16 scan — with
        // 8-byte loads the scan was bound by requests, 4.5 TB/s) and feeds two MFMAs from them. Lane group fg therefore
        // holds k = 16 fg .. 16 fg + 15 of a 64-wide span, not 8 fg .. + 7 of two 32-wide steps: a permutation of k inside
        // the span, applied to the query fragment as well (it only changes the order in which the APPROXIMATE score is
        // summed; the guard's bound does not depend on it and the returned distances are canonical).
        auto spans = [&](auto nsp_tag, int span0) {
            constexpr int NS = decltype(nsp_tag)::value;
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            typedef float f2 __attribute__((ext_vector_type(2)));
            const char* rp8 = reinterpret_cast<const char*>(a.rows) + (size_t)rload * D + fg * 16;
            const char* qp8 = sQ + fr * L.qstride + fg * 32;
            u32x4 araw[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) araw[s] = scan_row_load(reinterpret_cast<const u32x4*>(rp8 + (span0 + s) * 64));
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f16x8 a16;
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const f2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)araw[s][2 * h + w], false);
                        const f2 hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)araw[s][2 * h + w], true);
                        const h2 l2 = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(lo[0], lo[1]));   // (exact: e4m3 fits f16)
                        const h2 g2 = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(hi[0], hi[1]));
                        a16[4 * w + 0] = l2[0]; a16[4 * w + 1] = l2[1]; a16[4 * w + 2] = g2[0]; a16[4 * w + 3] = g2[1];
                    }
#pragma unroll
                    for (int qt = 0; qt < NQT; ++qt) {
                        const u32x4 braw = *reinterpret_cast<const u32x4*>(qp8 + qt * 16 * L.qstride + (span0 + s) * 128 + h * 16);
                        acc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, __builtin_bit_cast(f16x8, braw), acc[qt], 0, 0, 0);
                    }
                }
            }
        };
        if constexpr (ELT == 1) {
            const int nspans = D >> 6;   // D % 128 == 0: even
            int sp = 0;
            for (; sp + 8 <= nspans; sp += 8) spans(std::integral_constant<int, 8>{}, sp);
            if (sp + 4 <= nspans) { spans(std::integral_constant<int, 4>{}, sp); sp += 4; }
            if (sp < nspans) spans(std::integral_constant<int, 2>{}, sp);
        } else {
            const int row_bytes = D * QELT;  // of a QUERY row: a multiple of 256 (index dim % 128 == 0; f32: % 64)
            int byte0 = 0;
            if constexpr (GS == 16)
                for (; byte0 + 1024 <= row_bytes; byte0 += 1024) steps(std::integral_constant<int, 16>{}, byte0);
            for (; byte0 + 512 <= row_bytes; byte0 += 512) steps(std::integral_constant<int, 8>{}, byte0);
            if (byte0 < row_bytes) steps(std::integral_constant<int, 4>{}, byte0);
        }
        if constexpr (THR) {
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                const int q = qbase + qt * 16 + fr;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int64_t row = row0 + 4 * fg + reg;
                    const float s = acc[qt][reg] * rinv[reg];
                    if (q < a.Q && row < a.N && s >= tau_r[qt]) {
                        const int pos = atomicAdd(a.gcnt + q, 1);
                        if (pos < a.gcap) a.glist[(size_t)q * a.gcap + pos] = (int)row;
                    }
                }
            }
            continue;
        }
        // filter + append
        bool need = false;
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            const int ql = qt * 16 + fr;
            const bool qok = (qbase + ql) < a.Q;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t row = row0 + 4 * fg + reg;
                const float s = acc[qt][reg] * rinv[reg];   // (f16 / f32 rows: x 1; fp8 rows: x inv[row] / 128, one rounding)
                const int ri = (int)row;
                const bool after = (s < cur_s[qt]) || (s == cur_s[qt] && ri > cur_r[qt]);
                if (qok && row < a.N && s > tau_r[qt] && after) {
                    const int pos = atomicAdd(&cnt[ql], 1);
                    cs[ql * CAP + pos] = s;
                    cr[ql * CAP + pos] = ri;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) need |= cnt[qt * 16 + fr] > CAP - 16;
        if (__any(need)) compact_all();
    }

    if constexpr (THR) return;
    // block merge: the 4 waves' lists of one query -> one list of kp, written to out[blockIdx.x][q][:]
    compact_all();
    __syncthreads();
    float* cs_all = reinterpret_cast<float*>(smem + L.off_cs);
    int* cr_all = reinterpret_cast<int*>(smem + L.off_cr);
    int* cnt_all = reinterpret_cast<int*>(smem + L.off_cnt);
    for (int ql = wave; ql < NQ; ql += 4) {
        const int q = qbase + ql;
        if (q >= a.Q) continue;
        float s = SCAN_NEG_INF;
        int r = SCAN_ROW_NONE;
        const int c0 = cnt_all[0 * NQ + ql];
        if (lane < c0) { s = cs_all[(0 * NQ + ql) * CAP + lane]; r = cr_all[(0 * NQ + ql) * CAP + lane]; }
        for (int w = 1; w < 4; ++w) {
            const int cw = cnt_all[w * NQ + ql];
            // the running list sits in lanes [0, kp) (kp <= 32), the next wave's list goes to lanes [32, 32+cw)
            if (lane >= a.kp) { s = SCAN_NEG_INF; r = SCAN_ROW_NONE; }
            if (lane >= 32 && lane - 32 < cw) {
                s = cs_all[(w * NQ + ql) * CAP + lane - 32];
                r = cr_all[(w * NQ + ql) * CAP + lane - 32];
            }
            wave_sort64(s, r, lane);
        }
        if (lane < a.kp) {
            const size_t o = ((size_t)blockIdx.x * a.Q + q) * a.kp + lane;
            a.out_s[o] = s;
            a.out_r[o] = (r == SCAN_ROW_NONE) ? -1 : r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K11 stage 1 (batched path, Q > 16, f16 rows): the score pass is the f16 MFMA GEMM of gemm_bf16.h with the
// GROUPMAX epilogue — G[q][g] = max of the approximate scores of the 16 rows of group g — so only 1/16 of the
// Q x N score matrix ever leaves the registers. Every row of the true top-k' lies in one of the top-k' groups
// by group max (a group's max bounds its members from above), so selecting k' GROUPS and re-scoring their
// 16 x k' rows canonically in stage 2 loses nothing.
//
// select_topk_kernel: grid (Q, splits), 4 waves; each wave streams a contiguous run of G[q][:] and keeps its
// best 64 entries SORTED ACROSS ITS LANES (no LDS): a value beating the wave's k'-th best is inserted by
// ballot + popcount (its rank) + one shfl_up. After the first 64 values inserts are rare (~k' ln(n/k') per wave).
// ------------------------------------------------------------------------------------------------
struct SelectArgs {
    const float* G;       // [Qpad, ldg]
    int ldg;
    int ng;               // valid groups per query
    int kp;
    float* out_s;         // [gridDim.y][Q][kp]
    int32_t* out_r;       // group ids, -1 = none
    int Q;
};

__device__ __forceinline__ void lane_list_insert(float& ls, int& lr, float xs, int xr, int lane) {
    const unsigned long long better = __ballot(ls > xs || (ls == xs && lr < xr));
    const int p = __popcll(better);  // the list is sorted best-first, so the better entries are lanes [0, p)
    // lane i <- lane i-1 across the whole wave: DPP wave_shr:1 (gfx9 family), lane 0 keeps its own value (unused: p >= 0)
    const float us = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ls), __float_as_int(ls), 0x138, 0xF, 0xF, false));
    const int ur = __builtin_amdgcn_update_dpp(lr, lr, 0x138, 0xF, 0xF, false);
    if (lane > p) { ls = us; lr = ur; }
    else if (lane == p) { ls = xs; lr = xr; }
}

__global__ __launch_bounds__(256) void select_topk_kernel(SelectArgs a) {
    __shared__ float ms[4][32];
    __shared__ int mr[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const int nsplit = gridDim.y;
    // this block's column range, in units of 1024 columns
    const int units = (a.ng + 1023) / 1024;
    const int upb = (units + nsplit - 1) / nsplit;
    const int c_begin = blockIdx.y * upb * 1024;
    int c_end = c_begin + upb * 1024;
    if (c_end > a.ng) c_end = a.ng;
    const float* row = a.G + (size_t)q * a.ldg;

    float ls = SCAN_NEG_INF;
    int lr = SCAN_ROW_NONE;
    float tau = SCAN_NEG_INF;   // (tau, tau_r) = the list's kp-th entry: only values that sort before it are inserted
    int tau_r = SCAN_ROW_NONE;
    // wave w takes columns c_begin + 256*w + 4*lane .. +3 (one 16-byte load per lane, 1 KiB per wave), stepping 1024;
    // c_begin and ldg are multiples of 4, so the loads are aligned
    auto fetch = [&](int c0) {
        const int c = c0 + lane * 4;
        f32x4 v = {SCAN_NEG_INF, SCAN_NEG_INF, SCAN_NEG_INF, SCAN_NEG_INF};
        if (c0 >= c_end) return v;
        if (c + 3 < c_end) {
            v = *reinterpret_cast<const f32x4*>(row + c);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < c_end) v[e] = row[c + e];
        }
        return v;
    };
    f32x4 vnext = fetch(c_begin + wave * 256);
    for (int c0 = c_begin + wave * 256; c0 < c_end; c0 += 1024) {
        const int c = c0 + lane * 4;
        const f32x4 v = vnext;
        vnext = fetch(c0 + 1024);  // the next step's values fly while this step's are filed
        // The wave's first 64 values SEED the list through one bitonic sort (same list as 64 inserts one by one: the order
        // is total). With ~400 values per wave (6250 groups per query at 100k rows, 16 waves per query) the one-by-one
        // start-up was a quarter of the kernel: 39.9 -> 31.3 us at Q = 256. (Seeding with all 256 first values — four sorts
        // and a merge — costs what it saves: a 64-lane sort is 42 dependent ds_bpermutes, ~2 us.)
        const bool seed = (c0 == c_begin + wave * 256);
        if (seed) {
            ls = v[0];
            lr = (c < c_end) ? c : SCAN_ROW_NONE;
            wave_sort64(ls, lr, lane);
            tau = __shfl(ls, a.kp - 1);
            tau_r = __shfl(lr, a.kp - 1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (seed && e == 0) continue;
            const float s = v[e];
            unsigned long long m = __ballot(s > tau || (s == tau && c + e < tau_r));
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float xs = __shfl(s, src);
                const int xr = c0 + src * 4 + e;
                if (!(xs > tau || (xs == tau && xr < tau_r))) continue;
                lane_list_insert(ls, lr, xs, xr, lane);
                tau = __shfl(ls, a.kp - 1);
                tau_r = __shfl(lr, a.kp - 1);
            }
        }
    }
    if (lane < 32) { ms[wave][lane] = ls; mr[wave][lane] = lr; }
    __syncthreads();
    if (wave == 0) {
        float s = (lane < 32) ? ms[0][lane] : ms[1][lane - 32];
        int r = (lane < 32) ? mr[0][lane] : mr[1][lane - 32];
        wave_sort64(s, r, lane);
        for (int w = 2; w < 4; ++w) {
            if (lane >= 32) { s = ms[w][lane - 32]; r = mr[w][lane - 32]; }
            wave_sort64(s, r, lane);
        }
        if (lane < a.kp) {
            const size_t o = ((size_t)blockIdx.y * a.Q + q) * a.kp + lane;
            a.out_s[o] = s;
            a.out_r[o] = (r == SCAN_ROW_NONE) ? -1 : r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K11 stage 1b: merge the per-slab lists of one query into the query's top-k' (same filter +
// compact machinery, entries streamed 64 per wave-instruction). One block per query.
// Writes the page into cand[q][page_off .. page_off+kp) and advances the paging cursor.
// ------------------------------------------------------------------------------------------------
struct MergeArgs {
    const float* in_s;    // [L][Q][kp]
    const int32_t* in_r;
    int L, Q, kp;
    int32_t* cand;        // [Q][cand_stride] row ids, -1 = none
    int cand_stride, page_off;
    float* cur_s;         // [Q] cursor, updated to the page's last entry (or -inf when exhausted); may be null
    int32_t* cur_r;
    // first pass of a two-level merge (grid.y > 1): block (q, y) merges lists [y*lists_per_block, ...) and writes a
    // LIST [gridDim.y][Q][kp] (scores kept) instead of the final candidates
    int lists_per_block;
    float* out_s;
    int32_t* out_r;
    // threshold-filtered selection: besides the L lists, min(flat_cnt[q], flat_cap) loose (score, id) entries per query
    const float* flat_s;     // [Q][flat_cap]
    const int32_t* flat_r;
    const int32_t* flat_cnt; // [Q]
    int flat_cap;
};

// A wave streams 64 - kp entries per step, so a list of <= kp (<= 32) entries can never outgrow its 64-slot
// buffer before the compaction that follows the step.
__global__ __launch_bounds__(256) void merge_lists_kernel(MergeArgs a) {
    __shared__ float cs[4][64];
    __shared__ int cr[4][64];
    __shared__ int cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const int kp = a.kp;  // <= 32
    if (lane == 0) cnt[wave] = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    float tau = SCAN_NEG_INF;
    const int l_begin = a.out_s ? blockIdx.y * a.lists_per_block : 0;
    int l_end = a.out_s ? l_begin + a.lists_per_block : a.L;
    if (l_end > a.L) l_end = a.L;
    const int64_t list_total = (int64_t)(l_end > l_begin ? l_end - l_begin : 0) * kp;
    int64_t total = list_total;
    if (a.flat_s) {
        const int c = a.flat_cnt[q];
        total += c < a.flat_cap ? c : a.flat_cap;
    }
    // A wave takes NL = 64 - kp entries per step (its 64-slot buffer holds at most kp survivors + NL newcomers) and loads
    // the next step's entries before it files the current ones, so the global-load latency overlaps the LDS work.
    const int NL = 64 - kp;
    auto fetch = [&](int64_t e0, float& s, int& r) {
        s = SCAN_NEG_INF;
        r = -1;
        const int64_t e = e0 + lane;
        if (lane < NL && e < total) {
            if (e < list_total) {
                const int64_t l = l_begin + e / kp;
                const int slot = (int)(e % kp);
                const size_t o = ((size_t)l * a.Q + q) * kp + slot;
                s = a.in_s[o];
                r = a.in_r[o];
            } else {
                const size_t o = (size_t)q * a.flat_cap + (size_t)(e - list_total);
                s = a.flat_s[o];
                r = a.flat_r[o];
            }
        }
    };
    float s, sn;
    int r, rn;
    int64_t e0 = (int64_t)wave * NL;
    fetch(e0, s, r);
    while (e0 < total) {
        const int64_t e1 = e0 + 4 * (int64_t)NL;
        fetch(e1, sn, rn);
        if (r >= 0 && s > tau) {
            const int pos = atomicAdd(&cnt[wave], 1);
            cs[wave][pos] = s;
            cr[wave][pos] = r;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int c = cnt[wave];
        if (c > kp) {
            float ss = SCAN_NEG_INF;
            int rr = SCAN_ROW_NONE;
            if (lane < c) { ss = cs[wave][lane]; rr = cr[wave][lane]; }
            wave_sort64(ss, rr, lane);
            if (lane < kp) { cs[wave][lane] = ss; cr[wave][lane] = rr; }
            tau = __shfl(ss, kp - 1);
            if (lane == 0) cnt[wave] = kp;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        e0 = e1;
        s = sn;
        r = rn;
    }
    __syncthreads();
    if (wave == 0) {
        float s = SCAN_NEG_INF;
        int r = SCAN_ROW_NONE;
        const int c0 = cnt[0];
        if (lane < c0) { s = cs[0][lane]; r = cr[0][lane]; }
        // sort the first list too (it may never have been compacted)
        wave_sort64(s, r, lane);
        for (int w = 1; w < 4; ++w) {
            const int cw = cnt[w];
            if (cw == 0) continue;  // (wave-uniform; a short input leaves the later waves' lists empty)
            if (lane >= kp) { s = SCAN_NEG_INF; r = SCAN_ROW_NONE; }
            if (lane >= 32 && lane - 32 < cw) { s = cs[w][lane - 32]; r = cr[w][lane - 32]; }
            wave_sort64(s, r, lane);
        }
        const bool valid = lane < kp && r != SCAN_ROW_NONE;
        if (a.out_s) {
            if (lane < kp) {
                const size_t o = ((size_t)blockIdx.y * a.Q + q) * kp + lane;
                a.out_s[o] = valid ? s : SCAN_NEG_INF;
                a.out_r[o] = valid ? r : -1;
            }
            return;
        }
        if (lane < kp) a.cand[(size_t)q * a.cand_stride + a.page_off + lane] = valid ? r : -1;
        if (a.cur_s) {
            const unsigned long long m = __ballot(valid);
            const int nvalid = __popcll(m);
            const float ls = __shfl(s, nvalid > 0 ? nvalid - 1 : 0);
            const int lr = __shfl(r, nvalid > 0 ? nvalid - 1 : 0);
            if (lane == 0) {
                if (nvalid == kp) { a.cur_s[q] = ls; a.cur_r[q] = lr; }
                else { a.cur_s[q] = SCAN_NEG_INF; a.cur_r[q] = SCAN_ROW_NONE; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K11 stage 2: canonical rerank + final ordering. One block per query.
//   dist(q, c) = (float)(1.0 - canon_dot(qn, row))   — fp64, fixed order (see file header)
//   order      = (dist asc, label asc)
// ncand <= 2048 candidates per query (row ids, -1 = none).
// ------------------------------------------------------------------------------------------------
struct RerankArgs {
    const void* rows; int D;
    const float* qn;        // [Q, D] canonical-normalised queries
    const int32_t* cand;    // [blocks][cand_stride]
    int cand_stride, ncand;
    int group_mode;         // 1: cand holds GROUPMAX group ids, candidate c -> member c & 15 of group cand[c >> 4]
    int64_t nrows;          // rows in the index (group members beyond it are skipped)
    const int64_t* labels;  // [N] row -> label (strictly increasing with the row: (dist, row) order == (dist, label) order)
    int k;
    int64_t* out_labels;    // [Q, k]
    float* out_dist;        // [Q, k]
    int32_t* out_count;     // [Q]
    // widen pass: block b serves query qmap[b] (null: b) and also returns its k best ROWS (they seed the next round)
    const int32_t* qmap;
    int32_t* out_rows;      // [blocks][cand_stride], first k_eff entries written, or null
    // exactness guard (see api_index.hip "exactness contract"): tau[b] = stage 1's bound on the approximate score of every
    // row that is NOT among the candidates (-inf: every row is a candidate). The result is proven exact when the k-th
    // canonical score clears tau by more than eps; otherwise flags[b] = 1 and *nflag (when given) counts it.
    const float* tau;
    double eps;
    int32_t* flags;
    int32_t* nflag;
    int force_flag;         // tests: flag every query whose candidate list was full
    // threshold-filtered selection: a query whose candidate list overflowed (ovf_cnt[q] > ovf_cap) lost candidates: flag it
    const int32_t* ovf_cnt;
    int ovf_cap;
    const float* eps_q;     // per-query error bound (prep_queries_kernel), indexed by the ORIGINAL query; null: eps
    // out: the threshold of the widen pass, thr_out[q] = (k-th canonical score) - eps rounded down — every row that can
    // still belong to the top-k has an approximate score >= it (-inf when fewer than k candidates were valid)
    float* thr_out;
    const int32_t* cand_cnt;  // per-block candidate count (widen pass: the appended rows), null: ncand for every block
    const int32_t* bmap;      // block b reads candidate list / count bmap[b] (null: b)
    const float* inv;         // fp8 rows: inverse norm per row (f8_row_inv_kernel): canonical score = canon_dot x (double)inv[row]
    // widen pass: block b also leaves its list's RAW length cand_cnt[lb] here — PINNED HOST memory, one plain store per block, as
    // `flags` — so the host learns the lengths with the kernel's end instead of through a copy behind it (round 6); null: no
    int32_t* cnt_host;
};

// The four partial sums of one lane of the canonical dot product (lane p of a row's 16: elements d = 4p + j + 64 i, i
// ascending, acc[j] = sum over i in fp64 — the additions of canon_dot in oracle/, in its order). rv / qp point at the lane's
// first element of the row / the query. The loads of EIGHT 64-element pieces (row and query) are issued before the first
// multiply-add: with one piece per trip the loop was a chain of D/64 dependent memory round trips per row group — at
// D = 512 eight of them, ~1 us each: 33 us for the 256 candidate rows of a query (round 4; same bits).
template <typename T>
__device__ __forceinline__ void canon_partial_dots(const T* rv, bool live, const float* qp, int D, double (&acc)[4]) {
    acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
    typedef typename std::conditional<sizeof(T) == 1, uint32_t, typename std::conditional<sizeof(T) == 2, u32x2, f32x4>::type>::type piece_t;
    for (int d0 = 0; d0 < D; d0 += 512) {
        piece_t w[8];
        f32x4 qq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            w[u] = piece_t{};   // (zero: fp8 code 0 = +0.0)
            qq[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (d0 + 64 * u < D) {
                qq[u] = *reinterpret_cast<const f32x4*>(qp + d0 + 64 * u);
                if (live) w[u] = *reinterpret_cast<const piece_t*>(rv + d0 + 64 * u);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (d0 + 64 * u < D) {
                float rr[4];
                if constexpr (sizeof(T) == 1) {
                    f8x4_values(w[u], rr);
                } else if constexpr (sizeof(T) == 2) {
                    const _Float16* h = reinterpret_cast<const _Float16*>(&w[u]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) rr[j] = (float)h[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) rr[j] = w[u][j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = acc[j] + (double)qq[u][j] * (double)rr[j];
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void rerank_kernel(RerankArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int q = a.qmap ? a.qmap[b] : b;
    const int lb = a.bmap ? a.bmap[b] : b;   // which candidate list
    int nc = a.ncand;                        // candidates of THIS block (block-uniform)
    if (a.cand_cnt) {
        const int cc = a.cand_cnt[lb];
        if (a.cnt_host && tid == 0) a.cnt_host[b] = cc;
        nc = cc < nc ? cc : nc;
    }
    // sort buffer: npow2 entries of (dist f32, row i32), sized by the block's own candidate count (the launch provides LDS for
    // a.ncand): a widened query that collected 60 rows sorts 64 entries, not the launch's 1024
    int npow = 1;
    while (npow < nc) npow <<= 1;
    float* sd = reinterpret_cast<float*>(smem);
    int32_t* sr = reinterpret_cast<int32_t*>(smem + (size_t)npow * 4);
    const float* qv = a.qn + (size_t)q * a.D;
    const int nthreads = blockDim.x, nwaves = blockDim.x >> 6;
    for (int i = tid; i < npow; i += nthreads) { sd[i] = INFINITY; sr[i] = INT32_MAX; }
    __syncthreads();
    // Canonical dot products, FOUR candidate rows per wave at a time: the 16 lanes p of a quarter wave stand for the 64
    // lanes of the canonical reduction, lane p holding the partial sums of canonical lanes 4p .. 4p+3 (elements
    // d = 4p + j + 64 i, i ascending — the same additions in the same order). The butterfly's stages xor 32, 16, 8, 4 pair
    // canonical lanes 4p + j and 4(p ^ {8,4,2,1}) + j: a shuffle inside the quarter wave per partial sum; stages xor 2, 1
    // pair partial sums of one lane. Same bits as one row per wave (canon_dot in oracle/), a quarter of the dependent
    // steps, 8- / 16-byte row loads instead of 2- / 4-byte ones, and the query slice read once per four rows.
    const int p16 = lane & 15, sub = lane >> 4;
    for (int c0 = wave * 4; c0 < nc; c0 += nwaves * 4) {
        const int c = c0 + sub;
        int64_t row = -1;
        if (c < nc) {
            if (a.group_mode) {
                const int g = a.cand[(size_t)lb * a.cand_stride + (c >> 4)];
                if (g >= 0) {
                    row = groupmax_row(g, c & 15);
                    if (row >= a.nrows) row = -1;
                }
            } else {
                row = a.cand[(size_t)lb * a.cand_stride + c];
            }
        }
        const bool live = row >= 0;  // (no candidate, or an empty index: nothing is dereferenced)
        const T* rv = reinterpret_cast<const T*>(a.rows) + (size_t)(live ? row : 0) * a.D + 4 * p16;
        const float* qp = qv + 4 * p16;
        double acc[4];
        canon_partial_dots<T>(rv, live, qp, a.D, acc);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] + __shfl_xor(acc[j], o);
        }
        const double t0 = acc[0] + acc[2], t1 = acc[1] + acc[3];
        double dot = t0 + t1;
        if (p16 == 0 && row >= 0) {
            if constexpr (sizeof(T) == 1) dot = dot * (double)a.inv[row];   // fp8 rows: the represented row is values x inv
            sd[c] = (float)(1.0 - dot);
            sr[c] = (int32_t)row;
        }
    }
    __syncthreads();
    // block bitonic sort ascending by (dist, row); NaN distances sort last
    for (int k = 2; k <= npow; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow; i += nthreads) {
                const int p = i ^ j;
                if (p > i) {
                    const float di = sd[i], dp = sd[p];
                    const int32_t li = sr[i], lp = sr[p];
                    const bool up = (i & k) == 0;
                    // "i before p" in the final order?
                    const bool i_first = (di < dp) || (di == dp && li < lp) || (dp != dp && di == di);
                    const bool p_first = (dp < di) || (dp == di && lp < li) || (di != di && dp == dp);
                    const bool swap = up ? p_first : i_first;
                    if (swap) { sd[i] = dp; sd[p] = di; sr[i] = lp; sr[p] = li; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < a.k; i += nthreads) {
        const bool ok = i < npow && sr[i] != INT32_MAX;
        a.out_labels[(size_t)q * a.k + i] = ok ? a.labels[sr[i]] : -1;
        a.out_dist[(size_t)q * a.k + i] = ok ? sd[i] : INFINITY;
    }
    const int64_t k_eff = (int64_t)a.k < a.nrows ? (int64_t)a.k : a.nrows;  // results that exist
    if (a.out_rows) {
        for (int i = tid; i < a.cand_stride && i < (int)((k_eff + 31) & ~31LL); i += nthreads)
            a.out_rows[(size_t)b * a.cand_stride + i] = (i < k_eff && i < npow && sr[i] != INT32_MAX) ? sr[i] : -1;
    }
    if (tid == 0) {
        int nvalid = 0;
        const int lim = a.k < npow ? a.k : npow;
        for (int i = 0; i < lim; ++i) nvalid += (sr[i] != INT32_MAX);
        a.out_count[q] = nvalid;
        const double eps = a.eps_q ? (double)a.eps_q[q] : a.eps;
        if (a.flags) {
            int flag = 0;
            const float t = a.tau ? a.tau[b] : SCAN_NEG_INF;
            if (t > SCAN_NEG_INF) {  // stage 1 left rows out: every one of them has an approximate score <= t
                if (nvalid < k_eff || a.force_flag) flag = 1;
                else if (a.ovf_cnt && a.ovf_cnt[q] > a.ovf_cap) flag = 1;
                else {
                    const double ck = 1.0 - (double)sd[k_eff - 1];  // k-th canonical score (its float rounding is inside eps)
                    if (!(ck - (double)t > eps)) flag = 1;
                }
            }
            a.flags[b] = flag;
            if (flag && a.nflag) atomicAdd(a.nflag, 1);
        }
        if (a.thr_out) {
            float thr = SCAN_NEG_INF;
            if (k_eff > 0 && nvalid >= k_eff) {
                const double lim = (1.0 - (double)sd[k_eff - 1]) - eps;
                thr = (float)lim;
                if ((double)thr > lim) thr = nextafterf(thr, -INFINITY);
                if (!(lim == lim)) thr = SCAN_NEG_INF;   // (NaN distances: keep everything)
            }
            a.thr_out[q] = thr;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K11 last resort: the canonical distance of EVERY row to one query, dist[r] = (float)(1.0 - canon_dot(qn, row r)) — the
// arithmetic of rerank_kernel (four rows per wave, 16 lanes per row, same additions in the same order: same bits). The
// widen pass falls back on it when a query is still unproven after a few rounds (a plateau of hundreds of thousands of
// rows within eps of the k-th score: identical placeholder images, re-uploads); the host then selects by (dist, row).
// One pass over the index per query: N * D * elt bytes.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void canonical_scan_kernel(const void* __restrict__ rows, int64_t N, int D,
                                                             const float* __restrict__ qv, float* __restrict__ dist,
                                                             const float* __restrict__ inv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p16 = lane & 15, sub = lane >> 4;
    const float* qp = qv + 4 * p16;
    for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 4; r0 < N; r0 += (int64_t)gridDim.x * 16) {
        const int64_t row = r0 + sub;
        const bool live = row < N;
        const T* rv = reinterpret_cast<const T*>(rows) + (size_t)(live ? row : 0) * D + 4 * p16;
        double acc[4];
        canon_partial_dots<T>(rv, live, qp, D, acc);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] + __shfl_xor(acc[j], o);
        }
        const double t0 = acc[0] + acc[2], t1 = acc[1] + acc[3];
        double dot = t0 + t1;
        if (p16 == 0 && live) {
            if constexpr (sizeof(T) == 1) dot = dot * (double)inv[row];
            dist[row] = (float)(1.0 - dot);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K12: multimodal blend (backend/app/main.py:852-860), one wave per query:
//   i^ = i/|i|, t^ = t/|t| (canonical fp64 norms, rounded to f32 like the index rows),
//   c = (float)w * i^ + (float)(1-w) * t^   (two f32 multiplies and one f32 add, as numpy evaluates it),
//   out = c/|c|.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blend_kernel(const float* __restrict__ img, const float* __restrict__ txt,
                                                    float w_img, float w_txt, int Q, int D, float* __restrict__ out) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= Q) return;
    const float* xi = img + (size_t)r * D;
    const float* xt = txt + (size_t)r * D;
    double ai = 0.0, at = 0.0;
    for (int d = lane; d < D; d += 64) {
        const double vi = (double)xi[d], vt = (double)xt[d];
        ai = ai + vi * vi;
        at = at + vt * vt;
    }
    const double ni = sqrt(wave_butterfly_sum(ai)), nt = sqrt(wave_butterfly_sum(at));
    double ac = 0.0;
    for (int d = lane; d < D; d += 64) {
        float yi = (float)((double)xi[d] / ni), yt = (float)((double)xt[d] / nt);
        asm volatile("" : "+v"(yi), "+v"(yt));
        float pi = w_img * yi, pt = w_txt * yt;
        asm volatile("" : "+v"(pi), "+v"(pt));  // two rounded products, then one rounded add
        float c = pi + pt;
        asm volatile("" : "+v"(c));
        ac = ac + (double)c * (double)c;
    }
    const double nc = sqrt(wave_butterfly_sum(ac));
    for (int d = lane; d < D; d += 64) {
        float yi = (float)((double)xi[d] / ni), yt = (float)((double)xt[d] / nt);
        asm volatile("" : "+v"(yi), "+v"(yt));
        float pi = w_img * yi, pt = w_txt * yt;
        asm volatile("" : "+v"(pi), "+v"(pt));
        float c = pi + pt;
        asm volatile("" : "+v"(c));
        out[(size_t)r * D + d] = (float)((double)c / nc);
    }
}

// ------------------------------------------------------------------------------------------------
// X1 tail: merge S gathered per-shard result lists into the global top-k by (dist asc, label asc).
// One block per query, S*k <= 4096 entries.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shard_merge_kernel(const float* __restrict__ dist, const int64_t* __restrict__ labels,
                                                          int S, int Q, int k, float* __restrict__ out_dist,
                                                          int64_t* __restrict__ out_labels, int32_t* __restrict__ out_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = S * k;
    int npow = 1;
    while (npow < n) npow <<= 1;
    float* sd = reinterpret_cast<float*>(smem);
    int64_t* sl = reinterpret_cast<int64_t*>(smem + (size_t)npow * 4 + ((npow & 1) ? 4 : 0));
    const int tid = threadIdx.x, q = blockIdx.x;
    for (int i = tid; i < npow; i += 256) {
        float d = INFINITY;
        int64_t l = INT64_MAX;
        if (i < n) {
            const int s = i / k, j = i - s * k;
            const size_t o = ((size_t)s * Q + q) * k + j;
            const int64_t ll = labels[o];
            if (ll >= 0) { l = ll; d = dist[o]; }
        }
        sd[i] = d;
        sl[i] = l;
    }
    __syncthreads();
    for (int kk = 2; kk <= npow; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow; i += 256) {
                const int p = i ^ j;
                if (p > i) {
                    const float di = sd[i], dp = sd[p];
                    const int64_t li = sl[i], lp = sl[p];
                    const bool up = (i & kk) == 0;
                    const bool i_first = (di < dp) || (di == dp && li < lp) || (dp != dp && di == di);
                    const bool p_first = (dp < di) || (dp == di && lp < li) || (di != di && dp == dp);
                    const bool swap = up ? p_first : i_first;
                    if (swap) { sd[i] = dp; sd[p] = di; sl[i] = lp; sl[p] = li; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k; i += 256) {
        const bool ok = i < npow && sl[i] != INT64_MAX;
        out_labels[(size_t)q * k + i] = ok ? sl[i] : -1;
        out_dist[(size_t)q * k + i] = ok ? sd[i] : INFINITY;
    }
    if (tid == 0 && out_count) {
        int c = 0;
        const int lim = k < npow ? k : npow;
        for (int i = 0; i < lim; ++i) c += (sl[i] != INT64_MAX);
        out_count[q] = c;
    }
}

// ------------------------------------------------------------------------------------------------
// row movement for stable removal and for "get": dst[i] = src[map[i]] (rows of D elements)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, const int64_t* __restrict__ map,
                                                          T* __restrict__ dst, int64_t n, int D) {
    const int64_t total = n * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int d = (int)(i - r * D);
        dst[i] = src[map[r] * D + d];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const T* __restrict__ src, const int64_t* __restrict__ map,
                                                              float* __restrict__ dst, int64_t n, int D) {
    const int64_t total = n * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int d = (int)(i - r * D);
        dst[i] = (float)src[map[r] * D + d];
    }
}
// fp8 rows as the vectors they represent: value x inv[row] (one f32 multiply)
__global__ __launch_bounds__(256) void gather_rows_f8_f32_kernel(const F8* __restrict__ src, const float* __restrict__ inv,
                                                                 const int64_t* __restrict__ map, float* __restrict__ dst, int64_t n, int D) {
    const int64_t total = n * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / D;
        const int d = (int)(i - r * D);
        dst[i] = (float)src[map[r] * D + d] * inv[map[r]];
    }
}
__global__ void gather_i64_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ map,
                                  int64_t* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[map[i]];
}

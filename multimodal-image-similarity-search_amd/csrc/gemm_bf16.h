// gemm_bf16.h — bf16 MFMA GEMM for the CLIP towers:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue).
//
// Covers K1 (patch projection), K3 (QKV), K5 (out-proj + residual), K6 (FC1 + QuickGELU),
// K7 (FC2 + residual) and K8's projection of SURVEY.md §2.2. The arithmetic it replaces is
// nn.Linear / nn.Conv2d inside HF:modeling_clip.py:148-154,293-296,338-350,674-675.
//
// Both operands are K-contiguous (nn.Linear stores [out,in]), which is exactly the per-lane fragment
// shape of v_mfma_f32_16x16x32_bf16 (8 consecutive k per lane), so neither needs a transpose.
// The MFMA "A" operand is the WEIGHT tile and the "B" operand the ACTIVATION tile: the accumulator
// then holds 4 consecutive n (output features) per lane for one m (token), so epilogue stores are
// 8-byte (bf16) / 16-byte (f32) vectors along the contiguous output dimension.
//
// v1 structure (guide §5 "minimum 2-phase"): 128x128x64 tile, 4 waves (2x2, 64x64 per wave),
// global_load_lds dwordx4 staging into a double-buffered, XOR-swizzled LDS image (swizzle applied to
// the SOURCE address and to the read, destination linear — guide §5.4 rule 21), one barrier per K-tile,
// 2 workgroups per CU so one block's staging wait overlaps the other's MFMAs.
#pragma once
#include "common.h"

struct GemmEpi {
    void* out;          // f32 or bf16, row stride ldo elements
    const float* bias;  // [N] or null
    const float* aux;   // PATCH: position table [T, N]
    int ldo;            // output row stride (elements)
    int m_valid;        // rows >= m_valid are not stored
    int p0, p1;         // PATCH: p0 = patches per image (G), p1 = tokens per image (T)
};

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_LDS_BYTES (2 * (GEMM_BM + GEMM_BN) * GEMM_BK * 2)

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ float quick_gelu(float x) {
    // x * sigmoid(1.702 x)  — HF:activations.py:117-123
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
}

// XCD-aware, bijective block remap (guide §5 "XCD swizzle must be bijective"): blocks that share an
// XCD (equal bid % 8) get a contiguous run of tiles, so an activation row-panel is fetched into one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

template <int EPI>
__device__ __forceinline__ void gemm_store4(const GemmEpi& ep, int m, int n, f32x4 v) {
    if (m >= ep.m_valid) return;
    if constexpr (EPI == MMISS_EPI_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + (size_t)m * ep.ldo + n) = v;
    } else if constexpr (EPI == MMISS_EPI_BIAS_BF16 || EPI == MMISS_EPI_BIAS_QGELU_BF16) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(ep.bias + n);
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            y[i] = v[i] + b[i];
            if constexpr (EPI == MMISS_EPI_BIAS_QGELU_BF16) y[i] = quick_gelu(y[i]);
        }
        u32x2 pk;
        pk[0] = pack_bf16x2(y[0], y[1]);
        pk[1] = pack_bf16x2(y[2], y[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(ep.out) + (size_t)m * ep.ldo + n) = pk;
    } else if constexpr (EPI == MMISS_EPI_BIAS_RESID_F32) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(ep.bias + n);
        float* p = reinterpret_cast<float*>(ep.out) + (size_t)m * ep.ldo + n;
        f32x4 x = *reinterpret_cast<const f32x4*>(p);
        x += v + b;
        *reinterpret_cast<f32x4*>(p) = x;
    } else if constexpr (EPI == MMISS_EPI_PATCH_F32) {
        const int img = m / ep.p0, patch = m - img * ep.p0;
        const f32x4 pos = *reinterpret_cast<const f32x4*>(ep.aux + (size_t)(1 + patch) * ep.ldo + n);
        float* p = reinterpret_cast<float*>(ep.out) + ((size_t)img * ep.p1 + 1 + patch) * ep.ldo + n;
        *reinterpret_cast<f32x4*>(p) = v + pos;
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_128x128(const bf16_t* __restrict__ A,
                                                            const bf16_t* __restrict__ W, int M, int N,
                                                            int K, GemmEpi ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbn = N / GEMM_BN;
    const int nwg = (M / GEMM_BM) * nbn;
    const int wg = xcd_remap(blockIdx.x, nwg);
    const int bm = wg / nbn, bn = wg - bm * nbn;

    const bf16_t* Ab = A + (size_t)bm * GEMM_BM * K;
    const bf16_t* Wb = W + (size_t)bn * GEMM_BN * K;

    // staging: one wave-instruction = 8 rows x 128 B; lane -> (row r_in, 16-B slot p); slot p of
    // row r holds global chunk p ^ (r & 7)
    const int r_in = lane >> 3, p = lane & 7;
    const int src_chunk = (p ^ r_in) * 8;  // elements
    auto stage = [&](int buf, int kt) {
        char* sA = smem + buf * 32768;
        char* sW = sA + 16384;
        const size_t koff = (size_t)kt * GEMM_BK + src_chunk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rowblk = wave * 4 + i;
            const int row = rowblk * 8 + r_in;
            glds16(Ab + (size_t)row * K + koff, sA + rowblk * 1024);
            glds16(Wb + (size_t)row * K + koff, sW + rowblk * 1024);
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = K / GEMM_BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) stage(cur ^ 1, t + 1);
        const char* sA = smem + cur * 32768;
        const char* sW = sA + 16384;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 wf[4], af[4];
            const int chunk = 4 * s + fg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wn * 64 + i * 16 + fr;
                wf[i] = *reinterpret_cast<const bf16x8*>(sW + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 64 + j * 16 + fr;
                af[j] = *reinterpret_cast<const bf16x8*>(sA + row * 128 + ((chunk ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // acc[i][j][reg] = C[m = .. + j*16 + fr][n = .. + i*16 + 4*fg + reg]
    const int m_base = bm * GEMM_BM + wm * 64 + fr;
    const int n_base = bn * GEMM_BN + wn * 64 + 4 * fg;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) gemm_store4<EPI>(ep, m_base + j * 16, n_base + i * 16, acc[i][j]);
}

// algorithmic flops / bytes of one launch (for mmiss_prof_*)
static inline double gemm_flops(int M, int N, int K) { return 2.0 * M * N * K; }

static int launch_gemm(hipStream_t st, int epi, int variant, const void* A, const void* W, const GemmEpi& ep,
                       int M, int N, int K) {
    (void)variant;
    if (M <= 0 || N <= 0 || K <= 0 || (M % GEMM_BM) || (N % GEMM_BN) || (K % GEMM_BK))
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "gemm: M=%d N=%d K=%d must be multiples of %d/%d/%d", M, N, K, GEMM_BM,
                GEMM_BN, GEMM_BK);
    const int nwg = (M / GEMM_BM) * (N / GEMM_BN);
    const bf16_t* a = reinterpret_cast<const bf16_t*>(A);
    const bf16_t* w = reinterpret_cast<const bf16_t*>(W);
    static const char* names[] = {"gemm_bf16_f32", "gemm_bf16_bias", "gemm_bf16_bias_qgelu",
                                  "gemm_bf16_bias_resid", "gemm_bf16_patch"};
    if (epi < 0 || epi > 4) MM_FAIL(MMISS_ERR_ARG, "gemm: bad epilogue %d", epi);
    const int out_elt = (epi == MMISS_EPI_BIAS_BF16 || epi == MMISS_EPI_BIAS_QGELU_BF16) ? 2 : 4;
    const double bytes = 2.0 * ((double)M * K + (double)N * K) + (double)out_elt * M * N *
                                                                      (epi == MMISS_EPI_BIAS_RESID_F32 ? 2 : 1);
    MM_PROF(names[epi], st, gemm_flops(ep.m_valid < M ? ep.m_valid : M, N, K), bytes);
#define GEMM_LAUNCH(E)                                                                                    \
    case E: {                                                                                             \
        static bool attr_done = false;                                                                    \
        if (!attr_done) {                                                                                 \
            MM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_128x128<E>),              \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES));      \
            attr_done = true;                                                                             \
        }                                                                                                 \
        hipLaunchKernelGGL(gemm_bf16_128x128<E>, dim3(nwg), dim3(256), GEMM_LDS_BYTES, st, a, w, M, N, K, ep); \
    } break;
    switch (epi) {
        GEMM_LAUNCH(MMISS_EPI_F32)
        GEMM_LAUNCH(MMISS_EPI_BIAS_BF16)
        GEMM_LAUNCH(MMISS_EPI_BIAS_QGELU_BF16)
        GEMM_LAUNCH(MMISS_EPI_BIAS_RESID_F32)
        GEMM_LAUNCH(MMISS_EPI_PATCH_F32)
    }
#undef GEMM_LAUNCH
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

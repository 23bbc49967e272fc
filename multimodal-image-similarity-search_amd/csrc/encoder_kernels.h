// encoder_kernels.h — the non-GEMM kernels of the CLIP towers (K1 prologue, K2, K4, K8, K9 of
// SURVEY.md §2.2). Each kernel cites the HF arithmetic it restates.
#pragma once
#include "common.h"
#include "attention_stream.h"

// ------------------------------------------------------------------------------------------------
// K1 prologue: patchify.  pixels f32 [B,3,S,S] -> patches bf16 [B*G*G, Kp], k = c*P*P + ky*P + kx
// (the flatten order of nn.Conv2d's weight [d,3,P,P], HF:modeling_clip.py:148-154,209-211), zero
// padded from 3*P*P to Kp. One thread produces 8 consecutive k (one 16-byte store).
// SRC_U8: pixels are uint8 [B,S,S,3] HWC and the CLIP rescale+normalise
// ((x/255 - mean)/std, HF:image_processing_clip.py:23-34) is fused in.
// ------------------------------------------------------------------------------------------------
template <bool SRC_U8>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ pixels, uint16_t* __restrict__ out,
                                                     int B, int S, int P, int Kp) {
    const int G = S / P;
    const int PP = P * P;
    const int Kreal = 3 * PP;
    const int kgroups = Kp >> 3;
    const int64_t total = (int64_t)B * G * G * kgroups;
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float istd[3] = {1.0f / 0.26862954f, 1.0f / 0.26130258f, 1.0f / 0.27577711f};
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int kg = (int)(idx % kgroups);
        const int64_t m = idx / kgroups;
        const int b = (int)(m / (G * G));
        const int pr = (int)(m - (int64_t)b * G * G);
        const int py = pr / G, px = pr - py * G;
        const int k0 = kg << 3;
        float v[8];
        if (!SRC_U8 && (P & 7) == 0 && k0 < Kreal) {
            // 8 consecutive kx of one (c, ky): two float4 loads
            const int c = k0 / PP, rem = k0 - c * PP, ky = rem / P, kx = rem - ky * P;
            const float* src = reinterpret_cast<const float*>(pixels) +
                               (((size_t)b * 3 + c) * S + (size_t)py * P + ky) * S + (size_t)px * P + kx;
            const f32x4 a = *reinterpret_cast<const f32x4*>(src);
            const f32x4 d = *reinterpret_cast<const f32x4*>(src + 4);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
            v[4] = d[0]; v[5] = d[1]; v[6] = d[2]; v[7] = d[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e;
                float x = 0.f;
                if (k < Kreal) {
                    const int c = k / PP, rem = k - c * PP, ky = rem / P, kx = rem - ky * P;
                    const size_t y = (size_t)py * P + ky, xx = (size_t)px * P + kx;
                    if (SRC_U8) {
                        const uint8_t u = reinterpret_cast<const uint8_t*>(pixels)[(((size_t)b * S + y) * S + xx) * 3 + c];
                        // same op order as the HF processor: rescale (x * 1/255) then (x - mean) / std
                        x = ((float)u * (1.0f / 255.0f) - mean[c]) * istd[c];
                    } else {
                        x = reinterpret_cast<const float*>(pixels)[(((size_t)b * 3 + c) * S + y) * S + xx];
                    }
                }
                v[e] = x;
            }
        }
        u32x4 pk;
        pk[0] = pack_bf16x2(v[0], v[1]);
        pk[1] = pack_bf16x2(v[2], v[3]);
        pk[2] = pack_bf16x2(v[4], v[5]);
        pk[3] = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<u32x4*>(out + (size_t)m * Kp + k0) = pk;
    }
}

// CLS rows: x[b*T + 0][:] = class_embedding + position_embedding[0]   (HF:modeling_clip.py:213-216)
__global__ void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                int B, int T, int d) {
    const int64_t total = (int64_t)B * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / d), n = (int)(i - (int64_t)b * d);
        x[(size_t)b * T * d + n] = cls[n] + pos[n];
    }
}

// One request at a time (round 5): the vision tower's CLS rows, pre_layrnorm and the entry statistics of the skinny folded mode
// in ONE launch instead of three (cls_rows_kernel, layernorm_kernel in place, skinny_row_stats16_kernel). One wave per token row:
// row t = 0 of an image is class_embedding + position_embedding[0] (HF:modeling_clip.py:213-216) computed here, the others
// were written by the patch GEMM; LayerNorm with layernorm_kernel's arithmetic (two-pass mean / variance, the same lane
// layout and summation order), the result written back as the f32 residual stream, as its bf16 copy, and as the (sum, sumsq)
// of every 16-column slice (four consecutive lanes: skinny_row_stats16_kernel's order) — the same bits as the three kernels.
__global__ __launch_bounds__(256) void prelayernorm_skinny_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                                  const float* __restrict__ pos, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, uint16_t* __restrict__ xb,
                                                                  float* __restrict__ stats, int M, int T, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const bool is_cls = (r % T) == 0;
    float* xr = x + (size_t)r * d;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < d) {
            if (is_cls) v[i] = *reinterpret_cast<const f32x4*>(cls + c) + *reinterpret_cast<const f32x4*>(pos + c);
            else v[i] = *reinterpret_cast<const f32x4*>(xr + c);
        }
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[i][e] - mean;
                q += t * t;
            }
        }
    }
    const float var = wave_sum(q) / (float)d;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;   // (wave-uniform validity per i: d % 256 == 0 is not required, d % 16 == 0 is)
        f32x4 y = {0.f, 0.f, 0.f, 0.f};
        if (c < d) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + bb[e];
            *reinterpret_cast<f32x4*>(xr + c) = y;
            u32x2 pk;
            pk[0] = pack_bf16x2(y[0], y[1]);
            pk[1] = pack_bf16x2(y[2], y[3]);
            *reinterpret_cast<u32x2*>(xb + (size_t)r * d + c) = pk;
        }
        float ss = (y[0] + y[1]) + (y[2] + y[3]);
        float qq = (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
        ss += __shfl_xor(ss, 1); qq += __shfl_xor(qq, 1);
        ss += __shfl_xor(ss, 2); qq += __shfl_xor(qq, 2);
        if (c < d && (lane & 3) == 0) {
            float* o = stats + ((size_t)r * (d >> 4) + (c >> 4)) * 2;
            o[0] = ss;
            o[1] = qq;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K9: text embeddings.  x[b*T+t][:] = token_embedding[ids[b,t]] + position_embedding[t]
// (HF:modeling_clip.py:226-256). Also finds the pooled position per row: first id == eos_id, or
// argmax(ids) when eos_id == 2 (legacy checkpoints) — HF:modeling_clip.py:561-581.
// One block per (b); threads stride over t*d.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_embed_kernel(const int32_t* __restrict__ ids, const float* __restrict__ tok,
                                                         const float* __restrict__ pos, float* __restrict__ x,
                                                         int32_t* __restrict__ pool_row, int T, int d, int vocab,
                                                         int eos_id) {
    const int b = blockIdx.x;
    const int32_t* row = ids + (size_t)b * T;
    if (threadIdx.x == 0) {
        int pos_eos = 0;
        if (eos_id == 2) {
            int best = row[0];
            for (int t = 1; t < T; ++t)
                if (row[t] > best) { best = row[t]; pos_eos = t; }
        } else {
            // (ids == eos).int().argmax(): first match, 0 when there is none
            for (int t = 0; t < T; ++t)
                if (row[t] == eos_id) { pos_eos = t; break; }
        }
        pool_row[b] = b * T + pos_eos;
    }
    const int dv = d >> 2;
    for (int i = threadIdx.x; i < T * dv; i += blockDim.x) {
        const int t = i / dv, c = (i - t * dv) << 2;
        int id = row[t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4 e = *reinterpret_cast<const f32x4*>(tok + (size_t)id * d + c);
        const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (size_t)t * d + c);
        *reinterpret_cast<f32x4*>(x + ((size_t)b * T + t) * d + c) = e + p;
    }
}

// vision pooled row = token 0 of every image (HF:modeling_clip.py:650-651)
__global__ void vision_pool_rows_kernel(int32_t* pool_row, int B, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) pool_row[i] = i * T;
}

// ------------------------------------------------------------------------------------------------
// K2: LayerNorm (eps inside the sqrt, affine; fp32 statistics) — F.layer_norm as used by
// pre_layrnorm / layer_norm1 / layer_norm2 / post_layernorm / final_layer_norm
// (HF:modeling_clip.py:358-360,605-607,504). One wave per row, the row held in registers
// (d <= 1024), two-pass mean / variance. rowmap (optional) gathers input rows: out row r reads
// x row rowmap[r]  (K8's pooling).
// ------------------------------------------------------------------------------------------------
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ out,
                                                        const int32_t* __restrict__ rowmap, int M, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const size_t in_row = rowmap ? (size_t)rowmap[r] : (size_t)r;
    const float* xr = x + in_row * d;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = (c < d) ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[i][e] - mean;
                q += t * t;
            }
        }
    }
    const float var = wave_sum(q) / (float)d;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + bb[e];
            if constexpr (OUT_BF16) {
                u32x2 pk;
                pk[0] = pack_bf16x2(y[0], y[1]);
                pk[1] = pack_bf16x2(y[2], y[3]);
                *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(out) + (size_t)r * d + c) = pk;
            } else {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (size_t)r * d + c) = y;
            }
        }
    }
}

// The same LayerNorm for a bf16 residual stream (bf16 in, bf16 out; d % 8 == 0, d <= 1024): a lane owns 8 consecutive
// elements per 512-column half, one 16-byte load and one 16-byte store each (the 8-byte form of the f32 kernel's layout
// ran at 4.6 TB/s: half the bytes per load instruction in flight). Widening is exact; same two-pass statistics.
// A wave normalises TWO rows, all four of its loads issued before the first use (the kernel is latency-bound: bytes in
// flight per CU, not arithmetic, set its rate).
__global__ __launch_bounds__(256) void layernorm16_kernel(const uint16_t* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, uint16_t* __restrict__ out, int M,
                                                          int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (r0 >= M) return;
    u32x4 w[2][2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = (r0 + rr < M) ? r0 + rr : r0;  // (an odd last row is computed twice, stored once)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = (i * 64 + lane) * 8;
            w[rr][i] = u32x4{0u, 0u, 0u, 0u};
            if (c < d) w[rr][i] = *reinterpret_cast<const u32x4*>(x + (size_t)r * d + c);
        }
    }
    f32x4 g[2][2], bb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < d) {
            g[i][0] = *reinterpret_cast<const f32x4*>(gamma + c); g[i][1] = *reinterpret_cast<const f32x4*>(gamma + c + 4);
            bb[i][0] = *reinterpret_cast<const f32x4*>(beta + c); bb[i][1] = *reinterpret_cast<const f32x4*>(beta + c + 4);
        }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = r0 + rr;
        float v[2][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[i][2 * e] = __uint_as_float(w[rr][i][e] << 16);
                v[i][2 * e + 1] = __uint_as_float(w[rr][i][e] & 0xFFFF0000u);
                s += v[i][2 * e] + v[i][2 * e + 1];
            }
        const float mean = wave_sum(s) / (float)d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if ((i * 64 + lane) * 8 < d) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = v[i][e] - mean;
                    q += t * t;
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
        if (r >= M) continue;  // wave-uniform
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c < d) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = (v[i][e] - mean) * rstd * g[i][0][e] + bb[i][0][e];
                    y[4 + e] = (v[i][4 + e] - mean) * rstd * g[i][1][e] + bb[i][1][e];
                }
                u32x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(y[2 * e], y[2 * e + 1]);
                *reinterpret_cast<u32x4*>(out + (size_t)r * d + c) = pk;
            }
        }
    }
}

// pre_layrnorm for the large calls: LayerNorm in place on the f32 rows AND, in the same pass, what row_stats_kernel would
// produce from the result — the bf16 copy of the new rows (xb) and their (sum, sumsq) in statistics slot 0 of `parts`
// (the others zero). One launch and one 39 MB read less per ViT-B/32 step. d <= 1024, d % 4 == 0.
__global__ __launch_bounds__(256) void layernorm_stats_kernel(float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, uint16_t* __restrict__ xb,
                                                              float* __restrict__ stats, int M, int d, int parts, float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    float* xr = x + (size_t)r * d;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = (c < d) ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((i * 64 + lane) * 4 < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[i][e] - mean;
                q += t * t;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
    float ys = 0.f, yq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + bb[e];
            *reinterpret_cast<f32x4*>(xr + c) = y;
            u32x2 pk;
            pk[0] = pack_bf16x2(y[0], y[1]);
            pk[1] = pack_bf16x2(y[2], y[3]);
            *reinterpret_cast<u32x2*>(xb + (size_t)r * d + c) = pk;
            ys += (y[0] + y[1]) + (y[2] + y[3]);           // (same per-lane order as row_stats_kernel)
            yq += (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
        }
    }
    ys = wave_sum(ys);
    yq = wave_sum(yq);
    float* o = stats + (size_t)r * parts * 2;
    for (int i = lane; i < parts * 2; i += 64) o[i] = (i == 0) ? ys : (i == 1 ? yq : 0.f);
}

// (mean, rstd) of every row from its partial (sum, sumsq) per 64 columns — what the folded GEMM's epilogue applies. The
// persistent kernel finalises a tile's rows itself while K <= 768; for K = 1024 (ViT-L/14) the 32 KB of raw partials of a
// 256-row tile do not fit beside its staging buffers, so this pass runs in front of it: 128 bytes in, 8 out per row, one
// thread per row (4 MB per launch at 32 896 rows: a few us against the 25 us LayerNorm kernel it replaces).
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ stats, float* __restrict__ out, int M,
                                                          int parts, int d, float eps) {
    // eight lanes per row, one 16-byte load each per 16 partials (consecutive lanes read consecutive bytes), then three
    // xor-shuffles inside the group of eight
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int r = gid >> 3, l8 = gid & 7;
    const int rr = r < M ? r : M - 1;
    const f32x4* st = reinterpret_cast<const f32x4*>(stats + (size_t)rr * parts * 2);
    float s1 = 0.f, s2 = 0.f;
    for (int q = l8; q < parts / 2; q += 8) {
        const f32x4 v = st[q];
        s1 += v[0] + v[2];
        s2 += v[1] + v[3];
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (r < M && l8 == 0) {
        const float kd = (float)d;
        const float mean = s1 / kd;
        const float var = fmaxf(s2 / kd - mean * mean, 0.f);
        *reinterpret_cast<float2*>(out + 2 * (size_t)r) = make_float2(mean, 1.0f / sqrtf(var + eps));
    }
}

// Row statistics for the LayerNorm-fused GEMM (gemm_bf16.h, ALN): stats[r][0] = (sum, sumsq) of row r, the other
// parts zero. Only needed once per forward (the embeddings); afterwards the residual GEMM epilogues produce them.
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                        uint16_t* __restrict__ xb, int M, int d, int parts) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const float* xr = x + (size_t)r * d;
    float s = 0.f, q = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        s += (v[0] + v[1]) + (v[2] + v[3]);
        q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        if (xb) {  // bf16 copy of the row: the A operand of the LayerNorm-folded GEMMs
            u32x2 pk;
            pk[0] = pack_bf16x2(v[0], v[1]);
            pk[1] = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<u32x2*>(xb + (size_t)r * d + c) = pk;
        }
    }
    s = wave_sum(s);
    q = wave_sum(q);
    float* o = stats + (size_t)r * parts * 2;
    for (int i = lane; i < parts * 2; i += 64) o[i] = (i == 0) ? s : (i == 1 ? q : 0.f);
}

// LayerNorm folding (gemm_bf16.h, epilogues 7/8): W'[n,k] = bf16(W[n,k] * gamma[k]), c[n] = sum_k W'[n,k],
// b'[n] = b[n] + sum_k beta[k] * W[n,k]. One wave per output feature n; run once when the weights are finalised.
__global__ __launch_bounds__(256) void fold_ln_weights_kernel(const uint16_t* __restrict__ Wb, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ bias,
                                                              uint16_t* __restrict__ Wf, float* __restrict__ cvec,
                                                              float* __restrict__ bf, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float c = 0.f, bs = 0.f;
    for (int kk = lane; kk < K; kk += 64) {
        const float w = bf16_bits_to_f32(Wb[(size_t)n * K + kk]);
        const bf16_t wf = (bf16_t)(w * gamma[kk]);
        const uint16_t bits = __builtin_bit_cast(uint16_t, wf);
        Wf[(size_t)n * K + kk] = bits;
        c += bf16_bits_to_f32(bits);
        bs += beta[kk] * w;
    }
    c = wave_sum(c);
    bs = wave_sum(bs);
    if (lane == 0) { cvec[n] = c; bf[n] = bias[n] + bs; }
}

// ------------------------------------------------------------------------------------------------
// K4: multi-head attention core, head_dim 64:  ctx = softmax(Q K^T * 64^-0.5 (+causal mask)) V
// (HF:modeling_clip.py:259-277,280-335; text mask :543-548). Softmax statistics in fp32 as HF does.
//
// One workgroup per (image/text b, head h), 4 waves, each wave owns 16-query tiles.
// Per wave and query tile everything stays in registers:
//   S^T = K Q^T   (MFMA A = K tile from LDS, B = Q fragment from global)  -> lane holds, for ITS query
//                 (lane & 15), keys 4*(lane>>4)+reg of every 16-key tile: the softmax reduction over
//                 keys is in-lane plus two xor-shuffles (16, 32).
//   O^T = V^T P^T (A = V^T via ds_read_b64_tr_b16 from a row-major V image, B = P^T straight from the
//                 S^T accumulator registers: no LDS round trip, no lane movement — guide §3
//                 "An accumulator tile as the next MFMA's operand", with the k order of both
//                 operands permuted the same way).
// NKP = padded key count / 32.
// ------------------------------------------------------------------------------------------------
#define ATT_VSTRIDE 144  // bytes per V row in LDS (128 + 16 pad: spreads the tr-read's 8 rows over banks)

// One 16-query tile of the one-pass form (T <= 128 keys: all score tiles live in registers). qf = the tile's Q fragments,
// q = this lane's query row, (b, h) only enter through `orow` = ctx row of q at head h, column 4*fg.
// MXOUT (round 6: the fp8 tower of ViT-B/32 — 50 keys — feeds its out-projection MXFP8 rows, as attention_long_kernel does for
// ViT-L/14): instead of bf16 at `orow`, the tile's rows leave as e4m3 at o8row (= ctx8 row of q at head h, column 4*fg) with one
// E8M0 scale per (query, 32 columns) at srow[mx_scale_offset(2 h + block)] (srow = the row's permuted scale bytes).
__device__ __forceinline__ float att_max_over_lane_groups(float v);
template <int NKP, bool CAUSAL, bool MXOUT = false>
__device__ __forceinline__ void attention_onepass_tile(const char* sK, const char* sV, const bf16x8 (&qf)[2], int q, int T,
                                                       int fr, int fg, uint16_t* orow, uint8_t* o8row = nullptr, uint8_t* srow = nullptr,
                                                       int h = 0) {
    f32x4 sacc[2 * NKP];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2 * NKP; ++kt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        const int krow = kt * 16 + fr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int chunk = 4 * s + fg;
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + krow * 128 + ((chunk ^ (krow & 7)) << 4));
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], a, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * fg + r;
            const bool ok = (key < T) && (!CAUSAL || key <= q);
            a[r] = ok ? a[r] * 0.125f : -INFINITY;
            mx = fmaxf(mx, a[r]);
        }
        sacc[kt] = a;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2 * NKP; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pexp = __expf(sacc[kt][r] - mx);
            sacc[kt][r] = pexp;
            l += pexp;
        }
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);

    f32x4 oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tq = fr >> 2, tp = fr & 3;  // tr-read address role inside the 16-lane group
#pragma unroll
    for (int ks = 0; ks < NKP; ++ks) {
        // B fragment: element j<4 = key 32ks + 4fg + j, j>=4 = key 32ks + 16 + 4fg + (j-4)
        u32x4 praw;
        praw[0] = pack_bf16x2(sacc[2 * ks][0], sacc[2 * ks][1]);
        praw[1] = pack_bf16x2(sacc[2 * ks][2], sacc[2 * ks][3]);
        praw[2] = pack_bf16x2(sacc[2 * ks + 1][0], sacc[2 * ks + 1][1]);
        praw[3] = pack_bf16x2(sacc[2 * ks + 1][2], sacc[2 * ks + 1][3]);
        const bf16x8 pf = __builtin_bit_cast(bf16x8, praw);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const char* a0 = sV + (32 * ks + 4 * fg + tq) * ATT_VSTRIDE + (dt * 16 + 4 * tp) * 2;
            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(a0));
            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(a0 + 16 * ATT_VSTRIDE));
            bf16x8 vf;
            vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
            vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
            oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
        }
    }
    const float inv = 1.0f / l;
    if constexpr (MXOUT) {
        // (every lane takes part in the block maxima — the four lane groups of a query hold its 64 columns; only valid queries store)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float o[2][4];
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[i][r] = oacc[2 * blk + i][r] * inv;
                    amax = fmaxf(amax, fabsf(o[i][r]));
                }
            amax = att_max_over_lane_groups(amax);
            int e8;
            float sinv;
            mx_scale_of(amax, e8, sinv);
            if (q < T) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    *reinterpret_cast<uint32_t*>(o8row + (2 * blk + i) * 16) = pack_fp8x4(o[i][0] * sinv, o[i][1] * sinv, o[i][2] * sinv, o[i][3] * sinv);
                if (fg == 0) srow[mx_scale_offset(2 * h + blk)] = (uint8_t)e8;
            }
        }
    } else if (q < T) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            u32x2 pk;
            pk[0] = pack_bf16x2(oacc[dt][0] * inv, oacc[dt][1] * inv);
            pk[1] = pack_bf16x2(oacc[dt][2] * inv, oacc[dt][3] * inv);
            *reinterpret_cast<u32x2*>(orow + dt * 16) = pk;
        }
    }
}

template <int NKP, bool CAUSAL, bool MXOUT = false>
__global__ __launch_bounds__(256) void attention_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ ctx, int T, int H,
                                                        uint8_t* __restrict__ ctx8 = nullptr, uint8_t* __restrict__ ctxs = nullptr, int ld_s = 0) {
    static_assert(NKP <= 4, "sequences over 128 keys: attention_long_kernel");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = NKP * 32;
    char* sK = smem;             // [TP][128 B], 16-B chunks XOR-swizzled by (row & 7)
    char* sV = smem + TP * 128;  // [TP][144 B] row-major
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int dmodel = H * 64, ld = 3 * dmodel;
    const uint16_t* base = qkv + (size_t)b * T * ld + h * 64;

    constexpr int NW = 4;
    const int fr = lane & 15, fg = lane >> 4;
    // The Q fragments of this wave's first query tile are fetched BEFORE the K/V image is staged: their global latency then
    // overlaps the staging loads instead of following the barrier (the kernel is a chain of dependent latencies, not
    // bandwidth: 3072 workgroups of ~20 KB each at T = 50).
    const int qt_first = wave + NW * blockIdx.y;
    bf16x8 qf_first[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4 raw = {0u, 0u, 0u, 0u};
        const int q = qt_first * 16 + fr;
        if (q < T) raw = *reinterpret_cast<const u32x4*>(base + (size_t)q * ld + s * 32 + fg * 8);
        qf_first[s] = __builtin_bit_cast(bf16x8, raw);
    }
    for (int idx = tid; idx < TP * 8; idx += NW * 64) {
        const int row = idx >> 3, c = idx & 7;
        u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
        if (row < T) {
            kv = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + dmodel + c * 8);
            vv = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + 2 * dmodel + c * 8);
        }
        *reinterpret_cast<u32x4*>(sK + row * 128 + ((c ^ (row & 7)) << 4)) = kv;
        *reinterpret_cast<u32x4*>(sV + row * ATT_VSTRIDE + (c << 4)) = vv;
    }
    __syncthreads();

    const int nqt = (T + 15) >> 4;
    // gridDim.y workgroups share one (b, h): small batches split the query tiles so that the grid still fills the chip
    for (int qt = qt_first; qt < nqt; qt += NW * gridDim.y) {
        const int q = qt * 16 + fr;
        bf16x8 qf[2];
        if (qt == qt_first) {
            qf[0] = qf_first[0];
            qf[1] = qf_first[1];
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4 raw = {0u, 0u, 0u, 0u};
                if (q < T) raw = *reinterpret_cast<const u32x4*>(base + (size_t)q * ld + s * 32 + fg * 8);
                qf[s] = __builtin_bit_cast(bf16x8, raw);
            }
        }
        if constexpr (MXOUT) {
            const size_t row = (size_t)b * T + (q < T ? q : 0);
            attention_onepass_tile<NKP, CAUSAL, true>(sK, sV, qf, q, T, fr, fg, nullptr, ctx8 + row * dmodel + h * 64 + 4 * fg, ctxs + row * ld_s, h);
        } else {
            attention_onepass_tile<NKP, CAUSAL>(sK, sV, qf, q, T, fr, fg, ctx + ((size_t)b * T + q) * dmodel + h * 64 + 4 * fg);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4, long sequences (ViT-L/14: 257 keys, LongCLIP text: 248): 8 waves share one staged K/V image (78 KB for 288 padded keys:
// two workgroups per CU). Holding all score tiles of a query tile costs 288 VGPRs, so the scores are walked 32 keys at a time.
// Round 4 (measured on the round-3 form, 128 images x 16 heads, 97-107 us per layer: its two phases ADD — 27 us of K/V
// staging with nothing computing, 78 us of query tiles with nothing loading — and the query tiles are bound by vector +
// matrix issue together, which the second pass over Q K^T that only found the row maxima fed for nothing):
//   * ONE pass, online softmax: the running offset m of a query is raised — and the output tile and the denominator rescaled
//     by exp2((m_old - m_new) c) — only when a new score exceeds it by more than 8 / c (the probabilities then stay below 2^8:
//     bf16 keeps its 8 bits at any magnitude, the sums are f32). After the first key tiles that is rare; the branch is
//     wave-uniform (any lane). The four lanes that hold one query's keys agree on the maximum through two half-wave /
//     16-lane-row swaps in the vector unit (v_permlane32_swap, v_permlane16_swap), no LDS round trip.
//   * the softmax denominator comes from the matrix cores: a row tile of ones beside V^T sums the bf16 probabilities — the
//     ones the PV product uses — into every register of lacc (no vector add per score, no shuffle at the end).
//   (NOT kept: a workgroup walking several (item, head) pairs with the next pair's K/V rows in flight in registers — 40 more
//   VGPRs at the 128 that four waves per SIMD allow: 90-95 us against 83 without, the query tiles alone 74 against 64.)
//   * raw scores: the 1/8 scale and log2(e) ride in the ONE fma in front of v_exp_f32; key-validity / causal masks only on
//     boundary tiles; causal key tiles above the diagonal skipped; every LDS address a per-lane constant + a tile multiple.
// MXOUT (the fp8 vision tower's out-projection on the block-scaled fp8 GEMM): INSTEAD of the bf16 rows the kernel writes the
// output as MXFP8 — e4m3 bytes ctx8 [B*T, H*64] and one E8M0 scale per (row, 32 columns) in the permuted layout of
// gemm_fp8.h (ctxs, ld_s bytes per row). A (query, head) holds two 32-column blocks (output rows dt 0,1 / 2,3 of O^T); a lane
// has 8 values of each, the block maximum is one lane-local maximum and two xor-shuffles over the four lane groups.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float att_max_over_lane_groups(float v) {   // max over lanes l, l ^ 16, l ^ 32, l ^ 48
    uint32_t u = __float_as_uint(v);
    auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = mm_max2(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
    u = __float_as_uint(v);
    auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return mm_max2(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
}

template <int NKP, bool CAUSAL, bool MXOUT>
__global__ __launch_bounds__(512) void attention_long_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ ctx, int T,
                                                             int H, uint8_t* __restrict__ ctx8, uint8_t* __restrict__ ctxs, int ld_s,
                                                             int dbg) {
    static_assert(NKP > 4 && NKP <= 9, "129..288 keys");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = NKP * 32;
    constexpr int NW = 8;
    char* sK = smem;             // [TP][128 B], 16-B chunks XOR-swizzled by (row & 7)
    char* sV = smem + TP * 128;  // [TP][144 B] row-major
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dmodel = H * 64, ld = 3 * dmodel;
    const int fr = lane & 15, fg = lane >> 4;
    // dbg: timing ablations of tools/attention_l14_ablate.py (1 = no K/V loads, 2 = no query tiles; results wrong by
    // construction) — compiled into EXPERIMENTS builds only, a product build ignores the argument
#ifdef MMISS_EXPERIMENTS
    const int dbg_e = dbg;
#else
    constexpr int dbg_e = 0;
#endif
    const int nqt = (dbg_e & 2) ? 0 : (T + 15) >> 4;
    const float c_exp = 0.125f * 1.4426950408889634f;
    const float thr_raw = 8.0f / c_exp;              // deferred rescale: the offset may lag the maximum by 8 binary orders
    // per-lane constants of the LDS addresses (krow & 7 = fr & 7: a key tile starts at a multiple of 16 rows)
    const char* kbase0 = sK + fr * 128 + ((fg ^ (fr & 7)) << 4);
    const char* kbase1 = sK + fr * 128 + (((4 + fg) ^ (fr & 7)) << 4);
    const int tq = fr >> 2, tp = fr & 3;             // tr-read address role inside the 16-lane group
    const char* vbase = sV + (4 * fg + tq) * ATT_VSTRIDE + 8 * tp;   // + ks * 32 rows + dt * 32 bytes (+ 16 rows)
    u32x4 ones_raw = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    asm volatile("" : "+v"(ones_raw));   // (opaque: kept in four registers instead of three v_mov per key-pair step)
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_raw);

    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const uint16_t* base = qkv + (size_t)b * T * ld + h * 64;
    for (int idx = tid; idx < TP * 8; idx += NW * 64) {
        const int row = idx >> 3, c = idx & 7;
        u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
        if (row < T && !(dbg_e & 1)) {
            kv = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + dmodel + c * 8);
            vv = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + 2 * dmodel + c * 8);
        }
        *reinterpret_cast<u32x4*>(sK + row * 128 + ((c ^ (row & 7)) << 4)) = kv;
        *reinterpret_cast<u32x4*>(sV + row * ATT_VSTRIDE + (c << 4)) = vv;
    }
    __syncthreads();
    {
        // gridDim.y workgroups share one (b, h): small batches split the query tiles so that the grid still fills the chip
        for (int qt = wave + NW * blockIdx.y; qt < nqt; qt += NW * gridDim.y) {
            const int q = qt * 16 + fr;
            bf16x8 qf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4 raw = {0u, 0u, 0u, 0u};
                if (q < T) raw = *reinterpret_cast<const u32x4*>(base + (size_t)q * ld + s * 32 + fg * 8);
                qf[s] = __builtin_bit_cast(bf16x8, raw);
            }
            const int kt_end = CAUSAL ? (qt + 1 < nqt ? qt + 1 : nqt) : nqt;  // key tiles this query tile needs (nqt = ceil(T/16))
            // tiles [0, kt_clean) hold only valid keys for every query of the tile: no mask
            const int kt_clean = CAUSAL ? (qt < (T >> 4) ? qt : (T >> 4)) : (T >> 4);
            auto score_tile = [&](int kt) -> f32x4 {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                const bf16x8 kf0 = *reinterpret_cast<const bf16x8*>(kbase0 + kt * 2048);
                const bf16x8 kf1 = *reinterpret_cast<const bf16x8*>(kbase1 + kt * 2048);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[1], a, 0, 0, 0);
                if (kt >= kt_clean) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 16 + 4 * fg + r;
                        const bool ok = (key < T) && (!CAUSAL || key <= q);
                        a[r] = ok ? a[r] : -INFINITY;
                    }
                }
                return a;
            };
            float m = -INFINITY;   // this query's offset (raw score units): exp2((s - m) c) is what enters P
            f32x4 lacc = {0.f, 0.f, 0.f, 0.f};
            f32x4 oacc[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // one step = two key tiles (the 32 keys of one PV MFMA); an odd last tile is a step of its own (second tile -inf)
            auto step = [&](int ks, auto pair_tag) {
                constexpr bool PAIR = decltype(pair_tag)::value;
                f32x4 p0 = score_tile(2 * ks), p1;
                if constexpr (PAIR) p1 = score_tile(2 * ks + 1);
                else p1 = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                if constexpr (PAIR) mm_mfma_settle("+v"(p0), "+v"(p1));   // (the asm maxima below read MFMA results: common.h)
                else mm_mfma_settle("+v"(p0));
                float lm = mm_max3(p0[0], p0[1], p0[2]);
                if constexpr (PAIR) lm = mm_max3(mm_max3(lm, p0[3], p1[0]), p1[1], mm_max2(p1[2], p1[3]));
                else lm = mm_max2(lm, p0[3]);
                lm = att_max_over_lane_groups(lm);   // the same value in the four lanes of a query
                if (__any(lm > m + thr_raw)) {       // (m = -inf at the first step: taken, alpha = 0 on zeros)
                    const float mn = (lm > m + thr_raw) ? lm : m;
                    const float alpha = __builtin_amdgcn_exp2f((m - mn) * c_exp);   // 1 where the offset stays; exp2(-inf) = 0
                    m = mn;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) oacc[dt][r] *= alpha;
#pragma unroll
                    for (int r = 0; r < 4; ++r) lacc[r] *= alpha;
                }
                const float mc = m * c_exp;
#pragma unroll
                for (int r = 0; r < 4; ++r) p0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(p0[r], c_exp, -mc));
#pragma unroll
                for (int r = 0; r < 4; ++r) p1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(p1[r], c_exp, -mc));
                // B fragment of O^T = V^T P^T: element j < 4 = key 32 ks + 4 fg + j, j >= 4 = key 32 ks + 16 + 4 fg + (j - 4)
                u32x4 praw;
                praw[0] = pack_bf16x2(p0[0], p0[1]);
                praw[1] = pack_bf16x2(p0[2], p0[3]);
                praw[2] = pack_bf16x2(p1[0], p1[1]);
                praw[3] = pack_bf16x2(p1[2], p1[3]);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, praw);
                lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, lacc, 0, 0, 0);
                const char* vks = vbase + ks * (32 * ATT_VSTRIDE);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const char* a0 = vks + dt * 32;
                    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(a0));
                    const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(a0 + 16 * ATT_VSTRIDE));
                    bf16x8 vf;
                    vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                    vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
                }
            };
            const int npairs = kt_end >> 1;
#pragma unroll 1
            for (int ks = 0; ks < npairs; ++ks) step(ks, std::true_type{});
            if (kt_end & 1) step(npairs, std::false_type{});
            const float inv = 1.0f / lacc[0];
            if constexpr (MXOUT) {
                const size_t row = (size_t)b * T + (q < T ? q : 0);
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    float o[2][4];
                    float amax = 0.f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            o[i][r] = oacc[2 * blk + i][r] * inv;
                            amax = fmaxf(amax, fabsf(o[i][r]));
                        }
                    amax = att_max_over_lane_groups(amax);
                    int e8;
                    float sinv;
                    mx_scale_of(amax, e8, sinv);
                    if (q < T) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            *reinterpret_cast<uint32_t*>(ctx8 + row * dmodel + h * 64 + (2 * blk + i) * 16 + 4 * fg) =
                                pack_fp8x4(o[i][0] * sinv, o[i][1] * sinv, o[i][2] * sinv, o[i][3] * sinv);
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
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4, short sequences at large batch: one workgroup walks HPB heads of one item. With one (item, head) per workgroup the
// kernel is a chain of dependent latencies (PMC at B = 256, T = 50: 65 % of the wave cycles in s_waitcnt, matrix cores 9 %):
// Q/K/V global loads, LDS writes, barrier, 1 us of arithmetic, store - 3072 workgroups of 20 KB each. Here the K/V image of
// head h+1 (and its Q fragments) is in flight in registers while head h is computed from LDS (two LDS images, one barrier
// per head), so a workgroup pays the load latency once instead of HPB times. Same arithmetic, bit-identical output.
// ------------------------------------------------------------------------------------------------
template <int NKP, bool CAUSAL, int HPB, bool MXOUT = false>
__global__ __launch_bounds__(256) void attention_heads_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ ctx,
                                                              int T, int H, uint8_t* __restrict__ ctx8 = nullptr,
                                                              uint8_t* __restrict__ ctxs = nullptr, int ld_s = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = NKP * 32;
    constexpr int IMG = TP * (128 + ATT_VSTRIDE);  // one K + V image
    constexpr int NIT = TP * 8 / 256;              // 16-byte chunks per thread and operand (TP % 32 == 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int groups = H / HPB;
    const int b = blockIdx.x / groups, h0 = (blockIdx.x - b * groups) * HPB;
    const int dmodel = H * 64, ld = 3 * dmodel;
    const uint16_t* item = qkv + (size_t)b * T * ld;
    const int nqt = (T + 15) >> 4;

    u32x4 kr[NIT], vr[NIT];
    auto load_head = [&](int h) {
        const uint16_t* base = item + h * 64;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, c = idx & 7;
            kr[i] = u32x4{0u, 0u, 0u, 0u};
            vr[i] = u32x4{0u, 0u, 0u, 0u};
            if (row < T) {
                kr[i] = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + dmodel + c * 8);
                vr[i] = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + 2 * dmodel + c * 8);
            }
        }
    };
    auto store_head = [&](int buf) {
        char* sK = smem + buf * IMG;
        char* sV = sK + TP * 128;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, c = idx & 7;
            *reinterpret_cast<u32x4*>(sK + row * 128 + ((c ^ (row & 7)) << 4)) = kr[i];
            *reinterpret_cast<u32x4*>(sV + row * ATT_VSTRIDE + (c << 4)) = vr[i];
        }
    };
    auto load_q = [&](int h, int qt, bf16x8 (&qf)[2]) {
        const int q = qt * 16 + fr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (q < T) raw = *reinterpret_cast<const u32x4*>(item + h * 64 + (size_t)q * ld + s * 32 + fg * 8);
            qf[s] = __builtin_bit_cast(bf16x8, raw);
        }
    };

    bf16x8 qf[2], qn[2];
    load_q(h0, wave, qf);
    load_head(h0);
    store_head(0);
    __syncthreads();
#pragma unroll 1
    for (int hh = 0; hh < HPB; ++hh) {
        const int h = h0 + hh, cur = hh & 1;
        if (hh + 1 < HPB) {  // next head's operands fly during this head's arithmetic
            load_q(h + 1, wave, qn);
            load_head(h + 1);
        }
        const char* sK = smem + cur * IMG;
        const char* sV = sK + TP * 128;
        for (int qt = wave; qt < nqt; qt += 4) {
            const int q = qt * 16 + fr;
            if (qt != wave) load_q(h, qt, qf);  // (T > 64: a wave's second tile)
            if constexpr (MXOUT) {
                const size_t row = (size_t)b * T + (q < T ? q : 0);
                attention_onepass_tile<NKP, CAUSAL, true>(sK, sV, qf, q, T, fr, fg, nullptr, ctx8 + row * dmodel + h * 64 + 4 * fg, ctxs + row * ld_s, h);
            } else
            attention_onepass_tile<NKP, CAUSAL>(sK, sV, qf, q, T, fr, fg, ctx + ((size_t)b * T + q) * dmodel + h * 64 + 4 * fg);
        }
        if (hh + 1 < HPB) {
            store_head(cur ^ 1);  // image cur^1 was last read before the previous barrier
            qf[0] = qn[0];
            qf[1] = qn[1];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K8 tail / a7: out[r] = y[r] / ||y[r]||_2, fp32, no epsilon (backend/app/utils.py:78,98).
// One wave per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ y, float* __restrict__ out, int B,
                                                          int D, int ldy) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B) return;
    const float* yr = y + (size_t)r * ldy;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += yr[c] * yr[c];
    const float nrm = sqrtf(wave_sum(s));
    for (int c = lane; c < D; c += 64) out[(size_t)r * D + c] = yr[c] / nrm;
}

// f32 -> bf16 conversion (weight upload)
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        bf16_t v = (bf16_t)src[i];
        dst[i] = __builtin_bit_cast(uint16_t, v);
    }
}
// bf16 -> f32 widening (debug taps)
__global__ void bf16_to_f32_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = bf16_bits_to_f32(src[i]);
}

// ------------------------------------------------------------------------------------------------ launchers
static int launch_layernorm(hipStream_t st, const float* x, const float* g, const float* b, void* out, bool out_bf16,
                            const int32_t* rowmap, int M, int d, float eps) {
    if (d % 4 || d > 1024 || d <= 0) MM_FAIL(MMISS_ERR_UNSUPPORTED, "layernorm: d=%d (need d%%4==0, d<=1024)", d);
    if (M <= 0) return MMISS_OK;
    MM_PROF("layernorm", st, 8.0 * M * d, (double)M * d * (4 + (out_bf16 ? 2 : 4)));
    const int grid = (M + 3) / 4;
    if (out_bf16)
        hipLaunchKernelGGL(layernorm_kernel<true>, dim3(grid), dim3(256), 0, st, x, g, b, out, rowmap, M, d, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel<false>, dim3(grid), dim3(256), 0, st, x, g, b, out, rowmap, M, d, eps);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

// LayerNorm of a bf16 residual stream -> bf16 GEMM operand
static int launch_layernorm16(hipStream_t st, const uint16_t* x16, const float* g, const float* b, void* out_bf16, int M, int d,
                              float eps) {
    if (d % 8 || d > 1024 || d <= 0) MM_FAIL(MMISS_ERR_UNSUPPORTED, "layernorm16: d=%d (need d%%8==0, d<=1024)", d);
    if (M <= 0) return MMISS_OK;
    MM_PROF("layernorm16", st, 8.0 * M * d, (double)M * d * 4);
    hipLaunchKernelGGL(layernorm16_kernel, dim3((M + 7) / 8), dim3(256), 0, st, x16, g, b, reinterpret_cast<uint16_t*>(out_bf16),
                       M, d, eps);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

static int launch_im2col(hipStream_t st, const void* pixels, bool src_u8, void* out, int B, int S, int P, int Kp) {
    if (P <= 0 || S % P || Kp % 8 || Kp < 3 * P * P) MM_FAIL(MMISS_ERR_ARG, "im2col: S=%d P=%d Kp=%d", S, P, Kp);
    if (B <= 0) return MMISS_OK;
    const int G = S / P;
    const int64_t total = (int64_t)B * G * G * (Kp / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    MM_PROF("im2col", st, 0.0, (double)B * 3 * S * S * (src_u8 ? 1 : 4) + (double)B * G * G * Kp * 2);
    if (src_u8)
        hipLaunchKernelGGL(im2col_kernel<true>, dim3(grid), dim3(256), 0, st, pixels, (uint16_t*)out, B, S, P, Kp);
    else
        hipLaunchKernelGGL(im2col_kernel<false>, dim3(grid), dim3(256), 0, st, pixels, (uint16_t*)out, B, S, P, Kp);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

template <int NKP, bool CAUSAL, bool MXOUT>
static int launch_attention_long_t(hipStream_t st, const void* qkv, void* ctx, uint8_t* ctx8, uint8_t* ctxs, int ld_s, int B, int T,
                                   int H) {
    const int lds = NKP * 32 * (128 + ATT_VSTRIDE);
    const int items = B * H;
    // query-tile splits per (b, h): 1 once B*H covers the 256 CUs (one ViT-L/14 image: 16 heads x 17 query tiles ->
    // 3 splits = 48 workgroups instead of 16 walking 3 rounds each)
    const int rounds = ((T + 15) / 16 + 7) / 8;
    int qs = 256 / items;
    qs = qs < 1 ? 1 : (qs > rounds ? rounds : qs);
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&attention_long_kernel<NKP, CAUSAL, MXOUT>), lds));
    hipLaunchKernelGGL((attention_long_kernel<NKP, CAUSAL, MXOUT>), dim3(items, qs), dim3(512), lds, st, (const uint16_t*)qkv,
                       (uint16_t*)ctx, T, H, ctx8, ctxs, ld_s,
#ifdef MMISS_EXPERIMENTS
                       mmiss_option("att_dbg", 0));
#else
                       0);
#endif
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

template <bool CAUSAL, bool MXOUT>
static int launch_attention_long(hipStream_t st, const void* qkv, void* ctx, uint8_t* ctx8, uint8_t* ctxs, int ld_s, int B, int T, int H) {
    // round 5: the ViT-L/14 regime on the persistent streaming kernel (attention_stream.h); option attention_stream = 0: this kernel
    if constexpr (!CAUSAL)
        if (attention_stream_ok(B, T, H, false) && mmiss_option("attention_stream", 1))
            return launch_attention_stream<MXOUT>(st, qkv, ctx, ctx8, ctxs, ld_s, B, H);
    switch ((T + 31) / 32) {
        case 5: return launch_attention_long_t<5, CAUSAL, MXOUT>(st, qkv, ctx, ctx8, ctxs, ld_s, B, T, H);
        case 6: return launch_attention_long_t<6, CAUSAL, MXOUT>(st, qkv, ctx, ctx8, ctxs, ld_s, B, T, H);
        case 7: return launch_attention_long_t<7, CAUSAL, MXOUT>(st, qkv, ctx, ctx8, ctxs, ld_s, B, T, H);
        case 8: return launch_attention_long_t<8, CAUSAL, MXOUT>(st, qkv, ctx, ctx8, ctxs, ld_s, B, T, H);
        default: return launch_attention_long_t<9, CAUSAL, MXOUT>(st, qkv, ctx, ctx8, ctxs, ld_s, B, T, H);
    }
}

// attention writing MXFP8: non-causal only (the vision tower), 1..288 keys (round 6: the one-pass kernels too — ViT-B/32's 50 keys)
static bool attention_mx_ok(int T, int H) { return T > 0 && T <= 288 && H > 0; }
static int launch_attention_mx_short(hipStream_t st, const void* qkv, uint8_t* ctx8, uint8_t* ctxs, int ld_s, int B, int T, int H);
static int launch_attention_mx(hipStream_t st, const void* qkv, uint8_t* ctx8, uint8_t* ctxs, int ld_s, int B, int T, int H) {
    if (B <= 0) return MMISS_OK;
    if (!attention_mx_ok(T, H) || !ctx8 || !ctxs || ld_s < mx_scale_row_bytes(H * 64))
        MM_FAIL(MMISS_ERR_UNSUPPORTED, "attention (MXFP8 output): T=%d (1..288), H=%d", T, H);
    MM_PROF("attention_mx", st, 4.0 * B * H * (double)T * T * 64, (double)B * T * H * 64 * (2 * 3 + 1));
    if (T <= 128) return launch_attention_mx_short(st, qkv, ctx8, ctxs, ld_s, B, T, H);
    return launch_attention_long<false, true>(st, qkv, nullptr, ctx8, ctxs, ld_s, B, T, H);
}

template <int NKP>
static int launch_attention_nkp(hipStream_t st, const void* qkv, void* ctx, int B, int T, int H, bool causal) {
    const int lds = NKP * 32 * (128 + ATT_VSTRIDE);
    // query-tile splits per (b, h): 1 once B*H covers the 256 CUs
    const int rounds = ((T + 15) / 16 + 3) / 4;
    int qs = 256 / (B * H);
    qs = qs < 1 ? 1 : (qs > rounds ? rounds : qs);
    const dim3 grid(B * H, qs);
    if (causal) {
        MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&attention_kernel<NKP, true>), lds));
        hipLaunchKernelGGL((attention_kernel<NKP, true>), grid, dim3(256), lds, st, (const uint16_t*)qkv, (uint16_t*)ctx, T, H);
    } else {
        MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&attention_kernel<NKP, false>), lds));
        hipLaunchKernelGGL((attention_kernel<NKP, false>), grid, dim3(256), lds, st, (const uint16_t*)qkv, (uint16_t*)ctx, T, H);
    }
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

template <int NKP, int HPB>
static int launch_attention_heads(hipStream_t st, const void* qkv, void* ctx, int B, int T, int H, bool causal) {
    const int lds = 2 * NKP * 32 * (128 + ATT_VSTRIDE);
    const dim3 grid(B * (H / HPB));
    if (causal) {
        MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&attention_heads_kernel<NKP, true, HPB>), lds));
        hipLaunchKernelGGL((attention_heads_kernel<NKP, true, HPB>), grid, dim3(256), lds, st, (const uint16_t*)qkv,
                           (uint16_t*)ctx, T, H);
    } else {
        MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&attention_heads_kernel<NKP, false, HPB>), lds));
        hipLaunchKernelGGL((attention_heads_kernel<NKP, false, HPB>), grid, dim3(256), lds, st, (const uint16_t*)qkv,
                           (uint16_t*)ctx, T, H);
    }
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}

template <int NKP>
static int launch_attention_heads_hpb(hipStream_t st, int hpb, const void* qkv, void* ctx, int B, int T, int H, bool causal) {
    switch (hpb) {
        case 2: return launch_attention_heads<NKP, 2>(st, qkv, ctx, B, T, H, causal);
        case 3: return launch_attention_heads<NKP, 3>(st, qkv, ctx, B, T, H, causal);
        case 4: return launch_attention_heads<NKP, 4>(st, qkv, ctx, B, T, H, causal);
        default: return launch_attention_heads<NKP, 6>(st, qkv, ctx, B, T, H, causal);
    }
}

// T <= 128 with MXFP8 output (non-causal): the same choice between several heads per workgroup and one as launch_attention
template <int NKP>
static int launch_attention_mx_nkp(hipStream_t st, int hpb, const void* qkv, uint8_t* ctx8, uint8_t* ctxs, int ld_s, int B, int T, int H) {
    auto heads = [&](auto hpb_tag) -> int {
        constexpr int HPB = decltype(hpb_tag)::value;
        const int lds = 2 * NKP * 32 * (128 + ATT_VSTRIDE);
        MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&attention_heads_kernel<NKP, false, HPB, true>), lds));
        hipLaunchKernelGGL((attention_heads_kernel<NKP, false, HPB, true>), dim3(B * (H / HPB)), dim3(256), lds, st, (const uint16_t*)qkv,
                           (uint16_t*)nullptr, T, H, ctx8, ctxs, ld_s);
        MM_HIP(hipGetLastError());
        return MMISS_OK;
    };
    switch (hpb) {
        case 2: return heads(std::integral_constant<int, 2>{});
        case 3: return heads(std::integral_constant<int, 3>{});
        case 4: return heads(std::integral_constant<int, 4>{});
        case 6: return heads(std::integral_constant<int, 6>{});
    }
    const int lds = NKP * 32 * (128 + ATT_VSTRIDE);
    const int rounds = ((T + 15) / 16 + 3) / 4;
    int qs = 256 / (B * H);
    qs = qs < 1 ? 1 : (qs > rounds ? rounds : qs);
    MM_TRY(mmiss_ensure_dyn_lds(reinterpret_cast<const void*>(&attention_kernel<NKP, false, true>), lds));
    hipLaunchKernelGGL((attention_kernel<NKP, false, true>), dim3(B * H, qs), dim3(256), lds, st, (const uint16_t*)qkv, (uint16_t*)nullptr, T, H,
                       ctx8, ctxs, ld_s);
    MM_HIP(hipGetLastError());
    return MMISS_OK;
}
static int launch_attention_mx_short(hipStream_t st, const void* qkv, uint8_t* ctx8, uint8_t* ctxs, int ld_s, int B, int T, int H) {
    int hpb = mmiss_option("att_hpb", 0);
    if (hpb == 0) {
        hpb = 1;
        for (int c : {4, 6, 3, 2})
            if (H % c == 0 && (int64_t)B * (H / c) >= 512) { hpb = c; break; }
    }
    if (!((hpb == 2 || hpb == 3 || hpb == 4 || hpb == 6) && H % hpb == 0)) hpb = 1;
    switch ((T + 31) / 32) {
        case 1: return launch_attention_mx_nkp<1>(st, hpb, qkv, ctx8, ctxs, ld_s, B, T, H);
        case 2: return launch_attention_mx_nkp<2>(st, hpb, qkv, ctx8, ctxs, ld_s, B, T, H);
        case 3: return launch_attention_mx_nkp<3>(st, hpb, qkv, ctx8, ctxs, ld_s, B, T, H);
        default: return launch_attention_mx_nkp<4>(st, hpb, qkv, ctx8, ctxs, ld_s, B, T, H);
    }
}

static int launch_attention(hipStream_t st, const void* qkv, void* ctx, int B, int T, int H, bool causal) {
    if (B <= 0) return MMISS_OK;
    if (T <= 0 || T > 288 || H <= 0) MM_FAIL(MMISS_ERR_UNSUPPORTED, "attention: T=%d (1..288), H=%d", T, H);
    const int nkp = (T + 31) / 32;
    // algorithmic flops: QK^T and PV, unpadded, full (non-causal) count as SURVEY.md §8(d) does
    MM_PROF("attention", st, 4.0 * B * H * (double)T * T * 64, (double)B * T * H * 64 * 2 * 4);
    // short sequences, large batch: several heads per workgroup (attention_heads_kernel) while >= 512 workgroups remain.
    // Option att_hpb: 0 = automatic, 1 = never, 2/3/4/6 = forced (if it divides H)
    if (nkp <= 4) {
        int hpb = mmiss_option("att_hpb", 0);
        if (hpb == 0) {
            hpb = 1;
            for (int c : {4, 6, 3, 2})  // B = 256, T = 50, H = 12: 1 head 19.3 us, 2: 17.5, 3: 17.2, 4: 16.2, 6: 16.8 (4.9 TB/s)
                if (H % c == 0 && (int64_t)B * (H / c) >= 512) { hpb = c; break; }
        }
        if ((hpb == 2 || hpb == 3 || hpb == 4 || hpb == 6) && H % hpb == 0) {
            switch (nkp) {
                case 1: return launch_attention_heads_hpb<1>(st, hpb, qkv, ctx, B, T, H, causal);
                case 2: return launch_attention_heads_hpb<2>(st, hpb, qkv, ctx, B, T, H, causal);
                case 3: return launch_attention_heads_hpb<3>(st, hpb, qkv, ctx, B, T, H, causal);
                default: return launch_attention_heads_hpb<4>(st, hpb, qkv, ctx, B, T, H, causal);
            }
        }
    }
    switch (nkp) {
        case 1: return launch_attention_nkp<1>(st, qkv, ctx, B, T, H, causal);
        case 2: return launch_attention_nkp<2>(st, qkv, ctx, B, T, H, causal);
        case 3: return launch_attention_nkp<3>(st, qkv, ctx, B, T, H, causal);
        case 4: return launch_attention_nkp<4>(st, qkv, ctx, B, T, H, causal);
    }
    return causal ? launch_attention_long<true, false>(st, qkv, ctx, nullptr, nullptr, 0, B, T, H)
                  : launch_attention_long<false, false>(st, qkv, ctx, nullptr, nullptr, 0, B, T, H);
}

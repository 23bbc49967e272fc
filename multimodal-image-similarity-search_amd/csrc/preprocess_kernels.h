// SURVEY §8(f) N2 — the resize step of CLIPImageProcessor on the GPU, bit-identical to Pillow.
//
// Reference: backend/app/utils.py:76 calls CLIPProcessor(images=...), whose image half
// (HF:image_processing_clip.py:23-34) resizes the SHORTEST edge to S with PIL's BICUBIC filter and centre-crops
// S x S; rescale + normalise are fused into the patchify kernel (encoder_kernels.h, im2col_kernel<true>).
// Pillow's 8-bit resample (src/libImaging/Resample.c, restated and pinned in oracle/resize_oracle.py):
//   per axis  scale = in/out, filterscale = max(scale,1), support = 2*filterscale, taps [xmin, xmin+cnt) around
//             center = (xx+0.5)*scale, weights bicubic((x+xmin-center+0.5)/filterscale) normalised by their sum
//             (IEEE double), rounded half-away-from-zero to 22-bit fixed point;
//   a pass    out = clip8((2^21 + sum pixel*k) >> 22); horizontal pass first, its result ROUNDED TO UINT8, then
//             the vertical pass.
// Two kernels:
//   resize_coeffs_kernel   one thread per (image, axis, output index inside the crop window): the fixed-point taps,
//                          computed on the device in double with contraction off (same operation order as Pillow);
//   resize_crop_kernel     one block per (16 output rows, 256 output columns, image): streams the source rows its
//                          rows need; a thread computes the horizontally resampled uint8 RGB of its column for the
//                          row and feeds it to the 16 x 3 vertical accumulators it owns, so the intermediate image
//                          never exists in memory. Only the crop window is computed.
// Integer work bound by byte loads of the source (each source row is read once per 16-row tile it contributes to);
// at the sizes of the reference's uploads (<= a few MP) the whole batch costs a fraction of one encoder layer.
#pragma once
#include "common.h"

struct ResizeDesc {
    int64_t src_off;        // byte offset of the image (tightly packed RGB8, row stride 3*W) inside the source blob
    int32_t H, W;           // source size
    int32_t new_h, new_w;   // resized size (shortest edge = S)
    int32_t top, left;      // centre-crop offsets inside the resized image
    int32_t ksx, ksy;       // taps reserved per output index (Pillow's ksize) on x / y
    int64_t kx_off, ky_off; // int32 offsets into the coefficient pool: kx[tap][S] (tap-major), ky[S][ksy]
};

#define MMISS_RESIZE_PRECISION_BITS 22
#define MMISS_RESIZE_ROWS 16

__device__ __forceinline__ double pil_bicubic(double x) {
#pragma clang fp contract(off)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// grid (B, 2): axis 0 = x (taps over source columns), 1 = y. bounds: int32 [B][4][S] = xmin, xcnt, ymin, ycnt.
__global__ void resize_coeffs_kernel(const ResizeDesc* __restrict__ desc, int32_t* __restrict__ pool,
                                     int32_t* __restrict__ bounds, int S) {
#pragma clang fp contract(off)
    const ResizeDesc d = desc[blockIdx.x];
    const int axis = blockIdx.y;
    const int in_size = axis ? d.H : d.W, out_size = axis ? d.new_h : d.new_w, first = axis ? d.top : d.left;
    const int ks = axis ? d.ksy : d.ksx;
    int32_t* bnd = bounds + ((size_t)blockIdx.x * 4 + axis * 2) * S;
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        const int xx = first + i;
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) ww += pil_bicubic((x + xmin - center + 0.5) * ss);
        for (int x = 0; x < ks; ++x) {
            int32_t k = 0;
            if (x < xmax) {
                double w = pil_bicubic((x + xmin - center + 0.5) * ss);
                if (ww != 0.0) w /= ww;
                k = (w < 0) ? (int32_t)(-0.5 + w * (double)(1 << MMISS_RESIZE_PRECISION_BITS))
                            : (int32_t)(0.5 + w * (double)(1 << MMISS_RESIZE_PRECISION_BITS));
            }
            if (axis == 0) pool[d.kx_off + (int64_t)x * S + i] = k;
            else           pool[d.ky_off + (int64_t)i * ks + x] = k;
        }
        bnd[i] = xmin;
        bnd[S + i] = xmax;
    }
}

__device__ __forceinline__ int clip8_fixed(int32_t v) {
    v >>= MMISS_RESIZE_PRECISION_BITS;  // arithmetic shift, as Pillow's clip8
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// grid (ceil(S/16), ceil(S/256), B), block 256: a thread owns one output COLUMN (3 channels) of the tile's 16 rows.
// The kernel is bound by load latency and load-instruction count, not bandwidth, so:
//   * KMAX > 0 (every image of the launch has ksx <= KMAX): the column's taps live in registers, the tap loop is fully
//     unrolled and each tap is ONE unaligned 4-byte load holding R,G,B (scalar row base + 32-bit per-thread offset;
//     the last row of an image that ends the blob uses byte loads, so nothing is read past the blob); two source rows
//     per trip keep 2*KMAX loads in flight. Taps past xcnt re-read the last valid pixel with weight 0. Needs images
//     under 2 GiB (32-bit in-row offsets are always enough; the row base is 64-bit).
//   * KMAX == 0: generic (any ksx): taps streamed from the pool, three byte loads per tap.
//   * the vertical taps of the tile are expanded once per 64 source rows into a dense LDS table kyt[row][16]
//     (zero outside a row's window), so the vertical pass is 4 ds_read_b128 + 48 MADs per source row, no branches.
#define MMISS_RESIZE_CHUNK 64
template <int KMAX>
__global__ __launch_bounds__(256) void resize_crop_kernel(const uint8_t* __restrict__ src, int64_t blob_bytes,
                                                          const ResizeDesc* __restrict__ desc,
                                                          const int32_t* __restrict__ pool,
                                                          const int32_t* __restrict__ bounds,
                                                          uint8_t* __restrict__ dst, int S) {
    constexpr int RT = MMISS_RESIZE_ROWS, CH = MMISS_RESIZE_CHUNK;
    constexpr int32_t HALF = 1 << (MMISS_RESIZE_PRECISION_BITS - 1);
    __shared__ __attribute__((aligned(16))) int32_t kyt[CH][RT];
    const int b = blockIdx.z;
    const ResizeDesc d = desc[b];
    const int r0 = blockIdx.x * RT;
    const int col_raw = blockIdx.y * 256 + threadIdx.x;
    const bool live = col_raw < S;
    const int col = live ? col_raw : 0;
    const int32_t* bnd = bounds + (size_t)b * 4 * S;
    const int xmin = bnd[col], xcnt = bnd[S + col];
    const int rl = (r0 + RT < S ? r0 + RT : S) - 1;
    const int y0 = bnd[2 * S + r0], y1 = bnd[2 * S + rl] + bnd[3 * S + rl];
    const int32_t* kx = pool + d.kx_off + col;
    const int32_t* ky = pool + d.ky_off;
    const uint8_t* img = src + d.src_off;            // block-uniform
    const int64_t row_stride = (int64_t)d.W * 3;
    // a 4-byte load at the image's very last pixel would run one byte past the blob when the image ends the blob
    const bool tail_risk = d.src_off + (int64_t)d.H * row_stride + 1 > blob_bytes;
    int32_t acc[RT][3];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r][0] = acc[r][1] = acc[r][2] = HALF;

    int32_t kreg[KMAX > 0 ? KMAX : 1];
    uint32_t off[KMAX > 0 ? KMAX : 1];  // byte offset of tap x inside a source row
    if constexpr (KMAX > 0) {
        const int last = (xcnt - 1) * 3;
#pragma unroll
        for (int x = 0; x < KMAX; ++x) {
            kreg[x] = x < xcnt ? kx[(int64_t)x * S] : 0;
            off[x] = (uint32_t)(xmin * 3 + (x * 3 < last ? x * 3 : last));
        }
    }

    auto hpass = [&](int y, int32_t (&h)[3]) {
        const uint8_t* row = img + y * row_stride;    // block-uniform: scalar base + per-thread 32-bit offsets
        int32_t s0 = HALF, s1 = HALF, s2 = HALF;
        if constexpr (KMAX > 0) {
            uint32_t px[KMAX];
            if (tail_risk && y == d.H - 1) {
#pragma unroll
                for (int x = 0; x < KMAX; ++x)
                    px[x] = (uint32_t)row[off[x]] | ((uint32_t)row[off[x] + 1] << 8) | ((uint32_t)row[off[x] + 2] << 16);
            } else {
#pragma unroll
                for (int x = 0; x < KMAX; ++x) __builtin_memcpy(&px[x], row + off[x], 4);
            }
#pragma unroll
            for (int x = 0; x < KMAX; ++x) {
                s0 += (int32_t)(px[x] & 0xff) * kreg[x];
                s1 += (int32_t)((px[x] >> 8) & 0xff) * kreg[x];
                s2 += (int32_t)((px[x] >> 16) & 0xff) * kreg[x];
            }
        } else {
            const uint8_t* p = row + (int64_t)xmin * 3;
            for (int x = 0; x < xcnt; ++x) {
                const int32_t k = kx[(int64_t)x * S];
                s0 += (int32_t)p[x * 3] * k;
                s1 += (int32_t)p[x * 3 + 1] * k;
                s2 += (int32_t)p[x * 3 + 2] * k;
            }
        }
        h[0] = clip8_fixed(s0); h[1] = clip8_fixed(s1); h[2] = clip8_fixed(s2);  // Pillow's uint8 intermediate image
    };
    auto vpass = [&](int yy, const int32_t (&h)[3]) {
        const int4* kr = reinterpret_cast<const int4*>(&kyt[yy][0]);
#pragma unroll
        for (int q = 0; q < RT / 4; ++q) {
            const int4 k = kr[q];
            const int32_t kk[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[q * 4 + e][0] += h[0] * kk[e];
                acc[q * 4 + e][1] += h[1] * kk[e];
                acc[q * 4 + e][2] += h[2] * kk[e];
            }
        }
    };

#pragma nounroll
    for (int c0 = y0; c0 < y1; c0 += CH) {
        __syncthreads();
        for (int i = threadIdx.x; i < CH * RT; i += 256) {
            const int yy = i / RT, r = i - yy * RT;
            int32_t k = 0;
            if (r0 + r < S) {
                const int t = c0 + yy - bnd[2 * S + r0 + r];
                if ((unsigned)t < (unsigned)bnd[3 * S + r0 + r]) k = ky[(int64_t)(r0 + r) * d.ksy + t];
            }
            kyt[yy][r] = k;
        }
        __syncthreads();
        const int n = (y1 - c0 < CH) ? y1 - c0 : CH;
        int yy = 0;
#pragma nounroll
        for (; yy + 1 < n; yy += 2) {
            int32_t h0[3], h1[3];
            hpass(c0 + yy, h0);
            hpass(c0 + yy + 1, h1);
            vpass(yy, h0);
            vpass(yy + 1, h1);
        }
        if (yy < n) {
            int32_t h0[3];
            hpass(c0 + yy, h0);
            vpass(yy, h0);
        }
    }
    if (!live) return;
#pragma unroll
    for (int r = 0; r < RT; ++r)
        if (r0 + r < S) {
            uint8_t* o = dst + (((int64_t)b * S + r0 + r) * S + col) * 3;
            o[0] = (uint8_t)clip8_fixed(acc[r][0]);
            o[1] = (uint8_t)clip8_fixed(acc[r][1]);
            o[2] = (uint8_t)clip8_fixed(acc[r][2]);
        }
}

static inline void launch_resize_crop(hipStream_t st, int max_ksx, const uint8_t* src, int64_t blob_bytes,
                                      const ResizeDesc* desc, const int32_t* pool, const int32_t* bounds, uint8_t* dst,
                                      int S, int nb) {
    const dim3 grid((S + MMISS_RESIZE_ROWS - 1) / MMISS_RESIZE_ROWS, (S + 255) / 256, nb), block(256);
    if (blob_bytes >= 4 && max_ksx <= 12)
        hipLaunchKernelGGL(resize_crop_kernel<12>, grid, block, 0, st, src, blob_bytes, desc, pool, bounds, dst, S);
    else if (blob_bytes >= 4 && max_ksx <= 24)
        hipLaunchKernelGGL(resize_crop_kernel<24>, grid, block, 0, st, src, blob_bytes, desc, pool, bounds, dst, S);
    else
        hipLaunchKernelGGL(resize_crop_kernel<0>, grid, block, 0, st, src, blob_bytes, desc, pool, bounds, dst, S);
}

// Host geometry, the arithmetic of HF's get_resize_output_image_size (shortest edge -> S, long edge
// int(S * long / short)) and centre crop ((new - S) // 2), and Pillow's ksize.
static inline int resize_ksize(int in_size, int out_size) {
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}

static inline void resize_geometry(int H, int W, int S, ResizeDesc& d) {
    const int shortest = W <= H ? W : H, longest = W <= H ? H : W;
    const int new_long = (int)((double)((int64_t)S * longest) / (double)shortest);
    d.H = H; d.W = W;
    d.new_w = W <= H ? S : new_long;
    d.new_h = W <= H ? new_long : S;
    d.top = (d.new_h - S) / 2;
    d.left = (d.new_w - S) / 2;
    d.ksx = resize_ksize(W, d.new_w);
    d.ksy = resize_ksize(H, d.new_h);
}

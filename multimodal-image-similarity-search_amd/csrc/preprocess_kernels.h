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
//   resize_crop_kernel     one block per (16 output rows, 256 of the 3*S (column, channel) values, image): streams
//                          the source rows its rows need; a thread computes the horizontally resampled uint8 value
//                          of its (column, channel) for the row and feeds it to the <=16 vertical accumulators it
//                          owns, so the intermediate image never exists in memory. Only the crop window is computed.
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

// grid (ceil(S/16), ceil(3S/256), B), block 256. dst: uint8 [B,S,S,3].
__global__ __launch_bounds__(256) void resize_crop_kernel(const uint8_t* __restrict__ src,
                                                          const ResizeDesc* __restrict__ desc,
                                                          const int32_t* __restrict__ pool,
                                                          const int32_t* __restrict__ bounds,
                                                          uint8_t* __restrict__ dst, int S) {
    constexpr int RT = MMISS_RESIZE_ROWS;
    constexpr int32_t HALF = 1 << (MMISS_RESIZE_PRECISION_BITS - 1);
    const int b = blockIdx.z;
    const ResizeDesc d = desc[b];
    const int r0 = blockIdx.x * RT;
    const int v = blockIdx.y * 256 + threadIdx.x;  // (column, channel) value of the output row
    const bool live = v < 3 * S;
    const int col = live ? v / 3 : 0, ch = live ? v - col * 3 : 0;
    const int32_t* bnd = bounds + (size_t)b * 4 * S;
    const int xmin = bnd[col], xcnt = live ? bnd[S + col] : 0;
    // vertical windows of this tile's rows (block-uniform)
    int ymin[RT], ycnt[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const bool ok = r0 + r < S;
        ymin[r] = ok ? bnd[2 * S + r0 + r] : 0;
        ycnt[r] = ok ? bnd[3 * S + r0 + r] : 0;
    }
    const int rl = (r0 + RT < S ? r0 + RT : S) - 1;
    const int y0 = ymin[0], y1 = bnd[2 * S + rl] + bnd[3 * S + rl];
    const int32_t* kx = pool + d.kx_off + col;
    const int32_t* ky = pool + d.ky_off + (int64_t)r0 * d.ksy;
    const uint8_t* sp = src + d.src_off + (int64_t)xmin * 3 + ch;
    const int64_t row_stride = (int64_t)d.W * 3;
    int32_t acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = HALF;

    for (int y = y0; y < y1; ++y) {
        const uint8_t* row = sp + y * row_stride;
        int32_t s = HALF;
        for (int x = 0; x < xcnt; ++x) s += (int32_t)row[x * 3] * kx[(int64_t)x * S];
        const int32_t h = clip8_fixed(s);  // the uint8 pixel of Pillow's intermediate image
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int t = y - ymin[r];
            if ((unsigned)t < (unsigned)ycnt[r]) acc[r] += h * ky[r * d.ksy + t];
        }
    }
    if (!live) return;
#pragma unroll
    for (int r = 0; r < RT; ++r)
        if (r0 + r < S) dst[((int64_t)b * S + r0 + r) * S * 3 + v] = (uint8_t)clip8_fixed(acc[r]);
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

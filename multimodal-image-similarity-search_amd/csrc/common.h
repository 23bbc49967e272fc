// common.h — shared host/device helpers of libmmiss (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <string>
#include <vector>
#include <mutex>
#include "../../include/mmiss.h"
#include "../../include/mmiss_debug.h"

// ------------------------------------------------------------------ vector types
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MMISS_WAVE 64

// ------------------------------------------------------------------ error plumbing
void mmiss_set_error(const char* fmt, ...);

#define MM_FAIL(code, ...)            \
    do {                              \
        mmiss_set_error(__VA_ARGS__); \
        return (code);                \
    } while (0)

#define MM_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            mmiss_set_error("HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e),     \
                            __FILE__, __LINE__, #expr);                                           \
            return MMISS_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

#define MM_TRY(expr)             \
    do {                         \
        int _s = (expr);         \
        if (_s != MMISS_OK) return _s; \
    } while (0)

// is `p` device memory? (host pageable pointers make hipPointerGetAttributes fail -> host)
bool mmiss_is_device_ptr(const void* p);

// process-wide tuning knobs set through mmiss_dbg_set_option (tests / A-B experiments)
int mmiss_option(const char* key, int dflt);

// Make sure `device` is a usable gfx950 and current. Fails loudly otherwise (no CPU fallback).
int mmiss_use_device(int device);
// Raise a kernel's dynamic-LDS limit to `lds` bytes on the CURRENT device, once per (kernel, device): the attribute is
// per device, so a process-wide "done" flag would leave the second GPU of a process at the 64 KB default.
int mmiss_ensure_dyn_lds(const void* kernel, int lds);

// ------------------------------------------------------------------ kernel timing (mmiss_prof_*)
struct ProfScope {
    ProfScope(const char* name, hipStream_t s, double flops, double bytes);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};
#define MM_PROF(name, stream, flops, bytes) ProfScope _prof_scope_##__LINE__((name), (stream), (double)(flops), (double)(bytes))

// ------------------------------------------------------------------ small device buffer RAII
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n) {
        release();
        if (n == 0) return MMISS_OK;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            mmiss_set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
            return MMISS_ERR_NOMEM;
        }
        bytes = n;
        return MMISS_OK;
    }
    int ensure(size_t n) { return (n <= bytes) ? MMISS_OK : alloc(n); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    ~DevBuf() { release(); }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
    DevBuf() = default;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// max WITHOUT the canonicalising `v_max x, x` that fmaxf() emits in front of every use of a value the compiler cannot prove
// quiet (an MFMA result, a loaded float): maxnum must quiet a signalling NaN. Where the operands are never one — scores,
// magnitudes — these save one vector instruction per operand (round 4: 12 of ~60 per key-pair step of the long attention).
// CAUTION (guide 5.7 item 2): hipcc pads the MFMA-result -> VALU-read hazard only for consumers it models; an `asm` statement is
// opaque to its hazard recogniser. Values that may come STRAIGHT out of an MFMA must pass mm_mfma_settle() (an s_nop covering the
// 8-pass XDL write -> VALU read window) before they enter mm_max2 / mm_max3; without it the maxima read stale accumulator
// registers now and then (seen as run-to-run differences of the online softmax's offsets).
#define mm_mfma_settle(...) asm volatile("s_nop 11" : __VA_ARGS__)
__device__ __forceinline__ float mm_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float mm_max2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

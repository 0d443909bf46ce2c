// Shared helpers for libfrhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/frhip.h"

#define FR_WAVE 64

void fr_set_error(const char* fmt, ...);

#define FR_REQUIRE(cond, ...)                      \
    do {                                           \
        if (!(cond)) {                             \
            fr_set_error(__VA_ARGS__);             \
            return FR_E_INVALID;                   \
        }                                          \
    } while (0)

#define FR_CHECK_LAUNCH(name)                                                     \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            fr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return FR_E_LAUNCH;                                                   \
        }                                                                         \
    } while (0)

// Diagnostic switches (in-kernel cycle stamps, alternative kernel variants) exist only in the DEBUG build
// (`make debug` -> libfrhip_debug.so, -DFR_DEBUG_BUILD), where they are read from the environment on every call.
// The product library reads no environment variable and instantiates no stamped kernel.
#ifdef FR_DEBUG_BUILD
#include <cstdlib>
static inline int fr_dbg_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline void* fr_dbg_ptr(const char* name) { const char* e = getenv(name); return e ? (void*)strtoll(e, nullptr, 0) : nullptr; }
constexpr bool FR_DEBUG = true;
#else
static inline int fr_dbg_int(const char*, int dflt) { return dflt; }
static inline void* fr_dbg_ptr(const char*) { return nullptr; }
constexpr bool FR_DEBUG = false;
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: one atomic bit per device ordinal
// remembers where it has been raised (idempotent; the only process-wide state the library keeps).
struct FrDevLatch { std::atomic<unsigned long long> mask{0}; };
static inline bool fr_raise_lds(const void* kernel, size_t bytes, FrDevLatch& latch) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (bit && (latch.mask.load(std::memory_order_acquire) & bit)) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    if (bit) latch.mask.fetch_or(bit, std::memory_order_release);
    return true;
}

static inline hipStream_t fr_stream(fr_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int fr_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef int int4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

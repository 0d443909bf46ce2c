// Shared helpers for libfrhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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

// Dev tool (not part of the product): cross-wave interference probe on gfx950.
// Victims run packed-f32 VALU ops and check them against the scalar form in the same lane;
// aggressors run one instruction family in a tight loop on a second stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) float float2v;
typedef __attribute__((ext_vector_type(4))) float float4v;
typedef __attribute__((ext_vector_type(16))) float float16v;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;

// mode 0: v_pk_mul_f32 + v_pk_add_f32 ; mode 1: v_pk_fma_f32 ; mode 2: scalar v_mul/v_add only (control)
__global__ void victim(int mode, int iters, unsigned* __restrict__ errs_per_lane, unsigned* __restrict__ total) {
    const int lane = threadIdx.x & 63;
    float2v x = {(float)(threadIdx.x + 1), (float)(threadIdx.x * 2 + 3)};
    const float2v m = {1.5f, 0.75f}, one = {1.0f, 2.0f};
    unsigned err = 0;
    for (int i = 0; i < iters; ++i) {
        float2v p, q;
        float s0, s1, t0, t1;
        if (mode == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(m));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(q) : "v"(p), "v"(one));
        } else if (mode == 1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(q) : "v"(x), "v"(m), "v"(one));
            p = q;
        } else {
            float a, b;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(x.x), "v"(m.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(x.y), "v"(m.y));
            p = float2v{a, b};
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(p.x), "v"(one.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(p.y), "v"(one.y));
            q = float2v{a, b};
        }
        // reference through integer-exact arithmetic the compiler cannot pack (volatile asm scalar forms)
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(x.x), "v"(m.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(x.y), "v"(m.y));
        if (mode == 1) {
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(x.x), "v"(m.x), "v"(one.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(x.y), "v"(m.y), "v"(one.y));
            if (q.x != t0) err += 1;
            if (q.y != t1) err += 0x10000;
        } else {
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(t0) : "v"(s0), "v"(one.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(t1) : "v"(s1), "v"(one.y));
            if (p.x != s0 || q.x != t0) err += 1;
            if (p.y != s1 || q.y != t1) err += 0x10000;
        }
        x.x = x.x * 0.5f + 1.0f + (float)(i & 7);
        x.y = x.y * 0.25f + 3.0f + (float)(i & 3);
    }
    if (err) { atomicAdd(&errs_per_lane[lane], 1u); atomicAdd(total, err & 0xffff); atomicAdd(total + 1, err >> 16); }
}

// aggressors: kind 0 mfma_f32_16x16x32_f16, 1 mfma_f32_32x32x16_f16, 2 mfma_f32_16x16x16f16, 3 mfma_f32_16x16x4f32,
//             4 v_cvt_pk (f32->f16) + v_pk_mul_f16 VALU only, 5 ds_read_b128 loop
typedef __attribute__((ext_vector_type(4))) int int4v;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ __launch_bounds__(256) void aggressor_mem(int kind, int iters, const float* __restrict__ src, unsigned src_bytes,
                                                     float* __restrict__ sink) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    int4v acc = {0, 0, 0, 0};
    unsigned base = (blockIdx.x * 256 + tid) * 16u;
    for (int i = 0; i < iters; ++i) {
        unsigned off = (base + (unsigned)i * 65536u) % (src_bytes - 16u);
        off &= ~15u;
        if ((kind == 7 || kind == 9) && (lane & 1)) off = 0x80000000u;
        if (kind == 6 || kind == 7) {
            int4v v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            acc += v;
        } else if (kind == 8 || kind == 9) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + wave * 1024 + (i & 1) * 256), 16, off, 0, 0, 0);
            if ((i & 7) == 7) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc.x += (int)lds[wave * 1024 + lane]; }
        } else {          // 10: 16-byte global stores
            *reinterpret_cast<int4v*>(reinterpret_cast<char*>(sink) + 4096 + (off % (1u << 24))) = acc;
            acc.x += i;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) sink[tid] = 1.f;
#endif
}

__global__ __launch_bounds__(256) void aggressor(int kind, int iters, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    half8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (tid + i)); b8[i] = (_Float16)(0.02f * (tid - i)); }
    half4 a4 = {a8[0], a8[1], a8[2], a8[3]}, b4 = {b8[0], b8[1], b8[2], b8[3]};
    float4v c4 = {0, 0, 0, 0}, d4 = {0, 0, 0, 0};
    float16v c16 = {0};
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        if (kind == 0) {
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c4, 0, 0, 0);
            d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b8, a8, d4, 0, 0, 0);
        } else if (kind == 1) {
            c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c16, 0, 0, 0);
        } else if (kind == 2) {
            c4 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c4, 0, 0, 0);
            d4 = __builtin_amdgcn_mfma_f32_16x16x16f16(b4, a4, d4, 0, 0, 0);
        } else if (kind == 3) {
            c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(acc + 1.0f, 2.0f, c4, 0, 0, 0);
            d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(3.0f, acc, d4, 0, 0, 0);
        } else if (kind == 4) {
            half8 t = a8 * b8 + a8;
            a8 = t * (_Float16)0.5f + b8;
            acc += (float)a8[0];
        } else {
            float4v v = *reinterpret_cast<const float4v*>(&lds[((tid * 4 + i * 64) & 4095) & ~3]);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    float r = acc + c4.x + c4.y + d4.z + d4.w + c16[0] + c16[15];
    if (r == 123456.789f) sink[tid] = r;
}

extern "C" int race_victim(int mode, int iters, int blocks, unsigned* errs_per_lane, unsigned* total, void* stream) {
    victim<<<blocks, 256, 0, (hipStream_t)stream>>>(mode, iters, errs_per_lane, total);
    return (int)hipGetLastError();
}
// big-LDS aggressor: 512 threads, dynamic LDS of any size, ds_read_b128 over the whole allocation feeding f16 MFMAs
__global__ __launch_bounds__(512) void aggressor_big(int iters, int lds_bytes, int use_mfma, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char dl[];
    const int tid = threadIdx.x;
    for (int i = tid * 16; i < lds_bytes; i += 512 * 16) *reinterpret_cast<int4v*>(dl + i) = int4v{i, i + 1, i + 2, i + 3};
    __syncthreads();
    float4v c = {0, 0, 0, 0};
    int4v acc = {0, 0, 0, 0};
    unsigned off = tid * 16u;
    for (int i = 0; i < iters; ++i) {
        off = (off + 8192u + 16u * (i & 3)) % (unsigned)(lds_bytes - 16);
        off &= ~15u;
        int4v a = *reinterpret_cast<const int4v*>(dl + off);
        int4v b = *reinterpret_cast<const int4v*>(dl + ((off + 4096u) % (unsigned)(lds_bytes - 16) & ~15u));
        if (use_mfma) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
        else acc += a + b;
    }
    if (c.x + c.y + (float)(acc.x + acc.y) == 123456.789f) sink[tid] = 1.f;
}
extern "C" int race_aggressor_big(int iters, int blocks, int lds_bytes, int use_mfma, float* sink, void* stream) {
    hipFuncSetAttribute((const void*)aggressor_big, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    aggressor_big<<<blocks, 512, lds_bytes, (hipStream_t)stream>>>(iters, lds_bytes, use_mfma, sink);
    return (int)hipGetLastError();
}
extern "C" int race_aggressor_mem(int kind, int iters, int blocks, const float* src, unsigned src_bytes, float* sink, void* stream) {
    aggressor_mem<<<blocks, 256, 0, (hipStream_t)stream>>>(kind, iters, src, src_bytes, sink);
    return (int)hipGetLastError();
}
extern "C" int race_aggressor(int kind, int iters, int blocks, float* sink, void* stream) {
    aggressor<<<blocks, 256, 0, (hipStream_t)stream>>>(kind, iters, sink);
    return (int)hipGetLastError();
}

// Dev tool (not part of the product): minimal victim for DESIGN.md 4.7.
// Victim: a packed-f32 VALU result (v_pk_mul_f32) consumed by a global_store, the dependency the product's
// crop_resize_norm had when lanes 48-63 of its output went stale beside the embedder's conv kernels.  The pair
// (producer, consumer) sits in ONE asm block so that hipcc pads nothing between them; DIST s_nop's can be put in.
// A checker kernel recomputes every stored value from its indices and histograms mismatches by lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) float float2v;
typedef __attribute__((ext_vector_type(4))) float float4v;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;

__device__ __forceinline__ float2v vx(int gtid, int i) {
    const float a = (float)((gtid * 7 + i * 13) & 1023);
    return float2v{a, a + 1.0f};
}

// mode 0: v_pk_mul_f32 (VGPR srcs) -> store, no padding     mode 1: same + s_nop 7 between
// mode 2: v_pk_mul_f32 with an SGPR source and op_sel_hi:[1,0] (the product kernel's form) -> store
// mode 3: control, two v_mul_f32 -> store                    mode 4: v_pk_mul_f32 -> v_pk_add_f32 -> store (chain)
__global__ __launch_bounds__(256) void victim(int mode, int iters, float2v* __restrict__ out) {
    const int gtid = blockIdx.x * 256 + threadIdx.x;
    const int nthr = gridDim.x * 256;
    const float2v m = {1.5f, 0.75f};
    const unsigned long long sm = (unsigned long long)__float_as_uint(1.5f);      // SGPR pair: lo = 1.5, hi = 0 (unused)
    for (int i = 0; i < iters; ++i) {
        const float2v x = vx(gtid, i);
        float2v* p = out + (size_t)i * nthr + gtid;
        float2v r;
        if (mode == 0)
            asm volatile("v_pk_mul_f32 %0, %1, %2\n\tglobal_store_dwordx2 %3, %0, off" : "=&v"(r) : "v"(x), "v"(m), "v"(p) : "memory");
        else if (mode == 1)
            asm volatile("v_pk_mul_f32 %0, %1, %2\n\ts_nop 7\n\tglobal_store_dwordx2 %3, %0, off" : "=&v"(r) : "v"(x), "v"(m), "v"(p) : "memory");
        else if (mode == 2)
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\tglobal_store_dwordx2 %3, %0, off" : "=&v"(r) : "v"(x), "s"(sm), "v"(p) : "memory");
        else if (mode == 4)
            asm volatile("v_pk_mul_f32 %0, %1, %2\n\tv_pk_add_f32 %0, %0, %2\n\tglobal_store_dwordx2 %3, %0, off" : "=&v"(r) : "v"(x), "v"(m), "v"(p) : "memory");
        else if (mode == 6) {
            // the form the ISA bisect of the product victim singled out (pkisa_gen.py / pkisa_run.py): a packed add IN
            // PLACE on its src1 pair with the halves of src1 SWAPPED: D.lo = S0.lo + D.hi, D.hi = S0.hi + D.lo
            r = x;
            asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]\n\tglobal_store_dwordx2 %2, %0, off" : "+v"(r) : "v"(m), "v"(p) : "memory");
        } else if (mode == 7)       // the same swap, destination NOT one of the sources
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\tglobal_store_dwordx2 %3, %0, off" : "=&v"(r) : "v"(m), "v"(x), "v"(p) : "memory");
        else if (mode == 10)      // the swap on a multiply
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]\n\tglobal_store_dwordx2 %3, %0, off" : "=&v"(r) : "v"(m), "v"(x), "v"(p) : "memory");
        else if (mode == 11)      // the swap on SRC0 instead of src1
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]\n\tglobal_store_dwordx2 %3, %0, off" : "=&v"(r) : "v"(x), "v"(m), "v"(p) : "memory");
        else if (mode == 12)      // a packed MOVE that takes src0's high half for the low result
            asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]\n\tglobal_store_dwordx2 %2, %0, off" : "=&v"(r) : "v"(x), "v"(p) : "memory");
        else if (mode == 13)      // the swap on src1 of a packed fma
            asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\tglobal_store_dwordx2 %3, %0, off" : "=&v"(r) : "v"(m), "v"(x), "v"(p) : "memory");
        else if (mode == 14)      // only the HIGH result takes the other half (op_sel_hi:[1,0]); low result straight
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]\n\tglobal_store_dwordx2 %3, %0, off" : "=&v"(r) : "v"(m), "v"(x), "v"(p) : "memory");
        else if (mode == 15)      // only the LOW result takes the other half (op_sel:[0,1]); high result straight
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]\n\tglobal_store_dwordx2 %3, %0, off" : "=&v"(r) : "v"(m), "v"(x), "v"(p) : "memory");
        else if (mode == 8) {       // in place, halves NOT swapped
            r = x;
            asm volatile("v_pk_add_f32 %0, %1, %0\n\tglobal_store_dwordx2 %2, %0, off" : "+v"(r) : "v"(m), "v"(p) : "memory");
        } else if (mode == 9) {     // in place and swapped, a v_mov of the result between the op and the store
            r = x;
            float2v q;
            asm volatile("v_pk_add_f32 %0, %2, %0 op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_mov_b32 %1, %0, %0 op_sel:[0,1]\n\tglobal_store_dwordx2 %3, %1, off" : "+v"(r), "=&v"(q) : "v"(m), "v"(p) : "memory");
        }
        else {
            // mode 5: the product kernel's shape - a divergent region that masks lanes (here: lanes whose gtid+i is
            // odd, a different set every iteration), EXEC restored by s_or_b64, then the packed op and the store.  A lane the packed op does not write keeps the sentinel -1.
            const unsigned long long on_mask = __ballot(((gtid + i) & 1) == 0);
            r = float2v{-1.0f, -1.0f};
            asm volatile(
                    "s_and_saveexec_b64 s[20:21], %4\n\t"
                    "v_pk_add_f32 %0, %0, %0\n\t"
                    "s_or_b64 exec, exec, s[20:21]\n\t"
                    "v_pk_mul_f32 %0, %1, %2\n\t"
                    "global_store_dwordx2 %3, %0, off"
                    : "+v"(r) : "v"(x), "v"(m), "v"(p), "s"(on_mask) : "memory", "s20", "s21");
        }
    }
}

// control needs a register PAIR written by two scalar ops; done with explicit sub-registers in its own kernel
__global__ __launch_bounds__(256) void victim_ctl(int iters, float2v* __restrict__ out) {
    const int gtid = blockIdx.x * 256 + threadIdx.x;
    const int nthr = gridDim.x * 256;
    for (int i = 0; i < iters; ++i) {
        const float2v x = vx(gtid, i);
        float2v* p = out + (size_t)i * nthr + gtid;
        float a, b;
        asm volatile("v_mul_f32 %0, 1.5, %2\n\tv_mul_f32 %1, 0x3f400000, %3" : "=&v"(a), "=&v"(b) : "v"(x.x), "v"(x.y));
        *p = float2v{a, b};
    }
}

__global__ void checker(int mode, int iters, int nthr, const float2v* __restrict__ out, unsigned* __restrict__ hist,
                        unsigned* __restrict__ total) {
    const int gtid = blockIdx.x * 256 + threadIdx.x;
    if (gtid >= nthr) return;
    unsigned bad = 0, unsw = 0;
    for (int i = 0; i < iters; ++i) {
        const float2v x = vx(gtid, i);
        float2v w = (mode == 2) ? float2v{x.x * 1.5f, x.y * 1.5f} : float2v{x.x * 1.5f, x.y * 0.75f};
        if (mode == 4) { w.x += 1.5f; w.y += 0.75f; }
        if (mode == 6 || mode == 7 || mode == 9) w = float2v{1.5f + x.y, 0.75f + x.x};
        if (mode == 8) w = float2v{1.5f + x.x, 0.75f + x.y};
        if (mode == 10) w = float2v{1.5f * x.y, 0.75f * x.x};
        if (mode == 11) w = float2v{x.y + 1.5f, x.x + 0.75f};
        if (mode == 12) w = float2v{x.y, x.x};
        if (mode == 13) w = float2v{1.5f * x.y + 1.5f, 0.75f * x.x + 0.75f};
        if (mode == 14) w = float2v{1.5f + x.x, 0.75f + x.x};
        if (mode == 15) w = float2v{1.5f + x.y, 0.75f + x.y};
        const float2v g = out[(size_t)i * nthr + gtid];
        if (g.x != w.x) ++bad;
        if (g.y != w.y) ++bad;
        // what a wrong word holds: the result of the SAME add with the halves of src1 NOT swapped?
        if ((mode == 6 || mode == 7 || mode == 9) && g.x != w.x && g.x == 1.5f + x.x) ++unsw;
        if ((mode == 6 || mode == 7 || mode == 9) && g.y != w.y && g.y == 0.75f + x.y) ++unsw;
    }
    if (bad) { atomicAdd(&hist[threadIdx.x & 63], bad); atomicAdd(total, bad); atomicAdd(total + 1, unsw); }
}

// aggressors: kind 0 = back-to-back independent v_mfma_f32_16x16x32_f16, few registers; kind 1 = the same with ~100
// live accumulator registers per lane and 512-thread blocks (two fit a CU: the lean conv kernel's shape)
__global__ __launch_bounds__(256) void aggr_small(int iters, float* __restrict__ sink) {
    half8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (threadIdx.x + i)); b8[i] = (_Float16)(0.02f * (threadIdx.x - i)); }
    float4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b8, a8, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b8, b8, c3, 0, 0, 0);
    }
    if (c0.x + c1.y + c2.z + c3.w == 123456.789f) sink[threadIdx.x] = 1.f;
}
__global__ __launch_bounds__(512, 4) void aggr_big(int iters, float* __restrict__ sink) {
    half8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (threadIdx.x + i)); b8[i] = (_Float16)(0.02f * (threadIdx.x - i)); }
    float4v c[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) c[k] = float4v{0, 0, 0, (float)k};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 24; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 24; ++k) s += c[k].x + c[k].w;
    if (s == 123456.789f) sink[threadIdx.x] = s;
}

extern "C" int pk_victim(int mode, int iters, int blocks, void* out, void* stream) {
    if (mode == 3) victim_ctl<<<blocks, 256, 0, (hipStream_t)stream>>>(iters, (float2v*)out);
    else victim<<<blocks, 256, 0, (hipStream_t)stream>>>(mode, iters, (float2v*)out);
    return (int)hipGetLastError();
}
extern "C" int pk_check(int mode, int iters, int blocks, const void* out, unsigned* hist, unsigned* total, void* stream) {
    checker<<<blocks, 256, 0, (hipStream_t)stream>>>(mode, iters, blocks * 256, (const float2v*)out, hist, total);
    return (int)hipGetLastError();
}
// kind 2: the register footprint of aggr_big WITHOUT any MFMA (96 live VGPRs of plain v_fma work, 512-thread blocks);
// kind 3: MFMAs with few registers but 512-thread blocks; kind 4: aggr_big's registers and MFMAs in 256-thread blocks
__global__ __launch_bounds__(512, 4) void aggr_valu_big(int iters, float* __restrict__ sink) {
    float c[96];
#pragma unroll
    for (int k = 0; k < 96; ++k) c[k] = 0.001f * (float)(k + threadIdx.x);
    const float a = 1.0001f, b = 0.5f + (float)threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 96; ++k) c[k] = __builtin_fmaf(c[k], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 96; ++k) s += c[k];
    if (s == 123456.789f) sink[threadIdx.x] = s;
}
__global__ __launch_bounds__(512, 4) void aggr_small512(int iters, float* __restrict__ sink) {
    half8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (threadIdx.x + i)); b8[i] = (_Float16)(0.02f * (threadIdx.x - i)); }
    float4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b8, a8, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b8, b8, c3, 0, 0, 0);
    }
    if (c0.x + c1.y + c2.z + c3.w == 123456.789f) sink[threadIdx.x] = 1.f;
}
__global__ __launch_bounds__(256, 4) void aggr_big256(int iters, float* __restrict__ sink) {
    half8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (threadIdx.x + i)); b8[i] = (_Float16)(0.02f * (threadIdx.x - i)); }
    float4v c[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) c[k] = float4v{0, 0, 0, (float)k};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 24; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 24; ++k) s += c[k].x + c[k].w;
    if (s == 123456.789f) sink[threadIdx.x] = s;
}

// kind 5: f32 MFMAs (v_mfma_f32_16x16x4_f32, the detector's instruction) instead of f16 ones, few registers
__global__ __launch_bounds__(512, 4) void aggr_f32mfma(int iters, float* __restrict__ sink) {
    const float a = 0.01f * threadIdx.x, b = 0.02f * (threadIdx.x & 31);
    float4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, c3, 0, 0, 0);
    }
    if (c0.x + c1.y + c2.z + c3.w == 123456.789f) sink[threadIdx.x] = 1.f;
}

extern "C" int pk_aggr(int kind, int iters, int blocks, float* sink, void* stream) {
    if (kind == 0) aggr_small<<<blocks, 256, 0, (hipStream_t)stream>>>(iters, sink);
    else if (kind == 2) aggr_valu_big<<<blocks, 512, 0, (hipStream_t)stream>>>(iters, sink);
    else if (kind == 3) aggr_small512<<<blocks, 512, 0, (hipStream_t)stream>>>(iters, sink);
    else if (kind == 4) aggr_big256<<<blocks, 256, 0, (hipStream_t)stream>>>(iters, sink);
    else if (kind == 5) aggr_f32mfma<<<blocks, 512, 0, (hipStream_t)stream>>>(iters, sink);
    else aggr_big<<<blocks, 512, 0, (hipStream_t)stream>>>(iters, sink);
    return (int)hipGetLastError();
}

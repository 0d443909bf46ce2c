// DESIGN.md 4.7, from the failing end: the round-1 crop_resize_norm (the kernel whose lanes 48-63 came out stale beside
// the embedder's conv kernels) compiled WITH packed-f32 ops (this file is built without the library's NOPK flag), in
// variants that each remove ONE ingredient.  tools/hw/pkvictim.py runs every variant beside the product's conv
// kernels and counts mismatching words against the same variant run alone.
//   0 the kernel as it was          1 pixel bytes computed from the address, no global loads (no VMEM in flight)
//   2 four dword stores instead of one float4 store      3 s_nop 7 in front of the store
//   4 result moved through v_mov (asm) before the store   5 no `valid` branch around the blend
//   6 the store's 4th word from a register instead of the constant 0
#include <hip/hip_runtime.h>
#include <cstdint>

struct Lerp { int i0, i1; float w; };
__device__ __forceinline__ Lerp lerp_coord(int d, float ratio, int n) {
    float f = ((float)d + 0.5f) * ratio - 0.5f;
    float fl = floorf(f);
    Lerp r;
    r.w = f - fl;
    int i = (int)fl;
    r.i0 = min(max(i, 0), n - 1);
    r.i1 = min(max(i + 1, 0), n - 1);
    return r;
}
__device__ __forceinline__ float bilerp(float p00, float p01, float p10, float p11, float wx, float wy) {
    float top = (1.0f - wx) * p00 + wx * p01;
    float bot = (1.0f - wx) * p10 + wx * p11;
    return (1.0f - wy) * top + wy * bot;
}

template <int V>
__global__ __launch_bounds__(256) void crop_victim(const uint8_t* __restrict__ frames, int H, int W,
                                                   const float* __restrict__ boxes, const int32_t* __restrict__ counts,
                                                   int cap, int size, float* __restrict__ out) {
    const int slot = blockIdx.x;
    const int f = slot / cap, i = slot - f * cap;
    float* o = out + (int64_t)slot * size * size * 4;
    const float4 b = *reinterpret_cast<const float4*>(boxes + (int64_t)slot * 4);
    const int x1 = (int)truncf(b.x), y1 = (int)truncf(b.y), x2 = (int)truncf(b.z), y2 = (int)truncf(b.w);
    const int tw = x2 - x1 + 1, th = y2 - y1 + 1;
    const bool valid = V == 5 ? true : (i < counts[f] && tw > 0 && th > 0);
    const uint8_t* fr = frames + (int64_t)f * H * W * 3;
    const float ry = (float)th / (float)size, rx = (float)tw / (float)size;
    for (int t = threadIdx.x; t < size * size; t += 256) {
        const int oy = t / size, ox = t - oy * size;
        float v[3] = {0.f, 0.f, 0.f};
        if (valid) {
            Lerp ly = lerp_coord(oy, ry, th), lx = lerp_coord(ox, rx, tw);
            const int ys[2] = {y1 - 1 + ly.i0, y1 - 1 + ly.i1};
            const int xs[2] = {x1 - 1 + lx.i0, x1 - 1 + lx.i1};
            float p[2][2][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const bool in = ys[a] >= 0 && ys[a] < H && xs[c] >= 0 && xs[c] < W;
                    const int64_t off = ((int64_t)(in ? ys[a] : 0) * W + (in ? xs[c] : 0)) * 3;
                    const uint8_t* px = fr + off;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        float s;
                        if (V == 1) s = (float)(((unsigned)off * 2654435761u + (unsigned)ch * 40503u) >> 24);
                        else s = (float)px[2 - ch];
                        p[a][c][ch] = in ? s : 0.f;
                    }
                }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                v[ch] = (bilerp(p[0][0][ch], p[0][1][ch], p[1][0][ch], p[1][1][ch], lx.w, ly.w) - 127.5f) * 0.0078125f;
        }
        if (V == 2) {
            o[t * 4 + 0] = v[0]; o[t * 4 + 1] = v[1]; o[t * 4 + 2] = v[2]; o[t * 4 + 3] = 0.f;
        } else if (V == 3) {
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            *reinterpret_cast<float4*>(o + t * 4) = make_float4(v[0], v[1], v[2], 0.f);
        } else if (V == 4) {
            float a0, a1, a2;
            asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=v"(a0), "=v"(a1), "=v"(a2) : "v"(v[0]), "v"(v[1]), "v"(v[2]));
            *reinterpret_cast<float4*>(o + t * 4) = make_float4(a0, a1, a2, 0.f);
        } else if (V == 6) {
            *reinterpret_cast<float4*>(o + t * 4) = make_float4(v[0], v[1], v[2], (float)t);
        } else {
            *reinterpret_cast<float4*>(o + t * 4) = make_float4(v[0], v[1], v[2], 0.f);
        }
    }
}

extern "C" int pkv_run(int variant, const uint8_t* frames, int nframes, int H, int W, const float* boxes,
                       const int32_t* counts, int cap, int size, float* out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    dim3 g(nframes * cap);
    switch (variant) {
        case 0: crop_victim<0><<<g, 256, 0, s>>>(frames, H, W, boxes, counts, cap, size, out); break;
        case 1: crop_victim<1><<<g, 256, 0, s>>>(frames, H, W, boxes, counts, cap, size, out); break;
        case 2: crop_victim<2><<<g, 256, 0, s>>>(frames, H, W, boxes, counts, cap, size, out); break;
        case 3: crop_victim<3><<<g, 256, 0, s>>>(frames, H, W, boxes, counts, cap, size, out); break;
        case 4: crop_victim<4><<<g, 256, 0, s>>>(frames, H, W, boxes, counts, cap, size, out); break;
        case 5: crop_victim<5><<<g, 256, 0, s>>>(frames, H, W, boxes, counts, cap, size, out); break;
        case 6: crop_victim<6><<<g, 256, 0, s>>>(frames, H, W, boxes, counts, cap, size, out); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

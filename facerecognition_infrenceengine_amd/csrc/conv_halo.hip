// 3x3 / stride 1 / pad 1 convolution with an LDS-STAGED INPUT TILE (halo) reused by all nine taps:
// the IResNet body convs (82 % of the embed FLOPs) at 28x28 and 14x14.  Same math, operand maps and
// epilogue as conv_mfma.hip; what changes is the traffic: the generic implicit GEMM re-gathers the
// input tile for every tap (operand fetch = (BN+BM)*128 B per 64-deep K step, measured bound by the
// L2/Infinity-Cache gather rate), here the input rows + halo of one 64-channel chunk are brought in
// ONCE per chunk (9 K steps) and only the weight tile streams per step.
//
// Tile = 196 output pixels (TH rows x full width W: 14x14 or 7x28) of one image x BN couts
// (13 MFMA pixel tiles, 94 % useful) ; 512 threads = 8 waves = WN (64 couts each) x WP pixel groups.
// LDS: input halo chunk [2][320 rows][64 ch] (40 KB each, zero border by out-of-range LDS-DMA),
// weight tile [2][BN][64] (32 KB each at BN = 256); rows 128 B with 16-B chunk XOR (row & 7).
// Schedule per K step q = chunk*9 + tap: counted wait -> raw barrier -> issue W(q+1) (LDS-DMA, scalar
// offset only) and, at tap 0, the next chunk's halo (lands two steps later, vmcnt(5) keeps it in flight
// across one barrier) -> 2 x (NA A-fragments + PT B-fragments via ds_read_b128) -> 2*NA*PT MFMAs.
// No gather arithmetic in the loop: per-lane halo bases are fixed, taps are scalar row offsets.
#include "common.h"

struct HaloP {
    const half_t* x; const half_t* w; half_t* y;
    const float* bias; const float* slope; const half_t* res;
    int B, H, W, Cin, Cout, bias_mode;
    int TH, tiles_per_img;      // output rows per tile, H / TH
    unsigned xbytes, wbytes;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define HK 64          // channels per chunk
#define XROWS 320      // halo rows capacity (>= (TH+2)*(W+2))
#define NXI 5          // halo LDS-DMA instructions per thread per chunk (8 waves x 5 x 8 rows)
#define NPT 13         // pixel tiles (196 px)

template <int WN>
__global__ __launch_bounds__(512) void conv_halo_kernel(HaloP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BN = 64 * WN;
    constexpr int WP = 8 / WN;
    constexpr int PT = (NPT + WP - 1) / WP;
    constexpr int NWI = BN / 64;                      // weight LDS-DMA instructions per thread per step
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* xs = lds;                                 // [2][XROWS][HK]
    half_t* ws = lds + 2 * XROWS * HK;                // [2][BN][HK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wp = wave / WN;
    // XCD-aware order, cout tile innermost (see conv_mfma.hip)
    int tile;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int ntn = p.Cout / BN;
    const int cout0 = (tile % ntn) * BN;
    const int mt = tile / ntn;
    const int n = mt / p.tiles_per_img, y0 = (mt - n * p.tiles_per_img) * p.TH;
    const int HW = p.W + 2;
    const int nhalo = (p.TH + 2) * HW;

    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);

    const int lrow = lane >> 3;
    const int schunk = (lane & 7) ^ (lrow & 7);       // source chunk: LDS image chunk' = chunk ^ (row & 7)
    unsigned xoff[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int h = (wave * NXI + i) * 8 + lrow;
        const int hy = h / HW, hx = h - hy * HW;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = h < nhalo && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        xoff[i] = ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.Cin + schunk * 8) * 2) : 0x80000000u;
    }
    unsigned woff[NWI];
    const int K = 9 * p.Cin;
#pragma unroll
    for (int i = 0; i < NWI; ++i) woff[i] = (unsigned)(((cout0 + (wave * NWI + i) * 8 + lrow) * K + schunk * 8) * 2);

    auto issue_x = [&](int c) {
        half_t* dst = xs + (c & 1) * XROWS * HK;
#pragma unroll
        for (int i = 0; i < NXI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst + (wave * NXI + i) * 8 * HK), 16, xoff[i],
                                                     c * (HK * 2), 0, 0);
    };
    auto issue_w = [&](int q) {
        const int c = q / 9, tap = q - c * 9;
        half_t* dst = ws + (q & 1) * BN * HK;
#pragma unroll
        for (int i = 0; i < NWI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(dst + (wave * NWI + i) * 8 * HK), 16, woff[i],
                                                     (tap * p.Cin + c * HK) * 2, 0, 0);
    };

    // per-lane halo base (top-left of the 3x3 window) of each of this wave's pixel tiles
    const int fr = lane & 15, fq = lane >> 4;
    int hbase[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int px = (wp * PT + j) * 16 + fr;
        int hb = 0;
        if (px < p.TH * p.W) {
            const int oy = px / p.W, ox = px - oy * p.W;
            hb = oy * HW + ox;
        }
        hbase[j] = hb;
    }

    float4v acc[4][PT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nq = 9 * (p.Cin / HK);
    issue_x(0);
    issue_w(0);
    int tap = 0, c = 0, toff = 0, kw = 0;
    for (int q = 0; q < nq; ++q) {
        // W(q) (and everything older) landed; a halo issued at the previous step (tap == 1 now) may still fly
        if (tap == 1 && c + 1 < p.Cin / HK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NXI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (q + 1 < nq) issue_w(q + 1);
        if (tap == 0 && c + 1 < p.Cin / HK) issue_x(c + 1);

        const half_t* wl = ws + (q & 1) * BN * HK + (wn * 64) * HK;
        const half_t* xl = xs + (c & 1) * XROWS * HK;
        int xaddr[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int hr = hbase[j] + toff;
            xaddr[j] = hr * HK + ((fq ^ (hr & 7)) << 3);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 a[4], b[PT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 16 + fr;
                a[i] = *reinterpret_cast<const half8*>(wl + row * HK + (((kk * 4 + fq) ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < PT; ++j) b[j] = *reinterpret_cast<const half8*>(xl + (xaddr[j] ^ (kk << 5)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        // next tap (scalar bookkeeping, no divisions)
        ++tap; ++kw; ++toff;
        if (kw == 3) { kw = 0; toff += HW - 3; }
        if (tap == 9) { tap = 0; kw = 0; toff = 0; ++c; }
    }

    // ---- epilogue (same as conv_mfma.hip): lane owns pixel (tile j, fr) and couts i*16 + fq*4 .. +3
    const int HoWo = p.H * p.W;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int px = (wp * PT + j) * 16 + fr;
        if (px >= p.TH * p.W) continue;
        const int oy = px / p.W, ox = px - oy * p.W;
        const int ho = y0 + oy;
        const int m = n * HoWo + ho * p.W + ox;
        int bsel = 0;
        if (p.bias_mode == 1) {
            const int rc = ho == 0 ? 0 : (ho == p.H - 1 ? 2 : 1);
            const int cc = ox == 0 ? 0 : (ox == p.W - 1 ? 2 : 1);
            bsel = (rc * 3 + cc) * p.Cout;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = cout0 + wn * 64 + i * 16 + fq * 4;
            float4v v = acc[i][j];
            if (p.bias) v += *reinterpret_cast<const float4v*>(p.bias + bsel + co);
            if (p.slope) {
                const float4v sv = *reinterpret_cast<const float4v*>(p.slope + co);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sv[e];
            }
            const size_t o = (size_t)m * p.Cout + co;
            if (p.res) {
                const half4 rv = *reinterpret_cast<const half4*>(p.res + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            }
            const half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<half4*>(p.y + o) = hv;
        }
    }
#endif
}

template <int WN>
static int launch_halo(const HaloP& p, hipStream_t s) {
    constexpr int BN = 64 * WN;
    const size_t lds = (size_t)(2 * XROWS * HK + 2 * BN * HK) * sizeof(half_t);
    auto kern = conv_halo_kernel<WN>;
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            fr_set_error("conv_halo: cannot raise dynamic LDS to %zu bytes", lds);
            return FR_E_LAUNCH;
        }
        done = true;
    }
    const int blocks = p.B * p.tiles_per_img * (p.Cout / BN);
    kern<<<blocks, 512, lds, s>>>(p);
    return FR_OK;
}

// Returns 1 if the halo kernel handled the layer, 0 if the shape is not eligible, < 0 on error.
int fr_conv_halo_try(const fr_conv_args* a, hipStream_t s) {
    if (!(a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->H == a->W && a->out_f32_partial == nullptr))
        return 0;
    if (!(a->Cin % 64 == 0 && a->Cout % 128 == 0)) return 0;
    int TH;
    if (a->H == 14) TH = 14; else if (a->H == 28) TH = 7; else return 0;
    if ((int64_t)a->B * a->H * a->W * a->Cin * 2 >= (1ll << 31) || (int64_t)a->Cout * 9 * a->Cin * 2 >= (1ll << 31)) return 0;
    HaloP p;
    p.x = (const half_t*)a->x; p.w = (const half_t*)a->w; p.y = (half_t*)a->y;
    p.bias = a->bias; p.slope = a->slope; p.res = (const half_t*)a->residual;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.bias_mode = a->bias_mode;
    p.TH = TH; p.tiles_per_img = a->H / TH;
    p.xbytes = (unsigned)((int64_t)a->B * a->H * a->W * a->Cin * 2);
    p.wbytes = (unsigned)((int64_t)a->Cout * 9 * a->Cin * 2);
    int rc = (a->Cout % 256 == 0) ? launch_halo<4>(p, s) : launch_halo<2>(p, s);
    return rc == FR_OK ? 1 : rc;
}

// Implicit-GEMM convolution on the gfx950 matrix cores: NHWC f16 activations, f32 accumulate,
// fused (bias | border-class bias) + PReLU + residual epilogue, or split-K f32 partials (FC).
// This is the ArcFace IResNet conv stack that the reference runs inside
// FaceAnalysis.get (/root/reference/infrenceServer.py:528) through ONNX Runtime.
//
// GEMM view:  C[cout][pixel] = sum_k W[cout][k] * X[pixel][k],  k = (kh*KW + kw)*Cin + ci.
//   MFMA A operand = weights (rows = couts), B operand = gathered input pixels (cols = pixels):
//   with v_mfma_f32_16x16x32_f16 a lane then owns ONE pixel and 4 CONSECUTIVE couts per
//   accumulator quad, i.e. 8 contiguous bytes of the NHWC output row.
// Tiling: 256 threads = 4 waves, each wave 64 couts x 64 pixels (4x4 MFMA tiles, 64 acc VGPRs);
//   block = (64*WN couts) x (64*(4/WN) pixels); K step 64 (always inside one filter tap when
//   Cin >= 64); two LDS stages, register-prefetched (global->VGPR issued before the MFMAs of the
//   current stage, VGPR->LDS after them), one barrier per K step.
// LDS image: rows of 64 halves (128 B), 16-B chunk index XOR (row & 7): conflict-free for the
//   ds_read_b128 fragment reads of the 16x16x32 operand layout.
// Zero padding / M tail: buffer loads with the offset forced out of range return 0.
#include "common.h"

#define BK 64

struct ConvP {
    const half_t* x; const half_t* w; half_t* y;
    const float* bias; const float* slope; const half_t* res; float* partial;
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, bias_mode, splitk;
    int M, K, nk;            // M = B*Ho*Wo, K = row length of w (halves), nk = K / 64
    unsigned xbytes, wbytes;
    const half_t* x2; int C2, nk_main; unsigned x2bytes;      // second input through a 1x1 tap (fr_conv_args.x2): K steps nk_main .. nk-1
};

__device__ __forceinline__ int4v buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
}

template <int WN>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, float4v (&acc)[4][4], int cout0, int m0, int wc, int wp,
                                              int fr, int fq) {
    const int HoWo = p.Ho * p.Wo;
    // ---- epilogue: lane owns pixel (pj*16 + fr) and couts (ci*16 + fq*4 .. +3)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wp * 64 + j * 16 + fr;
        if (m >= p.M) continue;
        int bsel = 0;
        if (p.bias_mode == 1) {
            int r = m % HoWo;
            int ho = r / p.Wo, wo = r - ho * p.Wo;
            int rc = ho == 0 ? 0 : (ho == p.Ho - 1 ? 2 : 1);
            int cc = wo == 0 ? 0 : (wo == p.Wo - 1 ? 2 : 1);
            bsel = (rc * 3 + cc) * p.Cout;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = cout0 + wc * 64 + i * 16 + fq * 4;
            float4v v = acc[i][j];
            if (p.partial) {
                float* dst = p.partial + ((size_t)blockIdx.z * p.M + m) * p.Cout + co;
                *reinterpret_cast<float4v*>(dst) = v;
                continue;
            }
            if (p.bias) {
                float4v bv = *reinterpret_cast<const float4v*>(p.bias + bsel + co);
                v += bv;
            }
            if (p.slope) {
                float4v sv = *reinterpret_cast<const float4v*>(p.slope + co);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sv[e];
            }
            const size_t o = (size_t)m * p.Cout + co;
            if (p.res) {
                half4 rv = *reinterpret_cast<const half4*>(p.res + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            }
            half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *reinterpret_cast<half4*>(p.y + o) = hv;
        }
    }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// GLDS: stage tiles with buffer_load ... lds (16 B per lane straight into LDS, no VGPR round trip and
// no ds_write): one wave-instruction fills 8 rows x 128 B; the XOR swizzle moves to the SOURCE chunk.
template <int WN, bool SMALL_CIN, bool GLDS>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)      // device-only builtins / LDS address-space casts: keep the host pass to the stub
    constexpr int WP = 4 / WN;
    constexpr int BN = 64 * WN;      // couts per block
    constexpr int BM = 64 * WP;      // pixels per block
    constexpr int WROWS = BN / 32;   // 16-B chunk rows per thread (weights)
    constexpr int XROWS = BM / 32;
    __shared__ __attribute__((aligned(16))) half_t lds[2 * (BN + BM) * BK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave % WN, wp = wave / WN;
    const int cout0 = blockIdx.y * BN;
    const int m0 = blockIdx.x * BM;
    // loader mapping: thread -> (row, 16-B chunk) of a 128-B LDS row.  Register staging: rows tid/8 + 32*i.
    // GLDS: wave w fills row groups w*ROWS + i (8 rows each), lane -> row lane/8, LDS chunk lane%8, and
    // reads the source chunk (lane%8) ^ (row & 7) so that the linear DMA image IS the swizzled image.
    const int trow = GLDS ? (lane >> 3) : (tid >> 3);
    const int tchunk = GLDS ? ((lane & 7) ^ ((lane >> 3) & 7)) : (tid & 7);
    auto wrow = [&](int i) { return GLDS ? (wave * WROWS + i) * 8 + trow : trow + 32 * i; };
    auto xrow = [&](int i) { return GLDS ? (wave * XROWS + i) * 8 + trow : trow + 32 * i; };

    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);

    // ---- per-thread gather state for its XROWS pixels
    int xbase[XROWS];   // byte offset of (n, ho*s-pad, wo*s-pad, chunk*8)
    int xhw[XROWS];     // (base_h << 16) | (base_w & 0xffff); base_h = -32768 marks an invalid row
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < XROWS; ++i) {
        int m = m0 + xrow(i);
        if (m < p.M) {
            int n = m / HoWo, r = m - n * HoWo;
            int ho = r / p.Wo, wo = r - ho * p.Wo;
            int bh = ho * p.stride - p.pad, bw = wo * p.stride - p.pad;
            xbase[i] = (((n * p.H + bh) * p.W + bw) * p.Cin + (SMALL_CIN ? 0 : tchunk * 8)) * 2;
            xhw[i] = (bh << 16) | (bw & 0xffff);
        } else {
            xbase[i] = 0;
            xhw[i] = (int)0x80000000u;
        }
    }
    unsigned wbase[WROWS];
#pragma unroll
    for (int i = 0; i < WROWS; ++i) wbase[i] = ((unsigned)(cout0 + wrow(i)) * p.K + tchunk * 8) * 2;
    // second input (1x1 tap at the output pixel's stride position): same pixel raster, its own channel count
    __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x), 0, p.x2 ? p.x2bytes : 0u, 0x00020000);
    auto x2off = [&](int i, int c0) -> unsigned {
        if (xhw[i] == (int)0x80000000u) return 0x80000000u;
        const int bh = (xhw[i] >> 16) + p.pad, bw = (int)(short)(xhw[i] & 0xffff) + p.pad;      // = ho * stride, wo * stride
        const int n = (m0 + xrow(i)) / HoWo;
        return (unsigned)((((n * p.H + bh) * p.W + bw) * p.C2 + c0 + tchunk * 8) * 2);
    };

    // K range of this block (split-K over blockIdx.z)
    int ks = 0, ke = p.nk;
    if (p.splitk > 1) {
        int per = (p.nk + p.splitk - 1) / p.splitk;
        ks = blockIdx.z * per;
        ke = min(p.nk, ks + per);
    }

    int4v wreg[WROWS], xreg[XROWS];
    const int cin_steps = SMALL_CIN ? 1 : (p.Cin / BK);

    auto gload_lds = [&](int s, int buf) {
        int kh, kw, tapoff;
        if (SMALL_CIN) {
            int tap = s * 8 + tchunk;
            kh = tap / p.KW; kw = tap - kh * p.KW;
            if (tap >= p.KH * p.KW) kh = 1 << 14;
            tapoff = (kh * p.W + kw) * p.Cin * 2;
        } else {
            int tap = s / cin_steps, c0 = (s - tap * cin_steps) * BK;
            kh = tap / p.KW; kw = tap - kh * p.KW;
            tapoff = ((kh * p.W + kw) * p.Cin + c0) * 2;
        }
        half_t* wl = lds + buf * (BN + BM) * BK;
        half_t* xl = wl + BN * BK;
#pragma unroll
        for (int i = 0; i < WROWS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(wl + (wave * WROWS + i) * 8 * BK), 16,
                                                     wbase[i] + (unsigned)s * (BK * 2), 0, 0, 0);
        if (!SMALL_CIN && s >= p.nk_main) {                          // block-uniform: the second input's K steps
#pragma unroll
            for (int i = 0; i < XROWS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x2rs, (lds_ptr_t)(xl + (wave * XROWS + i) * 8 * BK), 16,
                                                         x2off(i, (s - p.nk_main) * BK), 0, 0, 0);
            return;
        }
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            int hi = (xhw[i] >> 16) + kh, wi = (int)(short)(xhw[i] & 0xffff) + kw;
            bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            unsigned off = ok ? (unsigned)(xbase[i] + tapoff) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(xl + (wave * XROWS + i) * 8 * BK), 16, off, 0, 0, 0);
        }
    };
    auto gload = [&](int s) {
        int kh, kw, tapoff;
        if (SMALL_CIN) {
            int tap = s * 8 + tchunk;              // one 16-B chunk (8 padded channels) per tap
            kh = tap / p.KW; kw = tap - kh * p.KW;
            if (tap >= p.KH * p.KW) kh = 1 << 14;  // beyond the filter: force out of range
            tapoff = (kh * p.W + kw) * p.Cin * 2;
        } else {
            int tap = s / cin_steps, c0 = (s - tap * cin_steps) * BK;
            kh = tap / p.KW; kw = tap - kh * p.KW;
            tapoff = ((kh * p.W + kw) * p.Cin + c0) * 2;
        }
#pragma unroll
        for (int i = 0; i < WROWS; ++i) wreg[i] = buf_load16(wrs, wbase[i] + (unsigned)s * (BK * 2));
        if (!SMALL_CIN && s >= p.nk_main) {
#pragma unroll
            for (int i = 0; i < XROWS; ++i) xreg[i] = buf_load16(x2rs, x2off(i, (s - p.nk_main) * BK));
            return;
        }
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            int hi = (xhw[i] >> 16) + kh, wi = (int)(short)(xhw[i] & 0xffff) + kw;
            bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            unsigned off = ok ? (unsigned)(xbase[i] + tapoff) : 0x80000000u;
            xreg[i] = buf_load16(xrs, off);
        }
    };
    auto lstore = [&](int buf) {
        half_t* wl = lds + buf * (BN + BM) * BK;
        half_t* xl = wl + BN * BK;
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            int row = trow + 32 * i;
            *reinterpret_cast<int4v*>(wl + row * BK + ((tchunk ^ (row & 7)) << 3)) = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            int row = trow + 32 * i;
            *reinterpret_cast<int4v*>(xl + row * BK + ((tchunk ^ (row & 7)) << 3)) = xreg[i];
        }
    };

    float4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    if (ks < ke) {
        if (GLDS) {
            gload_lds(ks, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            gload(ks);
            lstore(0);
        }
    }
    __syncthreads();
    for (int s = ks; s < ke; ++s) {
        const int buf = (s - ks) & 1;
        if (s + 1 < ke) {
            if (GLDS) gload_lds(s + 1, buf ^ 1); else gload(s + 1);
        }
        const half_t* wl = lds + buf * (BN + BM) * BK + (wc * 64) * BK;
        const half_t* xl = lds + buf * (BN + BM) * BK + BN * BK + (wp * 64) * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 a[4], b[4];
            const int ch = kk * 4 + fq;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = i * 16 + fr;
                a[i] = *reinterpret_cast<const half8*>(wl + row * BK + ((ch ^ (row & 7)) << 3));
                b[i] = *reinterpret_cast<const half8*>(xl + row * BK + ((ch ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (s + 1 < ke) lstore(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<WN>(p, acc, cout0, m0, wc, wp, fr, fq);
#endif
}

// ---------------------------------------------------------------------------------------------
// Four-stage ring variant (NOT on the product path: measured slower than the two-stage kernel above, compiled into
// the debug build only, FR_CONV_KERNEL=2): K step 32, ring of 4 LDS stages (16 KB each, 64 KB/block,
// 2 blocks/CU), tiles staged by LDS-DMA (buffer_load ... lds) that stay in flight ACROSS barriers:
// per step  s_waitcnt vmcnt(8) [stage s landed; s+1, s+2 may still fly] -> raw s_barrier ->
// issue stage s+3 into the buffer freed by step s-1 -> 8 ds_read_b128 + 16 MFMA on stage s.
// LDS rows are 64 B (4 chunks of 16 B); chunk index XOR swz[(row>>2)&3], swz = {0,2,3,1}: the four
// 16-lane groups of a ds_read_b128 each touch 16 distinct 16-B slots of the 256-B bank row.
// One LDS-DMA wave-instruction fills 16 rows; the swizzle is applied to the SOURCE chunk.
#define PK 32
#define PNS 4
template <int WN, bool SMALL_CIN>
__global__ __launch_bounds__(256, 2) void conv_mfma_pipe(ConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WP = 4 / WN;
    constexpr int BN = 64 * WN, BM = 64 * WP;
    constexpr int WI = BN / 64, XI = BM / 64;            // LDS-DMA instructions per thread per stage (16 rows each)
    constexpr int NLD = WI + XI;                          // = 4
    constexpr int STAGE = (BN + BM) * PK;                 // halves per stage
    __shared__ __attribute__((aligned(16))) half_t lds[PNS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave % WN, wp = wave / WN;
    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so XCD x walks the
    // contiguous tile range [x*per, ...) with the cout tile innermost: co-resident blocks of one XCD
    // then share input rows / halos and weights in that XCD's L2 (speed only, any placement is correct).
    int tile;
    {
        const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int ntn = p.Cout / BN;
    const int cout0 = (tile % ntn) * BN, m0 = (tile / ntn) * BM;
    const int lrow = lane >> 2;                                              // row inside a 16-row group
    const int swz_l = (0x1320 >> (4 * ((lane >> 4) & 3))) & 3;               // swz[(row>>2)&3] of the loader row
    const int schunk = (lane & 3) ^ swz_l;                                   // source chunk of this lane

    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);

    int xbase[XI], xhw[XI];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        int m = m0 + (wave * XI + i) * 16 + lrow;
        if (m < p.M) {
            int n = m / HoWo, r = m - n * HoWo;
            int ho = r / p.Wo, wo = r - ho * p.Wo;
            int bh = ho * p.stride - p.pad, bw = wo * p.stride - p.pad;
            xbase[i] = (((n * p.H + bh) * p.W + bw) * p.Cin + (SMALL_CIN ? 0 : schunk * 8)) * 2;
            xhw[i] = (bh << 16) | (bw & 0xffff);
        } else {
            xbase[i] = 0;
            xhw[i] = (int)0x80000000u;
        }
    }
    unsigned wbase[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) wbase[i] = ((unsigned)(cout0 + (wave * WI + i) * 16 + lrow) * p.K + schunk * 8) * 2;

    int ks = 0, ke = p.nk;                                 // nk counts 32-wide steps here
    if (p.splitk > 1) {
        int per = (p.nk + p.splitk - 1) / p.splitk;
        ks = blockIdx.z * per;
        ke = min(p.nk, ks + per);
    }
    const int cin_steps = SMALL_CIN ? 1 : (p.Cin / PK);

    auto issue = [&](int s) {                              // LDS-DMA of K step s into ring slot (s - ks) % PNS
        int kh, kw, tapoff;
        if (SMALL_CIN) {
            int tap = s * 4 + schunk;
            kh = tap / p.KW; kw = tap - kh * p.KW;
            if (tap >= p.KH * p.KW) kh = 1 << 14;
            tapoff = (kh * p.W + kw) * p.Cin * 2;
        } else {
            int tap = s / cin_steps, c0 = (s - tap * cin_steps) * PK;
            kh = tap / p.KW; kw = tap - kh * p.KW;
            tapoff = ((kh * p.W + kw) * p.Cin + c0) * 2;
        }
        half_t* wl = lds + ((s - ks) % PNS) * STAGE;
        half_t* xl = wl + BN * PK;
#pragma unroll
        for (int i = 0; i < WI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(wl + (wave * WI + i) * 16 * PK), 16,
                                                     wbase[i] + (unsigned)s * (PK * 2), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int hi = (xhw[i] >> 16) + kh, wi = (int)(short)(xhw[i] & 0xffff) + kw;
            bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            unsigned off = ok ? (unsigned)(xbase[i] + tapoff) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(xl + (wave * XI + i) * 16 * PK), 16, off, 0, 0, 0);
        }
    };

    float4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int rdoff = fr * PK + ((fq ^ ((0x1320 >> (4 * ((fr >> 2) & 3))) & 3)) << 3);   // halves, within a 16-row tile
#pragma unroll
    for (int d = 0; d < PNS - 1; ++d)
        if (ks + d < ke) issue(ks + d);
    for (int s = ks; s < ke; ++s) {
        const int rem = ke - 1 - s;                        // younger stages that exist
        if (rem >= PNS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD * (PNS - 2)) : "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + PNS - 1 < ke) issue(s + PNS - 1);
        const half_t* wl = lds + ((s - ks) % PNS) * STAGE + (wc * 64) * PK;
        const half_t* xl = lds + ((s - ks) % PNS) * STAGE + BN * PK + (wp * 64) * PK;
        half8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = *reinterpret_cast<const half8*>(wl + i * 16 * PK + rdoff);
            b[i] = *reinterpret_cast<const half8*>(xl + i * 16 * PK + rdoff);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    conv_epilogue<WN>(p, acc, cout0, m0, wc, wp, fr, fq);
#endif
}

// Generic-path variant (layers the halo kernel does not take): FR_CONV_KERNEL = 1 (default) two-stage
// LDS-DMA loader, K step 64; 2 = four-stage LDS-DMA ring, K step 32 (slower: a barrier per 16 MFMAs and ~100
// cycles of issue per LDS-DMA piece outweigh the deeper prefetch); 0 = two-stage register-staged loader.
static int conv_kernel_choice() { return fr_dbg_int("FR_CONV_KERNEL", 1); }      // product build: always 1

template <int WN, bool SMALL>
static void launch_conv(const ConvP& p, hipStream_t s) {
    constexpr int BN = 64 * WN, BM = 64 * (4 / WN);
    dim3 grid((p.M + BM - 1) / BM, p.Cout / BN, p.splitk > 1 ? p.splitk : 1);
    if constexpr (FR_DEBUG) {                  // measured alternatives, debug build only (none of them takes a second input)
        const int which = p.x2 ? 1 : conv_kernel_choice();
        if (which == 2) {
            ConvP q = p;
            q.nk = p.K / PK;
            dim3 g1(grid.x * grid.y, 1, grid.z);
            conv_mfma_pipe<WN, SMALL><<<g1, 256, 0, s>>>(q);
            return;
        }
        if (which == 0) { conv_mfma_kernel<WN, SMALL, false><<<grid, 256, 0, s>>>(p); return; }
    }
    conv_mfma_kernel<WN, SMALL, true><<<grid, 256, 0, s>>>(p);
}

int fr_conv_halo_try(const fr_conv_args* a, hipStream_t s);     // conv_halo.hip
int fr_conv_stem_try(const fr_conv_args* a, hipStream_t s);     // conv_stem.hip

static bool conv_halo_enabled() { return fr_dbg_int("FR_CONV_HALO", 1) != 0; }   // debug build: FR_CONV_HALO=0 for A/B

extern "C" int fr_conv_nhwc_f16(const fr_conv_args* a, fr_stream_t stream) {
    FR_REQUIRE(a, "fr_conv_nhwc_f16: null args");
    FR_REQUIRE(a->x && a->w && (a->y || a->out_f32_partial), "fr_conv_nhwc_f16: null tensor");
    FR_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->KH > 0 && a->KW > 0 && a->stride > 0 && a->pad >= 0,
               "fr_conv_nhwc_f16: bad geometry");
    FR_REQUIRE(a->Ho == (a->H + 2 * a->pad - a->KH) / a->stride + 1 && a->Wo == (a->W + 2 * a->pad - a->KW) / a->stride + 1,
               "fr_conv_nhwc_f16: Ho/Wo do not match the geometry");
    FR_REQUIRE(a->Cout % 64 == 0, "fr_conv_nhwc_f16: Cout must be a multiple of 64 (got %d)", a->Cout);
    const bool small = a->Cin == 8;
    FR_REQUIRE(small || a->Cin % 64 == 0, "fr_conv_nhwc_f16: Cin must be 8 or a multiple of 64 (got %d)", a->Cin);
    FR_REQUIRE(a->bias_mode == 0 || (a->bias_mode == 1 && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 &&
                                     a->Ho >= 2 && a->Wo >= 2),
               "fr_conv_nhwc_f16: bias_mode 1 needs a 3x3/s1/p1 conv");
    FR_REQUIRE(a->H < 32768 && a->W < 32768, "fr_conv_nhwc_f16: H/W too large");
    ConvP p;
    p.x = (const half_t*)a->x; p.w = (const half_t*)a->w; p.y = (half_t*)a->y;
    p.bias = a->bias; p.slope = a->slope; p.res = (const half_t*)a->residual; p.partial = a->out_f32_partial;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW;
    p.stride = a->stride; p.pad = a->pad; p.Ho = a->Ho; p.Wo = a->Wo; p.bias_mode = a->bias_mode;
    p.splitk = a->splitk > 1 ? a->splitk : 1;
    FR_REQUIRE(p.splitk == 1 || p.partial, "fr_conv_nhwc_f16: splitk > 1 needs out_f32_partial");
    int64_t M = (int64_t)a->B * a->Ho * a->Wo;
    int64_t xbytes = (int64_t)a->B * a->H * a->W * a->Cin * 2;
    int Kreal = a->KH * a->KW * a->Cin;
    FR_REQUIRE(!a->x2 || (!small && a->C2 > 0 && a->C2 % 64 == 0 && (a->H - 1) / a->stride + 1 >= a->Ho && (a->W - 1) / a->stride + 1 >= a->Wo),
               "fr_conv_nhwc_f16: x2 needs Cin %% 64 == 0, C2 %% 64 == 0 and every (ho * stride, wo * stride) inside the image");
    p.x2 = (const half_t*)a->x2; p.C2 = a->x2 ? a->C2 : 0;
    p.K = small ? ((Kreal + 127) / 128) * 128 : Kreal + p.C2;
    p.nk_main = Kreal / BK;
    int64_t x2bytes = (int64_t)a->B * a->H * a->W * p.C2 * 2;
    FR_REQUIRE(x2bytes < (1ll << 31), "fr_conv_nhwc_f16: x2 too large for 32-bit buffer offsets");
    p.x2bytes = (unsigned)x2bytes;
    int64_t wbytes = (int64_t)a->Cout * p.K * 2;
    FR_REQUIRE(M < (1ll << 31) && xbytes < (1ll << 31) && wbytes < (1ll << 31) && M * a->Cout * 2 < (1ll << 40),
               "fr_conv_nhwc_f16: tensor too large for 32-bit buffer offsets (split the batch)");
    p.M = (int)M; p.nk = p.K / BK; p.xbytes = (unsigned)xbytes; p.wbytes = (unsigned)wbytes;
    hipStream_t s = fr_stream(stream);
    if (conv_halo_enabled()) {
        int h = fr_conv_halo_try(a, s);
        if (h < 0) return h;
        if (h == 1) {
            FR_CHECK_LAUNCH("conv_halo_kernel");
            return FR_OK;
        }
    }
    if (small) {                                   // the packed stem has its own kernel (conv_stem.hip)
        if (fr_conv_stem_try(a, s) == 1) {
            FR_CHECK_LAUNCH("conv_stem_kernel");
            return FR_OK;
        }
    }
    // tile choice: 64-cout layers use 64x256 tiles, the rest 128x128
    if (small) {
        if (a->Cout % 128 == 0) launch_conv<2, true>(p, s); else launch_conv<1, true>(p, s);
    } else if (a->Cout % 128 == 0) {
        launch_conv<2, false>(p, s);
    } else {
        launch_conv<1, false>(p, s);
    }
    FR_CHECK_LAUNCH("conv_mfma_kernel");
    return FR_OK;
}

// ---- split-K tail of an ordinary conv (small batches: too few output tiles to fill 256 CUs, so K is cut into
// splitk slices that run side by side): sum the f32 partials, then the same epilogue as the fused path
__global__ void conv_splitk_epilogue(const float* __restrict__ partial, int splitk, int M, int Cout, int Ho, int Wo,
                                     const float* __restrict__ bias, int bias_mode, const float* __restrict__ slope,
                                     const half_t* __restrict__ res, half_t* __restrict__ y) {
    const int q4 = Cout / 4;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)M * q4) return;
    const int m = (int)(e / q4), co = (int)(e - (int64_t)m * q4) * 4;
    float4v v = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splitk; ++z) v += *reinterpret_cast<const float4v*>(partial + ((size_t)z * M + m) * Cout + co);
    if (bias) {
        int bsel = 0;
        if (bias_mode == 1) {
            const int r = m % (Ho * Wo), ho = r / Wo, wo = r - ho * Wo;
            const int rc = ho == 0 ? 0 : (ho == Ho - 1 ? 2 : 1), cc = wo == 0 ? 0 : (wo == Wo - 1 ? 2 : 1);
            bsel = (rc * 3 + cc) * Cout;
        }
        v += *reinterpret_cast<const float4v*>(bias + bsel + co);
    }
    if (slope) {
        const float4v sv = *reinterpret_cast<const float4v*>(slope + co);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : v[c] * sv[c];
    }
    if (res) {
        const half4 rv = *reinterpret_cast<const half4*>(res + (size_t)m * Cout + co);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += (float)rv[c];
    }
    *reinterpret_cast<half4*>(y + (size_t)m * Cout + co) = half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
}

extern "C" int fr_conv_splitk_epilogue(const float* partial, int splitk, int M, int Cout, int Ho, int Wo,
                                       const float* bias, int bias_mode, const float* slope, const void* residual,
                                       void* y, fr_stream_t stream) {
    FR_REQUIRE(partial && y && splitk >= 1 && M > 0 && Cout > 0 && Cout % 4 == 0 && Ho > 0 && Wo > 0,
               "fr_conv_splitk_epilogue: bad argument");
    FR_REQUIRE(bias_mode == 0 || (bias_mode == 1 && bias && Ho >= 2 && Wo >= 2), "fr_conv_splitk_epilogue: bad bias mode");
    const int64_t n = (int64_t)M * (Cout / 4);
    conv_splitk_epilogue<<<(unsigned)((n + 255) / 256), 256, 0, fr_stream(stream)>>>(
        partial, splitk, M, Cout, Ho, Wo, bias, bias_mode, slope, (const half_t*)residual, (half_t*)y);
    FR_CHECK_LAUNCH("conv_splitk_epilogue");
    return FR_OK;
}

extern "C" int fr_conv_sequence(const fr_conv_step* steps, int nsteps, fr_stream_t stream) {
    FR_REQUIRE(steps && nsteps > 0, "fr_conv_sequence: no steps");
    for (int i = 0; i < nsteps; ++i) {
        const fr_conv_step& st = steps[i];
        int rc;
        if (st.kind == 0) {
            rc = fr_conv_nhwc_f16(&st.args, stream);
        } else if (st.kind == 1) {
            const fr_conv_args& a = st.args;
            FR_REQUIRE(a.splitk > 1 && a.out_f32_partial && a.y, "fr_conv_sequence: step %d: split-K step without splitk / partial / y", i);
            fr_conv_args p = a;                         // the partials launch: no epilogue operands
            p.y = nullptr; p.bias = nullptr; p.slope = nullptr; p.residual = nullptr; p.bias_mode = 0;
            rc = fr_conv_nhwc_f16(&p, stream);
            if (rc == FR_OK)
                rc = fr_conv_splitk_epilogue(a.out_f32_partial, a.splitk, a.B * a.Ho * a.Wo, a.Cout, a.Ho, a.Wo, a.bias,
                                             a.bias_mode, a.slope, a.residual, a.y, stream);
        } else if (st.kind == 2) {
            rc = fr_conv_inblock_f16(&st.args, stream);
        } else {
            FR_REQUIRE(false, "fr_conv_sequence: step %d: unknown kind %d", i, st.kind);
        }
        if (rc != FR_OK) return rc;
    }
    return FR_OK;
}

// ---- FC tail: reduce split-K partials + bias -> embedding; L2-normalise (one wave per face)
__global__ void fc_reduce_l2norm(const float* __restrict__ partial, int splitk, int B, int dim,
                                 const float* __restrict__ bias, float* __restrict__ emb, float* __restrict__ normed) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    // Same sums in the same order as the plain loops (bias + slice 0 + slice 1 ...; squares chunk by chunk), but the slices'
    // loads are issued eight at a time instead of one dependent round trip per slice, and the embedding stays in registers for
    // the normalisation instead of being read back: a single face's tail was 15.6 us, all of it latency.
    float ss = 0.f;
    constexpr int MAXC = 4;                       // column chunks a lane keeps (dim <= 1024); larger dims take the plain path
    float4v keep[MAXC];
    const bool small = dim <= 256 * MAXC;
    int kc = 0;
    for (int c = lane * 4; c < dim; c += 256, ++kc) {
        float4v v = *reinterpret_cast<const float4v*>(bias + c);
        const float* p = partial + (size_t)row * dim + c;
        const size_t zs = (size_t)B * dim;
        int z = 0;
        for (; z + 8 <= splitk; z += 8) {
            float4v t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4v*>(p + (size_t)(z + u) * zs);
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; z < splitk; ++z) v += *reinterpret_cast<const float4v*>(p + (size_t)z * zs);
        *reinterpret_cast<float4v*>(emb + (size_t)row * dim + c) = v;
        if (small) {
#pragma unroll
            for (int u = 0; u < MAXC; ++u) if (u == kc) keep[u] = v;
        }
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    kc = 0;
    for (int c = lane * 4; c < dim; c += 256, ++kc) {
        float4v v;
        if (small) {
            v = keep[0];
#pragma unroll
            for (int u = 1; u < MAXC; ++u) if (u == kc) v = keep[u];
        } else {
            v = *reinterpret_cast<const float4v*>(emb + (size_t)row * dim + c);
        }
        v[0] /= nrm; v[1] /= nrm; v[2] /= nrm; v[3] /= nrm;
        *reinterpret_cast<float4v*>(normed + (size_t)row * dim + c) = v;
    }
}

extern "C" int fr_fc_reduce_l2norm(const float* partial, int splitk, int B, int dim, const float* bias,
                                   float* embedding, float* normed, fr_stream_t stream) {
    if (B <= 0) return FR_OK;
    FR_REQUIRE(partial && bias && embedding && normed, "fr_fc_reduce_l2norm: null pointer");
    FR_REQUIRE(splitk >= 1 && dim > 0 && dim % 4 == 0, "fr_fc_reduce_l2norm: bad splitk/dim");
    fc_reduce_l2norm<<<fr_cdiv(B, 4), 256, 0, fr_stream(stream)>>>(partial, splitk, B, dim, bias, embedding, normed);
    FR_CHECK_LAUNCH("fc_reduce_l2norm");
    return FR_OK;
}

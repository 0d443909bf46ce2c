// 3x3 / stride 1 / pad 1 convs of a forward of ONE to EIGHT faces (the embed half of a single-frame FaceAnalysis.get,
// /root/reference/infrenceServer.py:528): split along K INSIDE a workgroup, one launch per conv.
//
// Why: a single face's 14x14x256 conv is 231 MFLOP over 72 K steps; cut into slices that run side by side on different
// CUs (conv_mfma.hip's split-K mode) it is two launches - the partials, then fr_conv_splitk_epilogue - of 7.8 + 4.7 us,
// both at their launch-latency floor, 89 times per forward.  Here a workgroup is sixteen waves that own ONE small output
// tile (16 pixels x 32 couts) and a sixteenth of K each (K step ks goes to wave ks % 16): a wave issues ALL its operand
// loads at once - straight from global memory into MFMA fragments, no LDS staging: nothing is shared between waves -
// runs its 2 x (K steps) MFMAs, parks its f32 partial tile in LDS (2 KB), and after ONE barrier 128 threads add the
// sixteen partials in wave order and apply the fused epilogue ((border-class) bias -> PReLU -> + residual -> f16).  The
// tile is small on purpose: a CU's load path (~64 B / clock) is what bounds a workgroup that pulls a tile's whole K
// (16 px x 32 couts x 2304: 221 KB), and 14x14x256 is 104 such workgroups - one round over the CUs.
// GEMM view: D[cout][pixel] = sum_k W[cout][k] X[pixel][k], k = (tap, channel); A = weights (row = cout lane & 15,
// k = 8 (lane >> 4) .. + 7 of the K step), B = pixels (a tap's 32 channels; a tap outside the image reads zeros through
// an out-of-range buffer offset); a lane ends up with one pixel and 4 consecutive couts per cout tile.
// Results differ from the other batch-size modes by f32 summation order only (one rounding to f16 per output, as everywhere).
#include "common.h"

namespace {

struct InblockP {
    const half_t* x; const half_t* w; half_t* y;
    const float* bias; const float* slope; const half_t* res;
    int B, H, W, Cin, Cout, bias_mode;
    int M, K, nks, cpt;                   // M = B*H*W, K = 9*Cin, nks = K / 32, cpt = Cin / 32 (K steps per tap)
    unsigned xbytes, wbytes;
};

constexpr int IB_TN = 32, IB_WAVES = 16, IB_MAXS = 9;

// PT pixel tiles (of 16) per workgroup: 1 for one or two faces (14x14x256: 104 / 200 workgroups, one round over the CUs), 2 and 4 beyond
// (half the workgroups, each pixel fragment used by both cout tiles as before, each weight fragment by two pixel tiles).  The
// order in which an output element's products are summed does not depend on PT: the two forms give the same bits.
template <int PT>
__global__ __launch_bounds__(IB_WAVES * 64) void conv_inblock_kernel(InblockP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int IB_TM = 16 * PT, IB_RING = PT == 1 ? 5 : (PT == 2 ? 4 : 2);
    __shared__ __attribute__((aligned(16))) float part[IB_WAVES][IB_TM][IB_TN];      // 32 / 64 / 128 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * IB_TM, n0 = blockIdx.y * IB_TN;
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);

    // the epilogue threads fetch their residual first: it is the longest-latency operand of the tail
    const int epx = tid % IB_TM, ecq = tid / IB_TM;                // tid < 8 IB_TM: pixel epx, couts 4 ecq .. + 3 of the tile
    const int em = m0 + epx;
    half4 rv = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    const bool etail = tid < 8 * IB_TM && em < p.M;
    if (etail && p.res) rv = *reinterpret_cast<const half4*>(p.res + (size_t)em * p.Cout + n0 + 4 * ecq);
    // ... and their bias and slope rows: behind the barrier these were a second exposed round trip per launch
    float4v ebias = {0.f, 0.f, 0.f, 0.f}, eslope = {1.f, 1.f, 1.f, 1.f};
    if (etail) {
        const int co_ = n0 + 4 * ecq;
        if (p.bias) {
            int bsel = 0;
            if (p.bias_mode == 1) {
                const int hw_ = p.H * p.W, er = em % hw_, ho = er / p.W, wo = er - ho * p.W;
                const int rc = ho == 0 ? 0 : (ho == p.H - 1 ? 2 : 1), cc = wo == 0 ? 0 : (wo == p.W - 1 ? 2 : 1);
                bsel = (rc * 3 + cc) * p.Cout;
            }
            ebias = *reinterpret_cast<const float4v*>(p.bias + bsel + co_);
        }
        if (p.slope) eslope = *reinterpret_cast<const float4v*>(p.slope + co_);
    }

    // this lane's pixels (B operand column fr of pixel tile j)
    const int hw = p.H * p.W;
    bool mv[PT]; int pn[PT], py[PT], px[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int m = m0 + 16 * j + fr;
        mv[j] = m < p.M;
        pn[j] = m / hw;
        const int r = m - pn[j] * hw;
        py[j] = r / p.W; px[j] = r - py[j] * p.W;
    }
    const unsigned wrow0 = (unsigned)((n0 + fr) * p.K + 8 * fq) * 2u, wrow1 = wrow0 + (unsigned)(16 * p.K) * 2u;
    const int ns = (p.nks - wave + IB_WAVES - 1) / IB_WAVES;      // K steps of this wave: wave, wave + 16, ...

    int4v a0[IB_RING], a1[IB_RING], bx[IB_RING][PT];
    auto load = [&](int i, int slot) {
        const int ks = wave + IB_WAVES * i;
        const int tap = ks / p.cpt, cg = ks - tap * p.cpt;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        a0[slot] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wrow0 + (unsigned)ks * 64u, 0, 0);
        a1[slot] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wrow1 + (unsigned)ks * 64u, 0, 0);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int yy = py[j] + dy, xx = px[j] + dx;
            const bool ok = mv[j] && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            const unsigned xo = ok ? (unsigned)(((pn[j] * p.H + yy) * p.W + xx) * p.Cin + cg * 32 + 8 * fq) * 2u : 0x80000000u;
            bx[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < IB_RING; ++i)
        if (i < ns) load(i, i);
    float4v acc[PT][2];
#pragma unroll
    for (int j = 0; j < PT; ++j) acc[j][0] = acc[j][1] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < IB_MAXS; ++i) {
        if (i < ns) {
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a0[i % IB_RING]), __builtin_bit_cast(half8, bx[i % IB_RING][j]), acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a1[i % IB_RING]), __builtin_bit_cast(half8, bx[i % IB_RING][j]), acc[j][1], 0, 0, 0);
            }
            if (i + IB_RING < ns) load(i + IB_RING, i % IB_RING);
        }
    }
    // lane: pixel fr of tile j, couts 4 fq .. + 3 of cout tile 0 / 1
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        *reinterpret_cast<float4v*>(&part[wave][16 * j + fr][4 * fq]) = acc[j][0];
        *reinterpret_cast<float4v*>(&part[wave][16 * j + fr][16 + 4 * fq]) = acc[j][1];
    }
    __syncthreads();
    if (!etail) return;
    float4v v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < IB_WAVES; ++w) v += *reinterpret_cast<const float4v*>(&part[w][epx][4 * ecq]);
    const int co = n0 + 4 * ecq;
    if (p.bias) v += ebias;
    if (p.slope) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = v[c] > 0.f ? v[c] : v[c] * eslope[c];
    }
    if (p.res) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += (float)rv[c];
    }
    *reinterpret_cast<half4*>(p.y + (size_t)em * p.Cout + co) = half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
#endif
}

}  // namespace

extern "C" int fr_conv_inblock_f16(const fr_conv_args* a, fr_stream_t stream) {
    FR_REQUIRE(a && a->x && a->w && a->y, "fr_conv_inblock_f16: null pointer");
    FR_REQUIRE(a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->Ho == a->H && a->Wo == a->W && !a->x2 &&
               !a->out_f32_partial, "fr_conv_inblock_f16: a 3x3 / stride 1 / pad 1 conv with one input and a fused epilogue");
    FR_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cin % 32 == 0 && a->Cin <= 512 && a->Cout > 0 && a->Cout % IB_TN == 0,
               "fr_conv_inblock_f16: Cin %% 32 == 0, Cin <= 512, Cout %% 32 == 0 (got %d, %d)", a->Cin, a->Cout);
    FR_REQUIRE(a->bias_mode == 0 || (a->bias_mode == 1 && a->bias && a->H >= 2 && a->W >= 2), "fr_conv_inblock_f16: bad bias mode");
    const int64_t M = (int64_t)a->B * a->H * a->W;
    FR_REQUIRE(M * a->Cin * 2 < (1ll << 31) && (int64_t)a->Cout * 9 * a->Cin * 2 < (1ll << 31) && M / 16 < 65536,
               "fr_conv_inblock_f16: tensor too large for this small-batch kernel (B %d)", a->B);
    InblockP p;
    p.x = (const half_t*)a->x; p.w = (const half_t*)a->w; p.y = (half_t*)a->y;
    p.bias = a->bias; p.slope = a->slope; p.res = (const half_t*)a->residual;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.bias_mode = a->bias_mode;
    p.M = (int)M; p.K = 9 * a->Cin; p.nks = p.K / 32; p.cpt = a->Cin / 32;
    p.xbytes = (unsigned)(M * a->Cin * 2); p.wbytes = (unsigned)((int64_t)a->Cout * p.K * 2);
    static_assert(IB_MAXS * IB_WAVES >= 9 * 512 / 32, "every K step of the largest conv has a wave and a slot");
    // one pixel tile per workgroup while that is at most one round over the CUs' worth of workgroups, then two, then four
    const int64_t wg1 = ((M + 15) / 16) * (a->Cout / IB_TN), wg2 = ((M + 31) / 32) * (a->Cout / IB_TN);
    if (wg1 <= 256) {
        dim3 grid((unsigned)((M + 15) / 16), (unsigned)(a->Cout / IB_TN));
        conv_inblock_kernel<1><<<grid, IB_WAVES * 64, 0, fr_stream(stream)>>>(p);
    } else if (wg2 <= 256) {
        dim3 grid((unsigned)((M + 31) / 32), (unsigned)(a->Cout / IB_TN));
        conv_inblock_kernel<2><<<grid, IB_WAVES * 64, 0, fr_stream(stream)>>>(p);
    } else {
        dim3 grid((unsigned)((M + 63) / 64), (unsigned)(a->Cout / IB_TN));
        conv_inblock_kernel<4><<<grid, IB_WAVES * 64, 0, fr_stream(stream)>>>(p);
    }
    FR_CHECK_LAUNCH("conv_inblock_kernel");
    return FR_OK;
}

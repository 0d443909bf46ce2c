// ArcFace IResNet stem: 3x3 / stride 1 / pad 1 conv of the packed crop (f16 [B,H,W,8]: RGB in channels 0..2, zeros
// in 3..7 = one 16-byte pixel) to 64 channels, + bias + PReLU (the first conv inside FaceAnalysis.get,
// /root/reference/infrenceServer.py:528; weights in the packing of iresnet.py: [64][16 taps][8 channels] f16, taps
// 9..15 zero).  On the generic implicit-GEMM kernel this layer took 296 us per 256 faces against ~95 us of HBM time
// (411 MB of output): there every K step gathered 16-byte pieces per (pixel, tap) through LDS-DMA and wrote 8-byte
// pieces.  Here a block stages the input rows of its tile ONCE (4 output rows + halo: 6 x (W + 2) pixels x 16 B),
// the whole weight tensor lives in registers (12 fragments per lane), a K step is 4 taps x 8 channels so that a
// lane's B fragment is ONE ds_read_b128 of the tap's pixel, and the output leaves through an LDS transpose as whole
// 128-byte pixel rows.  v_mfma_f32_16x16x32_f16, A = weights (a lane owns one pixel and 4 consecutive couts per tile).
#include "common.h"

namespace {

constexpr int ST_TR = 4;                 // output rows per block

struct StemP {
    const half_t* x; const half_t* w; const float* bias; const float* slope; half_t* y;
    int B, H, W;
};

template <int WMAX>
__global__ __launch_bounds__(256) void conv_stem_kernel(StemP p) {
    constexpr int IWMAX = WMAX + 2;
    __shared__ __attribute__((aligned(16))) int4v xin[(ST_TR + 2) * IWMAX];      // one 16-byte pixel per entry
    constexpr int SROW = 72;                                                     // halves per staged pixel row: 64 + 8 (bank shift)
    __shared__ __attribute__((aligned(16))) half_t stage[4][16 * SROW];          // per wave: [16 pixels][64 couts]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int rows_per_img = p.H / ST_TR;
    const int n = blockIdx.x / rows_per_img, y0 = (blockIdx.x - n * rows_per_img) * ST_TR;
    const int IW = p.W + 2;

    // ---- input rows y0-1 .. y0+ST_TR (zero outside the image), columns -1 .. W
    const int4v* xg = reinterpret_cast<const int4v*>(p.x) + (int64_t)n * p.H * p.W;
    for (int e = tid; e < (ST_TR + 2) * IW; e += 256) {
        const int r = e / IW, c = e - r * IW;
        const int iy = y0 - 1 + r, ix = c - 1;
        int4v v = {0, 0, 0, 0};
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = xg[iy * p.W + ix];
        xin[r * IW + c] = v;
    }
    // ---- weights: K step s covers taps 4s .. 4s+3 (x 8 channels); lane (cout fr of tile ct, quarter fq) holds tap 4s+fq
    half8 wf[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
            wf[s][ct] = *reinterpret_cast<const half8*>(p.w + (ct * 16 + fr) * 128 + (4 * s + fq) * 8);
    // the lane's 4 consecutive couts of each cout tile: ct*16 + 4*fq + e
    float4v bias_r[4], slope_r[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        bias_r[ct] = *reinterpret_cast<const float4v*>(p.bias + ct * 16 + fq * 4);
        slope_r[ct] = *reinterpret_cast<const float4v*>(p.slope + ct * 16 + fq * 4);
    }
    // tap of this lane per K step -> LDS offset of the tap inside the window (taps 9..11: weights are zero; read tap 8's
    // pixel, finite data, instead of anything that might be uninitialised)
    int toff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int tap = min(4 * s + fq, 8);
        toff[s] = (tap / 3) * IW + (tap % 3);
    }
    __syncthreads();

    const int ntile = ST_TR * p.W / 16;                       // 16-pixel tiles of the block (W % 16 == 0: no row straddle)
    half_t* st = stage[wave];
    for (int t = wave; t < ntile; t += 4) {
        const int p0 = t * 16, ry = p0 / p.W, cx0 = p0 - ry * p.W;
        const int base = ry * IW + cx0 + fr;                  // top-left of the lane's pixel window
        float4v acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int4v raw = xin[base + toff[s]];
            half8 b;
            __builtin_memcpy(&b, &raw, 16);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][ct], b, acc[ct], 0, 0, 0);
        }
        // ---- bias + PReLU, one rounding to f16; transpose through LDS so that a pixel's 64 couts leave as 128 B
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            float4v v = acc[ct] + bias_r[ct];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope_r[ct][e];
            *reinterpret_cast<half4*>(st + fr * SROW + ct * 16 + fq * 4) = half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        }
        // (a wave's own LDS writes are visible to its own later reads in program order: no barrier inside the wave)
        half_t* yo = p.y + ((int64_t)n * p.H * p.W + (int64_t)(y0 + ry) * p.W + cx0) * 64;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = lane + k * 64;                  // 128 pieces of 16 B: pixel = piece / 8, chunk = piece % 8
            *reinterpret_cast<int4v*>(yo + piece * 8) = *reinterpret_cast<const int4v*>(st + (piece >> 3) * SROW + (piece & 7) * 8);
        }
    }
}

}  // namespace

// fr_conv_nhwc_f16 hands the packed stem (Cin == 8, 3x3 / s1 / p1, Cout == 64, bias + PReLU, no residual) to this
// kernel when the geometry fits; returns 1 when it launched, 0 when the generic kernel should run
int fr_conv_stem_try(const fr_conv_args* a, hipStream_t s) {
    if (!(a->Cin == 8 && a->Cout == 64 && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->bias_mode == 0 &&
          a->bias && a->slope && !a->residual && !a->out_f32_partial && a->y && a->W % 16 == 0 && a->H % ST_TR == 0 &&
          a->W <= 112))
        return 0;
    StemP p{(const half_t*)a->x, (const half_t*)a->w, a->bias, a->slope, (half_t*)a->y, a->B, a->H, a->W};
    const int64_t blocks = (int64_t)a->B * (a->H / ST_TR);
    if (blocks >= (1ll << 31)) return 0;
    conv_stem_kernel<112><<<(unsigned)blocks, 256, 0, s>>>(p);
    return 1;
}

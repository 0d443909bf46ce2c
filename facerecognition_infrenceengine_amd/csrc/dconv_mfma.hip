// Detector convolutions (MTCNN P/R/O-Net, inside FaceAnalysis.get: /root/reference/infrenceServer.py:528)
// as LDS-tiled implicit GEMMs on the f32-input matrix instruction v_mfma_f32_16x16x4_f32.
//
// Why f32 MFMA: the cascade thresholds scores (0.6/0.7/0.7) and floors box corners, so the
// detector keeps f32 operands end to end (an f16 P-Net flips borderline cells and moves crops);
// v_mfma_f32_16x16x4_f32 is an exact f32 fma chain at the f32 vector peak rate (64 FLOP/clk/SIMD)
// without the VALU's per-FMA operand traffic.  These nets have 3..256 channels: the kernels are
// bound by that f32 matrix rate / LDS feeding, not by the 2.5 PF f16 roof.
//
// GEMM view per block:  C[cout][pixel] = sum_{tap,ci} W[tap][ci][cout] * X[pixel + tap][ci]
//   A operand = weights: lane (i = l&15, kq = l>>4) supplies W[k0+kq][cout0+i]   (LDS, [k][CP])
//   B operand = pixels : lane (j = l&15, kq)        supplies X[pixel j][k0+kq]   (LDS input tile)
//   C: lane holds pixel j, couts 4*kq .. 4*kq+3 of each 16-cout tile -> 16-byte NHWC stores.
// Input tile: G images x (RH+KH-1) x (RW+KW-1) pixels, channel stride CINS = odd (bank spread),
// channels padded to a multiple of 4 with zeros.  Weights are staged per group of TG taps.
// Optional fused epilogues: PReLU, 2x2/s2 ceil-mode max pool (P-Net conv1), 1x1 head (P-Net
// conv3 -> 2 logits + 4 regressions).
#include "common.h"

struct DcArgs {
    const float* x; const float* w; const float* bias; const float* slope; float* y;
    const float* head_w; const float* head_b;
    int B, H, W;          // input images (for SRC == 1: the pyramid level size hs x ws)
    int Ho, Wo;           // conv output extent (H-KH+1, W-KW+1)
    int regions_x, regions_y;
    const uint8_t* frames; int FH, FW;   // SRC == 1: u8 BGR frames [B,FH,FW,3], resized on the fly
    unsigned long long* stamps;          // diagnostic (FR_DBG_STAMPS): per-wave phase cycle sums, else NULL
    const int32_t* counts; int cap;      // optional: image b is a real crop iff b % cap < counts[b / cap] (R-/O-Net slots)
    unsigned char* y_split;              // optional (layer 0): second copy of the output as split f16, 64 B per pixel =
                                         // [hi ch0-7 | hi ch8-15 | lo ch0-7 | lo ch8-15] (x = hi + lo; channels 12-15 zero): pnet_fused.hip
};

#define DSTAMP(var)                                                                               \
    do {                                                                                          \
        if (FR_DEBUG && a.stamps) {                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");            \
            __builtin_amdgcn_sched_barrier(0);                                                    \
        }                                                                                         \
    } while (0)

// POOL: 0 none | 1 fused 2x2/s2 ceil max pool in registers (needs RW % 16 == 0) | 2 fused PKxPK/s2 ceil max
// pool through an LDS conv-output tile (regions step by RSY x RSX conv pixels, RH x RW computed per region).
// RPB: regions per block along x (weights staged once per block).  SRC 1: input = bilinear resize of u8 frames.
template <int CIN, int COUT, int KH, int KW, int RH, int RW, int G, int NTB, int WN, int TG, int POOL, int PK,
          int RSY, int RSX, int NHEAD, int RPB, int SRC, int NW = 4>
struct DcCfg {
    static constexpr int CINP = (CIN + 3) / 4 * 4;
    static constexpr int CINS = CINP + 1;
    static constexpr int IH = RH + KH - 1, IW = RW + KW - 1;
    static constexpr int IMG = IH * IW * CINS;
    static constexpr int CP = NTB * 16 + ((NTB % 2 == 0) ? 16 : 0);
    static constexpr int NTAPS = KH * KW;
    static constexpr int NSTAGE = NTAPS / TG;
    static constexpr int WM = NW / WN, NT = NTB / WN;
    static constexpr int NTHR = NW * 64;
    static constexpr int NPIX = G * RH * RW;
    static constexpr int TP = (NPIX + 15) / 16;
    static constexpr int PT = (TP + WM - 1) / WM;
    static constexpr int CS = NTB * 16 + 4;                       // conv-output tile pixel stride (POOL == 2): 16-B rows, bank shift 4
    static constexpr int OUT_FLOATS = POOL == 2 ? NPIX * CS : 0;
    static constexpr int IN_FLOATS = G * IMG > OUT_FLOATS ? G * IMG : OUT_FLOATS;   // the two tiles share LDS
    static constexpr int W_OFF = (IN_FLOATS + 3) / 4 * 4;
    static constexpr int LDS_FLOATS = W_OFF + TG * CINP * CP;
    static_assert(NTAPS % TG == 0, "TG must divide the tap count");
    static_assert(NTB % WN == 0 && (WN == 1 || WN == 2 || WN == 4), "bad wave split");
    static_assert(POOL != 1 || (RW % 16 == 0 && RH % 2 == 0 && G == 1 && WN == 1 && (RH * RW / 16) % NW == 0), "pool layout");
    static_assert(CIN % 4 == 0 || SRC == 1, "f32 inputs are read as float4: pad channels to a multiple of 4");
    static_assert(SRC == 0 || CIN == 3, "fused resize feeds a 3-channel layer");
};

typedef unsigned long long u64_unaligned __attribute__((aligned(1)));
struct DLerp { int i0, i1; float w; };
__device__ __forceinline__ DLerp dlerp_coord(int d, float ratio, int n) {      // == detect_ops.hip lerp_coord
    float f = ((float)d + 0.5f) * ratio - 0.5f;
    float fl = floorf(f);
    DLerp r;
    r.w = f - fl;
    int i = (int)fl;
    r.i0 = min(max(i, 0), n - 1);
    r.i1 = min(max(i + 1, 0), n - 1);
    return r;
}
__device__ __forceinline__ float dbilerp(float p00, float p01, float p10, float p11, float wx, float wy) {
    float top = (1.0f - wx) * p00 + wx * p01;
    float bot = (1.0f - wx) * p10 + wx * p11;
    return (1.0f - wy) * top + wy * bot;
}

template <int CIN, int COUT, int KH, int KW, int RH, int RW, int G, int NTB, int WN, int TG, int POOL, int PK,
          int RSY, int RSX, int NHEAD, int RPB, int SRC, int NW = 4>
__global__ __launch_bounds__(NW * 64) void dconv_mfma(DcArgs a) {
    using C = DcCfg<CIN, COUT, KH, KW, RH, RW, G, NTB, WN, TG, POOL, PK, RSY, RSX, NHEAD, RPB, SRC, NW>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned long long dk = 0;
    DSTAMP(dk);
    float* xin = lds;
    float* wl = lds + C::W_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const int li = lane & 15, kq = lane >> 4;
    const int cg = blockIdx.y;
    const float* wsrc = a.w + (int64_t)cg * C::NTAPS * C::CINP * C::CP;
    // work items = (image group, region row, region col), RPB consecutive items per block
    const int per_img = a.regions_x * a.regions_y;
    const int nitems = per_img * ((a.B + G - 1) / G);
    const int item0 = blockIdx.x * RPB;
    if (a.counts) {
        // count-aware cascade: the R-Net / O-Net batch is nframes x cap crop SLOTS of which only counts[frame] hold a
        // candidate (a prefix).  A block whose items cover empty slots only exits before any barrier: a 1-face frame
        // no longer pays for 512 R-Net and 64 O-Net crops.  Outputs of empty slots stay unwritten (never read).
        const int zlo = item0 / per_img;
        int zhi = (item0 + RPB - 1) / per_img;
        const int zmax = (a.B + G - 1) / G - 1;
        if (zhi > zmax) zhi = zmax;
        const int b_lo = zlo * G, b_hi = min((zhi + 1) * G, a.B);             // images [b_lo, b_hi)
        bool any = false;
        for (int f = b_lo / a.cap; f * a.cap < b_hi && !any; ++f) {
            const int first = max(b_lo, f * a.cap);
            any = first - f * a.cap < a.counts[f];
        }
        if (!any) return;
    }

    auto stage_weights = [&](int st) {
        const float4v* src4 = reinterpret_cast<const float4v*>(wsrc + (int64_t)st * TG * C::CINP * C::CP);
        float4v* dst4 = reinterpret_cast<float4v*>(wl);
        for (int e = tid; e < TG * C::CINP * C::CP / 4; e += C::NTHR) dst4[e] = src4[e];
    };
    // tap groups that do not all fit in LDS (NSTAGE > 1) go through registers one stage ahead: the global loads of
    // stage st+1 fly under the MFMAs of stage st (staging them in place exposed an L2 round trip per stage)
    constexpr int W4 = TG * C::CINP * C::CP / 4;                 // float4 slots of one stage
    constexpr int NWPF = C::NSTAGE > 1 ? (W4 + C::NTHR - 1) / C::NTHR : 1;
    float4v wpf[NWPF];
    auto load_w = [&](int st) {
        const float4v* src4 = reinterpret_cast<const float4v*>(wsrc + (int64_t)st * TG * C::CINP * C::CP);
#pragma unroll
        for (int u = 0; u < NWPF; ++u) {
            const int e = tid + u * C::NTHR;
            wpf[u] = e < W4 ? src4[e] : float4v{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_w = [&]() {
        float4v* dst4 = reinterpret_cast<float4v*>(wl);
#pragma unroll
        for (int u = 0; u < NWPF; ++u) {
            const int e = tid + u * C::NTHR;
            if (e < W4) dst4[e] = wpf[u];
        }
    };

    // ---- input tile: global -> registers (issued early, lands under the previous item's MFMAs) -> LDS
    constexpr int C4 = C::CINP / 4;
    constexpr int NLOAD = SRC == 1 ? C::IH * C::IW : G * C::IH * C::IW * C4;     // float4 slots of one tile
    constexpr int NPF = (NLOAD + C::NTHR - 1) / C::NTHR;
    float4v pf[SRC == 1 ? 1 : NPF];
    // SRC == 1: the prefetch keeps the RAW source bytes (two 8-byte row pieces per level pixel) and the lerp weights;
    // conversion + bilinear blend happen in store_tile, AFTER the current tile's MFMAs - doing them in load_tile made
    // the "prefetch" wait for its own loads (stamps: 11 k cycles per tile in load_tile)
    unsigned long long rq0[SRC == 1 ? NPF : 1], rq1[SRC == 1 ? NPF : 1];
    float rwx[SRC == 1 ? NPF : 1], rwy[SRC == 1 ? NPF : 1];
    int rsh[SRC == 1 ? NPF : 1];                     // 24: x1 = x0 + 1; 0: clamped right border; -1: pixel outside the level
    int rpb[SRC == 1 ? NPF : 1];                     // pull-back of the two row loads in bits (last frame's last bytes)
    // slot e of the tile -> (image g, pixel iy/ix, channel group c4, LDS offset); decoded on the fly: keeping the
    // decode in registers across tiles cost 3 VGPRs per slot and bought nothing (measured)
    auto slot = [&](int e, int& iy, int& ix, int& g, int& c4, int& l) {
        iy = ix = g = c4 = 0; l = -1;
        if (e < NLOAD) {
            if constexpr (SRC == 1) {
                iy = e / C::IW; ix = e - iy * C::IW;
                l = e * C::CINS;
            } else {
                c4 = e % C4;
                const int pix_g = e / C4;
                g = pix_g / (C::IH * C::IW);
                const int pix = pix_g - g * (C::IH * C::IW);
                iy = pix / C::IW; ix = pix - iy * C::IW;
                l = g * C::IMG + pix * C::CINS + c4 * 4;
            }
        }
    };
    // SRC == 1: level -> frame scale factors and frame size, the same for every tile of the launch
    const float ryr = SRC == 1 ? (float)a.FH / (float)a.H : 0.f, rxr = SRC == 1 ? (float)a.FW / (float)a.W : 0.f;
    const int frame_bytes = a.FH * a.FW * 3;
    static_assert(SRC != 1 || G == 1, "fused resize: one frame per tile (block-uniform frame base)");
    auto load_tile = [&](int item) {
        const int zz = item / per_img, rem = item - zz * per_img;
        const int ry = rem / a.regions_x, rx = rem - ry * a.regions_x;
        const int y0 = ry * RSY, x0 = rx * RSX, img0 = zz * G;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            float4v v = {0.f, 0.f, 0.f, 0.f};
            int iy, ix, g, c4, l;
            slot(tid + u * C::NTHR, iy, ix, g, c4, l);
            const int yy = y0 + iy, xx = x0 + ix;
            const int n = img0 + g;
            if constexpr (SRC == 1) {
                rsh[u] = -1; rq0[u] = rq1[u] = 0; rwx[u] = rwy[u] = 0.f; rpb[u] = 0;
                if (l >= 0 && n < a.B && yy < a.H && xx < a.W) {
                    const DLerp ly = dlerp_coord(yy, ryr, a.FH), lx = dlerp_coord(xx, rxr, a.FW);
                    // both corners of a row are 6 adjacent bytes (BGR BGR): ONE unaligned 8-byte load per source row.
                    // 32-bit offsets inside the (block-uniform) frame; in the LAST frame the load is pulled back so
                    // that it never runs past the end of the buffer.  x1 == x0 (clamped right border) re-uses the
                    // first pixel's bytes.
                    const uint8_t* fbase = a.frames + (int64_t)n * frame_bytes;
                    const int lim = n == a.B - 1 ? frame_bytes - 8 : 0x7fffffff;
                    const int o0 = (ly.i0 * a.FW + lx.i0) * 3, o1 = (ly.i1 * a.FW + lx.i0) * 3;
                    const int c0 = min(o0, lim), c1 = min(o1, lim);
                    // raw loads only: any arithmetic on the loaded value here would make the prefetch wait for it
                    rq0[u] = *reinterpret_cast<const u64_unaligned*>(fbase + c0);
                    rq1[u] = *reinterpret_cast<const u64_unaligned*>(fbase + c1);
                    rpb[u] = ((o0 - c0) * 8) | (((o1 - c1) * 8) << 8);
                    rwx[u] = lx.w; rwy[u] = ly.w;
                    rsh[u] = lx.i1 == lx.i0 ? 0 : 24;
                }
            } else {
                if (l >= 0 && n < a.B && yy < a.H && xx < a.W)
                    v = *reinterpret_cast<const float4v*>(a.x + (((int64_t)n * a.H + yy) * a.W + xx) * CIN + c4 * 4);
                pf[u] = v;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            int iy, ix, g, c4, l;
            slot(tid + u * C::NTHR, iy, ix, g, c4, l);
            if (l >= 0) {
                float* d = xin + l;
                if constexpr (SRC == 1) {
                    float v[3] = {0.f, 0.f, 0.f};
                    if (rsh[u] >= 0) {
                        const unsigned long long q0 = rq0[u] >> (rpb[u] & 0xff), q1 = rq1[u] >> (rpb[u] >> 8);
                        const unsigned l0 = (unsigned)q0, h0 = (unsigned)(q0 >> 32), l1 = (unsigned)q1, h1 = (unsigned)(q1 >> 32);
                        const bool two = rsh[u] != 0;                 // x1 = x0 + 1 (else the clamped border: x1 = x0)
                        // bytes of a row piece: B0 G0 R0 B1 | G1 R1 . .   (v_cvt_f32_ubyteN each); output order R, G, B
                        const float a00[3] = {(float)((l0 >> 16) & 0xff), (float)((l0 >> 8) & 0xff), (float)(l0 & 0xff)};
                        const float a01[3] = {(float)((h0 >> 8) & 0xff), (float)(h0 & 0xff), (float)(l0 >> 24)};
                        const float a10[3] = {(float)((l1 >> 16) & 0xff), (float)((l1 >> 8) & 0xff), (float)(l1 & 0xff)};
                        const float a11[3] = {(float)((h1 >> 8) & 0xff), (float)(h1 & 0xff), (float)(l1 >> 24)};
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float sv = dbilerp(a00[c], two ? a01[c] : a00[c], a10[c], two ? a11[c] : a10[c], rwx[u], rwy[u]);
                            v[c] = (sv - 127.5f) * 0.0078125f;
                        }
                    }
                    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = 0.f;
                } else {
                    d[0] = pf[u][0]; d[1] = pf[u][1]; d[2] = pf[u][2]; d[3] = pf[u][3];
                }
            }
        }
    };

    if (item0 < nitems) load_tile(item0);
    if (C::NSTAGE == 1) stage_weights(0);
    else load_w(0);

    // ---- per-lane pixel bases (region-relative, the same for every item)
    int base[C::PT];
#pragma unroll
    for (int t = 0; t < C::PT; ++t) {
        const int p = (wm * C::PT + t) * 16 + li;
        int b = 0;
        if (p < C::NPIX) {
            const int g = p / (RH * RW), q = p - g * (RH * RW);
            const int ty = q / RW, tx = q - ty * RW;
            b = g * C::IMG + (ty * C::IW + tx) * C::CINS;
        }
        base[t] = b + kq;
    }
    const int coW = (cg * NTB + wn * C::NT) * 16 + kq * 4;      // first cout of this lane in tile i: coW + i*16
    // bias / PReLU slopes / head weights of this lane's channels: loaded once per block
    float4v bias_r[C::NT], slope_r[C::NT];
#pragma unroll
    for (int i = 0; i < C::NT; ++i) {
        bias_r[i] = *reinterpret_cast<const float4v*>(a.bias + coW + i * 16);
        slope_r[i] = a.slope ? *reinterpret_cast<const float4v*>(a.slope + coW + i * 16) : float4v{1.f, 1.f, 1.f, 1.f};
    }
    // fused head (P-Net conv3 -> conv4_1|conv4_2) on the matrix pipe: the activation tile sits in the C layout
    // (lane = pixel j, group kq, register e <-> channel 16*i + 4*kq + e), which IS a B operand with k = kq once one
    // register e is taken at a time; the A operand is the head weight W[h = lane&15][that channel].  8 MFMAs per
    // pixel tile replace 48 FMAs + 12 cross-lane reductions, and the weights need NT*4 registers instead of 48.
    float hw_a[NHEAD > 0 ? C::NT * 4 : 1];
    float hb_r[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NHEAD > 0) {
        static_assert(NHEAD <= 8 && WN == 1, "fused head: all channels of a pixel in one wave");
#pragma unroll
        for (int i = 0; i < C::NT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                hw_a[i * 4 + e] = li < NHEAD ? a.head_w[(coW + i * 16 + e) * NHEAD + li] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) hb_r[r] = (kq * 4 + r) < NHEAD ? a.head_b[kq * 4 + r] : 0.f;
    }
    if (item0 < nitems) store_tile();

    unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, q01 = 0, q12 = 0, q23 = 0, q34 = 0;
    for (int rr = 0; rr < RPB; ++rr) {
        const int item = item0 + rr;
        if (item >= nitems) break;
        DSTAMP(d0);
        if (FR_DEBUG && a.stamps && rr == 0) q01 += d0 - dk;          // block prologue (first tile load, weights) counted once
        const int zz = item / per_img, rem = item - zz * per_img;
        const int ry = rem / a.regions_x, rx = rem - ry * a.regions_x;
        const int y0 = ry * RSY, x0 = rx * RSX, img0 = zz * G;
        const bool more = rr + 1 < RPB && item + 1 < nitems;
        if (more) load_tile(item + 1);                   // global loads fly under this item's MFMAs
        DSTAMP(d1);

        float4v acc[C::NT][C::PT];
#pragma unroll
        for (int i = 0; i < C::NT; ++i)
#pragma unroll
            for (int t = 0; t < C::PT; ++t) acc[i][t] = float4v{0.f, 0.f, 0.f, 0.f};

        for (int st = 0; st < C::NSTAGE; ++st) {
            if (C::NSTAGE > 1) {
                if (st > 0) __syncthreads();                 // every wave is done with the previous stage's weights
                store_w();
                if (st + 1 < C::NSTAGE) load_w(st + 1);
                else if (more) load_w(0);
            }
            __syncthreads();
            // operands of k step s+1 are read before the MFMAs of step s are issued (the compiler's own order
            // issues each step's ds_reads right in front of their first use and exposes the LDS latency)
            constexpr int KSTEPS = TG * (C::CINP / 4);
            auto rd = [&](int ks, float (&av)[C::NT], float (&bv)[C::PT]) {
                const int tl = ks / (C::CINP / 4), c4 = ks - tl * (C::CINP / 4);
                const int tap = st * TG + tl;
                const int kh = tap / KW, kw = tap - kh * KW;
                const int toff = (kh * C::IW + kw) * C::CINS;
                const float* wt = wl + (tl * C::CINP + kq) * C::CP + wn * C::NT * 16 + li;
#pragma unroll
                for (int i = 0; i < C::NT; ++i) av[i] = wt[c4 * 4 * C::CP + i * 16];
#pragma unroll
                for (int t = 0; t < C::PT; ++t) bv[t] = xin[base[t] + toff + c4 * 4];
            };
            float av0[C::NT], bv0[C::PT], av1[C::NT], bv1[C::PT];
            rd(0, av0, bv0);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks += 2) {
                if (ks + 1 < KSTEPS) rd(ks + 1, av1, bv1);
#pragma unroll
                for (int i = 0; i < C::NT; ++i)
#pragma unroll
                    for (int t = 0; t < C::PT; ++t)
                        acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[i], bv0[t], acc[i][t], 0, 0, 0);
                if (ks + 1 < KSTEPS) {
                    if (ks + 2 < KSTEPS) rd(ks + 2, av0, bv0);
#pragma unroll
                    for (int i = 0; i < C::NT; ++i)
#pragma unroll
                        for (int t = 0; t < C::PT; ++t)
                            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[i], bv1[t], acc[i][t], 0, 0, 0);
                }
            }
        }

        DSTAMP(d2);
        // ---- epilogue: bias + PReLU
#pragma unroll
        for (int i = 0; i < C::NT; ++i) {
#pragma unroll
            for (int t = 0; t < C::PT; ++t) {
                float4v v = acc[i][t] + bias_r[i];
                if (a.slope) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope_r[i][e];
                }
                acc[i][t] = v;
            }
        }
        if constexpr (POOL == 1) {
            // tiles: t = TR*row_in_wave + half; pool rows (2a, 2a+1) in registers, lanes (2b, 2b+1) by one shuffle
            constexpr int TR = RW / 16;
            const int Hp = (a.Ho + 1) / 2, Wp = (a.Wo + 1) / 2;
#pragma unroll
            for (int i = 0; i < C::NT; ++i) {
#pragma unroll
                for (int t = 0; t < C::PT; ++t) {
                    const int tile = wm * C::PT + t;
                    const int ty = tile / TR, tx = (tile - ty * TR) * 16 + li;
                    if (y0 + ty >= a.Ho || x0 + tx >= a.Wo) acc[i][t] = float4v{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
                constexpr int ROWS = C::PT / TR;
                static_assert(POOL != 1 || ROWS % 2 == 0, "a wave must own whole row pairs");
#pragma unroll
                for (int rp = 0; rp < ROWS / 2; ++rp) {
#pragma unroll
                    for (int hf = 0; hf < TR; ++hf) {
                        const int t = 2 * rp * TR + hf;
                        float4v v = acc[i][t];
                        const float4v u = acc[i][t + TR];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float m = fmaxf(v[e], u[e]);
                            float o = __shfl_xor(m, 1, 64);
                            v[e] = fmaxf(m, o);
                        }
                        const int ty = wm * ROWS + 2 * rp;
                        const int tx = hf * 16 + li;
                        const int py = (y0 + ty) >> 1, px = (x0 + tx) >> 1;
                        const int n = img0;
                        if ((li & 1) == 0 && py < Hp && px < Wp && n < a.B) {
                            float* o = a.y + (((int64_t)n * Hp + py) * Wp + px) * COUT;
                            if constexpr (COUT % 4 == 0) {       // the lane's 4 consecutive channels: one 16-byte store
                                if (coW + i * 16 < COUT) *reinterpret_cast<float4v*>(o + coW + i * 16) = v;
                                if constexpr (COUT <= 16 && C::NT == 1) {
                                    if (a.y_split) {             // split-f16 copy for the fused P-Net kernel (8 B hi + 8 B lo per lane)
                                        const int ch0 = coW + i * 16;                 // 0, 4, 8, 12
                                        half4 hi, lo;
#pragma unroll
                                        for (int e = 0; e < 4; ++e) {
                                            const float x = ch0 + e < COUT ? v[e] : 0.f;
                                            const half_t h = (half_t)x;
                                            hi[e] = h; lo[e] = (half_t)(x - (float)h);
                                        }
                                        unsigned char* o2 = a.y_split + (((int64_t)n * Hp + py) * Wp + px) * 64 + (ch0 >> 3) * 16 + ((ch0 >> 2) & 1) * 8;
                                        *reinterpret_cast<half4*>(o2) = hi;
                                        *reinterpret_cast<half4*>(o2 + 32) = lo;
                                    }
                                }
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (coW + i * 16 + e < COUT) o[coW + i * 16 + e] = v[e];
                            }
                        }
                    }
                }
            }
        } else if constexpr (POOL == 2) {
            // conv outputs -> LDS tile [pixel][CS] (shares the input tile's space), then PKxPK/s2 ceil max pool
            __syncthreads();                               // every wave is done reading the input tile
            float* ot = lds;
#pragma unroll
            for (int t = 0; t < C::PT; ++t) {
                const int p = (wm * C::PT + t) * 16 + li;
                if (p >= C::NPIX) continue;
#pragma unroll
                for (int i = 0; i < C::NT; ++i)            // the lane's 4 consecutive channels: one 16-byte LDS store
                    *reinterpret_cast<float4v*>(ot + p * C::CS + (wn * C::NT + i) * 16 + kq * 4) = acc[i][t];
            }
            __syncthreads();
            const int PH = (a.Ho - PK + 1) / 2 + 1 - ((((a.Ho - PK + 1) / 2) * 2 >= a.Ho) ? 1 : 0);
            const int PW = (a.Wo - PK + 1) / 2 + 1 - ((((a.Wo - PK + 1) / 2) * 2 >= a.Wo) ? 1 : 0);
            constexpr int PRH = RSY / 2, PRW = RSX / 2;     // pooled rows / cols owned by one region
            constexpr int CG = NTB * 16, Q4 = CG / 4;
            static_assert(COUT % 4 == 0, "pooled layers store 4 channels per thread");
            // a thread owns (pooled pixel, 4 channels): PK*PK 16-byte LDS reads, no branches - window positions past
            // the region or the image edge are CLAMPED onto the last valid row / column (a ceil-mode window always
            // starts inside, so the clamp only repeats a value that is in the window anyway)
            const int vh = min(RH, a.Ho - y0), vw = min(RW, a.Wo - x0);
            for (int e = tid; e < G * PRH * PRW * Q4; e += C::NTHR) {
                const int qd = e % Q4, pp = e / Q4;
                const int g = pp / (PRH * PRW), q = pp - g * (PRH * PRW);
                const int pyl = q / PRW, pxl = q - pyl * PRW;
                const int py = y0 / 2 + pyl, px = x0 / 2 + pxl, n = img0 + g;
                const int cout = cg * CG + qd * 4;
                if (py >= PH || px >= PW || n >= a.B || cout >= COUT) continue;
                int ro[PK], cl[PK];
#pragma unroll
                for (int d = 0; d < PK; ++d) {
                    ro[d] = (g * RH + min(2 * pyl + d, vh - 1)) * RW;
                    cl[d] = min(2 * pxl + d, vw - 1);
                }
                float4v m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                for (int dy = 0; dy < PK; ++dy)
#pragma unroll
                    for (int dx = 0; dx < PK; ++dx) {
                        const float4v v = *reinterpret_cast<const float4v*>(ot + (ro[dy] + cl[dx]) * C::CS + qd * 4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) m[c] = fmaxf(m[c], v[c]);
                    }
                *reinterpret_cast<float4v*>(a.y + (((int64_t)n * PH + py) * PW + px) * COUT + cout) = m;
            }
        } else if constexpr (NHEAD > 0) {
            float4v hd[C::PT];
#pragma unroll
            for (int t = 0; t < C::PT; ++t) hd[t] = float4v{hb_r[0], hb_r[1], hb_r[2], hb_r[3]};
#pragma unroll
            for (int i = 0; i < C::NT; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < C::PT; ++t)
                        hd[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(hw_a[i * 4 + e], acc[i][t][e], hd[t], 0, 0, 0);
            // D layout: lane (pixel j = li, group kq) holds heads 4*kq .. 4*kq+3
#pragma unroll
            for (int t = 0; t < C::PT; ++t) {
                const int p = (wm * C::PT + t) * 16 + li;
                if (p >= C::NPIX || kq * 4 >= NHEAD) continue;
                const int g = p / (RH * RW), q = p - g * (RH * RW);
                const int ty = q / RW, tx = q - ty * RW;
                const int n = img0 + g, oy = y0 + ty, ox = x0 + tx;
                if (n < a.B && oy < a.Ho && ox < a.Wo) {
                    float* o = a.y + (((int64_t)n * a.Ho + oy) * a.Wo + ox) * NHEAD + kq * 4;
                    if constexpr (NHEAD % 2 == 0) {          // 8-byte stores (a pixel's NHEAD floats are 8-byte aligned)
                        typedef float float2v __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<float2v*>(o) = float2v{hd[t][0], hd[t][1]};
                        if (kq * 4 + 2 < NHEAD) *reinterpret_cast<float2v*>(o + 2) = float2v{hd[t][2], hd[t][3]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (kq * 4 + r < NHEAD) o[r] = hd[t][r];
                    }
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < C::PT; ++t) {
                const int p = (wm * C::PT + t) * 16 + li;
                if (p >= C::NPIX) continue;
                const int g = p / (RH * RW), q = p - g * (RH * RW);
                const int ty = q / RW, tx = q - ty * RW;
                const int n = img0 + g, oy = y0 + ty, ox = x0 + tx;
                if (n >= a.B || oy >= a.Ho || ox >= a.Wo) continue;
                float* o = a.y + (((int64_t)n * a.Ho + oy) * a.Wo + ox) * COUT;
#pragma unroll
                for (int i = 0; i < C::NT; ++i) {
                    const int co = coW + i * 16;
                    if constexpr (COUT % 4 == 0) {
                        if (co < COUT) *reinterpret_cast<float4v*>(o + co) = acc[i][t];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (co + e < COUT) o[co + e] = acc[i][t][e];
                    }
                }
            }
        }
        DSTAMP(d3);
        if (more) {
            __syncthreads();                             // every wave is done with this item's tiles
            store_tile();
        }
        DSTAMP(d4);
        if (FR_DEBUG && a.stamps) { q01 += d1 - d0; q12 += d2 - d1; q23 += d3 - d2; q34 += d4 - d3; }
    }
    if (FR_DEBUG && a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)(blockIdx.x % 4096) * NW + wave) * 4;
        o[0] = q01; o[1] = q12; o[2] = q23; o[3] = q34;
    }
}

template <int CIN, int COUT, int KH, int KW, int RH, int RW, int G, int NTB, int WN, int TG, int POOL, int PK,
          int RSY, int RSX, int NHEAD, int RPB, int SRC, int NW = 4>
static int launch_dc(const DcArgs& a0, hipStream_t s) {
    using C = DcCfg<CIN, COUT, KH, KW, RH, RW, G, NTB, WN, TG, POOL, PK, RSY, RSX, NHEAD, RPB, SRC, NW>;
    DcArgs a = a0;
    a.Ho = a.H - KH + 1; a.Wo = a.W - KW + 1;
    if constexpr (POOL == 2) {      // regions tile the POOLED output: a region owns RSY/2 x RSX/2 pooled pixels
        const int PH = (a.Ho - PK + 1) / 2 + 1 - ((((a.Ho - PK + 1) / 2) * 2 >= a.Ho) ? 1 : 0);
        const int PW = (a.Wo - PK + 1) / 2 + 1 - ((((a.Wo - PK + 1) / 2) * 2 >= a.Wo) ? 1 : 0);
        a.regions_x = (PW + RSX / 2 - 1) / (RSX / 2);
        a.regions_y = (PH + RSY / 2 - 1) / (RSY / 2);
    } else {
        a.regions_x = (a.Wo + RSX - 1) / RSX;
        a.regions_y = (a.Ho + RSY - 1) / RSY;
    }
    constexpr int COUTP = (COUT + 15) / 16 * 16;
    static_assert(COUTP % (NTB * 16) == 0, "cout groups must tile COUTP");
    const int nitems = a.regions_x * a.regions_y * ((a.B + G - 1) / G);
    dim3 grid((nitems + RPB - 1) / RPB, COUTP / (NTB * 16), 1);
    const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
    auto kern = dconv_mfma<CIN, COUT, KH, KW, RH, RW, G, NTB, WN, TG, POOL, PK, RSY, RSX, NHEAD, RPB, SRC, NW>;
    if (lds > 64 * 1024) {
        static FrDevLatch latch;        // per (kernel instantiation, device)
        if (!fr_raise_lds(reinterpret_cast<const void*>(kern), lds, latch)) {
            fr_set_error("fr_dconv_mfma_f32: cannot raise dynamic LDS to %zu bytes", lds);
            return FR_E_LAUNCH;
        }
    }
    kern<<<grid, NW * 64, lds, s>>>(a);
    return FR_OK;
}

int fr_pnet_conv1_launch(const uint8_t* frames, int B, int FH, int FW, int H, int W, const float* w, const float* bias,
                         const float* slope, float* y, void* y_split, hipStream_t s);

// Layer table (see mtcnn.py: layer ids).  Geometry is fixed by the MTCNN architecture.
extern "C" int fr_dconv_mfma_f32(int layer, const float* x, const float* w, const float* bias, const float* slope,
                                 float* y, int B, int H, int W, const float* head_w, const float* head_b,
                                 const uint8_t* frames, int FH, int FW, const int32_t* counts, int cap,
                                 void* y_split, fr_stream_t stream) {
    FR_REQUIRE(!y_split || layer == 0 || layer == 3, "fr_dconv_mfma_f32: y_split is an output of layer 0 only");
    FR_REQUIRE(!counts || (cap > 0 && B % cap == 0 && layer >= 10), "fr_dconv_mfma_f32: counts need cap | B and an R-/O-Net layer");
    FR_REQUIRE(w && bias && y && B > 0 && H > 0 && W > 0, "fr_dconv_mfma_f32: bad argument");
    FR_REQUIRE((layer == 0 || layer == 3) ? (frames && FH > 0 && FW > 0) : (x != nullptr), "fr_dconv_mfma_f32: no input");
    DcArgs a{x, w, bias, slope, y, head_w, head_b, B, H, W, 0, 0, 0, 0, frames, FH, FW,
             (unsigned long long*)fr_dbg_ptr("FR_DBG_STAMPS"),           // NULL in the product build
             counts, cap, (unsigned char*)y_split};
    hipStream_t s = fr_stream(stream);
    int rc = FR_OK;
    switch (layer) {
        //                  CIN COUT KH KW RH  RW  G NTB WN TG POOL PK RSY RSX NHEAD RPB SRC
        case 0:  FR_REQUIRE(H >= 3 && W >= 3 && frames && slope, "P1 needs frames, PReLU slopes and a level of at least 3x3");
                 rc = fr_pnet_conv1_launch(frames, B, FH, FW, H, W, w, bias, slope, y, y_split, s);      // pnet_conv1.hip
                 break;
        case 3:  FR_REQUIRE(H >= 3 && W >= 3 && frames, "P1 needs frames and a level of at least 3x3");
                 { const int v = fr_dbg_int("FR_P1_RPB", 8);
                 const bool big = (int64_t)((H - 2 + 15) / 16) * ((W - 2 + 31) / 32) * B >= 8192;
                 if (big && v == 2) rc = launch_dc<3, 12, 3, 3, 16, 32, 1, 1, 1, 9, 1, 2, 16, 32, 0, 2, 1>(a, s);
                 else if (big && v == 4) rc = launch_dc<3, 12, 3, 3, 16, 32, 1, 1, 1, 9, 1, 2, 16, 32, 0, 4, 1>(a, s);
                 else if (big && v == 8) rc = launch_dc<3, 12, 3, 3, 16, 32, 1, 1, 1, 9, 1, 2, 16, 32, 0, 8, 1>(a, s);
                 else rc = launch_dc<3, 12, 3, 3, 16, 32, 1, 1, 1, 9, 1, 2, 16, 32, 0, 1, 1>(a, s); }
                 break;                                                                                   // P-Net conv1 (+resize, PReLU, pool)
        case 1:  FR_REQUIRE(H >= 3 && W >= 3, "P2 input too small");
                 // small pyramid levels have too few tiles to fill 256 CUs: multi-tile blocks only add latency there
                 if ((int64_t)((H - 2 + 7) / 8) * ((W - 2 + 31) / 32) * B < 8192)
                     rc = launch_dc<12, 16, 3, 3, 8, 32, 1, 1, 1, 9, 0, 2, 8, 32, 0, 1, 0>(a, s);
                 else
                     rc = launch_dc<12, 16, 3, 3, 8, 32, 1, 1, 1, 9, 0, 2, 8, 32, 0, 8, 0>(a, s);
                 break;                                                                                   // P-Net conv2
        case 2:  FR_REQUIRE(H >= 3 && W >= 3 && head_w && head_b, "P3 needs head weights");
                 { const int v = fr_dbg_int("FR_P3_RPB", 8);
                 if (v == 1 || (int64_t)((H - 2 + 7) / 8) * ((W - 2 + 31) / 32) * B < 8192)
                     rc = launch_dc<16, 32, 3, 3, 8, 32, 1, 2, 1, 9, 0, 2, 8, 32, 6, 1, 0>(a, s);
                 else if (v == 2)
                     rc = launch_dc<16, 32, 3, 3, 8, 32, 1, 2, 1, 9, 0, 2, 8, 32, 6, 2, 0>(a, s);
                 else
                     rc = launch_dc<16, 32, 3, 3, 8, 32, 1, 2, 1, 9, 0, 2, 8, 32, 6, 8, 0>(a, s); }
                 break;                                                                                   // P-Net conv3+heads
        case 10: FR_REQUIRE(H == 24 && W == 24, "R1 expects 24x24");                        // conv1 + 3x3/s2 pool -> 11x11
                 rc = launch_dc<4, 28, 3, 3, 22, 22, 1, 2, 1, 9, 2, 3, 22, 22, 0, 4, 0>(a, s); break;
        case 11: FR_REQUIRE(H == 11 && W == 11, "R2 expects 11x11");                        // conv2 + 3x3/s2 pool -> 4x4
                 rc = launch_dc<28, 48, 3, 3, 9, 9, 2, 3, 1, 3, 2, 3, 8, 8, 0, 1, 0>(a, s); break;
        case 12: FR_REQUIRE(H == 4 && W == 4, "R3 expects 4x4");
                 rc = launch_dc<48, 64, 2, 2, 3, 3, 16, 4, 2, 1, 0, 2, 3, 3, 0, 1, 0>(a, s); break;
        case 13: FR_REQUIRE(H == 3 && W == 3, "R4 expects 3x3");
                 rc = launch_dc<64, 128, 3, 3, 1, 1, 32, 4, 4, 1, 0, 2, 1, 1, 0, 1, 0>(a, s); break;      // dense4
        case 14: FR_REQUIRE(H == 1 && W == 1, "R5 expects 1x1");
                 rc = launch_dc<128, 6, 1, 1, 1, 1, 64, 1, 1, 1, 0, 2, 1, 1, 0, 1, 0>(a, s); break;       // dense5_1|5_2
        case 20: FR_REQUIRE(H == 48 && W == 48, "O1 expects 48x48");                        // conv1 + 3x3/s2 pool -> 23x23
                 rc = launch_dc<4, 32, 3, 3, 9, 46, 1, 1, 1, 9, 2, 3, 8, 46, 0, 6, 0>(a, s); break;
        case 21: FR_REQUIRE(H == 23 && W == 23, "O2 expects 23x23");                        // conv2 + 3x3/s2 pool -> 10x10
                 rc = launch_dc<32, 64, 3, 3, 11, 21, 1, 2, 1, 3, 2, 3, 10, 20, 0, 1, 0>(a, s); break;   // 2 regions of 11 conv rows x 2 cout groups
        case 22: FR_REQUIRE(H == 10 && W == 10, "O3 expects 10x10");                        // conv3 + 2x2/s2 pool -> 4x4
                 rc = launch_dc<64, 64, 3, 3, 8, 8, 1, 4, 1, 1, 2, 2, 8, 8, 0, 1, 0>(a, s); break;
        case 23: FR_REQUIRE(H == 4 && W == 4, "O4 expects 4x4");
                 rc = launch_dc<64, 128, 2, 2, 3, 3, 16, 4, 2, 1, 0, 2, 3, 3, 0, 1, 0>(a, s); break;
        case 24: FR_REQUIRE(H == 3 && W == 3, "O5 expects 3x3");
                 rc = launch_dc<128, 256, 3, 3, 1, 1, 16, 4, 4, 1, 0, 2, 1, 1, 0, 1, 0>(a, s); break;     // dense5
        case 25: FR_REQUIRE(H == 1 && W == 1, "O6 expects 1x1");
                 rc = launch_dc<256, 16, 1, 1, 1, 1, 64, 1, 1, 1, 0, 2, 1, 1, 0, 1, 0>(a, s); break;      // dense6_*
        default: FR_REQUIRE(false, "fr_dconv_mfma_f32: unknown layer id %d", layer);
    }
    if (rc != FR_OK) return rc;
    FR_CHECK_LAUNCH("dconv_mfma");
    return FR_OK;
}

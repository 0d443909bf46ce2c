// MTCNN R-Net / O-Net first layer, fused with the crop that feeds it: zero-padded crop of trunc(box) from the u8 BGR
// frame -> bilinear resize to S x S (24 / 48) -> conv 3x3 (3 -> 28 / 32) -> PReLU -> 3x3 / stride 2 ceil-mode max pool
// (11 x 11 / 23 x 23), f32 end to end (part of FaceAnalysis.get, /root/reference/infrenceServer.py:528; the arithmetic
// of fr_crop_resize_norm and of layers 10 / 20 of fr_dconv_mfma_f32, which this replaces on the product path and
// must equal bit for bit: tests/test_gpu_detect.py).
//
// Why: the stand-alone crop kernel wrote 302 + 151 MB of f32 crops per 64-frame batch only for these layers to read
// them back, and on the 16x16x4 MFMA the layers used 27 of 36 K rows and 28 of 32 couts.  Here one block owns one crop
// slot: the resized crop is built in LDS (lerp tables per crop; the source pixels of a row pair come from one
// unaligned 8-byte load when both columns lie in the frame), the conv runs on v_mfma_f32_4x4x1 with a BROADCAST
// weight operand (see pnet_conv1.hip: K = 27 exactly, a conv pixel per lane, NG cout quads), conv pixels are
// flattened (64 per unit, a lane's 3x3 window = 27 LDS reads at constant offsets from its base), the conv map goes to
// an LDS tile [pixel][28 | 36] and is pooled from there, in bands of 4 (R-Net) / 2 (O-Net) pooled rows so that several
// blocks fit a CU.  A block walks 8 / 4 consecutive slots and fetches the next valid slot's source bytes (raw, into
// registers) under the current slot's conv.  When every PReLU slope is >= 0 the raw sums are pooled first and
// activated after (the same bits, a ninth of the bias / PReLU work); otherwise every pixel is activated before the pool.
// Count-aware like the other R-/O-Net layers: a slot >= counts[frame] does no work and leaves its output unwritten.
#include "common.h"
#include <type_traits>

#ifndef RC1_PB_R           // developer A/B (tools/abl_rc1.py): pooled rows per band / slots per block of the batch path's F16 form
#define RC1_PB_R 3         // measured (64 x 512 random boxes, us): PB x RPB 4x8 495, 3x8 417, 2x8 434, 4x16 429, 3x16 397, 3x32 see tools/abl_rc1.py
#define RC1_RPB_R 16
#define RC1_PB_O 2         // O-Net (64 x 64 boxes): 2x4 281, 2x8 279, 1x4 317, 3x4 446, 2x2 288
#define RC1_RPB_O 4
#endif

namespace {

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

struct RoArgs {
    const uint8_t* frames; int nframes, H, W;
    const float* boxes; const int32_t* counts; int cap;
    const float* w;                            // [27][NG * 4] f32: k = (kh * 3 + kw) * 3 + channel (R, G, B)
    const float* bias; const float* slope;     // [NG * 4]
    float* y;                                  // [slots][P][P][COUT]
    // SPLIT: the pooled map leaves as split f16 instead - [slots][P*P][hi 32 ch | lo 32 ch] (x = hi + lo, 128 B per pixel,
    // channels >= COUT zero): what fr_ro_conv2_split streams into LDS
    unsigned char* ys;
    // LIST: the block's slots are positions of a compact list of slot numbers (the crops the exact pass re-evaluates):
    // slot s -> box / frame of list[s], output row s; *list_count entries (clamped to list_cap)
    const int32_t* list; const int32_t* list_count; int list_cap;
};

typedef unsigned long long u64_unaligned_r __attribute__((aligned(1)));
struct LerpR { int i0, i1; float w; };
__device__ __forceinline__ LerpR lerp_coord_r(int d, float ratio, int n) {        // == detect_ops.hip lerp_coord
    float f = ((float)d + 0.5f) * ratio - 0.5f;
    float fl = floorf(f);
    LerpR r;
    r.w = f - fl;
    int i = (int)fl;
    r.i0 = min(max(i, 0), n - 1);
    r.i1 = min(max(i + 1, 0), n - 1);
    return r;
}
__device__ __forceinline__ float bilerp_r(float p00, float p01, float p10, float p11, float wx, float wy) {
    float top = (1.0f - wx) * p00 + wx * p01;
    float bot = (1.0f - wx) * p10 + wx * p11;
    return (1.0f - wy) * top + wy * bot;
}
__device__ __forceinline__ float vmax_r(float x, float y) {        // no canonicalising self-max (operands are never sNaN)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// S: crop size; NG: cout quads (COUT = 4 NG real channels written); PB: pooled rows per band (a band = 2 PB + 1 conv rows);
// RPB: consecutive crop slots per block.  The source bytes of the NEXT valid slot's crop are fetched (raw, into
// registers) before the current slot's conv and blended into LDS after it, so the frame gather's latency is covered.
// F16 (with SPLIT, the batch path): the conv runs on the f16 matrix cores with split-precision operands instead of the 4x4x1 f32
// form.  The resized crop is kept as hi | lo f16 planes of 4-channel pixels (8 B: R, G, B, 0), so the (kw = 0..3, channel 0..3)
// values of one kernel row are 32 contiguous bytes and K = (kh, kw', c') = 4 x 4 x 4 with zero weights at kh = 3, kw' = 3,
// c' = 3: two K = 32 steps, a B fragment = 16 B (2 pixels) at an 8-byte aligned address.  6 MFMAs of 16 cycles per 16 pixels x
// 16 couts against 27 x 4 = 108 4x4x1 MFMAs of 8 cycles per 64 pixels x 16 couts: ~2.8x fewer matrix cycles, and the work
// splits in tiles of 16 pixels instead of units of 64.  ~1e-6 from the f32 form (the exact pass covers the threshold).
template <int S, int NG, int COUT, int PB, int RPB, bool SPLIT = false, bool LIST = false, bool F16 = false>
__global__ __launch_bounds__(256) void crop_conv1_kernel(RoArgs a) {
    constexpr int C = S - 2;                                     // conv map size
    constexpr int P = (C - 3 + 1) / 2 + 1;                       // ceil((C - 3) / 2) + 1 pooled size (ceil mode)
    // conv tile pixel stride in floats: 16-B rows whose stride is NOT a multiple of 32 words, so that 8 consecutive lanes'
    // 16-byte accesses fall into 8 different bank groups (28 words for R-Net's 28 channels as they are; 32 + 4 for O-Net)
    constexpr int CS = (NG * 4) % 32 ? NG * 4 : NG * 4 + 4;
    constexpr int BR = 2 * PB + 1 < C ? 2 * PB + 1 : C;          // conv rows per band
    constexpr int NBAND = (P + PB - 1) / PB;
    constexpr int NPF = (S * S + 255) / 256;                     // crop pixels per thread
    static_assert(NG * 4 <= 32 && COUT <= NG * 4, "cout quads");
    constexpr int NW = (27 * NG + 15) / 16;                      // weight registers
    constexpr int XPX = S * S + S + 8;                           // F16: pixels per plane incl. the zero slack behind the image (kh = 3, kw' = 3 read there)
    constexpr int XFLOATS = F16 ? (2 * XPX * 8 / 4 + 3) & ~3 : (S * S * 3 + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) float lds_ro[];
    float* xin = lds_ro;                                         // [S*S][3] f32, or F16: hi | lo planes of [XPX][4] f16
    int4v* tab = reinterpret_cast<int4v*>(lds_ro + XFLOATS);     // [2][2 S] lerp tables (double-buffered)
    float4* sbox_mem = reinterpret_cast<float4*>(tab + 4 * S);  // [RPB] the block's boxes (one global read, off the per-slot path)
    int* sorig = reinterpret_cast<int*>(sbox_mem + RPB);         // [RPB] the slots' numbers in the cascade's slot space (LIST: list[s])
    float* ot = reinterpret_cast<float*>(sorig + ((RPB + 3) & ~3));      // [BR * C][CS] conv tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a block's RPB consecutive slots lie in ONE frame (the host picks RPB | cap), and a frame's candidates are a prefix of
    // its slots: the block's valid slots are the interval [slot, s_end) - one read of the count, no search
    const int s_first = blockIdx.x * RPB;
    int s_end;
    if constexpr (LIST) {
        s_end = min(s_first + RPB, min(*a.list_count, a.list_cap));
    } else {
        const int fblk = s_first / a.cap;
        s_end = min(s_first + RPB, fblk * a.cap + a.counts[fblk]);
    }
    int slot = s_first;
    if (slot >= s_end) return;                                   // only empty slots: no work, outputs unwritten
    auto next_valid = [&](int s) { return s < s_end ? s : -1; };
    float4* sbox = sbox_mem;
    if (tid < s_end - s_first) {
        const int o = LIST ? a.list[s_first + tid] : s_first + tid;
        sorig[tid] = o;
        sbox[tid] = *reinterpret_cast<const float4*>(a.boxes + (int64_t)o * 4);
    }
    __syncthreads();
    // ---- weights: slot c = k * NG + g -> lanes 4 (c % 16) + r of register c / 16 hold W[k][4g + r]
    float wreg[F16 ? 1 : NW];
    half8 wfh[F16 ? 2 : 1][2], wfl[F16 ? 2 : 1][2];              // F16: A fragments [K step][cout tile]: row = cout li, k = 8 kq + j
    if constexpr (F16) {
        unsigned char* xz = reinterpret_cast<unsigned char*>(xin);
        for (int e = tid; e < (S + 8) * 2; e += 256)            // the slack pixels of both planes: zero, once
            *reinterpret_cast<unsigned long long*>(xz + ((e & 1) * XPX + S * S + (e >> 1)) * 8) = 0ull;
        const int li_ = lane & 15, kq_ = lane >> 4;
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kh = 2 * st + (kq_ >> 1), kw = (kq_ & 1) * 2 + (j >> 2), c = j & 3, co = ct * 16 + li_;
                    const float w = (kh < 3 && kw < 3 && c < 3 && co < NG * 4) ? a.w[((kh * 3 + kw) * 3 + c) * (NG * 4) + co] : 0.f;
                    const half_t h = (half_t)w;
                    wfh[st][ct][j] = h; wfl[st][ct][j] = (half_t)(w - (float)h);
                }
    } else {
#pragma unroll
    for (int v = 0; v < NW; ++v) {
        const int c = v * 16 + (lane >> 2), r = lane & 3;
        wreg[v] = c < 27 * NG ? a.w[(c / NG) * (NG * 4) + (c % NG) * 4 + r] : 0.f;
    }
    }
    bool mono = true;
    for (int c = 0; c < NG * 4; ++c) mono = mono && a.slope[c] >= 0.f;

    // ---- the crop (fr_crop_resize_norm's arithmetic).  Per slot: tables of its S rows and S columns {first source
    // index in frame coordinates, second, weight, -}; per crop pixel two 8-byte row pieces fetched RAW plus flags:
    //   bits 0-1: what p(.,0) is (0 zero - column outside the frame, 1 first pixel of the piece), bits 2-3: what p(.,1)
    //   is (0 zero, 1 first pixel - clamped: x1 = x0, or x0 outside, 2 second pixel), bits 8-15 / 16-23: pull-back of the
    //   two loads in bits (last bytes of the last frame); -1: the whole pixel is zero (empty box)
    unsigned long long rq0[NPF], rq1[NPF];
    int rfl[NPF];
    bool edge = true;                                            // wave-uniform: some pixel is not the plain two-column case
    auto make_tab = [&](int sl, int buf) __attribute__((always_inline)) {
        if (tid < 2 * S) {
            const float4 b = sbox[sl - s_first];
            const int x1 = (int)truncf(b.x), y1 = (int)truncf(b.y), x2 = (int)truncf(b.z), y2 = (int)truncf(b.w);
            const int tw = x2 - x1 + 1, th = y2 - y1 + 1;
            int4v e = {0, 0, 0, 0};                              // e[3] = 1: the box is not empty
            if (tw > 0 && th > 0) {
                if (tid < S) {
                    const LerpR l = lerp_coord_r(tid, (float)th / (float)S, th);
                    e = int4v{y1 - 1 + l.i0, y1 - 1 + l.i1, __float_as_int(l.w), 1};
                } else {
                    const LerpR l = lerp_coord_r(tid - S, (float)tw / (float)S, tw);
                    e = int4v{x1 - 1 + l.i0, x1 - 1 + l.i1, __float_as_int(l.w), 1};
                }
            }
            tab[buf * 2 * S + tid] = e;
        }
    };
    auto load_crop = [&](int sl, int buf) __attribute__((always_inline)) {
        const int f = sorig[sl - s_first] / a.cap;
        const uint8_t* fr = a.frames + (int64_t)f * a.H * a.W * 3;
        const int lim = f == a.nframes - 1 ? a.H * a.W * 3 - 8 : 0x7fffffff;
        bool special = false;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int t = min(tid + u * 256, S * S - 1);         // threads past the crop: any valid pixel, never stored
            const int oy = t / S, ox = t - oy * S;
            const int4v re = tab[buf * 2 * S + oy], ce = tab[buf * 2 * S + S + ox];
            rq0[u] = rq1[u] = 0; rfl[u] = -1;
            if (re[3]) {
                const int xs0 = ce[0], xs1 = ce[1];
                const bool in0 = xs0 >= 0 && xs0 < a.W, in1 = xs1 >= 0 && xs1 < a.W;
                const int basecol = in0 ? xs0 : xs1;
                const int src0 = in0 ? 1 : 0;
                const int src1 = in1 ? (xs1 == basecol ? 1 : 2) : 0;
                int fl = src0 | (src1 << 2);
                if (src0 | src1) {
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int ys = r ? re[1] : re[0];
                        if (ys >= 0 && ys < a.H) {
                            const int off = (ys * a.W + basecol) * 3;
                            const int c8 = min(off, lim);
                            const unsigned long long q = *reinterpret_cast<const u64_unaligned_r*>(fr + c8);
                            if (r) rq1[u] = q; else rq0[u] = q;
                            fl |= ((off - c8) * 8) << (8 + 8 * r);
                        }
                    }
                }
                rfl[u] = fl;
                special = special || fl != (1 | (2 << 2));
            }
        }
        edge = __builtin_amdgcn_ballot_w64(special) != 0;
    };
    auto store_crop_as = [&](int buf, auto EDGE) __attribute__((always_inline)) {
        constexpr bool E = decltype(EDGE)::value;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int t = tid + u * 256;
            if (t >= S * S) continue;
            const int oy = t / S, ox = t - oy * S;
            const float wy = __int_as_float(tab[buf * 2 * S + oy][2]), wx = __int_as_float(tab[buf * 2 * S + S + ox][2]);
            float v[3] = {0.f, 0.f, 0.f};
            if (rfl[u] >= 0) {
                const unsigned long long q0 = E ? rq0[u] >> ((rfl[u] >> 8) & 0xff) : rq0[u];
                const unsigned long long q1 = E ? rq1[u] >> ((rfl[u] >> 16) & 0xff) : rq1[u];
                const unsigned l0 = (unsigned)q0, h0 = (unsigned)(q0 >> 32), l1 = (unsigned)q1, h1 = (unsigned)(q1 >> 32);
                // bytes of a row piece: B0 G0 R0 B1 | G1 R1 . .  ; channel order R, G, B
                const float f0[3] = {(float)((l0 >> 16) & 0xff), (float)((l0 >> 8) & 0xff), (float)(l0 & 0xff)};
                const float s0[3] = {(float)((h0 >> 8) & 0xff), (float)(h0 & 0xff), (float)(l0 >> 24)};
                const float f1[3] = {(float)((l1 >> 16) & 0xff), (float)((l1 >> 8) & 0xff), (float)(l1 & 0xff)};
                const float s1[3] = {(float)((h1 >> 8) & 0xff), (float)(h1 & 0xff), (float)(l1 >> 24)};
                const int src0 = rfl[u] & 3, src1 = (rfl[u] >> 2) & 3;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float p00 = E ? (src0 ? f0[ch] : 0.f) : f0[ch], p10 = E ? (src0 ? f1[ch] : 0.f) : f1[ch];
                    const float p01 = E ? (src1 == 2 ? s0[ch] : (src1 == 1 ? f0[ch] : 0.f)) : s0[ch];
                    const float p11 = E ? (src1 == 2 ? s1[ch] : (src1 == 1 ? f1[ch] : 0.f)) : s1[ch];
                    v[ch] = __builtin_fmaf(bilerp_r(p00, p01, p10, p11, wx, wy), 0.0078125f, -0.99609375f);      // == (s - 127.5) * 2^-7 bit for bit (pnet_conv1.hip)
                }
            }
            if constexpr (F16) {
                half4 hi, lo;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const half_t h = (half_t)v[ch];
                    hi[ch] = h; lo[ch] = (half_t)(v[ch] - (float)h);
                }
                hi[3] = lo[3] = (half_t)0.f;
                unsigned char* xz = reinterpret_cast<unsigned char*>(xin);
                *reinterpret_cast<half4*>(xz + t * 8) = hi;
                *reinterpret_cast<half4*>(xz + (XPX + t) * 8) = lo;
            } else {
                xin[t * 3 + 0] = v[0]; xin[t * 3 + 1] = v[1]; xin[t * 3 + 2] = v[2];
            }
        }
    };

    make_tab(slot, 0);
    __syncthreads();
    load_crop(slot, 0);
    for (int buf = 0; slot >= 0; buf ^= 1) {
    const int nslot = next_valid(slot + 1);
    if (edge) store_crop_as(buf, std::true_type{});
    else store_crop_as(buf, std::false_type{});
    if (nslot >= 0) make_tab(nslot, buf ^ 1);
    __syncthreads();                                             // the crop and the next slot's tables are in LDS
    if (nslot >= 0) load_crop(nslot, buf ^ 1);                   // global loads fly under this slot's conv

    float* yo = a.y + (int64_t)slot * P * P * COUT;
    for (int band = 0; band < NBAND; ++band) {
        const int r0 = band * 2 * PB;                                         // first conv row of the band
        const int nrow = min(BR, C - r0);
        const int npx = nrow * C;
        if constexpr (F16) {
            // ---- conv on the f16 matrix cores: tiles of 16 flattened conv pixels, one per wave at a time
            typedef half8 half8_a8 __attribute__((aligned(8)));
            const unsigned char* xz = reinterpret_cast<const unsigned char*>(xin);
            const int li_ = lane & 15, kq_ = lane >> 4;
            for (int tl = wave; tl * 16 < npx; tl += 4) {
                const int pidx = min(tl * 16 + li_, npx - 1);
                const int ly = pidx / C, lx = pidx - ly * C;
                const unsigned char* pb = xz + ((r0 + ly + (kq_ >> 1)) * S + lx + 2 * (kq_ & 1)) * 8;
                const half8 h0 = *reinterpret_cast<const half8_a8*>(pb), l0 = *reinterpret_cast<const half8_a8*>(pb + XPX * 8);
                const half8 h1 = *reinterpret_cast<const half8_a8*>(pb + 2 * S * 8), l1 = *reinterpret_cast<const half8_a8*>(pb + (XPX + 2 * S) * 8);
                float4v acc[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    float4v d = {0.f, 0.f, 0.f, 0.f};
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfl[0][ct], h0, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[0][ct], l0, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfl[1][ct], h1, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[1][ct], l1, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[0][ct], h0, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[1][ct], h1, d, 0, 0, 0);
                    acc[ct] = d;
                }
                if (tl * 16 + li_ < npx) {
                    float* o = ot + (tl * 16 + li_) * CS;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const int co = ct * 16 + 4 * kq_;        // the lane's 4 couts of this tile
                        if (co < NG * 4) {
                            float4v v = acc[ct];
                            if (!mono) {
                                v += *reinterpret_cast<const float4v*>(a.bias + co);
                                const float4v sv = *reinterpret_cast<const float4v*>(a.slope + co);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sv[e];
                            }
                            *reinterpret_cast<float4v*>(o + co) = v;
                        }
                    }
                }
            }
        } else
        // ---- conv: units of 64 flattened conv pixels, one unit per wave at a time
        for (int u = wave; u * 64 < npx; u += 4) {
            const int pidx = min(u * 64 + lane, npx - 1);                    // lanes past the band: any valid pixel, not stored
            const int ly = pidx / C, lx = pidx - ly * C;
            const float* xb = xin + ((r0 + ly) * S + lx) * 3;
            float4v acc[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = float4v{0.f, 0.f, 0.f, 0.f};
            // k = (kh, kw, channel) ascending, one fma per k: the chain of the 16x16x4 form
            static_for<27>([&](auto K) {
                constexpr int k = decltype(K)::value;
                constexpr int kh = k / 9, q = k - kh * 9;                    // q = kw * 3 + channel
                const float xv = xb[kh * S * 3 + q];
                static_for<NG>([&](auto G) {
                    constexpr int g = decltype(G)::value;
                    constexpr int c = k * NG + g;
                    acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[c / 16], xv, acc[g], 4, c % 16, 0);
                });
            });
            if (u * 64 + lane < npx) {
                float* o = ot + (u * 64 + lane) * CS;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float4v v = acc[g];
                    if (!mono) {
                        v += *reinterpret_cast<const float4v*>(a.bias + g * 4);
                        const float4v sv = *reinterpret_cast<const float4v*>(a.slope + g * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sv[e];
                    }
                    *reinterpret_cast<float4v*>(o + g * 4) = v;
                }
            }
        }
        __syncthreads();
        // ---- 3x3 / stride 2 ceil-mode max pool of the band's pooled rows; window positions past the map are CLAMPED onto
        // the last valid row / column (a ceil-mode window always starts inside: the clamp repeats a value of the window)
        const int p0 = band * PB;
        const int prow = min(PB, P - p0);
        constexpr int Q4 = COUT / 4;
        for (int e = tid; e < prow * P * Q4; e += 256) {
            const int qd = e % Q4, pp = e / Q4;
            const int pyl = pp / P, px = pp - pyl * P;
            float4v m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int ry = min(2 * pyl + dy, nrow - 1);                   // band-local conv row
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int cx = min(2 * px + dx, C - 1);
                    const float4v v = *reinterpret_cast<const float4v*>(ot + (ry * C + cx) * CS + qd * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) m[k] = vmax_r(m[k], v[k]);
                }
            }
            if (mono) {
                m += *reinterpret_cast<const float4v*>(a.bias + qd * 4);
                const float4v sv = *reinterpret_cast<const float4v*>(a.slope + qd * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = m[k] > 0.f ? m[k] : m[k] * sv[k];
            }
            if constexpr (SPLIT) {
                half4 hi, lo;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const half_t h = (half_t)m[k];
                    hi[k] = h; lo[k] = (half_t)(m[k] - (float)h);
                }
                unsigned char* o2 = a.ys + ((int64_t)slot * P * P + (p0 + pyl) * P + px) * 128 + qd * 8;
                *reinterpret_cast<half4*>(o2) = hi;
                *reinterpret_cast<half4*>(o2 + 64) = lo;
                if constexpr (COUT < 32) {                                   // the K padding (channels COUT .. 31) is zero
                    if (qd == Q4 - 1) {
#pragma unroll
                        for (int z = COUT / 4; z < 8; ++z) {
                            *reinterpret_cast<half4*>(o2 + (z - qd) * 8) = half4{0, 0, 0, 0};
                            *reinterpret_cast<half4*>(o2 + 64 + (z - qd) * 8) = half4{0, 0, 0, 0};
                        }
                    }
                }
            } else {
                *reinterpret_cast<float4v*>(yo + ((int64_t)(p0 + pyl) * P + px) * COUT + qd * 4) = m;
            }
        }
        if (band + 1 < NBAND) __syncthreads();                               // the next band overwrites the conv tile
    }
    __syncthreads();                                             // every wave is done with this slot's crop and conv tile
    slot = nslot;
    }
}

template <int S, int NG, int COUT, int PB, int RPB, bool SPLIT = false, bool LIST = false, bool F16 = false>
int launch_ro(const RoArgs& a, int nslots, hipStream_t s) {
    constexpr int C = S - 2;
    constexpr int rows = 2 * PB + 1 < C ? 2 * PB + 1 : C;
    constexpr int CS = (NG * 4) % 32 ? NG * 4 : NG * 4 + 4;
    constexpr int XFLOATS = F16 ? (2 * (S * S + S + 8) * 8 / 4 + 3) & ~3 : (S * S * 3 + 3) & ~3;
    const size_t lds = (size_t)(XFLOATS + 4 * S * 4 + RPB * 4 + ((RPB + 3) & ~3) + rows * C * CS) * sizeof(float);
    auto kern = crop_conv1_kernel<S, NG, COUT, PB, RPB, SPLIT, LIST, F16>;
    if (lds > 64 * 1024) {
        static FrDevLatch latch;
        if (!fr_raise_lds(reinterpret_cast<const void*>(kern), lds, latch)) {
            fr_set_error("fr_crop_conv1_f32: cannot raise dynamic LDS to %zu bytes", lds);
            return FR_E_LAUNCH;
        }
    }
    kern<<<(nslots + RPB - 1) / RPB, 256, lds, s>>>(a);
    return FR_OK;
}

}  // namespace

extern "C" int fr_crop_conv1_f32(int net, const uint8_t* frames, int nframes, int H, int W, const float* boxes,
                                 const int32_t* counts, int cap, const float* w, const float* bias, const float* slope,
                                 float* y, fr_stream_t stream) {
    FR_REQUIRE(frames && boxes && counts && w && bias && slope && y, "fr_crop_conv1_f32: null pointer");
    FR_REQUIRE(nframes > 0 && cap > 0 && H > 0 && W > 0 && (int64_t)H * W * 3 < (1ll << 31) && (int64_t)H * W * 3 >= 8,
               "fr_crop_conv1_f32: bad frame size");
    FR_REQUIRE((int64_t)nframes * cap < (1ll << 31), "fr_crop_conv1_f32: too many slots");
    RoArgs a{frames, nframes, H, W, boxes, counts, cap, w, bias, slope, y, nullptr, nullptr, nullptr, 0};
    int rc;
    // bands of 4 / 2 pooled rows keep the conv tile at 22 / 33 KB (5 / 2 blocks per CU); 8 / 4 slots per block (measured:
    // one band 785 / 626 us, these bands with one slot per block 569 / 369, as below 508 / 356 us per 64-frame batch)
    hipStream_t s = fr_stream(stream);
    // several slots per block only when there are blocks to spare (a single frame's 512 / 64 slots want one block each)
    const int64_t nslots = (int64_t)nframes * cap;
    if (net == 0) rc = cap % 8 == 0 && nslots >= 8192 ? launch_ro<24, 7, 28, 4, 8>(a, nframes * cap, s) : launch_ro<24, 7, 28, 4, 1>(a, nframes * cap, s);      // R-Net: 24 -> 22 -> 11
    else if (net == 1) rc = cap % 4 == 0 && nslots >= 2048 ? launch_ro<48, 8, 32, 2, 4>(a, nframes * cap, s) : launch_ro<48, 8, 32, 2, 1>(a, nframes * cap, s);  // O-Net: 48 -> 46 -> 23
    else { FR_REQUIRE(false, "fr_crop_conv1_f32: net must be 0 (R-Net) or 1 (O-Net)"); }
    if (rc != FR_OK) return rc;
    FR_CHECK_LAUNCH("crop_conv1_kernel");
    return FR_OK;
}

// The same layer with the pooled map written as SPLIT f16 (x = hi + lo; [slots][P*P][hi 32 ch | lo 32 ch], 128 B per pixel):
// the input format of fr_ro_conv2_split.  Same arithmetic up to the split, which is exact to 22 mantissa bits.
extern "C" int fr_crop_conv1_split(int net, const uint8_t* frames, int nframes, int H, int W, const float* boxes,
                                   const int32_t* counts, int cap, const float* w, const float* bias, const float* slope,
                                   void* y_split, int conv_f16, fr_stream_t stream) {
    FR_REQUIRE(frames && boxes && counts && w && bias && slope && y_split, "fr_crop_conv1_split: null pointer");
    FR_REQUIRE(nframes > 0 && cap > 0 && H > 0 && W > 0 && (int64_t)H * W * 3 < (1ll << 31) && (int64_t)H * W * 3 >= 8,
               "fr_crop_conv1_split: bad frame size");
    FR_REQUIRE((int64_t)nframes * cap < (1ll << 31), "fr_crop_conv1_split: too many slots");
    RoArgs a{frames, nframes, H, W, boxes, counts, cap, w, bias, slope, nullptr, (unsigned char*)y_split, nullptr, nullptr, 0};
    hipStream_t s = fr_stream(stream);
    const int64_t nslots = (int64_t)nframes * cap;
    int rc;
    if (conv_f16) {       // the conv itself on the f16 matrix cores (split precision)
        if (net == 0) rc = cap % RC1_RPB_R == 0 && nslots >= 8192 ? launch_ro<24, 7, 28, RC1_PB_R, RC1_RPB_R, true, false, true>(a, nframes * cap, s) : launch_ro<24, 7, 28, 4, 1, true, false, true>(a, nframes * cap, s);
        else if (net == 1) rc = cap % RC1_RPB_O == 0 && nslots >= 2048 ? launch_ro<48, 8, 32, RC1_PB_O, RC1_RPB_O, true, false, true>(a, nframes * cap, s) : launch_ro<48, 8, 32, 2, 1, true, false, true>(a, nframes * cap, s);
        else { FR_REQUIRE(false, "fr_crop_conv1_split: net must be 0 (R-Net) or 1 (O-Net)"); }
    } else
    if (net == 0) rc = cap % 8 == 0 && nslots >= 8192 ? launch_ro<24, 7, 28, 4, 8, true>(a, nframes * cap, s) : launch_ro<24, 7, 28, 4, 1, true>(a, nframes * cap, s);
    else if (net == 1) rc = cap % 4 == 0 && nslots >= 2048 ? launch_ro<48, 8, 32, 2, 4, true>(a, nframes * cap, s) : launch_ro<48, 8, 32, 2, 1, true>(a, nframes * cap, s);
    else { FR_REQUIRE(false, "fr_crop_conv1_split: net must be 0 (R-Net) or 1 (O-Net)"); }
    if (rc != FR_OK) return rc;
    FR_CHECK_LAUNCH("crop_conv1_kernel (split)");
    return FR_OK;
}

// The f32 layer for a compact LIST of slots (the exact pass over the crops whose approximate logit lies within the margin of
// the threshold, fr_ro_margin_list): row i of `y` is the pooled conv1 map of slot list[i], i < min(*list_count, list_cap).
extern "C" int fr_crop_conv1_list_f32(int net, const uint8_t* frames, int nframes, int H, int W, const float* boxes, int cap,
                                      const int32_t* list, const int32_t* list_count, int list_cap, const float* w,
                                      const float* bias, const float* slope, float* y, fr_stream_t stream) {
    FR_REQUIRE(frames && boxes && list && list_count && w && bias && slope && y, "fr_crop_conv1_list_f32: null pointer");
    FR_REQUIRE(nframes > 0 && cap > 0 && list_cap > 0 && H > 0 && W > 0 && (int64_t)H * W * 3 < (1ll << 31) && (int64_t)H * W * 3 >= 8,
               "fr_crop_conv1_list_f32: bad argument");
    RoArgs a{frames, nframes, H, W, boxes, nullptr, cap, w, bias, slope, y, nullptr, list, list_count, list_cap};
    hipStream_t s = fr_stream(stream);
    int rc;
    if (net == 0) rc = launch_ro<24, 7, 28, 4, 1, false, true>(a, list_cap, s);
    else if (net == 1) rc = launch_ro<48, 8, 32, 2, 1, false, true>(a, list_cap, s);
    else { FR_REQUIRE(false, "fr_crop_conv1_list_f32: net must be 0 (R-Net) or 1 (O-Net)"); }
    if (rc != FR_OK) return rc;
    FR_CHECK_LAUNCH("crop_conv1_kernel (list)");
    return FR_OK;
}

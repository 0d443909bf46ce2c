// MTCNN P-Net conv1 (3 -> 10 channels, 3x3) + PReLU + 2x2/s2 ceil max pool, fed by the pyramid's bilinear resize
// of the u8 BGR frame inside the tile load (the detector half of FaceAnalysis.get, /root/reference/infrenceServer.py:528;
// architecture: SURVEY.md appendix A).  f32 end to end (the detector thresholds and truncates).
//
// Why its own kernel: on the 16x16x4 f32 MFMA (dconv_mfma.hip) this layer used 10 of 16 output columns and 27 of 36
// K rows (47 % of the matrix work useful), and the epilogue ran on 16 pixels x 4 cout groups per wave instruction
// with 37 % of the lanes on padding channels.  Here the matrix op is v_mfma_f32_4x4x1_16B_f32 with the A operand
// BROADCAST (cbsz = 4): one instruction is a 4 (couts) x 64 (pixels, one per lane) x 1 (k) outer product,
//   acc[cout 4g..4g+3][pixel lane] += W[k][4g+r] * X[pixel lane][k],
// so K = 27 exactly, couts 12 of 12 (10 real), every lane of every VALU instruction of the epilogue is a pixel, and a
// pixel's 12 channels sit in one lane: the pooled map leaves as 48 contiguous bytes per pixel (and 64 for the
// split-f16 copy).  All 81 (k, cout group) weight quads live in SIX VGPRs: slot c = 3k + g sits in lanes
// 4 (c % 16) .. +3 of register c / 16 and is selected by the instruction's ABID field.
// The sum runs over k = (kh, kw, channel) ascending, one fma per k: the same chain as the 16x16x4 form.
//
// Tile: (4 RPW) rows x 64 columns of conv pixels per 4-wave block; wave w owns rows 4w' = RPW w .. + RPW - 1, a lane
// owns one column.  Input tile (rows + 2) x 66 level pixels x 3 floats in LDS (lane stride 3 words: conflict-free);
// an input row's 9 values (kw, channel) are read once and feed the up to 3 output rows that use it.
// Blocks are persistent over RPB tiles; the next tile's SOURCE BYTES are fetched under the current tile's MFMAs and
// blended afterwards (same scheme and same arithmetic as the form this replaces).
#include "common.h"
#include <type_traits>

namespace {

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

struct P1Args {
    const uint8_t* frames; int B, FH, FW;      // u8 BGR frames [B,FH,FW,3]
    int H, W;                                  // pyramid level size (resized on the fly)
    const float* w;                            // [9 taps][4][16] f32 (the layer-0 packing of mtcnn._MConv)
    const float* bias; const float* slope;     // [16]
    float* y;                                  // pooled map f32 [B,Hp,Wp,12]
    unsigned char* y_split;                    // optional split-f16 copy, 64 B per pixel (pnet_fused.hip)
    int Ho, Wo, Hp, Wp, regions_x, regions_y;
    const int32_t* list; const int32_t* list_count; int list_cap;     // LIST: the tiles to compute (numbers in the full launch's tile order)
};

typedef unsigned long long u64_unaligned __attribute__((aligned(1)));
struct Lerp { int i0, i1; float w; };
__device__ __forceinline__ Lerp lerp_coord(int d, float ratio, int n) {        // == detect_ops.hip lerp_coord
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

typedef int int2v __attribute__((ext_vector_type(2)));
constexpr int P1_TW = 64, P1_IW = P1_TW + 2;
#ifndef P1_OCC
#define P1_OCC 3
#endif
#ifndef P1_CTU
#define P1_CTU 4
#endif
#ifndef P1_ABL
#define P1_ABL 0
#endif

__device__ __forceinline__ float dpp_xor1(float v) {        // value of lane ^ 1 (quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shr1(float v) {        // value of lane - 1 inside a row of 16 lanes (row_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}

// max(x, y) as v_med3_f32(x, y, +inf) with the +inf in a register the compiler cannot see through: ONE instruction - fmaxf (and
// med3 with a literal +inf, which hipcc folds to it) puts a canonicalising self-max in front of every operand of unknown origin
// (MFMA results; never NaN here).  NOT an inline-asm v_max_f32: the hazard recogniser does not see an asm statement's operands,
// and an asm max placed right behind the MFMA that produces its operand read the register before the matrix pipe had written it
// (round 4: the first of four maxima wrong, in interior tiles only - elsewhere a select sat in between).
__device__ __forceinline__ float opaque_inf() {
    float r;
    asm("v_mov_b32 %0, 0x7f800000" : "=v"(r));
    return r;
}
__device__ __forceinline__ float vmax(float x, float y, float pinf) { return __builtin_amdgcn_fmed3f(x, y, pinf); }

// Lerp tables of a tile: the source rows / columns and weights of its (rows + 2) level rows and 66 level columns are
// computed ONCE per tile by 84 threads (not once per pixel by everybody) and read back from LDS.
//   row entry {i0 * FW * 3, i1 * FW * 3, wy, valid}    column entry {x0 * 3, wx, x1 != x0, valid}
// F16 (round 4, the batch path's band mode): the conv runs on the f16 matrix cores with split-precision operands (x = hi + lo in
// f16, three MFMAs per product, f32 accumulate) and only the split map is written.  The level tile is kept as hi | lo planes of
// 4-channel pixels (8 B: R, G, B, 0) so that the (kw' = 0..3, c' = 0..3) values of a kernel row are 32 contiguous bytes and
// K = (kh, kw', c') = 4 x 4 x 4 with zero weights at kh = 3, kw' = 3, c' = 3 (the packing of ro_conv1.hip's F16 form): two K = 32
// steps per 16 pixels x 16 couts, 6 MFMAs of 16 cycles, against 81 4x4x1 MFMAs of 8 cycles per 64 pixels - 2.6x fewer matrix
// cycles, and a 4x4x1 MFMA holds the SIMD's vector issue for its whole 8 cycles.  The map differs from the f32 form's by ~1e-6; the
// cells whose decision that could touch are re-evaluated from an EXACT map: the f32 form below, run over the tiles such a cell's
// window touches (LIST: tile numbers from fr_pnet_band_tiles).
template <int RPW, int RPB, bool F16 = false, bool LIST = false>
__global__ __launch_bounds__(256, RPW == 4 ? (F16 ? P1_OCC : 3) : 6) void pnet_conv1_kernel(P1Args a) {
    constexpr int TH = 4 * RPW, IH = TH + 2, NPX = IH * P1_IW;
    constexpr int NPF = (NPX + 255) / 256;
    constexpr int NTAB = IH + P1_IW;
    constexpr int XPX = NPF * 256;            // pixel slots of the LDS tile; F16: per plane (the slack behind NPX is what kh = 3 / kw' = 3 read)
    static_assert(!F16 || XPX >= NPX + P1_IW + 4, "slack behind the tile for the zero-weight taps");
    // TWO tile buffers when a block walks several tiles: tile i + 1 is blended into the other buffer BEFORE this wave's conv of tile i, so a
    // tile costs one barrier (two before) and the blend (VALU) of some waves runs under the MFMAs of others
    constexpr int NXB = RPB > 1 ? 2 : 1, NTB = RPB > 1 ? 3 : 1;
    constexpr int XFL = F16 ? XPX * 4 : XPX * 3;
    __shared__ __attribute__((aligned(16))) float xin2[NXB][XFL];                     // slots past NPX (the last slot of some threads) land in the padding
    __shared__ __attribute__((aligned(16))) int4v tab[NTB][NTAB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float pinf = opaque_inf();

    // ---- weights: slot c = 3k + g (k = tap * 3 + channel) -> lanes 4 (c % 16) + r of register c / 16 hold W[k][4g + r]
    float wreg[6];
    half8 wfh[2], wfl[2];                     // F16: A fragments of the two K steps: row = cout lane & 15, k = 8 (lane >> 4) + j
    if constexpr (F16) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kq_ = lane >> 4, kh = 2 * st + (kq_ >> 1), kw = (kq_ & 1) * 2 + (j >> 2), c = j & 3;
                const float w = (kh < 3 && kw < 3 && c < 3) ? a.w[((kh * 3 + kw) * 4 + c) * 16 + (lane & 15)] : 0.f;
                const half_t h = (half_t)w;
                wfh[st][j] = h; wfl[st][j] = (half_t)(w - (float)h);
            }
    } else {
#pragma unroll
    for (int v = 0; v < 6; ++v) {
        const int c = v * 16 + (lane >> 2), r = lane & 3;
        float x = 0.f;
        if (c < 81) {
            const int k = c / 3, g = c - k * 3, tap = k / 3, ch = k - tap * 3;
            x = a.w[(tap * 4 + ch) * 16 + g * 4 + r];
        }
        wreg[v] = x;
    }
    }

    const int per_img = a.regions_x * a.regions_y;
    const int nitems = LIST ? min(*a.list_count, a.list_cap) : per_img * a.B;      // LIST: positions of the tile list
    const int item0 = blockIdx.x * RPB;
    auto tile_of = [&](int item) { return LIST ? a.list[item] : item; };
    const float ryr = (float)a.FH / (float)a.H, rxr = (float)a.FW / (float)a.W;
    const int frame_bytes = a.FH * a.FW * 3;

    auto make_tables = [&](int item, int buf) __attribute__((always_inline)) {
        if (tid < NTAB) {
            const int tile = tile_of(item);
            const int n = tile / per_img, rem = tile - n * per_img;
            const int ry = rem / a.regions_x, rx = rem - ry * a.regions_x;
            int4v e;
            if (tid < IH) {
                const int yy = ry * TH + tid;
                const Lerp l = lerp_coord(yy, ryr, a.FH);
                e = int4v{l.i0 * a.FW * 3, l.i1 * a.FW * 3, __float_as_int(l.w), yy < a.H ? 1 : 0};
            } else {
                const int xx = rx * P1_TW + (tid - IH);
                const Lerp l = lerp_coord(xx, rxr, a.FW);
                e = int4v{l.i0 * 3, __float_as_int(l.w), l.i1 != l.i0 ? 1 : 0, xx < a.W ? 1 : 0};
            }
            tab[buf][tid] = e;
        }
    };

    // ---- input tile: the prefetch keeps the RAW source bytes (two 8-byte row pieces per level pixel); conversion +
    // blend happen in store_tile, after the current tile's MFMAs.  Per slot: 4 data registers + 1 of flags.
    unsigned long long rq0[NPF], rq1[NPF];
    int rfl[NPF];            // -1: pixel outside the level; else bit 0: x1 = x0 + 1 (0: clamped right border),
                             // bits 8..15 / 16..23: pull-back of the two row loads in bits (last frame's last bytes)
    int pos[NPF];            // table slots of the pixel: row | column << 16   (the same for every tile)
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int e = min(tid + u * 256, NPX - 1), iy = e / P1_IW;         // slots past NPX: any valid table entry
        pos[u] = iy | ((IH + e - iy * P1_IW) << 16);
    }
    // Both phases are branch-free per slot (a pixel outside the level loads offset 0 and is zeroed at the end), so the
    // slots' table reads, loads and blends interleave instead of running as five dependent chains
    bool edge = true;        // wave-uniform: some staged pixel is outside the level, sits on the clamped right border or
                             // was loaded with a pull-back (otherwise the blend needs no selects and no shifts)
    auto load_tile = [&](int item, int buf) __attribute__((always_inline)) {
        const int n = tile_of(item) / per_img;
        const uint8_t* fbase = a.frames + (int64_t)n * frame_bytes;
        const bool last = n == a.B - 1;
        const int lim = frame_bytes - 8;
        int4v re[NPF], ce[NPF];
        bool special = false;
#pragma unroll
        for (int u = 0; u < NPF; ++u) { re[u] = tab[buf][pos[u] & 0xffff]; ce[u] = tab[buf][pos[u] >> 16]; }
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const bool valid = (re[u][3] & ce[u][3]) != 0;
            // both corners of a row are 6 adjacent bytes (BGR BGR): ONE unaligned 8-byte load per source row; in the
            // LAST frame the load is pulled back so that it never runs past the end of the buffer
            const int o0 = valid ? re[u][0] + ce[u][0] : 0, o1 = valid ? re[u][1] + ce[u][0] : 0;
            int c0 = o0, c1 = o1, fl = ce[u][2];
            if (last) {
                c0 = min(o0, lim); c1 = min(o1, lim);
                fl |= ((o0 - c0) << 11) | ((o1 - c1) << 19);              // (bytes * 8) << 8 and << 16
            }
            if (P1_ABL & 1) { rq0[u] = c0; rq1[u] = c1; (void)fbase; }
            else {
            rq0[u] = *reinterpret_cast<const u64_unaligned*>(fbase + c0);
            rq1[u] = *reinterpret_cast<const u64_unaligned*>(fbase + c1);
            }
            rfl[u] = valid ? fl : -1;
            special = special || !valid || !(fl & 1);
        }
        edge = last || __builtin_amdgcn_ballot_w64(special) != 0;
    };
    auto store_tile_as = [&](int buf, float* xin, auto EDGE) __attribute__((always_inline)) {
        constexpr bool E = decltype(EDGE)::value;
        float wy[NPF], wx[NPF];
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            wy[u] = __int_as_float(tab[buf][pos[u] & 0xffff][2]);
            wx[u] = __int_as_float(tab[buf][pos[u] >> 16][1]);
        }
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = tid + u * 256;
            float v[3];
            if ((P1_ABL & 16) && F16) {                                                                        // (ablation: no conversion / blend)
                unsigned char* xz = reinterpret_cast<unsigned char*>(xin);
                half4 hh; hh[0] = hh[1] = hh[2] = hh[3] = (half_t)__int_as_float((int)(rq0[u] ^ rq1[u]) & 0x3fffffff);
                *reinterpret_cast<half4*>(xz + e * 8) = hh;
                *reinterpret_cast<half4*>(xz + (XPX + e) * 8) = hh;
                continue;
            }
            const unsigned long long q0 = E ? rq0[u] >> ((rfl[u] >> 8) & 0xff) : rq0[u];
            const unsigned long long q1 = E ? rq1[u] >> ((rfl[u] >> 16) & 0xff) : rq1[u];
            const unsigned l0 = (unsigned)q0, h0 = (unsigned)(q0 >> 32), l1 = (unsigned)q1, h1 = (unsigned)(q1 >> 32);
            const bool two = !E || (rfl[u] & 1) != 0;     // x1 = x0 + 1 (else the clamped border: x1 = x0)
            // bytes of a row piece: B0 G0 R0 B1 | G1 R1 . .  ; output order R, G, B
            const float a00[3] = {(float)((l0 >> 16) & 0xff), (float)((l0 >> 8) & 0xff), (float)(l0 & 0xff)};
            const float a01[3] = {(float)((h0 >> 8) & 0xff), (float)(h0 & 0xff), (float)(l0 >> 24)};
            const float a10[3] = {(float)((l1 >> 16) & 0xff), (float)((l1 >> 8) & 0xff), (float)(l1 & 0xff)};
            const float a11[3] = {(float)((h1 >> 8) & 0xff), (float)(h1 & 0xff), (float)(l1 >> 24)};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float sv = bilerp(a00[c], two ? a01[c] : a00[c], a10[c], two ? a11[c] : a10[c], wx[u], wy[u]);
                // (sv - 127.5) * 2^-7 as ONE fma: scaling by a power of two commutes with the rounding of the difference (no
                // subnormals here: |sv - 127.5| >= 2^-17 or 0), so the bits are those of the two-operation form
                v[c] = (!E || rfl[u] >= 0) && !(P1_ABL & 4) ? __builtin_fmaf(sv, 0.0078125f, -0.99609375f) : 0.f;
            }
            if constexpr (F16) {
                half4 hi, lo;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const half_t h = (half_t)v[c];
                    hi[c] = h; lo[c] = (half_t)(v[c] - (float)h);
                }
                hi[3] = lo[3] = (half_t)0.f;
                unsigned char* xz = reinterpret_cast<unsigned char*>(xin);
                *reinterpret_cast<half4*>(xz + e * 8) = hi;
                *reinterpret_cast<half4*>(xz + (XPX + e) * 8) = lo;
            } else {
                xin[e * 3 + 0] = v[0]; xin[e * 3 + 1] = v[1]; xin[e * 3 + 2] = v[2];
            }
        }
    };
    auto store_tile = [&](int buf, float* xdst) __attribute__((always_inline)) {
        if (edge) store_tile_as(buf, xdst, std::true_type{});
        else store_tile_as(buf, xdst, std::false_type{});
    };

    float4v bias_r[3], slope_r[3];
    bool mono = true;        // every PReLU slope >= 0: bias + PReLU is non-decreasing and commutes with the max pool
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        bias_r[g] = *reinterpret_cast<const float4v*>(a.bias + g * 4);
        slope_r[g] = *reinterpret_cast<const float4v*>(a.slope + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) mono = mono && slope_r[g][e] >= 0.f;
    }
    if (item0 < nitems) {
        make_tables(item0, 0);
        __syncthreads();
        load_tile(item0, 0);
        if (RPB > 1 && item0 + 1 < nitems) make_tables(item0 + 1, 1);
        if (RPB > 2 && item0 + 2 < nitems) make_tables(item0 + 2, 2);
        store_tile(0, xin2[0]);
    }
    __syncthreads();
    if (RPB > 1 && item0 + 1 < nitems) load_tile(item0 + 1, 1);        // raw bytes of tile 1: blended at the top of iteration 0

    for (int rr = 0; rr < RPB; ++rr) {
        const int item = item0 + rr;
        if (item >= nitems) break;
        const int tile_ = tile_of(item);
        const int n = tile_ / per_img, rem = tile_ - n * per_img;
        const int ry = rem / a.regions_x, rx = rem - ry * a.regions_x;
        const int y0 = ry * TH, x0 = rx * P1_TW;
        const float* xin = xin2[rr & (NXB - 1)];
        if constexpr (RPB > 1) {
            // tile rr + 1: raw bytes (fetched one iteration ago) -> the other buffer; tile rr + 2: global loads fly under this tile's MFMAs;
            // tile rr + 3: tables.  Buffers (rr + 1) & 1 / rr % 3 were last read before the barrier that ended iteration rr - 1.
            if (rr + 1 < RPB && item + 1 < nitems) store_tile((rr + 1) % 3, xin2[(rr + 1) & 1]);
            if (rr + 2 < RPB && item + 2 < nitems) load_tile(item + 2, (rr + 2) % 3);
            if (rr + 3 < RPB && item + 3 < nitems) make_tables(item + 3, rr % 3);
        }

        if constexpr (F16) {
            // ---- conv on the f16 matrix cores: the wave's RPW rows x four 16-column tiles; lane (li, kq): pixel column 16 ct + li,
            // couts 4 kq .. 4 kq + 3
            typedef half8 half8_a8 __attribute__((aligned(8)));
            const unsigned char* xz = reinterpret_cast<const unsigned char*>(xin);
            const int li = lane & 15, kq = lane >> 4;
            const float4v b4 = *reinterpret_cast<const float4v*>(a.bias + 4 * kq), s4 = *reinterpret_cast<const float4v*>(a.slope + 4 * kq);
            const int64_t fbase = (int64_t)n * a.Hp;
            auto conv16 = [&](int row, int ct) __attribute__((always_inline)) {          // 16 pixels x 16 couts: raw sums
                const unsigned char* pb = xz + ((row + (kq >> 1)) * P1_IW + ct * 16 + li + 2 * (kq & 1)) * 8;
                half8 h0, l0, h1, l1;
                if (P1_ABL & 8) { h0 = l0 = h1 = l1 = wfh[0]; asm volatile("" :: "v"(pb)); }                 // (ablation: no fragment reads)
                else {
                h0 = *reinterpret_cast<const half8_a8*>(pb); l0 = *reinterpret_cast<const half8_a8*>(pb + XPX * 8);
                h1 = *reinterpret_cast<const half8_a8*>(pb + 2 * P1_IW * 8); l1 = *reinterpret_cast<const half8_a8*>(pb + (XPX + 2 * P1_IW) * 8);
                }
                float4v q = {0.f, 0.f, 0.f, 0.f};
                if (P1_ABL & 2) { asm volatile("" :: "v"(h0), "v"(l0), "v"(h1), "v"(l1)); return q; }      // (ablation: no MFMAs)
                q = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfl[0], h0, q, 0, 0, 0);
                q = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[0], l0, q, 0, 0, 0);
                q = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfl[1], h1, q, 0, 0, 0);
                q = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[1], l1, q, 0, 0, 0);
                q = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[0], h0, q, 0, 0, 0);
                q = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfh[1], h1, q, 0, 0, 0);
                return q;
            };
            auto put16 = [&](const float4v& m, int py, int px) __attribute__((always_inline)) {   // a pooled pixel's four couts
                const int64_t pix = (fbase + py) * a.Wp + px;
                if (P1_ABL & 32) { asm volatile("" :: "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(pix)); return; }   // (ablation: no split / stores)
                half4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const half_t hh = (half_t)m[e];
                    hi[e] = hh; lo[e] = (half_t)(m[e] - (float)hh);
                }
                // 16 B per lane: v_permlane16_swap trades the odd kq rows' hi halves with the even rows' lo halves - an even-kq lane then
                // holds hi of couts 4 kq .. 4 kq + 7, the odd one beside it their lo (partners share the pixel, hence the branch);
                // the four lanes of a pixel write its whole 64-B entry in ONE instruction (two 8-B stores per lane before)
                const int2v hp = __builtin_bit_cast(int2v, hi), lp = __builtin_bit_cast(int2v, lo);
                const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)hp[0], (unsigned)lp[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)hp[1], (unsigned)lp[1], false, false);
                *reinterpret_cast<int4v*>(a.y_split + pix * 64 + (kq & 1) * 32 + (kq >> 1) * 16) = int4v{(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
                if (a.y && kq < 3) *reinterpret_cast<float4v*>(a.y + pix * 12 + 4 * kq) = m;      // (tests: the f32 view of the same map)
            };
            if (mono) {
                // Every slope >= 0: bias + PReLU commute with the max pool.  Pool the RAW sums, then move row pair 1's pooled pixels
                // (even lanes) into the odd lanes beside row pair 0's: bias, PReLU, split and store run once per FOUR conv rows with
                // every lane on a pooled pixel.  PReLU as x + (s - 1) min(x, 0): two operations (the split map is ~1e-6 anyway).
                static_assert(RPW == 4 || !F16, "the packed epilogue pairs the wave's two row pairs");
                const bool interior = y0 + TH <= a.Ho && x0 + P1_TW <= a.Wo;       // block-uniform: no pixel of the tile is outside
                float4v s4m, b4z;
#pragma unroll
                for (int e = 0; e < 4; ++e) { s4m[e] = 4 * kq + e < 10 ? s4[e] - 1.f : -1.f; b4z[e] = 4 * kq + e < 10 ? b4[e] : 0.f; }
                // (couts 10..15: zero weights and zero bias -> 0 + (-1) min(0, 0) = 0, as the map wants them)
                const int odd = li & 1;
                const int py = ((y0 + wave * RPW) >> 1) + odd;
#pragma unroll P1_CTU
                for (int ct = 0; ct < 4; ++ct) {
                    float4v m[2];
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        float4v d0 = conv16(wave * RPW + 2 * rp, ct), d1 = conv16(wave * RPW + 2 * rp + 1, ct);
                        if (!interior) {
                            const bool cin = x0 + ct * 16 + li < a.Wo;
                            const bool in0 = cin && y0 + wave * RPW + 2 * rp < a.Ho, in1 = cin && y0 + wave * RPW + 2 * rp + 1 < a.Ho;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { d0[e] = in0 ? d0[e] : -INFINITY; d1[e] = in1 ? d1[e] : -INFINITY; }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = vmax(d0[e], d1[e], pinf);
                            m[rp][e] = vmax(t, dpp_xor1(t), pinf);
                        }
                    }
                    float4v v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // lane - 1's m[1] for the odd lanes.  The move is pinned in front of the select: written as `odd ? dpp(m1) : m0`
                        // hipcc runs it under the odd lanes' EXEC mask, where its source lanes are disabled and read as 0
                        float sh = dpp_shr1(m[1][e]);
                        asm volatile("" : "+v"(sh));
                        const float x = (odd ? sh : m[0][e]) + b4z[e];
                        v[e] = __builtin_fmaf(fminf(x, 0.f), s4m[e], x);
                    }
                    const int px = ((x0 + ct * 16) >> 1) + (li >> 1);
                    if (py < a.Hp && px < a.Wp) put16(v, py, px);
                }
            } else {
#pragma unroll
            for (int rp = 0; rp < RPW / 2; ++rp) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    float4v d[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = wave * RPW + 2 * rp + h;
                        float4v q = conv16(row, ct);
                        // bias + PReLU, then the pool (a conv pixel outside the map: -inf): the order that is right for any slope
                        q += b4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) q[e] = q[e] > 0.f ? q[e] : q[e] * s4[e];
                        const bool inside = y0 + row < a.Ho && x0 + ct * 16 + li < a.Wo;
                        d[h] = inside ? q : float4v{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    }
                    float4v m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = vmax(d[0][e], d[1][e], pinf);
                        m[e] = 4 * kq + e < 10 ? vmax(t, dpp_xor1(t), pinf) : 0.f;
                    }
                    const int py = (y0 + wave * RPW + 2 * rp) >> 1, px = (x0 + ct * 16 + li) >> 1;
                    if ((li & 1) == 0 && py < a.Hp && px < a.Wp) put16(m, py, px);
                }
            }
            }
        } else {
        float4v acc[RPW][3];
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[r][g] = float4v{0.f, 0.f, 0.f, 0.f};
        const float* xb = xin + ((wave * RPW) * P1_IW + lane) * 3;
        // input row ir of the wave's band feeds output rows ir - kh (kh = 0..2): for an output row the k order is
        // kh, then kw, then channel - ascending k
        static_for<RPW + 2>([&](auto IR) {
            constexpr int ir = decltype(IR)::value;
            float xv[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) xv[q] = xb[ir * P1_IW * 3 + q];          // q = kw * 3 + channel
            static_for<9>([&](auto Q) {
                constexpr int q = decltype(Q)::value;
                static_for<3>([&](auto KH) {
                    constexpr int kh = decltype(KH)::value, r = ir - kh;
                    if constexpr (r >= 0 && r < RPW) {
                        static_for<3>([&](auto G) {
                            constexpr int g = decltype(G)::value;
                            constexpr int c = (kh * 9 + q) * 3 + g;          // 3k + g, k = (kh*3 + kw)*3 + channel = kh*9 + q
                            if (P1_ABL & 2) { if (g == 0) asm volatile("" :: "v"(xv[q])); }
                            else
                            acc[r][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[c / 16], xv[q], acc[r][g], 4, c % 16, 0);
                        });
                    }
                });
            });
        });

        // ---- epilogue: bias + PReLU, 2x2/s2 ceil-mode max pool (rows in registers, columns by one lane exchange)
        const int xcol = x0 + lane;
        const bool interior = y0 + TH <= a.Ho && x0 + P1_TW <= a.Wo;       // block-uniform: no pixel of the tile is outside
        auto put = [&](const float4v (&pv)[3], int py, int px) __attribute__((always_inline)) {
            const int64_t pix = ((int64_t)n * a.Hp + py) * a.Wp + px;
            float* o = a.y + pix * 12;
#pragma unroll
            for (int g = 0; g < 3; ++g) *reinterpret_cast<float4v*>(o + g * 4) = pv[g];
            if (a.y_split) {
                // split-f16 copy for the fused conv2/conv3 kernel: [hi ch0-7 | hi ch8-15 | lo ch0-7 | lo ch8-15],
                // channels 10..15 zero
                half8 hi[2], lo[2];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float x = c < 10 ? pv[c >> 2][c & 3] : 0.f;
                    const half_t h = (half_t)x;
                    hi[c >> 3][c & 7] = h;
                    lo[c >> 3][c & 7] = (half_t)(x - (float)h);
                }
                unsigned char* o2 = a.y_split + pix * 64;
                *reinterpret_cast<half8*>(o2) = hi[0];
                *reinterpret_cast<half8*>(o2 + 16) = hi[1];
                *reinterpret_cast<half8*>(o2 + 32) = lo[0];
                *reinterpret_cast<half8*>(o2 + 48) = lo[1];
            }
        };
        if (mono) {
            // pool the raw accumulators first (a quarter of the bias / PReLU work), then pack the wave's two pooled rows
            // into even / odd lanes so that the rest of the epilogue runs with every lane on a pooled pixel
            if (!interior) {
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    const bool inside = y0 + wave * RPW + r < a.Ho && xcol < a.Wo;
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[r][g] = inside ? acc[r][g] : float4v{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
            }
            float4v pv[RPW / 2][3];
#pragma unroll
            for (int rp = 0; rp < RPW / 2; ++rp)
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float m = vmax(acc[2 * rp][g][e], acc[2 * rp + 1][g][e], pinf);
                        pv[rp][g][e] = vmax(m, dpp_xor1(m), pinf);
                    }
            float4v q[3];
            int py = (y0 + wave * RPW) >> 1;
            bool writer = (lane & 1) == 0;
            if constexpr (RPW == 4) {
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = dpp_shr1(pv[1][g][e]);           // odd lane 2j+1 <- second pooled row, pixel j
                        q[g][e] = (lane & 1) ? t : pv[0][g][e];
                    }
                py += lane & 1;
                writer = true;
            } else {
#pragma unroll
                for (int g = 0; g < 3; ++g) q[g] = pv[0][g];
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                q[g] += bias_r[g];
#pragma unroll
                for (int e = 0; e < 4; ++e) q[g][e] = q[g][e] > 0.f ? q[g][e] : q[g][e] * slope_r[g][e];
            }
            const int px = (x0 >> 1) + (lane >> 1);
            if (writer && py < a.Hp && px < a.Wp) put(q, py, px);
        } else {
            // a negative slope makes PReLU non-monotonic: activate every pixel, then pool
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const bool inside = y0 + wave * RPW + r < a.Ho && xcol < a.Wo;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    float4v v = acc[r][g] + bias_r[g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope_r[g][e];
                    acc[r][g] = inside ? v : float4v{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
            }
#pragma unroll
            for (int rp = 0; rp < RPW / 2; ++rp) {
                float4v pv[3];
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float m = vmax(acc[2 * rp][g][e], acc[2 * rp + 1][g][e], pinf);
                        pv[g][e] = vmax(m, dpp_xor1(m), pinf);
                    }
                const int py = (y0 + wave * RPW + 2 * rp) >> 1, px = xcol >> 1;
                if ((lane & 1) == 0 && py < a.Hp && px < a.Wp) put(pv, py, px);
            }
        }
        }   // !F16
        if (rr + 1 < RPB && item + 1 < nitems) __syncthreads();      // tile rr + 1 and the tables of tile rr + 3 are in LDS; every wave is done with tile rr
    }
}

template <int RPW, int RPB, bool F16 = false, bool LIST = false>
int launch_p1(P1Args a, hipStream_t s) {
    constexpr int TH = 4 * RPW;
    a.regions_x = (a.Wo + P1_TW - 1) / P1_TW;
    a.regions_y = (a.Ho + TH - 1) / TH;
    const int64_t nitems = LIST ? a.list_cap : (int64_t)a.regions_x * a.regions_y * a.B;
    if (nitems >= (1ll << 31)) return FR_E_INVALID;
    pnet_conv1_kernel<RPW, RPB, F16, LIST><<<(unsigned)((nitems + RPB - 1) / RPB), 256, 0, s>>>(a);
    return FR_OK;
}

}  // namespace

// called by fr_dconv_mfma_f32 (layer 0); arguments checked there
int fr_pnet_conv1_launch(const uint8_t* frames, int B, int FH, int FW, int H, int W, const float* w, const float* bias,
                         const float* slope, float* y, void* y_split, hipStream_t s) {
    P1Args a{frames, B, FH, FW, H, W, w, bias, slope, y, (unsigned char*)y_split, H - 2, W - 2, 0, 0, 0, 0, nullptr, nullptr, 0};
    a.Hp = (a.Ho + 1) / 2; a.Wp = (a.Wo + 1) / 2;
    // big launches: 16-row tiles, blocks persistent over 8 tiles; small pyramid levels / single frames: 8-row tiles, one
    // tile per block (too few tiles to fill 256 CUs: more, shorter blocks)
    const int64_t tiles16 = (int64_t)((a.Ho + 15) / 16) * ((a.Wo + P1_TW - 1) / P1_TW) * B;
    if (tiles16 >= 4096) return launch_p1<4, 8>(a, s);
    if (tiles16 >= 512) return launch_p1<4, 1>(a, s);
    return launch_p1<2, 1>(a, s);
}

// The band mode's pair (round 4).  mode 0: the conv on the f16 matrix cores with split-precision operands, the split map only
// (y optional: the f32 view of the same values, tests).  mode 1: the exact f32 form over a LIST of tiles of 16 x 64 conv pixels
// (tile numbers in the order ((frame * regions_y) + ry) * regions_x + rx with regions of 16 rows x 64 columns: fr_pnet_band_tiles)
// - writes y (and y_split) for those tiles only.
extern "C" int fr_pnet_conv1_band(int mode, const uint8_t* frames, int B, int FH, int FW, int H, int W, const float* w,
                                  const float* bias, const float* slope, float* y, void* y_split, const int32_t* list,
                                  const int32_t* list_count, int list_cap, fr_stream_t stream) {
    FR_REQUIRE(frames && w && bias && slope && B > 0 && FH > 0 && FW > 0 && H >= 3 && W >= 3, "fr_pnet_conv1_band: bad argument");
    P1Args a{frames, B, FH, FW, H, W, w, bias, slope, y, (unsigned char*)y_split, H - 2, W - 2, 0, 0, 0, 0, list, list_count, list_cap};
    a.Hp = (a.Ho + 1) / 2; a.Wp = (a.Wo + 1) / 2;
    hipStream_t s = fr_stream(stream);
    int rc;
    if (mode == 0) {
        FR_REQUIRE(y_split, "fr_pnet_conv1_band: mode 0 writes the split map");
        const int64_t tiles16 = (int64_t)((a.Ho + 15) / 16) * ((a.Wo + P1_TW - 1) / P1_TW) * B;
        rc = tiles16 >= 4096 ? launch_p1<4, 8, true>(a, s) : launch_p1<4, 1, true>(a, s);
    } else if (mode == 1) {
        FR_REQUIRE(y && list && list_count && list_cap > 0, "fr_pnet_conv1_band: mode 1 needs the f32 map and a tile list");
        rc = launch_p1<4, 1, false, true>(a, s);
    } else { FR_REQUIRE(false, "fr_pnet_conv1_band: mode must be 0 or 1"); }
    if (rc != FR_OK) { fr_set_error("fr_pnet_conv1_band: too many tiles"); return rc; }
    FR_CHECK_LAUNCH("pnet_conv1_kernel (band)");
    return FR_OK;
}

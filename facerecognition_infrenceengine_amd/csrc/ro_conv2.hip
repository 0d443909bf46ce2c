// MTCNN R-Net / O-Net SECOND conv layer (28 -> 48 at 11x11, 32 -> 64 at 23x23) + PReLU + 3x3 / stride-2 max pool on the f16
// matrix cores with SPLIT-PRECISION operands, and the two small kernels of the exact pass that goes with it
// (detector half of FaceAnalysis.get, /root/reference/infrenceServer.py:528; conventions of oracle/detect.py).
//
// Why: on the f32 matrix instruction (v_mfma_f32_16x16x4_f32, 1/16 of the f16 rate) these two layers were 1.42 ms of a
// 6.5 ms detector batch (64 x 1080p: 32 768 R-Net and 4 096 O-Net crops), at 0.56 of that instruction's roof.  Here every f32
// operand is x = hi + lo (hi = f16(x), lo = f16(x - hi): 22 mantissa bits) and a product is hi*hi + lo*hi + hi*lo on
// v_mfma_f32_16x16x32_f16 with f32 accumulation - three MFMAs at 16x the rate, the dropped lo*lo term is 2^-22 relative.
// The input arrives ALREADY split (fr_crop_conv1_split writes [slot][pixel][hi 32 ch | lo 32 ch], 128 B per pixel), so it
// goes to LDS by LDS-DMA and no conversion runs here; K step = one tap x 32 channels (R-Net's 28 are zero-padded).
//
// Parity: the sums differ from the f32 fma chain in the last bits (measured on the heads: ~1e-6).  The cascade THRESHOLDS the
// face probability, so the crops whose logit difference lies within `margin` of the threshold are re-evaluated by the all-f32
// layers (fr_ro_margin_list -> fr_crop_conv1_list_f32 -> fr_dconv_mfma_f32 on the compact list -> fr_ro_scatter_rows): every
// keep / reject decision is that of exact f32 arithmetic.  A kept crop further from the threshold keeps the split-precision
// score and regression (the same distance from the oracle as a different f32 summation order).
//
// Geometry.  GEMM view D[cout][pixel] = sum_{tap, ch} W[cout][tap][ch] X[pixel + tap][ch]; A = weights (a wave owns ONE
// 16-cout tile and keeps its nine taps' hi and lo fragments in registers: 72 VGPRs), B = pixels (LDS).  One workgroup per CU,
// wave = (cout tile, pixel-tile group), persistent over consecutive items, the NEXT item's image arriving by LDS-DMA in a
// second buffer under the current item's K loop:
//   R-Net: item = three crops (3 x 9 x 9 = 243 conv pixels = 16 MFMA pixel tiles), 3 cout tiles x 4 pixel groups = 12 waves;
//   O-Net: item = one band of a crop (11 conv rows x 21 = 231 pixels = 15 tiles; two bands per crop share conv row 10),
//          4 cout tiles x 2 pixel groups = 8 waves.
// (First version: 3 / 4 waves per workgroup, two workgroups per CU, one buffer: the two workgroups ran in lockstep - load, K
// loop and pool ADDED up, 148 + 167 + 83 us for the R-Net batch by compile-time ablation - and hipcc sank every fragment
// read to its first use behind an lgkmcnt(0).)
// LDS image: a hi plane and a lo plane of 64-B pixel rows, image pitch W + 2 pixels; 16-B chunk c of pixel (Y, X, crop g) sits
// at chunk c ^ (((X + KC Y + GK g) >> 1) & 3) with KC = W_out mod 8, GK = pixels per crop mod 8.  Then the key of the pixel a
// lane reads for output pixel q and tap (kh, kw) is (((q + kw + KC kh) & 7) >> 1) - and q = 16 t + lane & 15, so it is ONE
// value per (lane, tap), whatever the tile: a fragment address is (per-tile pixel base) + (per-tap chunk term) + immediate.
// Brute-forced over the real tiles (tools/lds_swizzle_search.py): every ds_read_b128 is conflict-free (R-Net: but for the tiles that straddle two
// crops), where plain or padded rows cost 1.7 - 1.9x the LDS cycles.
// The conv map goes through an LDS tile (aliasing the item's own input buffer; O-Net: 32 couts at a time) to the pool;
// pool-before-activation when every PReLU slope is >= 0 (the same bits, a ninth of the bias / PReLU work), as in the other
// detector kernels.
#include "common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
#ifndef RC2_CFG
#define RC2_CFG 0          // developer A/B of the workgroup shapes
#endif
#ifndef RC2_STAGGER
#define RC2_STAGGER 0      // start delay of the second workgroup of a CU, in units of 1 024 cycles (measured 0 / 3 / 6: no difference)
#endif
#ifndef RC2_ABL
#define RC2_ABL 0          // developer ablations (tools/abl_rc2.py builds its own objects): 1 no MFMAs, 2 no fragment reads, 4 no input DMA, 8 no pool
#endif

namespace {

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

struct Rc2Args {
    const unsigned char* xs;               // split map [slots][WIN * WIN][128 B] (fr_crop_conv1_split)
    const float* w;                        // [COUT][9][32] f32 (channels >= Cin zero)
    const float* bias; const float* slope; // [COUT]
    float* y;                              // [slots][PO][PO][COUT] f32
    const int32_t* counts; int cap;        // slot s holds a crop iff s % cap < counts[s / cap]
    int nslots, per_block;                 // items per block (consecutive)
    int32_t* zero;                         // optional: a device word block 0 clears (the exact pass's list counter, used next)
    unsigned long long* stamps;            // developer builds (RC2_ABL & 16): per-wave phase cycle sums, else unused
};
#define RC2_STAMP(var)                                                                   \
    do {                                                                                 \
        if (RC2_ABL & 16) {                                                              \
            __builtin_amdgcn_sched_barrier(0);                                           \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");  \
            __builtin_amdgcn_sched_barrier(0);                                           \
        }                                                                                \
    } while (0)

// NPH: pixel-tile groups per cout tile (waves = NCT * NPH: wave = (cout tile, pixel group)); PROUNDS: the conv map goes through
// the LDS pool tile in PROUNDS rounds of NCT / PROUNDS cout tiles (the tile aliases ONE input buffer)
template <int WIN, int HB, int WOUT, int HC, int G, int NCT, int NPH, int NBAND, int BROWS, int PR, int PO, int PROUNDS, int NBUFS, int DEPTH_>
struct Rc2Cfg {
    static constexpr int P = WIN + 2;                          // LDS image pitch in pixels
    static constexpr int NPX = G * HC * WOUT;                  // conv pixels of an item
    static constexpr int NT = (NPX + 15) / 16;                 // MFMA pixel tiles
    static constexpr int NTW = (NT + NPH - 1) / NPH;           // ... per wave (tile k of group ph: ph + k NPH)
    static constexpr int PLANE = G * HB * P * 64;              // bytes of one LDS plane
    static constexpr int NW = NCT * NPH;
    static constexpr int NPIECE = (2 * PLANE + 1023) / 1024;   // LDS-DMA pieces of an item
    static constexpr int NPW = (NPIECE + NW - 1) / NW;         // pieces per wave
    static constexpr int COUT = NCT * 16;
    static constexpr int PC = COUT / PROUNDS;                  // couts of a pool round
    static constexpr int CS = PC + 4;                          // pool tile pixel stride (floats)
    static constexpr int POOL_BYTES = NPX * CS * 4;
    static constexpr int BUF = NPIECE * 1024 > POOL_BYTES ? NPIECE * 1024 : (POOL_BYTES + 1023) / 1024 * 1024;   // one input buffer (it also holds the item's pool tile)
    static constexpr int PRM_OFF = NBUFS * BUF;                // bias[COUT] | slope[COUT] f32 behind the buffers
    static constexpr int LDS_BYTES = PRM_OFF + 2 * COUT * 4;
    static constexpr int SLOT_BYTES = WIN * WIN * 128;
    static constexpr int KC = WOUT % 8, GK = (HC * WOUT) % 8;
    static constexpr int NTHR = NW * 64;
    static constexpr int NSTEP = 9 * NTW;                      // (tap, tile) steps of the K loop
    static constexpr int DEPTH = DEPTH_;                       // fragment pairs in flight ahead of the MFMAs
    static constexpr int DMA_EVERY = (NSTEP * 5 / 8) / NPW > 0 ? (NSTEP * 5 / 8) / NPW : 1;     // K-loop steps between two LDS-DMA pieces of the next item (all in the loop's first 5/8: they must have landed at its end)
    static constexpr int Q4 = PC / 4;                          // cout quads of a pool round
    static_assert(NBUFS == 1 || NSTEP - 4 >= NPW, "every piece of the next item is issued inside the K loop");
    static_assert(NCT % PROUNDS == 0, "pool rounds");
    static_assert(2 * PLANE + 3 * P * 64 < 65536, "ds_read immediates");
    static_assert(2 * (PR - 1) + 2 < HC && 2 * (PO - 1) + 2 < WOUT + 0 * BROWS, "pool windows lie inside the band");
};

template <int WIN, int HB, int WOUT, int HC, int G, int NCT, int NPH, int NBAND, int BROWS, int PR, int PO, int PROUNDS, int NBUFS, int DEPTH_, int MINW>
__global__ __launch_bounds__(NCT * NPH * 64, MINW) void ro_conv2_split_kernel(Rc2Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = Rc2Cfg<WIN, HB, WOUT, HC, G, NCT, NPH, NBAND, BROWS, PR, PO, PROUNDS, NBUFS, DEPTH_>;
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int ct = wave % NCT, ph = wave / NCT;
    const int nitems = (a.nslots + G - 1) / G * NBAND;
    const int it_end = min(nitems, (int)(blockIdx.x + 1) * a.per_block);
    if (a.zero && blockIdx.x == 0 && tid == 0) *a.zero = 0;
    // the next item at or behind `it` that holds a crop, or -1.  A frame's candidates are a prefix of its slots: behind an
    // item without one the search jumps to the first slot of the next frame
    auto next_valid = [&](int it) {
        while (it < it_end) {
            const int s0 = it / NBAND * G;
            bool any = false;
            int last = s0;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int sl = s0 + g;
                if (sl < a.nslots) {
                    const int f = sl / a.cap;
                    any = any || sl - f * a.cap < a.counts[f];
                    last = sl;
                }
            }
            if (any) return it;
            const int nxt = ((last / a.cap + 1) * a.cap) / G * NBAND;
            it = nxt > it ? nxt : it + 1;
        }
        return -1;
    };
    int cur = next_valid(blockIdx.x * a.per_block);
    if (cur < 0) return;

    // ---- weights of this wave's cout tile: A fragments (row = cout li, k = 8 kq + j of the tap), hi and lo, in registers
    half8 wh[9], wl[9];
    {
        const float* wrow = a.w + (size_t)(ct * 16 + li) * 9 * 32 + 8 * kq;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float4v v0 = *reinterpret_cast<const float4v*>(wrow + tap * 32);
            const float4v v1 = *reinterpret_cast<const float4v*>(wrow + tap * 32 + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = j < 4 ? v0[j] : v1[j - 4];
                const half_t h = (half_t)v;
                wh[tap][j] = h;
                wl[tap][j] = (half_t)(v - (float)h);
            }
        }
    }
    const float4v bias_r = *reinterpret_cast<const float4v*>(a.bias + ct * 16 + 4 * kq);
    const float4v slope_r = *reinterpret_cast<const float4v*>(a.slope + ct * 16 + 4 * kq);
    bool mono = true;
    for (int c = 0; c < C::COUT; ++c) mono = mono && a.slope[c] >= 0.f;

    // ---- LDS-DMA source offsets of this lane's chunk in each of the wave's pieces (the same for every item): LDS chunk n of
    // the image = (plane, pixel row, position c'); it holds source chunk c' ^ key of that pixel, or zeros (padding columns,
    // the tail of the last piece)
    unsigned voff[C::NPW];
#pragma unroll
    for (int i = 0; i < C::NPW; ++i) {
        const int n = (wave + C::NW * i) * 64 + lane;
        unsigned o = 0x80000000u;
        if (n < 2 * C::PLANE / 16) {
            const int pl = n / (C::PLANE / 16), m = n - pl * (C::PLANE / 16);
            const int row = m >> 2, cp = m & 3;
            const int g = row / (HB * C::P), rr = row - g * (HB * C::P);
            const int Y = rr / C::P, X = rr - Y * C::P;
            if (X < WIN) {
                const int key = ((X + C::KC * Y + C::GK * g) >> 1) & 3;
                o = (unsigned)(g * C::SLOT_BYTES + (Y * WIN + X) * 128 + pl * 64 + ((cp ^ key) << 4));
            }
        }
        voff[i] = o;
    }
    auto item_rsrc = [&](int it) {
        const int sg = it / NBAND, band = it - sg * NBAND;
        const int s0 = sg * G;
        const unsigned char* base = a.xs + (size_t)s0 * C::SLOT_BYTES + (size_t)band * BROWS * WIN * 128;
        const int nsl = min(G, a.nslots - s0);
        const unsigned bytes = NBAND == 1 ? (unsigned)nsl * C::SLOT_BYTES : (unsigned)(HB * WIN * 128);
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
    };
    auto piece = [&](__amdgpu_buffer_rsrc_t rs, int buf, int i) {
        const int j = wave + C::NW * i;
        if (j < C::NPIECE && !(RC2_ABL & 4))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds2 + buf * C::BUF + j * 1024), 16, voff[i], 0, 0, 0);
    };
    // ---- per-tile LDS addresses (buffer 0) of this lane's output pixel; lanes / tiles past the item's pixels: the last pixel
    const unsigned lbase = (unsigned)reinterpret_cast<uintptr_t>((lds_ptr_t)lds2);
    unsigned rt[C::NTW];
#pragma unroll
    for (int k = 0; k < C::NTW; ++k) {
        const int q = min((ph + k * NPH) * 16 + li, C::NPX - 1);
        const int g = q / (HC * WOUT), qq = q - g * (HC * WOUT);
        const int y = qq / WOUT, x = qq - y * WOUT;
        rt[k] = lbase + ((g * HB + y) * C::P + x) * 64;
    }

    // Workgroups that share a CU start together and, left alone, stay in lockstep: they load together, share the matrix pipe
    // together (each at half speed) and pool together - the phases ADD up (measured).  The workgroup whose waves sit in the
    // odd wave slots of their SIMDs (the second to arrive on the CU) starts half an item late; from then on one computes
    // while the other loads / pools.
    if (RC2_STAGGER) {
        const unsigned hwid = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((4 - 1) << 11));     // wave slot in its SIMD
        if (__builtin_amdgcn_readfirstlane(hwid) & 1) {
#pragma unroll 1
            for (int i = 0; i < RC2_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);      // 16 x 64 cycles each
        }
    }
    // pool: bias / slope of every cout in LDS (a thread reads its quad's there: no global latency behind the barrier)
    float* prm = reinterpret_cast<float*>(lds2 + C::PRM_OFF);
    for (int c = tid; c < C::COUT; c += C::NTHR) { prm[c] = a.bias[c]; prm[C::COUT + c] = a.slope[c]; }
    if (NBUFS == 2) {
        const __amdgpu_buffer_rsrc_t rs0 = item_rsrc(cur);
#pragma unroll
        for (int i = 0; i < C::NPW; ++i) piece(rs0, 0, i);
    }
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, ph_sum[6] = {0, 0, 0, 0, 0, 0};
    int nxt = next_valid(cur + 1);
    for (int b = 0; cur >= 0; b ^= (NBUFS - 1)) {
        RC2_STAMP(st[0]);
        const int sg = cur / NBAND, band = cur - sg * NBAND;
        const int s0 = sg * G;
        // which of the item's slots hold a crop (scalar; the pool stores only those)
        unsigned smask = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int sl = s0 + g;
            if (sl < a.nslots) {
                const int f = sl / a.cap;
                if (sl - f * a.cap < a.counts[f]) smask |= 1u << g;
            }
        }
        if (NBUFS == 1) {                                   // one buffer: the item's image is fetched here, behind the previous pool
            __syncthreads();
            const __amdgpu_buffer_rsrc_t rs0 = item_rsrc(cur);
#pragma unroll
            for (int i = 0; i < C::NPW; ++i) piece(rs0, 0, i);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of the item have landed ...
        __syncthreads();                                    // ... everybody's have, and the other buffer's pool tile has been read
        RC2_STAMP(st[1]);
        // the next item's image arrives under this item's K loop: its LDS-DMA pieces are issued from INSIDE the MFMA stream, a
        // piece every DMA_EVERY steps (all of a wave's pieces in front of the loop held it for ~1 000 cycles: stamps)
        const __amdgpu_buffer_rsrc_t rsn = item_rsrc(nxt >= 0 ? nxt : cur);
        const int nn = nxt >= 0 ? next_valid(nxt + 1) : -1;  // (scalar loads: their latency hides under the K loop)
        RC2_STAMP(st[2]);

        // ---- K loop: 9 taps x NTW pixel tiles.  The fragment reads run DEPTH (tap, tile) steps ahead of the MFMAs that use them,
        // by hand: ds_read_b128 as inline asm, released by counted lgkmcnt waits that carry the fragments as operands (hipcc
        // otherwise sinks every read to its first use and waits lgkmcnt(0) in front of every MFMA)
        float4v acc[C::NTW];
#pragma unroll
        for (int k = 0; k < C::NTW; ++k) acc[k] = float4v{0.f, 0.f, 0.f, 0.f};
        const unsigned boff = b ? (unsigned)C::BUF : 0u;
        unsigned xd[8];                                     // chunk term by the tap's key shift d = (kw + KC kh) & 7 (unused ones vanish)
#pragma unroll
        for (int d = 0; d < 8; ++d) xd[d] = boff + (unsigned)((kq ^ (((li + d) & 7) >> 1)) << 4);
        half8 fh[C::DEPTH + 1], fl[C::DEPTH + 1];
        auto rd = [&](auto S) {
            constexpr int s = decltype(S)::value;
            constexpr int tap = s / C::NTW, k = s - tap * C::NTW, kh = tap / 3, kw = tap - kh * 3;
            constexpr int imm = (kh * C::P + kw) * 64;
            if (RC2_ABL & 2) { fh[s % (C::DEPTH + 1)] = wh[tap]; fl[s % (C::DEPTH + 1)] = wl[tap]; return; }
            // (the address add sits inside the asm: left to hipcc, the 9 x NTW sums are hoisted in front of the loop and spill)
            unsigned ad;
            const unsigned r0 = rt[k], x0 = xd[(kw + C::KC * kh) & 7];
            asm volatile("v_add_u32 %2, %3, %4\n\tds_read_b128 %0, %2 offset:%5\n\tds_read_b128 %1, %2 offset:%6"
                         : "=&v"(fh[s % (C::DEPTH + 1)]), "=&v"(fl[s % (C::DEPTH + 1)]), "=&v"(ad)
                         : "v"(r0), "v"(x0), "n"(imm), "n"(imm + C::PLANE));
        };
        static_for<C::DEPTH>([&](auto S) { rd(S); });
        static_for<C::NSTEP>([&](auto S) {
            constexpr int s = decltype(S)::value;
            constexpr int tap = s / C::NTW, k = s - tap * C::NTW;
            if constexpr (s + C::DEPTH < C::NSTEP) rd(std::integral_constant<int, s + C::DEPTH>{});
            if constexpr (NBUFS == 2 && s >= 2 && (s - 2) % C::DMA_EVERY == 0 && (s - 2) / C::DMA_EVERY < C::NPW) {
                if (nxt >= 0) piece(rsn, b ^ 1, (s - 2) / C::DMA_EVERY);
            }
            constexpr int later = (C::NSTEP - 1 - s) < C::DEPTH ? (C::NSTEP - 1 - s) : C::DEPTH;      // reads issued behind this step's
            half8& bh = fh[s % (C::DEPTH + 1)];
            half8& bl = fl[s % (C::DEPTH + 1)];
            if (!(RC2_ABL & 2)) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh), "+v"(bl) : "n"(2 * later));
            if (RC2_ABL & 1) { asm volatile("" :: "v"(bh), "v"(bl)); }
            else {
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[tap], bh, acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[tap], bh, acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[tap], bl, acc[k], 0, 0, 0);
            }
        });

        RC2_STAMP(st[3]);
        if (RC2_ABL & 8) {
            float4v sum = acc[0];
#pragma unroll
            for (int k = 1; k < C::NTW; ++k) sum += acc[k];
            if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) a.y[tid] = sum[0];
            cur = nxt; nxt = nn;
            continue;
        }
        // ---- conv map -> LDS tile [pixel][CS] (over this item's input buffer) in PROUNDS rounds of cout tiles, each followed
        // by the 3x3 / s2 pool of its couts
        float* pt = reinterpret_cast<float*>(lds2 + b * C::BUF);
#pragma unroll
        for (int round = 0; round < PROUNDS; ++round) {
            __syncthreads();                                // every wave is done reading the image / the previous round's tile
            if (ct / (NCT / PROUNDS) == round) {
#pragma unroll
                for (int k = 0; k < C::NTW; ++k) {
                    const int q = (ph + k * NPH) * 16 + li;
                    if (q < C::NPX) {
                        float4v v = acc[k];
                        if (!mono) {
                            v += bias_r;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope_r[e];
                        }
                        *reinterpret_cast<float4v*>(pt + q * C::CS + (ct % (NCT / PROUNDS)) * 16 + 4 * kq) = v;
                    }
                }
            }
            __syncthreads();
            if (round == 0) RC2_STAMP(st[4]);
            constexpr int Q4 = C::Q4;
            for (int e = tid; e < G * PR * PO * Q4; e += C::NTHR) {
                const int qd = e % Q4, pp = e / Q4;
                const int g = pp / (PR * PO), p2 = pp - g * (PR * PO);
                const int pr = p2 / PO, pc = p2 - pr * PO;
                const int slot = s0 + g;
                if (!((smask >> g) & 1)) continue;
                const float* src = pt + ((g * HC + 2 * pr) * WOUT + 2 * pc) * C::CS + qd * 4;
                float4v m = *reinterpret_cast<const float4v*>(src);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        if (dy == 0 && dx == 0) continue;
                        const float4v v = *reinterpret_cast<const float4v*>(src + (dy * WOUT + dx) * C::CS);
#pragma unroll
                        for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
                    }
                const int co = round * C::PC + qd * 4;
                if (mono) {
                    m += *reinterpret_cast<const float4v*>(prm + co);
                    const float4v sv = *reinterpret_cast<const float4v*>(prm + C::COUT + co);
#pragma unroll
                    for (int k = 0; k < 4; ++k) m[k] = m[k] > 0.f ? m[k] : m[k] * sv[k];
                }
                *reinterpret_cast<float4v*>(a.y + (((size_t)slot * PO + band * PR + pr) * PO + pc) * C::COUT + co) = m;
            }
        }
        RC2_STAMP(st[5]);
        if (RC2_ABL & 16) {
#pragma unroll
            for (int i = 0; i < 5; ++i) ph_sum[i] += st[i + 1] - st[i];
            ph_sum[5] += 1;
        }
        cur = nxt; nxt = nn;
    }
    if ((RC2_ABL & 16) && a.stamps && lane == 0) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * C::NW + wave) * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = ph_sum[i];
    }
#endif
}

template <int WIN, int HB, int WOUT, int HC, int G, int NCT, int NPH, int NBAND, int BROWS, int PR, int PO, int PROUNDS, int NBUFS, int DEPTH_, int MINW>
int launch_rc2(Rc2Args a, hipStream_t s) {
    using C = Rc2Cfg<WIN, HB, WOUT, HC, G, NCT, NPH, NBAND, BROWS, PR, PO, PROUNDS, NBUFS, DEPTH_>;
    auto kern = ro_conv2_split_kernel<WIN, HB, WOUT, HC, G, NCT, NPH, NBAND, BROWS, PR, PO, PROUNDS, NBUFS, DEPTH_, MINW>;
    const size_t lds = C::LDS_BYTES;
    if (lds > 64 * 1024) {
        static FrDevLatch latch;
        if (!fr_raise_lds(reinterpret_cast<const void*>(kern), lds, latch)) {
            fr_set_error("fr_ro_conv2_split: cannot raise dynamic LDS to %zu bytes", lds);
            return FR_E_LAUNCH;
        }
    }
    // workgroups persistent over consecutive items, about three rounds of the workgroups a chip holds (12 waves per CU)
    const int nitems = (a.nslots + G - 1) / G * NBAND;
    const int resident = 256 * (MINW * 4 / C::NW);
    int per = (nitems + 2 * resident - 1) / (2 * resident);
    if (per < 1) per = 1;
    a.per_block = per;
    kern<<<(nitems + per - 1) / per, C::NTHR, lds, s>>>(a);
    return FR_OK;
}

__global__ void ro_margin_list_kernel(const float* head, int nhead, const int32_t* counts, int nslots, int cap, float lthr,
                                      float margin, int32_t* list, int32_t* lcount, int lcap) {
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= nslots) return;
    const int f = slot / cap;
    if (slot - f * cap >= counts[f]) return;
    const float d = head[(size_t)slot * nhead + 1] - head[(size_t)slot * nhead];
    if (fabsf(d - lthr) <= margin) {
        const int pos = atomicAdd(lcount, 1);
        if (pos < lcap) list[pos] = slot;
    }
}

__global__ void ro_scatter_rows_kernel(const float* src, const int32_t* list, const int32_t* lcount, int lcap, int ncols, float* dst) {
    const int n = min(*lcount, lcap);
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int i = e / ncols, c = e - i * ncols;
    if (i < n) dst[(size_t)list[i] * ncols + c] = src[(size_t)i * ncols + c];
}

}  // namespace

extern "C" int fr_ro_conv2_split(int net, const void* x_split, const float* w, const float* bias, const float* slope, float* y,
                                 int nslots, const int32_t* counts, int cap, int32_t* zero_word, fr_stream_t stream) {
    FR_REQUIRE(x_split && w && bias && slope && y && counts, "fr_ro_conv2_split: null pointer");
    FR_REQUIRE(nslots > 0 && cap > 0 && nslots % cap == 0, "fr_ro_conv2_split: nslots must be frames x cap");
    Rc2Args a{(const unsigned char*)x_split, w, bias, slope, y, counts, cap, nslots, 1, zero_word,
              (RC2_ABL & 16) ? reinterpret_cast<unsigned long long*>(zero_word) : nullptr};
    if (RC2_ABL & 16) a.zero = nullptr;      // developer build: the pointer argument carries the stamp buffer
    hipStream_t s = fr_stream(stream);
    int rc;
    //                       WIN HB WOUT HC G NCT NPH NBAND BROWS PR PO PROUNDS MINW
    const int cfg = RC2_CFG;
    //                                    WIN HB WOUT HC G NCT NPH NBAND BROWS PR PO PROUNDS NBUFS DEPTH MINW
    // Measured (64 x 1080p worth of slots, tools/abl_rc2.py; us): R-Net 6 waves x 2 workgroups per CU 340 - 380, 12 waves x 1: 277 - 302
    // (first version, 3 waves x 2, one buffer, compiler-scheduled reads: 391); O-Net 4 waves x 2 (one buffer) 212 - 215, 8 waves x 1
    // (two buffers) 232 - 283 (first version 237)
    if (net == 0 && cfg == 1) rc = launch_rc2<11, 11, 9, 9, 2, 3, 2, 1, 0, 4, 4, 1, 2, 2, 3>(a, s);        // two crops, 3 cout tiles x 2 pixel groups = 6 waves, two workgroups per CU
    else if (net == 0) rc = launch_rc2<11, 11, 9, 9, 3, 3, 4, 1, 0, 4, 4, 1, 2, 2, 3>(a, s);                 // three crops, 3 x 4 = 12 waves, one per CU
    else if (net == 1 && cfg == 0) rc = launch_rc2<23, 13, 21, 11, 1, 4, 1, 2, 10, 5, 10, 1, 1, 3, 2>(a, s);   // one band, 4 waves (cout tiles), one buffer, two workgroups per CU
    else if (net == 1) rc = launch_rc2<23, 13, 21, 11, 1, 4, 2, 2, 10, 5, 10, 1, 2, 3, 2>(a, s);            // one band, 4 x 2 = 8 waves, two buffers, one per CU
    else { FR_REQUIRE(false, "fr_ro_conv2_split: net must be 0 (R-Net) or 1 (O-Net)"); }
    if (rc != FR_OK) return rc;
    FR_CHECK_LAUNCH("ro_conv2_split_kernel");
    return FR_OK;
}

extern "C" int fr_ro_margin_list(const float* head, int nhead, const int32_t* counts, int nframes, int cap, float logit_thr,
                                 float margin, int32_t* list, int32_t* list_count, int list_cap, fr_stream_t stream) {
    FR_REQUIRE(head && counts && list && list_count, "fr_ro_margin_list: null pointer");
    FR_REQUIRE(nhead >= 2 && nframes > 0 && cap > 0 && list_cap > 0 && margin >= 0.f, "fr_ro_margin_list: bad argument");
    hipStream_t s = fr_stream(stream);
    const int nslots = nframes * cap;
    ro_margin_list_kernel<<<(nslots + 255) / 256, 256, 0, s>>>(head, nhead, counts, nslots, cap, logit_thr, margin, list, list_count, list_cap);
    FR_CHECK_LAUNCH("ro_margin_list_kernel");
    return FR_OK;
}

extern "C" int fr_ro_scatter_rows(const float* src, const int32_t* list, const int32_t* list_count, int list_cap, int ncols,
                                  float* dst, fr_stream_t stream) {
    FR_REQUIRE(src && list && list_count && dst && list_cap > 0 && ncols > 0, "fr_ro_scatter_rows: bad argument");
    ro_scatter_rows_kernel<<<(list_cap * ncols + 255) / 256, 256, 0, fr_stream(stream)>>>(src, list, list_count, list_cap, ncols, dst);
    FR_CHECK_LAUNCH("ro_scatter_rows_kernel");
    return FR_OK;
}

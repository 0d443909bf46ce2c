// 3x3 / stride 1 / pad 1 convolution with an LDS-STAGED INPUT TILE (halo) reused by all nine taps:
// the IResNet body convs (82 % of the embed FLOPs) at 112x112 .. 14x14.  Same math, operand maps and
// epilogue as conv_mfma.hip; what changes is the traffic: the generic implicit GEMM re-gathers the
// input tile for every tap (operand fetch = (BN+BM)*128 B per 64-deep K step, measured bound by the
// L2/Infinity-Cache gather rate), here the input rows + halo of one 64-channel chunk are brought in
// ONCE per chunk (9 K steps) and only the weight tile streams per step.
//
// Tile = TH whole image rows of one image (196 pixels at 14x14 / 7x28, 224 at 4x56 / 2x112) x BN couts;
// 512 threads = 8 waves = WN (64 couts each) x WP pixel groups.  LDS: halo chunk [NXBUF][XROWS][64 ch]
// (zero border by out-of-range LDS-DMA), weight tile [2][BN][64]; rows of 128 B with a 16-B chunk XOR.
// Variants in use (fr_conv_halo_try): LEAN <2,13,256|320,1,4,true> for 14x14 / 28x28 (BN = 128, one halo
// buffer, <= 128 VGPRs, two blocks per CU, 13th pixel tile shared by cout = SPLIT); single-chunk
// <1,14,384|512,1,4> for Cin = 64 at 56x56 / 112x112; the pipelined two-halo-buffer schedule
// (<4|2,13,320,2,2>, one block per CU) is kept behind FR_HALO_LEAN=0.
// Lean schedule per K step q = chunk*9 + tap: vmcnt(0) -> raw barrier -> issue W(q+1) (LDS-DMA, scalar
// offset only) -> 2 x (A and B fragments via ds_read_b128, 13 MFMAs); at a chunk end: barrier + halo reload.
// No gather arithmetic in the loop: per-lane halo bases are fixed, taps are scalar row offsets.
#include "common.h"
#include <type_traits>

struct HaloP {
    const half_t* x; const half_t* w; half_t* y;     // F8 variant: x, w are fp8 e4m3 bytes; y (f16) may be NULL
    const float* bias; const float* slope; const half_t* res;
    const float* oscale; unsigned char* y8; float y8_mul;   // F8: acc * oscale[cout] first; optional fp8 copy of the output: (y - y8_sub[cout]) * y8_mul
    const float* y8_sub;                                    // per-channel centre of the consumer's input (or NULL)
    int B, H, W, Cin, Cout, bias_mode;
    int TH, tiles_per_img;      // output rows per tile, H / TH
    int G;                      // images per tile (1; 4 at 7x7, where TH = H: a tile = G whole images, each with its own halo)
    unsigned xbytes, wbytes;
    unsigned long long* stamps; // diagnostic build only (FR_DBG_STAMPS=<device ptr>): per-wave segment cycle sums
    int stagger;                // diagnostic build only (FR_HALO_STAGGER): start delay of every second block, in 512-cycle units
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int int8v __attribute__((ext_vector_type(8)));

// 4 floats -> 4 fp8 e4m3 (OCP) bytes, saturating at +-448 (the conversion itself would produce NaN past the range)
__device__ __forceinline__ int pack_fp8x4(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return v;
}

#define HK 64          // channels per chunk
#define HALO_NW4_DEFAULT 0


// scheduling pattern for one half step: 4+PT ds_read_b128 (+ ~3 address VALU each) spread over 4*PT MFMAs
#define INTERLEAVE_READS_MFMA()                                            \
    do {                                                                   \
        _Pragma("unroll") for (int g_ = 0; g_ < NA + PT; ++g_) {            \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);             \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);             \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);             \
        }                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * PT - 2 * (4 + PT), 0); \
    } while (0)

#define STAMP(var)                                                                          \
    do {                                                                                    \
        if (STAMPS) {                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                              \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");      \
            __builtin_amdgcn_sched_barrier(0);                                              \
        }                                                                                   \
    } while (0)

// WN: 64-cout groups per block; NPT: 16-pixel MFMA tiles per M tile; XROWS: halo row capacity (multiple of 64,
// >= (TH+2)*(W+2)); NXBUF: halo buffers (1 when Cin == 64: a single chunk); MINW: waves per SIMD to fit
// (4 = two blocks per CU).
// LEAN: one halo buffer, one fragment set, <= 128 VGPRs -> TWO blocks per CU (16 waves): the fixed cost of a
// block (first loads, epilogue: ~30 % of a 28x28 tile) and its barrier stalls are covered by the neighbour.
// NA: 16-cout MFMA tiles per wave (4 = 64 couts).  (A 32-cout / 2-pixel-group split was tried for balance: it spilled at
// 128 VGPRs and was replaced by SPLIT below.)
// F8: fp8 e4m3 activations and weights on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales; twice the f16 MFMA
// rate).  A chunk is 128 CHANNELS (still 128 B per LDS row), a K step is one tap x 128 channels = ONE MFMA per tile
// pair, fed by the same two ds_read_b128 per operand as the f16 step's two K = 32 MFMAs (lane quarter fq holds bytes
// [16 fq, +16) and [64 + 16 fq, +16) of the row: the k order inside an MFMA is free as long as A and B agree), so the
// LDS image, swizzle, DMA pattern and barrier structure are the f16 kernel's; the K loop has half as many steps.
// NW: waves per block.  8 (512 threads) everywhere except the "lean4" variant <2,13,*,1,2,true,4,false,4>: FOUR waves
// per block, each owning 64 couts x 6 pixel tiles + two cout tiles of the shared 13th pixel tile (26 accumulator tiles,
// <= 256 VGPRs), still two blocks per CU.  Same LDS image and barrier structure, but a wave's register tile is twice as
// large, so a K step reads 26 KB of fragments for 52 MFMAs instead of 18 KB for 26 (0.50 vs 0.69 KB of LDS traffic per
// MFMA) and a barrier joins 4 waves instead of 8.
template <int WN, int NPT, int XROWS, int NXBUF, int MINW, bool STAMPS, bool LEAN = false, int NA = 4, bool F8 = false, int NW = 8, int ABL = 0>
__global__ __launch_bounds__(NW * 64, MINW) void conv_halo_kernel(HaloP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(!F8 || LEAN, "the fp8 variant exists for the lean schedule only");
    constexpr int ES = F8 ? 1 : 2;                     // bytes per element
    constexpr int CH = 128 / ES;                       // channels per 128-B chunk
    constexpr int BN = 16 * NA * WN;
    constexpr int WP = NW / WN;
    constexpr int NT = NW * 64;                       // threads per block
    // SPLIT (lean 196-pixel tiles): 13 pixel tiles over 4 pixel groups used to be 4+4+4+(1 real + 3 padding) tiles,
    // i.e. 16 MFMA tiles per wave for 12.25 useful.  Now every wave owns 3 pixel tiles x 64 couts and the 13th pixel
    // tile is shared by cout: wave (wn, wp) computes its couts [wn*64 + wp*16, +16) -> 13 MFMA tiles per wave.
    constexpr bool SPLIT = LEAN && (WN == 2 || WN == 4) && NPT == 13 && NA == 4;
    constexpr int PT = SPLIT ? 12 / WP : (NPT + WP - 1) / WP;
    constexpr int XT = SPLIT ? NA / WP : 1;           // cout tiles of the shared 13th pixel tile per wave (1 or 2)
    constexpr int NWI = BN / (8 * NW);                // weight LDS-DMA instructions per thread per step
    constexpr int NXI = XROWS / (8 * NW);             // halo LDS-DMA instructions per thread per chunk
    static_assert(BN % (8 * NW) == 0 && XROWS % (8 * NW) == 0 && NW % WN == 0, "tile / wave split");
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0;
    STAMP(tA);
    extern __shared__ __attribute__((aligned(16))) half_t lds[];
    half_t* xs = lds;                                 // [NXBUF][XROWS][HK]
    half_t* ws = lds + NXBUF * XROWS * HK;            // [2][BN][HK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: LDS-DMA bases stay in SGPRs
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
    const int n = (mt / p.tiles_per_img) * p.G, y0 = (mt - (mt / p.tiles_per_img) * p.tiles_per_img) * p.TH;   // first image of the tile
    const int HW = p.W + 2;
    const int IH = (p.TH + 2) * HW;                    // halo rows of ONE image's part of the tile
    const int nhalo = p.G * IH;
    const int IPX = p.TH * p.W;                        // output pixels of one image's part

    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);

    // LDS image: row = halo pixel, 16-B chunk' = chunk ^ key(row).  Weights: key = row & 7.  Halo: key =
    // (row - 2*hy) & 7 = the pixel's index in a W-pitch raster: 16 consecutive output pixels (any tap) then
    // hit keys p..p+15 with the row parity of a dense tile -> ds_read_b128 fragment reads stay conflict-free
    // across image-row wraps (with key = row & 7 the wrap lanes collide: measured 35 % extra LDS cycles).
    const int lrow = lane >> 3;
    const int schunk = (lane & 7) ^ (lrow & 7);
    unsigned xoff[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int h = (wave * NXI + i) * 8 + lrow;
        const int gi = h / IH, hr = h - gi * IH;                 // image of the tile, halo row inside its part
        const int hy = hr / HW, hx = hr - hy * HW;
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = h < nhalo && n + gi < p.B && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const int xch = (lane & 7) ^ ((h - 2 * (gi * (p.TH + 2) + hy)) & 7);
        xoff[i] = ok ? (unsigned)((((n + gi) * p.H + iy) * p.W + ix) * p.Cin * ES + xch * 16) : 0x80000000u;
    }
    unsigned woff[NWI];
    const int K = 9 * p.Cin;
#pragma unroll
    for (int i = 0; i < NWI; ++i) woff[i] = (unsigned)((cout0 + (wave * NWI + i) * 8 + lrow) * K * ES + schunk * 16);

    auto issue_x = [&](int c) {
        half_t* dst = xs + (c & (NXBUF - 1)) * XROWS * HK;
#pragma unroll
        for (int i = 0; i < NXI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(dst + (wave * NXI + i) * 8 * HK), 16, xoff[i],
                                                     c * (HK * 2), 0, 0);
    };
    // W(q): K offset (tap*Cin + chunk*64) halves; tracked incrementally for q+2 (no divisions in the loop)
    int wq_tap = 0, wq_c = 0;
    auto issue_w = [&](int q) {
        half_t* dst = ws + (q & 1) * BN * HK;
        const int soff = (wq_tap * p.Cin + wq_c * CH) * ES;
#pragma unroll
        for (int i = 0; i < NWI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(dst + (wave * NWI + i) * 8 * HK), 16, woff[i], soff, 0, 0);
        if (++wq_tap == 9) { wq_tap = 0; ++wq_c; }
    };

    // per-lane halo base (top-left of the 3x3 window) of each of this wave's pixel tiles
    const int fr = lane & 15, fq = lane >> 4;
    int hoff[PT], kbs[PT];       // byte offset of the window's top-left halo row; swizzle key base
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int px = (wp * PT + j) * 16 + fr;
        int hb = 0, kb = 0;
        if (px < p.G * IPX) {
            const int gi = px / IPX, r = px - gi * IPX;
            const int oy = r / p.W, ox = r - oy * p.W;
            hb = gi * IH + oy * HW + ox;
            kb = hb - 2 * (gi * (p.TH + 2) + oy);
        }
        hoff[j] = hb * (HK * 2);
        kbs[j] = kb;
    }

    int hoffx = 0, kbx = 0;      // SPLIT: the shared 13th pixel tile
    if constexpr (SPLIT) {
        const int px = 12 * 16 + fr;
        if (px < p.G * IPX) {
            const int gi = px / IPX, r = px - gi * IPX;
            const int oy = r / p.W, ox = r - oy * p.W;
            const int hb = gi * IH + oy * HW + ox;
            hoffx = hb * (HK * 2);
            kbx = hb - 2 * (gi * (p.TH + 2) + oy);
        }
    }

    float4v acc[NA][PT];
    float4v accx[XT];
#pragma unroll
    for (int t = 0; t < XT; ++t) accx[t] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

    const int nq = 9 * (p.Cin / CH), nchunk = p.Cin / CH;
    // fragment sets: (a0,b0) = K half 0 of the current step, (a1,b1) = K half 1.  Software pipeline:
    //   top:  read (a1,b1)(q)            | MFMA half 0 (q)      <- LDS reads fly under the MFMAs
    //   mid:  lgkmcnt(0), counted vmcnt, barrier: W(q+1) landed, W(q)'s buffer free
    //         issue W(q+2) [+ next halo] | read (a0,b0)(q+1)    | MFMA half 1 (q)
    int4v a0[NA], b0[PT], a1[NA], b1[PT];       // 8 halves each, kept as 4 dwords (no per-element repacking)
    int4v ax0[XT], ax1[XT], bx0 = {0, 0, 0, 0}, bx1 = {0, 0, 0, 0};   // SPLIT: 13th tile operands
#pragma unroll
    for (int t = 0; t < XT; ++t) ax0[t] = ax1[t] = int4v{0, 0, 0, 0};
    auto read_x = [&](int4v (&ax)[XT], int4v& bx, int q, int c, int toff, int kh, int kk) {
        // A: this wave's 16-cout group(s) of the shared tile (re-read from LDS: a register select by wp would be dynamic)
        const half_t* wl = ws + (q & 1) * BN * HK + (wn * NA * 16) * HK;
#pragma unroll
        for (int t = 0; t < XT; ++t) {
            const int row = (wp * XT + t) * 16 + fr;
            ax[t] = *reinterpret_cast<const int4v*>(wl + row * HK + (((kk * 4 + fq) ^ (row & 7)) << 3));
        }
        const char* xl = reinterpret_cast<const char*>(xs + (c & (NXBUF - 1)) * XROWS * HK) + toff * (HK * 2);
        const int key = (kbx + toff - 2 * kh) & 7;
        bx = *reinterpret_cast<const int4v*>(xl + hoffx + (((kk * 4 + fq) ^ key) << 4));
    };
    auto read_frags = [&](int4v (&a)[NA], int4v (&b)[PT], int q, int c, int toff, int kh, int kk) {
        const half_t* wl = ws + (q & 1) * BN * HK + (wn * NA * 16) * HK;
        const char* xl = reinterpret_cast<const char*>(xs + (c & (NXBUF - 1)) * XROWS * HK) + toff * (HK * 2);   // scalar part
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = i * 16 + fr;
            a[i] = *reinterpret_cast<const int4v*>(wl + row * HK + (((kk * 4 + fq) ^ (row & 7)) << 3));
        }
        const int ks = toff - 2 * kh;
        const int cq = kk * 4 + fq;
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int key = (kbs[j] + ks) & 7;
            b[j] = *reinterpret_cast<const int4v*>(xl + hoff[j] + ((cq ^ key) << 4));
        }
    };
    auto mfma_all = [&](int4v (&a)[NA], int4v (&b)[PT]) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a[i]), __builtin_bit_cast(half8, b[j]),
                                                                   acc[i][j], 0, 0, 0);
    };
    auto mfma8 = [&](const int4v& alo, const int4v& ahi, const int4v& blo, const int4v& bhi, float4v c) {
        const int8v a = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
        const int8v b = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
        return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    };
#pragma unroll
    for (int i = 0; i < NA; ++i) a0[i] = a1[i] = int4v{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < PT; ++j) b0[j] = b1[j] = int4v{0, 0, 0, 0};

    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, s01 = 0, s12 = 0, s23 = 0, s34 = 0, s45 = 0;
    if constexpr (F8) {
        // fp8 schedule: chunk loop x 9-tap inner loop; the tap step is ONE basic block (the weight DMA of the next step
        // is issued unconditionally - the step after the last reads rows that are never consumed and is drained before
        // the epilogue), so that the sched_group_barrier pipeline below can order reads against MFMAs.
        issue_x(0);
        issue_w(0);
        int qq = 0;
#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
            int toff = 0, kw = 0, kh = 0;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap, ++qq) {
                const int q = qq;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // W(q) (and the halo issued at the last chunk end)
                __builtin_amdgcn_s_barrier();
                issue_w(q + 1);
                // an 8-VGPR operand per tile: keep the pixel fragments (B) of the step live and stream the weight
                    // fragments (A) one cout tile at a time, so that the kernel stays within 128 VGPRs (two blocks per CU)
                    {
                        const char* xl = reinterpret_cast<const char*>(xs) + toff * (HK * 2);
                        const int ks = toff - 2 * kh;
#pragma unroll
                        for (int j = 0; j < PT; ++j) {
                            const int key = (kbs[j] + ks) & 7;
                            b0[j] = *reinterpret_cast<const int4v*>(xl + hoff[j] + ((fq ^ key) << 4));
                            b1[j] = *reinterpret_cast<const int4v*>(xl + hoff[j] + (((4 + fq) ^ key) << 4));
                        }
                    }
                    const half_t* wl = ws + (q & 1) * BN * HK + (wn * NA * 16) * HK;
                    const int alo_off = fr * HK + ((fq ^ (fr & 7)) << 3), ahi_off = fr * HK + (((4 + fq) ^ (fr & 7)) << 3);
                    int4v alo = *reinterpret_cast<const int4v*>(wl + alo_off), ahi = *reinterpret_cast<const int4v*>(wl + ahi_off);
#pragma unroll
                    for (int i = 0; i < NA; ++i) {
                        int4v nlo = alo, nhi = ahi;
                        if (i + 1 < NA) {                           // next cout tile's fragment flies under this tile's MFMAs
                            nlo = *reinterpret_cast<const int4v*>(wl + (i + 1) * 16 * HK + alo_off);
                            nhi = *reinterpret_cast<const int4v*>(wl + (i + 1) * 16 * HK + ahi_off);
                        }
#pragma unroll
                        for (int j = 0; j < PT; ++j) acc[i][j] = mfma8(alo, ahi, b0[j], b1[j], acc[i][j]);
                        alo = nlo; ahi = nhi;
                    }
                    if constexpr (SPLIT) {
                        read_x(ax0, bx0, q, c, toff, kh, 0); read_x(ax1, bx1, q, c, toff, kh, 1);
                        accx[0] = mfma8(ax0[0], ax1[0], bx0, bx1, accx[0]);
                    }
                    // pin the order above (hipcc otherwise hoists all 18 fragment reads = 72 VGPRs to the top of the step
                    // and spills): B + first A fragment, then per cout tile 3 MFMAs with the next tile's 2 reads between them
#define FR_SGB_READ() do { __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } while (0)
#define FR_SGB_MFMA() __builtin_amdgcn_sched_group_barrier(0x008, 1, 0)
                    static_assert(!F8 || (PT == 3 && NA == 4 && SPLIT && NW == 8), "the fp8 schedule is written for the 4 x 3 + 1 tile split");
#pragma unroll
                    for (int g_ = 0; g_ < 2 * PT + 2; ++g_) FR_SGB_READ();
#pragma unroll
                    for (int i = 0; i < NA - 1; ++i) { FR_SGB_MFMA(); FR_SGB_READ(); FR_SGB_MFMA(); FR_SGB_READ(); FR_SGB_MFMA(); }
                    FR_SGB_MFMA(); FR_SGB_READ(); FR_SGB_MFMA(); FR_SGB_READ(); FR_SGB_MFMA(); FR_SGB_READ(); FR_SGB_READ(); FR_SGB_MFMA();

                ++kw; ++toff;
                if (kw == 3) { kw = 0; toff += HW - 3; ++kh; }
            }
            if (c + 1 < nchunk) {                                    // reload the single halo buffer
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                        // every wave is done reading chunk c
                issue_x(c + 1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the surplus weight DMA must not land in the epilogue tile
    } else
    if constexpr (LEAN) {
        static_assert(NXBUF == 1 || NXBUF == 2, "lean variant: one halo buffer, or two (the 16-wave variant) so that the next chunk's halo lands under this chunk's MFMAs");
        STAMP(tB);
        if (FR_DEBUG && (p.stagger & 0xffff)) {   // experiment: de-phase blocks that may share a CU
            const int mode = (p.stagger >> 8) & 255, n = p.stagger & 255;
            const bool late = mode == 0 ? (int)blockIdx.x >= (int)gridDim.x / 2 : mode == 1 ? (blockIdx.x & 1) : ((blockIdx.x >> 3) & 1);
            if (late) for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(8);
        }
        issue_x(0);
        issue_w(0);
        constexpr int abl = ABL;      // debug build only: compile-time ablation bits (1 no MFMA, 2 no reads, 4 no W DMA, 8 no barrier)
        int tap = 0, c = 0, toff = 0, kw = 0, kh = 0;
        for (int q = 0; q < nq; ++q) {
            STAMP(t0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // W(q) (and a halo issued at the last chunk end)
            STAMP(t1);
            if constexpr (!(abl & 8)) __builtin_amdgcn_s_barrier();
            STAMP(t2);
            if (q + 1 < nq && !(abl & 4)) issue_w(q + 1);
            if constexpr (NXBUF == 2) { if (tap == 0 && c + 1 < nchunk) issue_x(c + 1); }   // lands during this chunk's 9 steps
            STAMP(t3);
            if constexpr (!(abl & 2)) {
                read_frags(a0, b0, q, c, toff, kh, 0);
                if constexpr (SPLIT) read_x(ax0, bx0, q, c, toff, kh, 0);
            }
            if constexpr ((abl & 1) != 0) {
#pragma unroll
                for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(a0[i]));
#pragma unroll
                for (int j = 0; j < PT; ++j) asm volatile("" ::"v"(b0[j]));
                asm volatile("" ::"v"(ax0[0]), "v"(bx0));
            } else {
            mfma_all(a0, b0);
            if constexpr (SPLIT) {
#pragma unroll
                for (int t = 0; t < XT; ++t)
                    accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ax0[t]), __builtin_bit_cast(half8, bx0), accx[t], 0, 0, 0);
            }
            }
            if constexpr (!(abl & 2)) {
                read_frags(a1, b1, q, c, toff, kh, 1);
                if constexpr (SPLIT) read_x(ax1, bx1, q, c, toff, kh, 1);
            }
            if constexpr ((abl & 1) != 0) {
#pragma unroll
                for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(a1[i]));
#pragma unroll
                for (int j = 0; j < PT; ++j) asm volatile("" ::"v"(b1[j]));
                asm volatile("" ::"v"(ax1[0]), "v"(bx1));
            } else {
            mfma_all(a1, b1);
            if constexpr (SPLIT) {
#pragma unroll
                for (int t = 0; t < XT; ++t)
                    accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ax1[t]), __builtin_bit_cast(half8, bx1), accx[t], 0, 0, 0);
            }
            }
            STAMP(t4);
            ++tap; ++kw; ++toff;
            if (kw == 3) { kw = 0; toff += HW - 3; ++kh; }
            if (tap == 9) {
                tap = 0; kw = 0; toff = 0; kh = 0; ++c;
                if constexpr (NXBUF == 1) {
                if (c < nchunk) {                                    // reload the single halo buffer
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                    // every wave is done reading chunk c-1
                    issue_x(c);
                }
                }
            }
            STAMP(t5);
            if (STAMPS) { s01 += t1 - t0; s12 += t2 - t1; s23 += t3 - t2; s34 += t4 - t3; s45 += t5 - t4; }
        }
    } else {
    issue_x(0);
    issue_w(0);
    if (nq > 1) issue_w(1);
    // W(0) and the first halo must be visible before the first fragment reads
    if (nq > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWI) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(a0, b0, 0, 0, 0, 0, 0);
    int tap = 0, c = 0, toff = 0, kw = 0, kh = 0;
    bool x_inflight = false;               // a halo was issued at the previous mid-step (after the W loads)
    // One K step.  LDS-DMA issue costs ~100 cycles per piece and sits in the wave's in-order stream, so the
    // four W pieces of step q+2 are scheduled INTO the MFMA stream of half 1 (stamps: 390 of 2680 cycles per
    // step when issued as a block after the barrier); fragment reads are front-loaded in each MFMA stream so
    // the lgkmcnt(0) before the barrier does not expose the last read's latency.
    auto step = [&](int q, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        STAMP(t0);
        read_frags(a1, b1, q, c, toff, kh, 1);
        mfma_all(a0, b0);
#pragma unroll
        for (int g_ = 0; g_ < NA + PT; ++g_) {
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NA * PT - (NA + PT), 0);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(t1);
        int ntap = tap + 1, nkw = kw + 1, ntoff = toff + 1, nkh = kh, nc = c;
        if (nkw == 3) { nkw = 0; ntoff += HW - 3; ++nkh; }
        if (ntap == 9) { ntap = 0; nkw = 0; ntoff = 0; nkh = 0; ++nc; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's reads of W(q)'s buffer are in registers
        if (x_inflight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NXI) : "memory");   // W(q+1) landed, halo may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(t2);
        __builtin_amdgcn_s_barrier();
        STAMP(t3);
        STAMP(t4);
        __builtin_amdgcn_sched_barrier(0);
        if (ISSUE) issue_w(q + 2);
        read_frags(a0, b0, q + 1, nc, ntoff, nkh, 0);      // harmless on the last step (stays inside the LDS buffers)
        mfma_all(a1, b1);
#pragma unroll
        for (int g_ = 0; g_ < NA + PT; ++g_) {
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if (ISSUE) {
#pragma unroll
            for (int g_ = 0; g_ < NWI; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x004, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NA * PT - (NA + PT) - 3 * NWI, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, NA * PT - (NA + PT), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // next chunk's halo: issued AFTER W(q+2) so that "all but the youngest NXI" at the next step's wait
        // still means W(q+1)..W(q+2) landed; once per 9 steps
        x_inflight = false;
        if (tap == 0 && c + 1 < nchunk) { issue_x(c + 1); x_inflight = true; }
        STAMP(t5);
        if (STAMPS) { s01 += t1 - t0; s12 += t2 - t1; s23 += t3 - t2; s34 += t4 - t3; s45 += t5 - t4; }
        tap = ntap; kw = nkw; toff = ntoff; kh = nkh; c = nc;
    };
    int q = 0;
    for (; q + 2 < nq; ++q) step(q, std::true_type{});
    for (; q < nq; ++q) step(q, std::false_type{});
    }
    STAMP(tC);

    // ---- epilogue through LDS: the tile's output is 196 consecutive pixels x BN couts, so it is staged
    // as [px][BN] f16 (row pitch + 16 B against bank conflicts) and moved with 16-byte fully coalesced
    // accesses: (A) residual tile global -> LDS, (B) each lane: acc + bias -> PReLU -> + residual -> f16,
    // in place, (C) LDS -> global.  One rounding to f16, as in conv_mfma.hip.
    constexpr int OP = BN + 8;                         // row pitch in halves
    constexpr int CPR = BN / 8;                        // 16-B chunks per row
    half_t* ot = lds;
    const int npx = min(p.G, p.B - n) * IPX;           // a tile's images are consecutive: one contiguous span of pixels
    const int m_base = n * p.H * p.W + y0 * p.W;
    __syncthreads();                                   // every wave is done with the operand buffers
    if (p.res) {
        for (int e = tid; e < npx * CPR; e += NT) {
            const int px = e / CPR, cc = e - px * CPR;
            const int4v v = *reinterpret_cast<const int4v*>(p.res + (size_t)(m_base + px) * p.Cout + cout0 + cc * 8);
            *reinterpret_cast<int4v*>(ot + px * OP + cc * 8) = v;
        }
        __syncthreads();
    }
    auto bias_sel = [&](int px) {
        int bsel = 0;
        if (p.bias_mode == 1) {
            const int r = px % IPX;
            const int oy = r / p.W, ox = r - oy * p.W, ho = y0 + oy;
            const int rc = ho == 0 ? 0 : (ho == p.H - 1 ? 2 : 1);
            const int cc = ox == 0 ? 0 : (ox == p.W - 1 ? 2 : 1);
            bsel = (rc * 3 + cc) * p.Cout;
        }
        return bsel;
    };
    auto finish = [&](float4v v, int px, int col, int bsel) {       // bias -> PReLU -> + residual -> f16, in place in LDS
        const int co = cout0 + col;
        if constexpr (F8) v *= *reinterpret_cast<const float4v*>(p.oscale + co);     // dequantise: sw[cout] * sx
        if (p.bias) v += *reinterpret_cast<const float4v*>(p.bias + bsel + co);
        if (p.slope) {
            const float4v sv = *reinterpret_cast<const float4v*>(p.slope + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sv[e];
        }
        half4* slot = reinterpret_cast<half4*>(ot + px * OP + col);
        if (p.res) {
            const half4 rv = *slot;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
        }
        *slot = half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    };
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const int px = (wp * PT + j) * 16 + fr;
        if (px >= npx) continue;
        const int bsel = bias_sel(px);
#pragma unroll
        for (int i = 0; i < NA; ++i) finish(acc[i][j], px, wn * NA * 16 + i * 16 + fq * 4, bsel);
    }
    if constexpr (SPLIT) {
        const int px = 12 * 16 + fr;
        if (px < npx) {
            const int bsel = bias_sel(px);
#pragma unroll
            for (int t = 0; t < XT; ++t) finish(accx[t], px, wn * NA * 16 + (wp * XT + t) * 16 + fq * 4, bsel);
        }
    }
    __syncthreads();
    if (!F8 || p.y) {
        for (int e = tid; e < npx * CPR; e += NT) {
            const int px = e / CPR, cc = e - px * CPR;
            const int4v v = *reinterpret_cast<const int4v*>(ot + px * OP + cc * 8);
            *reinterpret_cast<int4v*>(p.y + (size_t)(m_base + px) * p.Cout + cout0 + cc * 8) = v;
        }
    }
    if constexpr (F8) {
        if (p.y8) {                                    // fp8 copy for the next conv: 8 channels = 8 bytes per thread
            for (int e = tid; e < npx * CPR; e += NT) {
                const int px = e / CPR, cc = e - px * CPR;
                const half8 h = *reinterpret_cast<const half8*>(ot + px * OP + cc * 8);
                float4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
                if (p.y8_sub) {
                    s0 = *reinterpret_cast<const float4v*>(p.y8_sub + cout0 + cc * 8);
                    s1 = *reinterpret_cast<const float4v*>(p.y8_sub + cout0 + cc * 8 + 4);
                }
                int2 o;
                o.x = pack_fp8x4(((float)h[0] - s0[0]) * p.y8_mul, ((float)h[1] - s0[1]) * p.y8_mul, ((float)h[2] - s0[2]) * p.y8_mul, ((float)h[3] - s0[3]) * p.y8_mul);
                o.y = pack_fp8x4(((float)h[4] - s1[0]) * p.y8_mul, ((float)h[5] - s1[1]) * p.y8_mul, ((float)h[6] - s1[2]) * p.y8_mul, ((float)h[7] - s1[3]) * p.y8_mul);
                *reinterpret_cast<int2*>(p.y8 + (size_t)(m_base + px) * p.Cout + cout0 + cc * 8) = o;
            }
        }
    }
    if (STAMPS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(tD);
    if (STAMPS && p.stamps && lane == 0) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = s01; o[1] = s12; o[2] = s23; o[3] = s34; o[4] = s45;
        o[5] = tB - tA; o[6] = tC - tB; o[7] = tD - tC;
    }
#endif
}

template <int WN, int NPT, int XROWS, int NXBUF, int MINW, bool LEAN = false, int NA = 4, bool F8 = false, int NW = 8, int ABL = 0>
static int launch_halo(const HaloP& p, hipStream_t s) {
    constexpr int BN = 16 * NA * WN;
    constexpr size_t opnd = (size_t)(NXBUF * XROWS * HK + 2 * BN * HK) * sizeof(half_t);
    constexpr size_t epi = (size_t)NPT * 16 * (BN + 8) * sizeof(half_t);          // output staging tile
    constexpr size_t lds = opnd > epi ? opnd : epi;
    const int blocks = ((p.B + p.G - 1) / p.G) * p.tiles_per_img * (p.Cout / BN);
    if constexpr (FR_DEBUG) {                  // stamped twin: debug build only
        if (p.stamps) {
            static FrDevLatch dl;
            auto dk = conv_halo_kernel<WN, NPT, XROWS, NXBUF, MINW, true, LEAN, NA, F8, NW, ABL>;
            if (!fr_raise_lds(reinterpret_cast<const void*>(dk), lds, dl)) { fr_set_error("conv_halo: cannot raise dynamic LDS"); return FR_E_LAUNCH; }
            dk<<<blocks, NW * 64, lds, s>>>(p);
            return FR_OK;
        }
    }
    static FrDevLatch latch;
    auto kern = conv_halo_kernel<WN, NPT, XROWS, NXBUF, MINW, false, LEAN, NA, F8, NW, ABL>;
    if (!fr_raise_lds(reinterpret_cast<const void*>(kern), lds, latch)) {
        fr_set_error("conv_halo: cannot raise dynamic LDS to %zu bytes", lds);
        return FR_E_LAUNCH;
    }
    kern<<<blocks, NW * 64, lds, s>>>(p);
    return FR_OK;
}

// Returns 1 if the halo kernel handled the layer, 0 if the shape is not eligible, < 0 on error.
int fr_conv_halo_try(const fr_conv_args* a, hipStream_t s) {
    if (!(a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->H == a->W && a->out_f32_partial == nullptr) || a->x2)
        return 0;
    if (a->Cin % 64 != 0 || a->Cout % 64 != 0) return 0;
    int TH, G = 1;
    if (a->H == 7 && a->Cout % 128 == 0) { TH = 7; G = 4; }          // four whole 7x7 images = 196 pixels per tile
    else if (a->H == 14 && a->Cout % 128 == 0) TH = 14;
    else if (a->H == 28 && a->Cout % 128 == 0) TH = 7;
    else if (a->H == 56 && a->Cin == 64) TH = 4;                         // single-chunk variants (one halo buffer)
    else if (a->H == 112 && a->Cin == 64 && a->Cout == 64) TH = 2;
    else return 0;
    if ((int64_t)a->B * a->H * a->W * a->Cin * 2 >= (1ll << 31) || (int64_t)a->Cout * 9 * a->Cin * 2 >= (1ll << 31)) return 0;
    HaloP p;
    p.x = (const half_t*)a->x; p.w = (const half_t*)a->w; p.y = (half_t*)a->y;
    p.bias = a->bias; p.slope = a->slope; p.res = (const half_t*)a->residual;
    p.oscale = nullptr; p.y8 = nullptr; p.y8_mul = 0.f; p.y8_sub = nullptr;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.bias_mode = a->bias_mode;
    p.TH = TH; p.tiles_per_img = a->H / TH; p.G = G;
    p.xbytes = (unsigned)((int64_t)a->B * a->H * a->W * a->Cin * 2);
    p.wbytes = (unsigned)((int64_t)a->Cout * 9 * a->Cin * 2);
    p.stamps = (unsigned long long*)fr_dbg_ptr("FR_DBG_STAMPS");        // always NULL in the product build
    p.stagger = fr_dbg_int("FR_HALO_STAGGER", 0);
    int rc;
    if (a->H == 7) rc = launch_halo<2, 13, 384, 1, 4, true>(p, s);        // lean schedule; halo rows: 4 x 9 x 9 = 324
    else if (FR_DEBUG && (a->H == 56 || a->H == 112) && fr_dbg_int("FR_HALO_LEAN64", 0)) {
        if constexpr (FR_DEBUG) {
            if (a->H == 112) rc = launch_halo<1, 14, 512, 1, 4, true>(p, s);
            else switch (fr_dbg_int("FR_HALO_LEAN64", 0)) {     // 56x56: the lean schedule and its compile-time ablations
                case 2: rc = launch_halo<1, 14, 384, 1, 4, true, 4, false, 8, 4>(p, s); break;    // no W DMA after the first
                case 3: rc = launch_halo<1, 14, 384, 1, 4, true, 4, false, 8, 1>(p, s); break;    // no MFMA
                case 4: rc = launch_halo<1, 14, 384, 1, 4, true, 4, false, 8, 2>(p, s); break;    // no fragment reads
                case 5: rc = launch_halo<1, 14, 384, 1, 4, true, 4, false, 8, 7>(p, s); break;    // skeleton
                case 6: rc = launch_halo<1, 14, 384, 1, 4, true, 4, false, 8, 8>(p, s); break;    // no barrier
                default: rc = launch_halo<1, 14, 384, 1, 4, true>(p, s);
            }
        }
        else rc = FR_E_INVALID;
    }
    else if (a->H == 56) rc = launch_halo<1, 14, 384, 1, 4>(p, s);
    else if (a->H == 112) rc = launch_halo<1, 14, 512, 1, 4>(p, s);
    else {
        // 28x28 and 14x14 layers run as two lean blocks per CU.  Debug build: FR_HALO_LEAN bit 0 / bit 1 = 0 selects
        // the pipelined one-block-per-CU schedule for 28x28 / 14x14 instead (kept as a measured alternative).
        const int lean = fr_dbg_int("FR_HALO_LEAN", 3);
        const int nw4 = fr_dbg_int("FR_HALO_NW4", HALO_NW4_DEFAULT);      // bit 0: 28x28, bit 1: 14x14 on the 4-wave lean variant
        if ((lean & 1) && a->H == 28) rc = (nw4 & 1) ? launch_halo<2, 13, 320, 1, 2, true, 4, false, 4>(p, s) : launch_halo<2, 13, 320, 1, 4, true>(p, s);
        else if (FR_DEBUG && a->H == 14 && (p.stagger >> 16)) {
            if constexpr (FR_DEBUG) {
                switch (p.stagger >> 16) {
                    case 1: rc = launch_halo<2, 13, 256, 1, 4, true, 4, false, 8, 1>(p, s); break;
                    case 2: rc = launch_halo<2, 13, 256, 1, 4, true, 4, false, 8, 2>(p, s); break;
                    case 3: rc = launch_halo<2, 13, 256, 1, 4, true, 4, false, 8, 3>(p, s); break;
                    case 4: rc = launch_halo<2, 13, 256, 1, 4, true, 4, false, 8, 4>(p, s); break;
                    case 6: rc = launch_halo<2, 13, 256, 1, 4, true, 4, false, 8, 6>(p, s); break;
                    case 7: rc = launch_halo<2, 13, 256, 1, 4, true, 4, false, 8, 7>(p, s); break;
                    case 8: rc = launch_halo<2, 13, 256, 1, 4, true, 4, false, 8, 8>(p, s); break;
                    default: rc = launch_halo<2, 13, 256, 1, 4, true, 4, false, 8, 15>(p, s); break;
                }
            } else rc = FR_E_INVALID;
        }
        else if (FR_DEBUG && a->H == 14 && a->Cout % 256 == 0 && fr_dbg_int("FR_HALO_W16", 0)) {
            if constexpr (FR_DEBUG) rc = launch_halo<4, 13, 256, 2, 4, true, 4, false, 16>(p, s);   // one 16-wave block per CU, two halo buffers
            else rc = FR_E_INVALID;
        }
        else if ((lean & 2) && a->H == 14) rc = (nw4 & 2) ? launch_halo<2, 13, 256, 1, 2, true, 4, false, 4>(p, s) : launch_halo<2, 13, 256, 1, 4, true>(p, s);
        else if constexpr (FR_DEBUG) rc = (a->Cout % 256 == 0) ? launch_halo<4, 13, 320, 2, 2>(p, s) : launch_halo<2, 13, 320, 2, 2>(p, s);
        else rc = FR_E_INVALID;
    }
    return rc == FR_OK ? 1 : rc;
}

// ---------------------------------------------------------------- fp8 body convs (BASELINE config C5)
extern "C" int fr_conv_nhwc_f8(const fr_conv_f8_args* a, fr_stream_t stream) {
    FR_REQUIRE(a, "fr_conv_nhwc_f8: null args");
    FR_REQUIRE(a->x8 && a->w8 && a->oscale && (a->y16 || a->y8), "fr_conv_nhwc_f8: null tensor");
    FR_REQUIRE(a->B > 0 && a->H == a->W && (a->H == 14 || a->H == 28), "fr_conv_nhwc_f8: 3x3/s1/p1 layers at 14x14 or 28x28 only (got %dx%d)", a->H, a->W);
    FR_REQUIRE(a->Cin % 128 == 0 && a->Cout % 128 == 0, "fr_conv_nhwc_f8: Cin and Cout must be multiples of 128 (got %d, %d)", a->Cin, a->Cout);
    FR_REQUIRE(a->bias_mode == 0 || a->bias_mode == 1, "fr_conv_nhwc_f8: bad bias_mode");
    FR_REQUIRE((int64_t)a->B * a->H * a->W * a->Cin < (1ll << 31) && (int64_t)a->Cout * 9 * a->Cin < (1ll << 31), "fr_conv_nhwc_f8: tensor too large");
    HaloP p;
    p.x = (const half_t*)a->x8; p.w = (const half_t*)a->w8; p.y = (half_t*)a->y16;
    p.bias = a->bias; p.slope = a->slope; p.res = (const half_t*)a->residual;
    p.oscale = a->oscale; p.y8 = (unsigned char*)a->y8; p.y8_mul = a->y8_mul; p.y8_sub = a->y8_sub;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.bias_mode = a->bias_mode;
    p.TH = a->H == 14 ? 14 : 7; p.tiles_per_img = a->H / p.TH; p.G = 1;
    p.xbytes = (unsigned)((int64_t)a->B * a->H * a->W * a->Cin);
    p.wbytes = (unsigned)((int64_t)a->Cout * 9 * a->Cin);
    p.stamps = (unsigned long long*)fr_dbg_ptr("FR_DBG_STAMPS");
    p.stagger = 0;
    hipStream_t s = fr_stream(stream);
    const int rc = a->H == 28 ? launch_halo<2, 13, 320, 1, 4, true, 4, true>(p, s) : launch_halo<2, 13, 256, 1, 4, true, 4, true>(p, s);
    if (rc != FR_OK) return rc;
    FR_CHECK_LAUNCH("conv_halo_f8");
    return FR_OK;
}

// f16 -> fp8 e4m3 ((x - sub[channel]) * mul, saturating): input of the first fp8 conv of a chain
__global__ void quantize_f16_f8(const half_t* __restrict__ x, unsigned char* __restrict__ out, int64_t n8, float mul,
                                const float* __restrict__ sub, int C8) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const half8 h = *reinterpret_cast<const half8*>(x + i * 8);
        float4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        if (sub) {
            const int c = (int)(i % C8) * 8;
            s0 = *reinterpret_cast<const float4v*>(sub + c);
            s1 = *reinterpret_cast<const float4v*>(sub + c + 4);
        }
        int2 o;
        o.x = pack_fp8x4(((float)h[0] - s0[0]) * mul, ((float)h[1] - s0[1]) * mul, ((float)h[2] - s0[2]) * mul, ((float)h[3] - s0[3]) * mul);
        o.y = pack_fp8x4(((float)h[4] - s1[0]) * mul, ((float)h[5] - s1[1]) * mul, ((float)h[6] - s1[2]) * mul, ((float)h[7] - s1[3]) * mul);
        *reinterpret_cast<int2*>(out + i * 8) = o;
    }
}

static int quantize_launch(const void* x16, void* out8, int64_t n, float mul, const float* sub, int C, fr_stream_t stream) {
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    quantize_f16_f8<<<(int)blocks, 256, 0, fr_stream(stream)>>>((const half_t*)x16, (unsigned char*)out8, n / 8, mul, sub, C / 8);
    FR_CHECK_LAUNCH("quantize_f16_f8");
    return FR_OK;
}

extern "C" int fr_quantize_f16_f8(const void* x16, void* out8, int64_t n, float mul, fr_stream_t stream) {
    if (n <= 0) return FR_OK;
    FR_REQUIRE(x16 && out8 && n % 8 == 0, "fr_quantize_f16_f8: null pointer or n not a multiple of 8");
    return quantize_launch(x16, out8, n, mul, nullptr, 8, stream);
}

extern "C" int fr_quantize_f16_f8_centred(const void* x16, void* out8, int64_t n, int C, const float* sub, float mul,
                                          fr_stream_t stream) {
    if (n <= 0) return FR_OK;
    FR_REQUIRE(x16 && out8 && sub && C > 0 && C % 8 == 0 && n % C == 0, "fr_quantize_f16_f8_centred: bad argument (n %lld, C %d)", (long long)n, C);
    return quantize_launch(x16, out8, n, mul, sub, C, stream);
}

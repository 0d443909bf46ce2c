// THE 14x14 STAGE AS ONE LAUNCH ON THE FP8 MATRIX CORES (BASELINE config C5: "fp8 ArcFace conv path, CDNA4 fp8 MFMA";
// the embed half of `FaceAnalysis.get`, /root/reference/infrenceServer.py:528).  The fp8 twin of conv_stage14.hip: same
// ownership (one workgroup = one face for the whole run of residual blocks), same wave / tile split (8 waves, 26
// accumulator tiles each), same ring protocol - with these differences:
//
//   image   the conv's input as CENTRED e4m3 CODES, fp8((x - mu[c]) / sx) (IResNetHIP.enable_fp8): 2 planes of 128
//           channels x 200 rows x 128 B = 51 KB; row = pixel, 16-B chunk' = chunk ^ (pixel & 7); rows 196..199 zero.
//           Every conv's epilogue writes the NEXT conv's codes (its mu and 1 / sx ride in this conv's parameters).
//   step    one tap x 128 channels = ONE v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) per tile pair: 18 steps
//           per conv instead of 72; an operand is two ds_read_b128 (bytes [16 fq, +16) and [64 + 16 fq, +16) of the row)
//   ring    3 slots x [256 couts][128 channels] e4m3 = 32 KB, 4 LDS-DMA pieces per wave per step; 128-B rows,
//           chunk' = chunk ^ (row & 7)
//   stream  the residual stream stays f16 and does NOT fit beside image + ring (100 KB): a block's second conv writes
//           it to HBM (y16, 8 B per lane and tile) and the next block's second conv reads it back (L1-bypassing loads)
//           into its accumulators before its K loop, scaled by 1 / oscale[cout] so that the dequantising multiply of
//           the epilogue restores it
//   epilogue v = acc * oscale[cout] + bias9[class][cout] -> PReLU (first conv) -> f16 (second conv: stored) ->
//           next conv's code fp8((f16 - mu_next[cout]) * inv_sx_next) -> the image in place: the arithmetic of
//           fr_conv_nhwc_f8's epilogue, in its order
//
// Parameters per conv: f32 [14][256] = oscale, 1 / oscale, 9 border-class biases, PReLU slope (1 = none), mu of the
// next conv's input, and a row whose first element is 1 / sx of the next conv.  LDS = 51 200 + 98 304 + 14 336 =
// 163 840 B: all of it.
#include "common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int int2v __attribute__((ext_vector_type(2)));
typedef int int8v __attribute__((ext_vector_type(8)));

namespace {

constexpr int F14_PX = 196, F14_C = 256, F14_ROWS = 200;
constexpr int F14_PLANE = F14_ROWS * 128;                           // 25 600 B: 128 channels x 1 B per row
constexpr int F14_IMG = 2 * F14_PLANE;                              // 51 200
constexpr int F14_SLOT = 256 * 128;                                 // 32 768
constexpr int F14_PRM_ROWS = 14;
constexpr int F14_PRM = F14_PRM_ROWS * 256 * 4;                     // 14 336
constexpr int F14_LDS = F14_IMG + 3 * F14_SLOT + F14_PRM;           // 163 840
constexpr int F14_STEPS = 18;                                       // per conv
enum { PR_OSCALE = 0, PR_INVOSCALE = 1, PR_BIAS = 2, PR_SLOPE = 11, PR_MU = 12, PR_INVSX = 13 };

struct StageF8P {
    const unsigned char* x8;    // [B][196][256] centred e4m3 codes of the first conv's input
    const half_t* x16;          // [B][196][256] the same tensor in f16: the first block's residual
    half_t* y16;                // [B][196][256] residual stream / output (written by every block's second conv)
    const unsigned char* w;     // [nconv][18][32768] pre-swizzled e4m3 weight stream
    const float* prm;           // [nconv][14][256]
    int B, nconv;
    unsigned xbytes8, xbytes16, wbytes;
    unsigned long long* stamps; // diagnostic build only
};

__device__ __forceinline__ int pack_fp8x4_sat(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return v;
}

// an MFMA operand: 8 consecutive VGPRs, built from its two 16-B halves at load time (one 256-bit value from the start:
// joining the halves at the MFMA instead made hipcc copy them into fresh 8-register tuples and spill 500 VGPRs)
typedef int8v Frag;
__device__ __forceinline__ Frag frag_of(const int4v& lo, const int4v& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

__device__ __forceinline__ float4v mfma8(const Frag& a, const Frag& b, float4v c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

}  // namespace

#define F14_STAMP(var)                                                                     \
    do {                                                                                    \
        if (STAMPS) {                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                              \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");      \
            __builtin_amdgcn_sched_barrier(0);                                              \
        }                                                                                   \
    } while (0)
#define F14_PIN() __builtin_amdgcn_sched_barrier(0)

template <int STAMPS>
__global__ __launch_bounds__(512, 2) void conv_stage14_f8_kernel(StageF8P p) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long tA = 0, tB = 0, tC = 0, t0p = 0, t1 = 0, se = 0, sp = 0, sl_ = 0, rA = 0, rB = 0;
    F14_STAMP(tA);
    if (STAMPS) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rA)::"memory");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* img = lds;
    char* ring = lds + F14_IMG;
    const float* lprm = reinterpret_cast<const float*>(lds + F14_IMG + 3 * F14_SLOT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wp = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;
    const int n = blockIdx.x;

    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x8, 0, p.xbytes8, 0x00020000);
    __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)p.prm, 0, (unsigned)p.nconv * F14_PRM, 0x00020000);

    // ---- weight stream: W(s) = 32 KB at s * 32 KB; this wave moves pieces 4 * wave .. 4 * wave + 3 (1 KB each)
    const unsigned wlane = (unsigned)(wave * 4096 + lane * 16);
    unsigned wsrc = 0;
    auto issue_w = [&](int slot) {
        char* dst = ring + slot * F14_SLOT + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(dst + i * 1024), 16, wlane + i * 1024, wsrc, 0, 0);
        wsrc += F14_SLOT;
    };
    issue_w(0);
    issue_w(1);
    issue_w(2);
    // a conv's parameters (14 KB) -> LDS: pieces 0..7 by the 8 waves, 8..13 by waves 0..5
    auto issue_prm = [&](int conv, int ln) {
        char* dst = lds + F14_IMG + 3 * F14_SLOT;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (lds_ptr_t)(dst + wave * 1024), 16, (unsigned)(wave * 1024 + ln * 16), conv * F14_PRM, 0, 0);
        if (wave < 6)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (lds_ptr_t)(dst + (8 + wave) * 1024), 16, (unsigned)((8 + wave) * 1024 + ln * 16), conv * F14_PRM, 0, 0);
    };

    // ---- image: HBM codes [196][256 B] -> 2 planes x 200 rows x 128 B (rows >= 196: out of range -> zeros)
    {
        const int lrow = lane >> 3, ch = lane & 7;
        for (int pc = wave; pc < 50; pc += 8) {                      // 50 pieces of 8 rows: 25 per plane
            const int plane = pc / 25, row = (pc - plane * 25) * 8 + lrow;
            const unsigned off = row < F14_PX ? (unsigned)(((size_t)n * F14_PX + row) * 256 + plane * 128 + ((ch ^ (row & 7)) << 4))
                                              : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(img + plane * F14_PLANE + (pc - plane * 25) * 1024), 16, off, 0, 0, 0);
        }
    }

    // Everything from here on exists in TWO copies, one per wave role (WP = wave >> 2: pixel group, cout tiles of the
    // shared pixel tile, place of the weight DMA in a step): with the role as a run-time (wave-uniform) branch inside the
    // unrolled steps hipcc spilled 480 VGPRs - 8-register MFMA operands live across a dozen basic blocks per step.
    auto run = [&](auto wp_tag) {
    constexpr int WP = decltype(wp_tag)::value;
    // A fragment: row (cout) fr of a 16-cout tile: lo = 16-B chunk fq ^ (row & 7), hi = that ^ 4 (byte offset ^ 64)
    const int a_own = wn * 8192 + fr * 128 + ((fq ^ (fr & 7)) << 4);  // + i * 2048: cout tile i of the wave's 64 couts
    const int px0 = WP * 96 + fr;

    float4v acc[6][4], accx[2];
    // pixel fragments live in a ring of THREE tiles (tile t of every step in bt[t % 3]): a tile is re-read - three tiles
    // ahead: this step's t + 3, or the next step's t - 3 - right behind the four MFMAs that used it.  All six at once
    // (48 VGPRs) beside 104 accumulator and 64 weight-fragment registers do not fit the register file.
    Frag a0[4], a1[4], bt[3], bx;
    int boff[7];
    int px0e = px0, fre = fr, fqe = fq;
    auto set_tap_one = [&](int j, int dy, int dx) {
        const int px = j < 6 ? px0e + 16 * j : 192 + fre;
        const int oy = (px * 4682) >> 16, ox = px - oy * 14;
        const bool ok = px < F14_PX && (unsigned)(oy + dy) < 14u && (unsigned)(ox + dx) < 14u;
        const int pxn = ok ? px + dy * 14 + dx : F14_PX;
        boff[j] = pxn * 128 + ((fqe ^ (pxn & 7)) << 4);
    };
    auto rd_a = [&](int slot, int i) {
        return frag_of(*reinterpret_cast<const int4v*>(ring + slot * F14_SLOT + a_own + i * 2048),
                       *reinterpret_cast<const int4v*>(ring + slot * F14_SLOT + ((a_own + i * 2048) ^ 64)));
    };
    auto rd_b = [&](int g, int j) {
        return frag_of(*reinterpret_cast<const int4v*>(img + g * F14_PLANE + boff[j]),
                       *reinterpret_cast<const int4v*>(img + g * F14_PLANE + (boff[j] ^ 64)));
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // image, W(0..2)
    __builtin_amdgcn_s_barrier();

    // One K step (local index k of a 6-step group: slot k % 3, plane g = k & 1); see conv_stage14.hip for the protocol.
    auto step = [&](Frag (&ac)[4], Frag (&an)[4], int k, int dyn, int dxn) {
        const int g = k & 1, ng = (k + 1) & 1, nslot = (k + 1) % 3;
        if constexpr (WP == 0) { issue_w(k % 3); F14_PIN(); }
        accx[0] = mfma8(ac[2 * WP], bx, accx[0]); accx[1] = mfma8(ac[2 * WP + 1], bx, accx[1]);
        F14_PIN();
#pragma unroll
        for (int t = 0; t < 6; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i] = mfma8(ac[i], bt[t % 3], acc[t][i]);
            // reads are placed LATE (live ranges are what the register file is short of): the next step's weight fragments
            // behind tiles 3 and 4 - first the two the shared tile needs at the next step's start - the shared tile's pixel
            // fragment behind tile 5; weight reads precede the pixel read of their tile, so the last six LDS reads of a step
            // are pixel reads (lgkmcnt(6) at the next top)
            if (t == 3) { an[2 * WP] = rd_a(nslot, 2 * WP); an[2 * WP + 1] = rd_a(nslot, 2 * WP + 1); }
            if (t == 4) { an[2 - 2 * WP] = rd_a(nslot, 2 - 2 * WP); an[3 - 2 * WP] = rd_a(nslot, 3 - 2 * WP); }
            if (t < 3) {                                             // this step's tile t + 3 (same tap, same plane)
                bt[t % 3] = rd_b(g, t + 3);
            } else {                                                 // the next step's tile t - 3: the tap moves behind plane 1
                if (g == 1) set_tap_one(t - 3, dyn, dxn);
                bt[t % 3] = rd_b(ng, t - 3);
            }
            if (t == 5) { if (g == 1) set_tap_one(6, dyn, dxn); bx = rd_b(ng, 6); }
            F14_PIN();
            if constexpr (WP == 1) { if (t == 3) { issue_w(k % 3); F14_PIN(); } }
        }
        if (g == 1) { set_tap_one(3, dyn, dxn); set_tap_one(4, dyn, dxn); set_tap_one(5, dyn, dxn); }      // read inside the next step
    };

#pragma unroll 1
    for (int conv = 0; conv < p.nconv; ++conv) {
        F14_STAMP(tC);
        if (STAMPS && conv) se += tC - tB;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        fre = lane_e & 15; fqe = lane_e >> 4; px0e = WP * 96 + fre;
        issue_prm(conv, lane_e);
        const int co_own = wn * 64 + fqe * 4, co_sh = wn * 64 + WP * 32 + fqe * 4;      // + 16 per cout tile
        if (conv & 1) {
            // second conv of a block: accumulators start as residual / oscale (the epilogue's dequantising multiply restores
            // the residual).  The block's input: x16 for the first block, else what this workgroup wrote to y16 one block
            // ago - L1-bypassing loads (sc1): a CU's vector L1 is never refreshed by stores.
            __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(conv == 1 ? p.x16 : (const half_t*)p.y16), 0, p.xbytes16, 0x00020000);
            const unsigned rb = (unsigned)n * (unsigned)(F14_PX * F14_C * 2);
            // 16 B per lane = 8 consecutive couts: lanes fq = 0, 2 / 1, 3 fetch couts 0..7 / 8..15 of cout tiles (2 ip, 2 ip + 1) and
            // v_permlane16_swap puts them back into the accumulator layout (a vector-memory instruction costs the same whatever
            // its width, and L1-bypassing 8-byte loads at a 512-byte stride are one L2 request per lane)
            int4v r[6][2], rx;
            const int co8 = wn * 64 + (fqe & 1) * 16 + (fqe >> 1) * 8;      // + 32 per tile pair
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
                    r[j][ip] = __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(rrs, (unsigned)((px0e + 16 * j) * F14_C + co8 + ip * 32) * 2, rb, 16));
            rx = __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(rrs, 192 + fre < F14_PX ? (unsigned)((192 + fre) * F14_C + co8 + WP * 32) * 2 : 0x80000000u, rb, 16));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the parameters (1 / oscale) and the residual
            __builtin_amdgcn_s_barrier();
            auto unswap = [&](const int4v& v, float4v& lo, float4v& hi, int co_lo) {
                const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)v[0], (unsigned)v[2], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)v[1], (unsigned)v[3], false, false);
                const half4 ha = __builtin_bit_cast(half4, int2v{(int)s0[0], (int)s1[0]});
                const half4 hb = __builtin_bit_cast(half4, int2v{(int)s0[1], (int)s1[1]});
                const float4v ia = *reinterpret_cast<const float4v*>(lprm + PR_INVOSCALE * F14_C + co_lo);
                const float4v ib = *reinterpret_cast<const float4v*>(lprm + PR_INVOSCALE * F14_C + co_lo + 16);
                lo = float4v{(float)ha[0] * ia[0], (float)ha[1] * ia[1], (float)ha[2] * ia[2], (float)ha[3] * ia[3]};
                hi = float4v{(float)hb[0] * ib[0], (float)hb[1] * ib[1], (float)hb[2] * ib[2], (float)hb[3] * ib[3]};
            };
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) unswap(r[j][ip], acc[j][2 * ip], acc[j][2 * ip + 1], co_own + ip * 32);
            unswap(rx, accx[0], accx[1], co_sh);
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = float4v{0.f, 0.f, 0.f, 0.f};
            accx[0] = accx[1] = float4v{0.f, 0.f, 0.f, 0.f};
        }
        // prologue: fragments of the conv's first step (its slot, 0, landed before the previous conv's last barrier)
#pragma unroll
        for (int j = 0; j < 7; ++j) set_tap_one(j, -1, -1);
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[i] = rd_a(0, i);
#pragma unroll
        for (int t = 0; t < 3; ++t) bt[t] = rd_b(0, t);
        bx = rd_b(0, 6);
        F14_STAMP(t0p);
        if (STAMPS) sp += t0p - tC;
#pragma unroll 1
        for (int it = 0; it < 3; ++it) {                             // kernel row dy = it - 1
#pragma unroll
            for (int k = 0; k < 6; ++k) {                            // 3 taps x 2 planes; 6 % 3 == 0: slots are compile-time
                // own pieces of W(k+1) landed (all but the 4 youngest: W(k+2)); own weight reads of slot k % 3 returned
                // (all but the 6 youngest LDS reads: pixel fragments)
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(6)" ::: "memory");          // the last weight read is followed by bx and tiles 2..5: >= 6 pixel reads
                __builtin_amdgcn_s_barrier();
                F14_PIN();
                const int tt = k >> 1;
                const int dyn = tt < 2 ? it - 1 : it, dxn = tt < 2 ? tt : -1;
                if ((k & 1) == 0) step(a0, a1, k, dyn, dxn);
                else step(a1, a0, k, dyn, dxn);
            }
        }
        F14_STAMP(tB);
        if (STAMPS) sl_ += tB - t0p;
        // ---- epilogue: dequantise, bias, PReLU (first conv), f16; second conv: f16 -> HBM; next conv's codes -> the image
        const bool second = conv & 1;
        asm volatile("" : "+v"(px0e), "+v"(fre), "+v"(fqe));
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             // the parameters: older than the 4 weight pieces in flight
        __builtin_amdgcn_s_barrier();                                // every wave has consumed its last fragments of the old image
        int rowoff[7], key[7], clsoff[7], gpix[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int px = j < 6 ? px0e + 16 * j : 192 + fre;
            const int oy = (px * 4682) >> 16, ox = px - oy * 14;
            const int cls = (oy == 0 ? 0 : (oy >= 13 ? 2 : 1)) * 3 + (ox == 0 ? 0 : (ox == 13 ? 2 : 1));
            clsoff[j] = (PR_BIAS + cls) * F14_C;
            const int pxr = px < F14_PX ? px : F14_PX;
            rowoff[j] = (wn >> 1) * F14_PLANE + pxr * 128 + fqe * 4;
            key[j] = pxr & 7;
            gpix[j] = pxr * F14_C;
        }
        const int co_o = wn * 64 + fqe * 4, co_s = wn * 64 + WP * 32 + fqe * 4;
        const int ch_own = (wn & 1) * 4, ch_sh = (wn & 1) * 4 + WP * 2;                  // 16-B chunk inside the plane row: + 1 per cout tile
        const float inv_sx = lprm[PR_INVSX * F14_C];
        half_t* ybase = p.y16 + (size_t)n * (F14_PX * F14_C);
        // Tile pairs (cout tiles 2 ip, 2 ip + 1) outermost: the per-cout parameters (oscale, centre, slope) are read once per
        // PAIR - 12 + 6 b128 LDS reads per wave instead of 78 - and only 24 registers of them are live at a time (hoisting all
        // six cout tiles' parameters in front of the loop cost the K loop 16 more spilled registers and 80 % more cycles per step);
        // a second conv's f16 result leaves as one 16-byte store per pair (v_permlane16_swap, as the residual comes in).
        const int co8e = wn * 64 + (fqe & 1) * 16 + (fqe >> 1) * 8;
        auto tiles = [&](auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
            for (int ip = 0; ip < 3; ++ip) {                         // ip 2: the shared pixel tile's two cout tiles
                float4v osc[2], mu2[2], slp[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int co = ip < 2 ? co_o + (2 * ip + u) * 16 : co_s + u * 16;
                    osc[u] = *reinterpret_cast<const float4v*>(lprm + PR_OSCALE * F14_C + co);
                    mu2[u] = *reinterpret_cast<const float4v*>(lprm + PR_MU * F14_C + co);
                    if constexpr (FIRST) slp[u] = *reinterpret_cast<const float4v*>(lprm + PR_SLOPE * F14_C + co);
                }
#pragma unroll
                for (int jj = 0; jj < (ip < 2 ? 6 : 1); ++jj) {
                    const int j = ip < 2 ? jj : 6;
                    const bool live = ip < 2 || 192 + fre < F14_PX;
                    int2v hp[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float4v a_ = ip < 2 ? acc[jj][2 * ip + u] : accx[u];
                        const int co = ip < 2 ? co_o + (2 * ip + u) * 16 : co_s + u * 16;
                        const int ch = ip < 2 ? ch_own + 2 * ip + u : ch_sh + u;
                        float4v v = a_ * osc[u];
                        v += *reinterpret_cast<const float4v*>(lprm + clsoff[j] + co);
                        if constexpr (FIRST) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slp[u][e];
                        }
                        const half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        hp[u] = __builtin_bit_cast(int2v, h);
                        const int code = pack_fp8x4_sat(((float)h[0] - mu2[u][0]) * inv_sx, ((float)h[1] - mu2[u][1]) * inv_sx,
                                                        ((float)h[2] - mu2[u][2]) * inv_sx, ((float)h[3] - mu2[u][3]) * inv_sx);
                        if (live) *reinterpret_cast<int*>(img + rowoff[j] + ((ch ^ key[j]) << 4)) = code;
                    }
                    if constexpr (!FIRST) {                          // the residual stream / the result
                        const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)hp[0][0], (unsigned)hp[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)hp[0][1], (unsigned)hp[1][1], false, false);
                        if (live) *reinterpret_cast<int4v*>(ybase + gpix[j] + co8e + (ip < 2 ? ip * 32 : WP * 32)) = int4v{(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
                    }
                    F14_PIN();
                }
            }
        };
        if (second) tiles(std::false_type{}); else tiles(std::true_type{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // the new image is complete
    }
    };
    if (wp == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    F14_STAMP(tC);
    if (STAMPS) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rB)::"memory");
    if (STAMPS && p.stamps && lane == 0) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = 0; o[1] = 0; o[2] = 0; o[3] = se + (tC - tB); o[4] = sp; o[5] = tC - tA; o[6] = rB - rA; o[7] = sl_;
    }
    (void)t1;
#endif
}

// ---------------------------------------------------------------- host side
extern "C" size_t fr_conv_stage14_f8_weight_bytes(int nconv) { return (size_t)(nconv > 0 ? nconv : 0) * F14_STEPS * F14_SLOT; }
extern "C" size_t fr_conv_stage14_f8_param_floats(void) { return (size_t)F14_PRM_ROWS * 256; }

// ONE conv's e4m3 weights [256][9 * 256] bytes (K = tap-major, as fr_conv_nhwc_f8 takes them) -> its 18 slot images
__global__ void stage14_f8_pack_weights(const unsigned char* __restrict__ w, unsigned char* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;             // one thread per 16-B chunk: 18 slots x 256 rows x 8 chunks
    if (e >= F14_STEPS * 256 * 8) return;
    const int q = e / 2048, r = e - q * 2048, row = r >> 3, cp = r & 7;
    const int chunk = cp ^ (row & 7);
    const int tap = q >> 1, g = q & 1;
    const int4v v = *reinterpret_cast<const int4v*>(w + (size_t)row * 2304 + tap * 256 + g * 128 + chunk * 16);
    *reinterpret_cast<int4v*>(out + (size_t)q * F14_SLOT + row * 128 + cp * 16) = v;
}

extern "C" int fr_conv_stage14_f8_pack(const void* w8, void* out, fr_stream_t stream) {
    FR_REQUIRE(w8 && out, "fr_conv_stage14_f8_pack: null pointer");
    stage14_f8_pack_weights<<<fr_cdiv(F14_STEPS * 2048, 256), 256, 0, fr_stream(stream)>>>((const unsigned char*)w8, (unsigned char*)out);
    FR_CHECK_LAUNCH("stage14_f8_pack_weights");
    return FR_OK;
}

extern "C" int fr_conv_stage14_f8(const void* x8, const void* x16, void* y16, const void* wstream, const float* params, int B,
                                  int nblocks, fr_stream_t stream) {
    FR_REQUIRE(x8 && x16 && y16 && wstream && params && B > 0 && nblocks > 0, "fr_conv_stage14_f8: bad argument");
    FR_REQUIRE(x16 != y16, "fr_conv_stage14_f8: x16 and y16 must be different buffers");
    FR_REQUIRE((int64_t)B * F14_PX * F14_C * 2 < (1ll << 31) && (int64_t)nblocks * 2 * F14_STEPS * F14_SLOT < (1ll << 32) - (1 << 20),
               "fr_conv_stage14_f8: tensor too large (B %d, blocks %d)", B, nblocks);
    StageF8P p;
    p.x8 = (const unsigned char*)x8; p.x16 = (const half_t*)x16; p.y16 = (half_t*)y16;
    p.w = (const unsigned char*)wstream; p.prm = params;
    p.B = B; p.nconv = 2 * nblocks;
    p.xbytes8 = (unsigned)((int64_t)B * F14_PX * F14_C);
    p.xbytes16 = 2 * p.xbytes8;
    p.wbytes = (unsigned)((int64_t)p.nconv * F14_STEPS * F14_SLOT);
    p.stamps = (unsigned long long*)fr_dbg_ptr("FR_DBG_STAMPS");
    if constexpr (FR_DEBUG) {
        if (p.stamps) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_stage14_f8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, F14_LDS) != hipSuccess) { fr_set_error("fr_conv_stage14_f8: cannot raise dynamic LDS"); return FR_E_LAUNCH; }
            conv_stage14_f8_kernel<1><<<B, 512, F14_LDS, fr_stream(stream)>>>(p);
            FR_CHECK_LAUNCH("conv_stage14_f8_kernel<stamps>");
            return FR_OK;
        }
    }
    static FrDevLatch latch;
    if (!fr_raise_lds(reinterpret_cast<const void*>(conv_stage14_f8_kernel<0>), F14_LDS, latch)) {
        fr_set_error("fr_conv_stage14_f8: cannot raise dynamic LDS to %d bytes", F14_LDS);
        return FR_E_LAUNCH;
    }
    conv_stage14_f8_kernel<0><<<B, 512, F14_LDS, fr_stream(stream)>>>(p);
    FR_CHECK_LAUNCH("conv_stage14_f8_kernel");
    return FR_OK;
}

// A RUN OF RESIDUAL BLOCKS AT 14x14x256 AS ONE LAUNCH, THE IMAGE RESIDENT IN LDS (IResNet-100 stage 3: 29 stride-1
// blocks = 58 of the network's 105 convs and 57 % of its FLOPs; the embed half of `FaceAnalysis.get`,
// /root/reference/infrenceServer.py:528).
//
// Why: as one launch per conv (conv_halo.hip) a 14x14 layer takes ~65 us of which ~19 us are a memory skeleton - halo
// loads, residual read, output store - that nothing overlaps, because a launch is exactly one lockstep wave of blocks
// (DESIGN.md 4.1).  Here ONE workgroup owns ONE image for the whole run: a 14x14x256 f16 map is 100 KB, so it stays in
// the CU's LDS from block to block; a conv's output overwrites its input in place (accumulators hold the whole output
// map: 196 pixels x 256 couts = 26 MFMA tiles per wave); only the weights stream, and they stream through the
// launch boundary-free: all convs' weights are ONE pre-swizzled stream in memory, read by linear LDS-DMA.
//
//   LDS  image   4 planes (64 channels each) x 200 rows x 128 B: row = pixel (14-pitch raster, NO halo), 16-B chunk'
//                = chunk ^ (pixel & 7) (conflict-free ds_read_b128 for every tap, as conv_halo.hip); rows 196..199 of a
//                plane are zero: a lane whose tap falls outside the image reads row 196
//        ring    3 slots x [256 couts][32 channels] f16 (16 KB): one K step = one tap x 32 channels = ONE MFMA per
//                tile pair; 64-B rows, chunk' = chunk ^ (((row >> 3) & 1) << 1) (conflict-free)
//   waves 8 = 4 cout groups (wn = plane of the output) x 2 pixel groups (6 tiles of 16 pixels each); the 13th pixel
//         tile (pixels 192..195) is shared by cout: 2 cout tiles per wave (with the wave's own weight fragments) -> 26
//         accumulator tiles per wave, every wave; 256 VGPRs, no spill inside the K loop
//   step  s: wait own W(s+1) pieces (W(s+2) stays in flight) -> barrier -> 26 MFMAs on the fragments read during step
//         s-1, with the 11 fragment reads of step s+1 between them and the DMA of W(s+3) into step s's own slot before
//         them (one wave of a SIMD pair) or in their middle (the other)
//   conv  end: barrier -> bias (9 border classes) -> PReLU -> f16 into the image in place -> barrier.  A block's second
//         conv starts from accumulators that hold the block's input (the residual): the first conv's epilogue reads
//         each old image element just before it overwrites it.  The residual stream never leaves the CU: HBM sees the
//         run's input and, after the last conv, its result (coalesced 16-B stores).
//
// Weight stream (host: iresnet.py _pack_stage14 -> fr_conv_stage14_pack): per conv 72 slots, slot = step q = tap * 8 + g (g = 32-channel
// group), each 16 KB in LDS image order.  Parameters per conv: f32 [10][256] = 9 border-class biases + PReLU slope.
#include "common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int int2v __attribute__((ext_vector_type(2)));

namespace {

constexpr int S14_PX = 196, S14_C = 256, S14_ROWS = 200;            // rows per plane (196 pixels + zero rows)
constexpr int S14_PLANE = S14_ROWS * 128;                           // bytes
constexpr int S14_IMG = 4 * S14_PLANE;                              // 102 400
constexpr int S14_SLOT = 256 * 64;                                  // 16 384
constexpr int S14_PRM = 10 * 256 * 4;                               // one conv's parameters: 10 240
constexpr int S14_LDS = S14_IMG + 3 * S14_SLOT + S14_PRM;           // 161 792
constexpr int S14_STEPS = 72;                                       // per conv

struct StageP {
    const half_t* x;        // [B][196][256] input of the first block
    half_t* y;              // [B][196][256] residual stream / output (written after every block)
    const half_t* w;        // [nconv][72][8192] pre-swizzled weight stream
    const float* prm;       // [nconv][10][256]
    int B, nconv;
    unsigned xbytes, wbytes;
    unsigned long long* stamps;     // diagnostic build only (FR_DBG_STAMPS=<device ptr>): per-wave cycle sums
    int dephase;                    // diagnostic build only (FR_S14_DEPHASE): start delay per XCD-local block index, in units of 64 s_sleep(127)
};

__device__ __forceinline__ float4v mfma16(const int4v& a, const int4v& b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

}  // namespace

#define S14_STAMP_L(var, level)                                                            \
    do {                                                                                    \
        if (STAMPS >= level) {                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                              \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");      \
            __builtin_amdgcn_sched_barrier(0);                                              \
        }                                                                                   \
    } while (0)
#define S14_STAMP(var) S14_STAMP_L(var, 1)

// STAMPS (diagnostic build): 1 = per-conv phases only (3 stamps per conv), 2 = per-step segments too.
// ABL (diagnostic build): compile-time ablation bits of the K loop - 1 no MFMA, 2 no fragment reads, 4 no weight DMA, 8 no barrier
template <int STAMPS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void conv_stage14_kernel(StageP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, tA = 0, tB = 0, tC = 0, sw = 0, sb = 0, sm = 0, se = 0, sp = 0, rA = 0, rB = 0, sl_ = 0, t0p = 0;
    S14_STAMP(tA);
    if (STAMPS) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rA)::"memory");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* img = lds;
    char* ring = lds + S14_IMG;
    const float* lprm = reinterpret_cast<const float*>(lds + S14_IMG + 3 * S14_SLOT);      // the current conv's parameters
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wp = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;
    const int n = blockIdx.x;                                        // the image
    const size_t img_elems = (size_t)S14_PX * S14_C;

    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);

    // ---- weight stream: W(s) = 16 KB at s * 16 KB; this wave moves pieces 2 * wave and 2 * wave + 1 (1 KB each).
    // Unconditional: past the end of the stream the buffer resource returns zeros (into a slot nobody reads).
    const unsigned wlane = (unsigned)(wave * 2048 + lane * 16);
    unsigned wsrc = 0;                                               // byte offset of the next slot to fetch
    auto issue_w = [&](int slot) {
        char* dst = ring + slot * S14_SLOT + wave * 2048;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)dst, 16, wlane, wsrc, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(dst + 1024), 16, wlane + 1024, wsrc, 0, 0);
        wsrc += S14_SLOT;
    };
    issue_w(0);
    issue_w(1);
    issue_w(2);
    // a conv's parameters (10 KB) -> LDS, by LDS-DMA too: pieces 0..7 by the 8 waves, 8..9 by waves 0 and 1
    __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)p.prm, 0, (unsigned)p.nconv * S14_PRM, 0x00020000);
    auto issue_prm = [&](int conv, int ln) {
        char* dst = lds + S14_IMG + 3 * S14_SLOT;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (lds_ptr_t)(dst + wave * 1024), 16, (unsigned)(wave * 1024 + ln * 16), conv * S14_PRM, 0, 0);
        if (wave < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (lds_ptr_t)(dst + (8 + wave) * 1024), 16, (unsigned)((8 + wave) * 1024 + ln * 16), conv * S14_PRM, 0, 0);
    };

    // ---- image: HBM [196][512 B] -> 4 planes x 200 rows x 128 B (rows >= 196 and the tail: out of range -> zeros)
    {
        const int lrow = lane >> 3, ch = lane & 7;
        for (int pc = wave; pc < 100; pc += 8) {                     // 100 pieces of 8 rows: 25 per plane
            const int plane = pc / 25, row = (pc - plane * 25) * 8 + lrow;
            const unsigned off = row < S14_PX ? (unsigned)(((size_t)n * S14_PX + row) * 512 + plane * 128 + ((ch ^ (row & 7)) << 4))
                                              : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(img + plane * S14_PLANE + (pc - plane * 25) * 1024), 16, off, 0, 0, 0);
        }
    }

    // Everything from here on exists in TWO copies, one per wave role (WP = wave >> 2: pixel group, cout tiles of the
    // shared pixel tile, place of the weight DMA in a step): no wave-uniform branches inside the unrolled steps (the
    // fp8 twin spilled 480 VGPRs with them).
    auto run = [&](auto wp_tag) {
    constexpr int WP = decltype(wp_tag)::value;
    // A fragment: row (cout) fr of a 16-cout tile, 16-B chunk fq ^ key(row)
    const int a_lane = fr * 64 + ((fq ^ (((fr >> 3) & 1) << 1)) << 4);
    const int a_own = wn * 4096 + a_lane;                            // + i * 1024: cout tile i of the wave's 64 couts
    const int px0 = WP * 96 + fr;                                    // pixel of tile 0; tile j: + 16 j; the shared 13th tile: 192 + fr

    float4v acc[6][4], accx[2];
    // fragments: the weights (A) are double-buffered across steps; a pixel fragment (B) is re-read for the NEXT step
    // into its own registers as soon as the current step's MFMAs of that pixel tile are issued
    int4v a0[4], a1[4], b[6], bx;
    int boff[7];        // per tap: byte offset of the lane's tap pixel row + chunk bits, for even g; odd g: ^ 64

    // per-conv copies of the lane constants, re-derived from the lane id at every conv start behind an opaque asm:
    // hipcc otherwise hoists the ~100 addresses they determine (the first tap's pixel offsets, residual loads, epilogue
    // stores) out of the conv loop, keeps them in scratch and reloads them one by one, each reload behind a
    // vmcnt(0) that also waits for every weight DMA in flight (measured: 6 000 cycles per conv start)
    int px0e = px0, fre = fr, fqe = fq;
    auto set_tap_one = [&](int j, int dy, int dx) {                  // dy, dx in -1..1
        const int px = j < 6 ? px0e + 16 * j : 192 + fre;
        const int oy = px / 14, ox = px - oy * 14;
        const bool ok = px < S14_PX && (unsigned)(oy + dy) < 14u && (unsigned)(ox + dx) < 14u;
        const int pxn = ok ? px + dy * 14 + dx : S14_PX;             // the zero row
        boff[j] = pxn * 128 + ((fqe ^ (pxn & 7)) << 4);
    };
    auto set_tap = [&](int dy, int dx) {
#pragma unroll
        for (int j = 0; j < 7; ++j) set_tap_one(j, dy, dx);
    };
    auto rd_a = [&](int slot, int i) {
        if constexpr (ABL & 2) { int4v v = {slot, i, 0, 0}; asm volatile("" : "+v"(v)); return v; }
        else return *reinterpret_cast<const int4v*>(ring + slot * S14_SLOT + a_own + i * 1024);
    };
    auto rd_b = [&](int g, int j) {
        if constexpr (ABL & 2) { int4v v = {g, boff[j], 0, 0}; asm volatile("" : "+v"(v)); return v; }
        else return *reinterpret_cast<const int4v*>(img + (g >> 1) * S14_PLANE + ((g & 1) ? (boff[j] ^ 64) : boff[j]));
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // image, W(0), W(1)
    __builtin_amdgcn_s_barrier();
    if (STAMPS && p.dephase) {                                       // experiment: de-phase the workgroups that share an XCD (and its L2)
        const int nd = (blockIdx.x >> 3) * p.dephase;
        for (int i = 0; i < nd; ++i)
            for (int q_ = 0; q_ < 64; ++q_) __builtin_amdgcn_s_sleep(127);
        __builtin_amdgcn_s_barrier();
        S14_STAMP(tA);
    }

#define S14_PIN() __builtin_amdgcn_sched_barrier(0)
    // One K step (local index k of a 24-step group: slot k % 3, channel group g = k & 7) on the fragments (ac, axc, b,
    // bx); meanwhile the NEXT step's fragments are read: weights into (an, axn) from slot (k + 1) % 3, pixel tile j
    // into b[j] right behind the MFMAs that used it.  Before a step with g == 7 reads its successor's pixels, boff
    // moves to the next tap (dyn, dxn).  The weight DMA of step k + 3 goes into this step's own slot (its fragments
    // are in registers: the caller waited lgkmcnt(0) before the barrier) from INSIDE the MFMA stream - an LDS-DMA
    // piece holds the wave's in-order issue for ~100 cycles - early for one wave of a SIMD pair (waves w, w + 4), late
    // for the other, so that the partner's MFMAs cover it.  No branches besides that: the last step of a conv
    // prefetches too - its weight fragments are the next conv's first (that slot has landed), its pixel fragments
    // are dead (the prologue re-reads them).
    auto mm = [&](const int4v& a, const int4v& b_, float4v c) {
        if constexpr (ABL & 1) { asm volatile("" ::"v"(a), "v"(b_)); return c; }
        else return mfma16(a, b_, c);
    };
    auto step = [&](int4v (&ac)[4], int4v (&an)[4], int k, int dyn, int dxn) {
        const int g = k & 7, ng = (k + 1) & 7, nslot = (k + 1) % 3;
        // read order: the four weight fragments first, the seven pixel fragments behind them, b[5]' last - LDS returns in
        // order, so `lgkmcnt(1)` at the next step's top proves every weight read (the slot about to be overwritten) done
        // while b[5]' is still in flight: the image does not change inside a conv, a pixel read may cross the barrier.
        // The weight DMA of step k + 3 (into this step's own slot): an LDS-DMA piece holds the issuing wave for 100+ cycles,
        // so the two waves of a SIMD (w and w + 4) take turns - the first issues its two pieces before its MFMAs, while
        // the partner has the matrix pipe to itself; the partner issues its pieces behind its 18th MFMA (about when the
        // first wave's pieces are out), while the first wave has the pipe (stamps: DMA in the middle of both streams
        // 1 470 cycles per step, first / last 1 180, first / 18th: see DESIGN.md).
        if constexpr (WP == 0 && !(ABL & 4)) { issue_w(k % 3); S14_PIN(); }
        // the shared 13th pixel tile: this wave's cout tiles 2 WP, 2 WP + 1 of its own four - the same weight fragments
        accx[0] = mm(ac[2 * WP], bx, accx[0]); accx[1] = mm(ac[2 * WP + 1], bx, accx[1]);
        S14_PIN();
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mm(ac[i], b[j], acc[j][i]);
            if (j == 0) { an[0] = rd_a(nslot, 0); an[1] = rd_a(nslot, 1); }
            if (j == 1) { an[2] = rd_a(nslot, 2); an[3] = rd_a(nslot, 3); }
            if (j == 2) {
                if (g == 7) { set_tap_one(6, dyn, dxn); set_tap_one(0, dyn, dxn); }
                bx = rd_b(ng, 6); b[0] = rd_b(ng, 0);
            }
            if (j == 3) {
                if (g == 7) { set_tap_one(1, dyn, dxn); set_tap_one(2, dyn, dxn); }
                b[1] = rd_b(ng, 1); b[2] = rd_b(ng, 2);
            }
            if (j == 4) {
                if (g == 7) { set_tap_one(3, dyn, dxn); set_tap_one(4, dyn, dxn); }
                b[3] = rd_b(ng, 3); b[4] = rd_b(ng, 4);
            }
            if (j == 5) {
                if (g == 7) set_tap_one(5, dyn, dxn);
                b[5] = rd_b(ng, 5);
            }
            S14_PIN();
            if constexpr (WP == 1 && !(ABL & 4)) { if (j == 3) { issue_w(k % 3); S14_PIN(); } }
        }
    };

#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = float4v{0.f, 0.f, 0.f, 0.f};
    accx[0] = accx[1] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int conv = 0; conv < p.nconv; ++conv) {
        S14_STAMP(tC);
        if (STAMPS && conv) se += tC - tB;                           // the previous conv's epilogue
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        fre = lane_e & 15; fqe = lane_e >> 4; px0e = WP * 96 + fre;
        issue_prm(conv, lane_e);                                     // read in this conv's epilogue, 72 steps from here
        // prologue: fragments of the conv's first step (its slot, 0, landed before the previous conv's last barrier)
        set_tap(-1, -1);
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[i] = rd_a(0, i);
#pragma unroll
        for (int j = 0; j < 6; ++j) b[j] = rd_b(0, j);
        bx = rd_b(0, 6);
        S14_STAMP(t0p);
        if (STAMPS) sp += t0p - tC;                                  // prologue
#pragma unroll 1
        for (int it = 0; it < 3; ++it) {                             // kernel row dy = it - 1: taps 3 it .. 3 it + 2
#pragma unroll
            for (int k = 0; k < 24; ++k) {                           // 3 taps x 8 channel groups; 24 % 3 == 0: slots are compile-time
                // top of step k: this wave's pieces of W(k+1) have landed (all but its youngest DMA, W(k+2), 2 pieces)
                // and its weight-fragment reads of slot k % 3 have returned (all but its youngest LDS read, a pixel
                // fragment); after the barrier that holds for every wave:
                // slot (k + 1) % 3 may be read, slot k % 3 may be overwritten
                S14_STAMP_L(t0, 2);
                asm volatile("s_waitcnt vmcnt(2) lgkmcnt(1)" ::: "memory");
                S14_STAMP_L(t1, 2);
                if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
                S14_STAMP_L(t2, 2);
                S14_PIN();
                const int tt = k >> 3;                               // next tap: (it, tt + 1), or (it + 1, 0) after the row's last
                const int dyn = tt < 2 ? it - 1 : it, dxn = tt < 2 ? tt : -1;
                if ((k & 1) == 0) step(a0, a1, k, dyn, dxn);
                else step(a1, a0, k, dyn, dxn);
                S14_STAMP_L(t3, 2);
                if (STAMPS >= 2) { sw += t1 - t0; sb += t2 - t1; sm += t3 - t2; }
            }
        }
        S14_STAMP(tB);
        if (STAMPS) sl_ += tB - t0p;
        // ---- epilogue of the conv: parameters from LDS, residual (second conv of a block), then the image in place
        const bool second = conv & 1;
        half_t* ybase = p.y + (size_t)n * img_elems;
        asm volatile("" : "+v"(px0e), "+v"(fre), "+v"(fqe));
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");             // the parameters: older than the 2 weight pieces in flight
        __builtin_amdgcn_s_barrier();                                // every wave has consumed its last fragments of the old image
        S14_STAMP(t1);
        if (STAMPS == 1) sw += t1 - tB;                              // level-1 stamps: epilogue segments in the per-step slots
        // First conv of a block: the NEXT conv's accumulators start as the block's input (the residual), so that its
        // epilogue needs no second operand.  That input is the OLD image - still in LDS, and the element a lane is about
        // to overwrite with its output is exactly the residual element it needs: read, then write.  The residual
        // stream never leaves the CU; HBM sees the run's input and its last block's output only.
        // per pixel tile j (6 own + the shared one): LDS row offset, swizzle key, parameter row of the pixel's border class
        // (in floats); per cout tile: channel offsets.  px < 196 holds for the own tiles by construction.
        int rowoff[7], key[7], clsoff[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int px = j < 6 ? px0e + 16 * j : 192 + fre;
            const int oy = (px * 4682) >> 16, ox = px - oy * 14;     // px / 14 for px < 256
            const int cls = (oy == 0 ? 0 : (oy >= 13 ? 2 : 1)) * 3 + (ox == 0 ? 0 : (ox == 13 ? 2 : 1));
            clsoff[j] = cls * S14_C;
            const int pxr = px < S14_PX ? px : S14_PX;               // dead lanes of the shared tile: the plane's zero row (never written)
            rowoff[j] = wn * S14_PLANE + pxr * 128 + (fqe & 1) * 8;
            key[j] = pxr & 7;
        }
        const int co_own = wn * 64 + fqe * 4, co_sh = wn * 64 + WP * 32 + fqe * 4;      // + 16 per cout tile
        const int ch_own = fqe >> 1, ch_sh = WP * 4 + (fqe >> 1);                       // 16-B chunk inside the plane row: + 2 per cout tile
        // PRELU is a compile-time flag of two copies of the tile loop: a block's second conv has none (slope 1)
        auto tiles = [&](auto prelu_tag) {
            constexpr bool FIRST = decltype(prelu_tag)::value;       // first conv of a block: PReLU, and the residual refill
            float4v sl[4], slx[2];                                   // slopes depend on the cout only: once per conv
            if constexpr (FIRST) {
#pragma unroll
                for (int i = 0; i < 4; ++i) sl[i] = *reinterpret_cast<const float4v*>(lprm + 9 * S14_C + co_own + i * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) slx[t] = *reinterpret_cast<const float4v*>(lprm + 9 * S14_C + co_sh + t * 16);
            }
#pragma unroll
            for (int t = 0; t < 26; ++t) {
                float4v& a_ = t < 24 ? acc[t >> 2][t & 3] : accx[t - 24];
                const int j = t < 24 ? t >> 2 : 6, co = t < 24 ? co_own + (t & 3) * 16 : co_sh + (t - 24) * 16;
                const int ch = t < 24 ? ch_own + (t & 3) * 2 : ch_sh + (t - 24) * 2;
                half4* dst = reinterpret_cast<half4*>(img + rowoff[j] + ((ch ^ key[j]) << 4));
                float4v v = a_ + *reinterpret_cast<const float4v*>(lprm + clsoff[j] + co);
                if constexpr (FIRST) {
                    const half4 old = *dst;                          // the block's input at the element this lane overwrites (dead lanes: a zero row)
                    const float4v s_ = t < 24 ? sl[t & 3] : slx[t - 24];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * s_[e];
                    a_ = float4v{(float)old[0], (float)old[1], (float)old[2], (float)old[3]};
                } else {
                    a_ = float4v{0.f, 0.f, 0.f, 0.f};
                }
                const half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                if (t < 24) *dst = h;
                else if (192 + fre < S14_PX) *dst = h;
                if ((t & 3) == 3) S14_PIN();
            }
        };
        if (second) tiles(std::false_type{}); else tiles(std::true_type{});
        S14_STAMP(t2);
        if (STAMPS == 1) sb += t2 - t1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // the new image is complete
        S14_STAMP(t3);
        if (STAMPS == 1) sm += t3 - t2;
        if (conv == p.nconv - 1) {                                   // the run's result -> HBM
            for (int e = tid; e < S14_PX * 32; e += 512) {           // 16-B chunks: pixel x 32 chunks
                const int px = e >> 5, c = e & 31;                   // c = plane * 8 + chunk
                const int4v v = *reinterpret_cast<const int4v*>(img + (c >> 3) * S14_PLANE + px * 128 + (((c & 7) ^ (px & 7)) << 4));
                *reinterpret_cast<int4v*>(ybase + (size_t)px * S14_C + c * 8) = v;
            }
        }
    }
    };
    if (wp == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S14_STAMP(tC);
    if (STAMPS) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rB)::"memory");
    if (STAMPS && p.stamps && lane == 0) {
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = sw; o[1] = sb; o[2] = sm; o[3] = se + (tC - tB); o[4] = sp; o[5] = tC - tA; o[6] = rB - rA; o[7] = sl_;
    }
#endif
}

// ---------------------------------------------------------------- host side
extern "C" size_t fr_conv_stage14_weight_bytes(int nconv) { return (size_t)(nconv > 0 ? nconv : 0) * S14_STEPS * S14_SLOT; }

// Re-orders ONE conv's folded weights [256][9 * 256] f16 (K = tap-major, as fr_conv_nhwc_f16 takes them) into its 72
// LDS slot images (see the file header).  Device-side, at model load.
__global__ void stage14_pack_weights(const half_t* __restrict__ w, half_t* __restrict__ out) {
    // one thread per 16-B chunk of the output: 72 slots x 256 rows x 4 chunks
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S14_STEPS * 256 * 4) return;
    const int q = e / 1024, r = e - q * 1024, row = r >> 2, cp = r & 3;          // cp = chunk' in the slot row
    const int chunk = cp ^ (((row >> 3) & 1) << 1);
    const int tap = q >> 3, g = q & 7;
    const int4v v = *reinterpret_cast<const int4v*>(w + (size_t)row * 2304 + tap * 256 + g * 32 + chunk * 8);
    *reinterpret_cast<int4v*>(out + (size_t)q * 8192 + row * 32 + cp * 8) = v;
}

extern "C" int fr_conv_stage14_pack(const void* w, void* out, fr_stream_t stream) {
    FR_REQUIRE(w && out, "fr_conv_stage14_pack: null pointer");
    stage14_pack_weights<<<fr_cdiv(S14_STEPS * 1024, 256), 256, 0, fr_stream(stream)>>>((const half_t*)w, (half_t*)out);
    FR_CHECK_LAUNCH("stage14_pack_weights");
    return FR_OK;
}

extern "C" int fr_conv_stage14_f16(const void* x, void* y, const void* wstream, const float* params, int B, int nblocks,
                                   fr_stream_t stream) {
    FR_REQUIRE(x && y && wstream && params && B > 0 && nblocks > 0, "fr_conv_stage14_f16: bad argument");
    FR_REQUIRE(x != y, "fr_conv_stage14_f16: x and y must be different buffers");
    FR_REQUIRE((int64_t)B * S14_PX * S14_C * 2 < (1ll << 31) && (int64_t)nblocks * 2 * S14_STEPS * S14_SLOT < (1ll << 32) - (1 << 20),
               "fr_conv_stage14_f16: tensor too large (B %d, blocks %d)", B, nblocks);
    StageP p;
    p.x = (const half_t*)x; p.y = (half_t*)y; p.w = (const half_t*)wstream; p.prm = params;
    p.B = B; p.nconv = 2 * nblocks;
    p.xbytes = (unsigned)((int64_t)B * S14_PX * S14_C * 2);
    p.wbytes = (unsigned)((int64_t)p.nconv * S14_STEPS * S14_SLOT);
    p.stamps = (unsigned long long*)fr_dbg_ptr("FR_DBG_STAMPS");    // always NULL in the product build
    p.dephase = fr_dbg_int("FR_S14_DEPHASE", 0);
    if constexpr (FR_DEBUG) {                                       // stamped twin: debug build only
        if (p.stamps) {
            static FrDevLatch dl;
            if (fr_dbg_int("FR_S14_STAMP_LEVEL", 2) == 1) {
                const int abl = fr_dbg_int("FR_S14_ABL", 0);
                auto k1 = abl == 1 ? conv_stage14_kernel<1, 1> : abl == 2 ? conv_stage14_kernel<1, 2> : abl == 3 ? conv_stage14_kernel<1, 3>
                        : abl == 4 ? conv_stage14_kernel<1, 4> : abl == 8 ? conv_stage14_kernel<1, 8> : abl == 7 ? conv_stage14_kernel<1, 7>
                        : conv_stage14_kernel<1, 0>;
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, S14_LDS) != hipSuccess) { fr_set_error("fr_conv_stage14_f16: cannot raise dynamic LDS"); return FR_E_LAUNCH; }
                k1<<<B, 512, S14_LDS, fr_stream(stream)>>>(p);
                FR_CHECK_LAUNCH("conv_stage14_kernel<stamps 1>");
                return FR_OK;
            }
            if (!fr_raise_lds(reinterpret_cast<const void*>(conv_stage14_kernel<2>), S14_LDS, dl)) { fr_set_error("fr_conv_stage14_f16: cannot raise dynamic LDS"); return FR_E_LAUNCH; }
            conv_stage14_kernel<2><<<B, 512, S14_LDS, fr_stream(stream)>>>(p);
            FR_CHECK_LAUNCH("conv_stage14_kernel<stamps>");
            return FR_OK;
        }
    }
    static FrDevLatch latch;
    if (!fr_raise_lds(reinterpret_cast<const void*>(conv_stage14_kernel<0>), S14_LDS, latch)) {
        fr_set_error("fr_conv_stage14_f16: cannot raise dynamic LDS to %d bytes", S14_LDS);
        return FR_E_LAUNCH;
    }
    conv_stage14_kernel<0><<<B, 512, S14_LDS, fr_stream(stream)>>>(p);
    FR_CHECK_LAUNCH("conv_stage14_kernel");
    return FR_OK;
}

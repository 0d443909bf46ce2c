// The run of stride-1 residual blocks of IResNet's 28x28 x 128 stage (r100: 12 blocks = 24 convs, 23 % of the network's
// FLOPs; the embed half of `FaceAnalysis.get`, /root/reference/infrenceServer.py:528) as ONE launch, one workgroup per FACE.
//
// A 28x28x128 f16 map is 200 KB: unlike the 14x14 stage (conv_stage14.hip) it cannot stay in LDS.  Launched layer by layer
// (conv_halo.hip, or this file's loop as a one-layer kernel: 77 - 78 us per 256 faces either way) a layer is two lockstep
// rounds of workgroups whose halo loads, residual reads and output stores all hit HBM together and overlap with nothing:
// ~30 of the 77 us.  Here a workgroup walks its own face through all the convs - a conv of one face needs nothing from
// another face - so the maps only travel between the CU and its XCD's L2 / the 256 MB memory-side cache (2 x 51 MB live
// per 256 faces), workgroups drift out of phase, and there is no launch boundary for the weight stream to stop at.
//
//   pass  = (conv, half of the image): 14 rows x 28 = 392 output pixels x 128 couts, 8 waves x 256 VGPRs:
//           2 cout groups (64 couts) x 4 pixel groups (6 tiles of 16 pixels) + the 25th pixel tile (8 valid pixels) shared
//           by cout: a wave computes ONE cout tile of it, with its first weight fragment (fragment i = cout tile (i + wp) & 3)
//   LDS   halo  2 planes (64 channels each) x 512 rows x 128 B: the half's 16 x 30 input pixels at a pitch of 32 (zero
//               border by out-of-range LDS-DMA; a 1 KB DMA piece = 8 pixels of ONE halo row, and everything lane-dependent
//               in a piece's source address is one constant per lane), row = halo pixel, 16-B chunk' = chunk ^ key, key =
//               the pixel's index in a 28-pitch raster & 7 (conflict-free ds_read_b128 for every tap, as conv_halo.hip)
//         ring  3 slots x [128 couts][32 channels] f16 (8 KB): a K step = one tap x 32 channels = ONE MFMA per tile pair,
//               one LDS-DMA piece per wave per step from ONE pre-swizzled stream for all convs (fr_conv_stage28_pack)
//         prm   the conv's 9 border-class biases + PReLU slope, f32 [10][128]
//   K     PLANE-major: 18 steps (9 taps x 2 channel groups) on plane 0, then 18 on plane 1 - so that plane 0's buffer is
//         dead during the second half of a pass and the NEXT pass's plane 0 is fetched under it; plane 1 follows under the
//         epilogue.  The next pass is the image's other half or, behind a conv's second half, the first half of the NEXT conv's
//         input: 15 of its 16 halo rows were stored a pass ago, the last one follows behind this pass's store drain.  Step as conv_stage14.hip: W(s+3) into step s's own slot, counted
//         vmcnt / lgkmcnt, weights double-buffered, pixel fragments re-read in place, the two waves of a SIMD issue
//         their DMA pieces at opposite ends of a step, role-specialised copies of the unrolled loop.
//   end   PReLU -> f16 straight from the accumulators (which started the pass as the pixel's border-class bias) to HBM (conv1 -> the scratch map `mid`, conv2 -> `x` in place);
//         a block's second conv starts from accumulators that hold the residual (its own tile of `x`, L1-bypassing loads
//         issued with the halo loads).  A workgroup reads back what it wrote: stores are drained (vmcnt(0) + barrier)
//         before the next conv's halo loads, which bypass the vector L1 (sc1).
// `x` is overwritten with the run's output.
#include "common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int int2v __attribute__((ext_vector_type(2)));

namespace {

constexpr int S28_W = 28, S28_TH = 14, S28_C = 128;
constexpr int S28_HROWS = 16 * 32;                                             // halo rows of a half image: 16 x (30 px at a pitch of 32)
constexpr int S28_PLANE = S28_HROWS * 128;                                     // 65 536 B
constexpr int S28_SLOT = 128 * 64;                                             // 8 192
constexpr int S28_STEPS = 36;
constexpr int S28_PRM = 10 * 128 * 4;                                          // 5 120
constexpr int S28_RING = 2 * S28_PLANE, S28_PRMO = S28_RING + 3 * S28_SLOT;
constexpr int S28_LDS = S28_PRMO + S28_PRM;                                    // 160 768
constexpr int S28_PX = S28_TH * S28_W;                                         // 392 output pixels per pass
constexpr int S28_IMG = 784 * S28_C * 2;                                       // bytes per image

struct Stage28P {
    half_t* x; half_t* mid; const half_t* w; const float* prm;
    int B, nconv;
    unsigned xbytes, wbytes;
    unsigned long long* stamps;     // diagnostic build only (FR_DBG_STAMPS=<device ptr>): per-wave cycle sums
};

__device__ __forceinline__ float4v mm16(const int4v& a, const int4v& b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

}  // namespace

#define S28_PIN() __builtin_amdgcn_sched_barrier(0)
#define S28_STAMP(var)                                                                      \
    do {                                                                                    \
        if (STAMPS) {                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                              \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");      \
            __builtin_amdgcn_sched_barrier(0);                                              \
        }                                                                                   \
    } while (0)

// STAMPS (diagnostic build): s_memtime sums per wave - load wait at a pass start, K loop, epilogue issue, drain
template <int STAMPS>
__global__ __launch_bounds__(512, 2) void conv_stage28_kernel(Stage28P p) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long tA = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, rA = 0, rB = 0, s_wait = 0, s_k = 0, s_epi = 0, s_drain = 0, s_pro = 0;
    S28_STAMP(tA);
    if (STAMPS) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rA)::"memory");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ring = lds + S28_RING;
    const float* lprm = reinterpret_cast<const float*>(lds + S28_PRMO);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wp = wave >> 1;                          // SIMD partners: waves w and w + 4 = (wn, wp) and (wn, wp + 2)
    const int n = blockIdx.x;

    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.mid, 0, p.xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)p.prm, 0, (unsigned)p.nconv * S28_PRM, 0x00020000);

    // ---- weight stream: conv c's 36 slots at c * 36 * 8 KB, read once per half; this wave moves piece `wave` (1 KB) of a slot
    unsigned wsrc = 0, wwrap = S28_STEPS * S28_SLOT, wback = 0;       // at wsrc == wwrap continue at wback (second half: the conv again)
    auto issue_w = [&](int slot, int ln) {
        // past the last conv: an out-of-range offset (zeros into a slot nobody reads)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(ring + slot * S28_SLOT + wave * 1024), 16,
                                                 wsrc < p.wbytes ? (unsigned)(wave * 1024 + ln * 16) : 0x80000000u, wsrc, 0, 0);
        wsrc += S28_SLOT;
        if (wsrc == wwrap) wsrc = wback;
    };
    // One 1 KB piece = 8 pixels x 64 channels of halo row hy (0..15), quarter (wave & 3) of the row; this wave's pieces are
    // always rows of parity (wave >> 2) & 1 - so a lane's part of the source offset (pixel in the piece, swizzled chunk,
    // "left of / right of the image") is ONE constant, `lt`; the rest is scalar.  src_ok false or a row outside the image: zeros.
    auto halo_lane = [&](int ln) -> unsigned {
        const int lrow = ln >> 3, ch = ln & 7, q = wave & 3;
        const unsigned out = (unsigned)((q == 0) & (lrow == 0)) | (unsigned)((q == 3) & (lrow >= 5));        // branch-free
        return ((unsigned)((lrow - 1) * 256 + ((ch ^ ((lrow + 4 * ((wave >> 2) & 1)) & 7)) << 4)) & (out - 1u)) | (out << 31);      // out: exactly 2^31
    };
    auto issue_halo = [&](__amdgpu_buffer_rsrc_t rs, int plane, int i, int y0, unsigned lt, bool src_ok) {      // piece i (0..7) of this wave
        int n_ = n;
        asm volatile("" : "+s"(n_));                                  // computed here (a few scalar ops), not kept in SGPRs per piece across the kernel
        const int hy = 2 * i + (wave >> 2), iy = y0 - 1 + hy;
        const bool ok = src_ok && (unsigned)iy < 28u;
        const unsigned so = (unsigned)(((n_ * 28 + iy) * 28 + (wave & 3) * 8) * 256 + plane * 128);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + plane * S28_PLANE + (hy * 4 + (wave & 3)) * 1024), 16,
                                                 ok ? lt + so : 0x80000000u, 0, 0, 16);
    };
    auto issue_prm = [&](int conv, int ln) {                          // 5 KB: pieces 0..4 by waves 0..4
        if (wave < 5)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (lds_ptr_t)(lds + S28_PRMO + wave * 1024), 16, (unsigned)(wave * 1024 + ln * 16), conv * S28_PRM, 0, 0);
    };

    {
        const int ln = tid & 63;
        issue_w(0, ln);
        issue_w(1, ln);
        issue_w(2, ln);
        issue_prm(0, ln);
        const unsigned lt = halo_lane(ln);
        for (int i = 0; i < 16; ++i) issue_halo(xrs, i >> 3, i & 7, 0, lt, true);
    }

    float4v acc[6][4], accx;
    int4v a0[4], a1[4], bt[3], bx;
    int boff[7];

    auto run = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;                  // wp >> 1: which SIMD partner this wave is
#pragma unroll 1
    for (int pass = 0; pass < 2 * p.nconv; ++pass) {
        const int conv = pass >> 1, hf = pass & 1, y0 = hf * S28_TH;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));                               // lane constants re-derived per pass: nothing hoisted into scratch
        const int fr = lane & 15, fq = lane >> 4;
        const bool second = conv & 1;
        __amdgpu_buffer_rsrc_t srs = second ? mrs : xrs;              // this conv's input map
        // The NEXT pass's halo: the other half of the same input - or, behind a conv's second half, the first half of the map
        // this conv writes: 15 of its 16 halo rows come from rows 0..13, stored (and drained) a pass ago; only halo row 15 =
        // image row 14 belongs to the half being computed now and follows behind this pass's store drain.
        __amdgpu_buffer_rsrc_t nxr = hf == 0 ? srs : (second ? xrs : mrs);
        const int nxy0 = hf == 0 ? S28_TH : 0;
        const bool nx_any = hf == 0 || conv + 1 < p.nconv;
        auto nx_ok = [&](int i) { return nx_any && (hf == 0 || !(i == 7 && (wave >> 2) == 1)); };
        half_t* dst = second ? p.x : p.mid;
        const unsigned tile0 = (unsigned)((n * 28 + y0) * 28);        // first output pixel of the half (global pixel index)
        // weight fragment i = cout tile (i + wp) & 3 of the wave's 64 couts: a wave-uniform offset on ONE lane register
        int a_own = fr * 64 + ((fq ^ (((fr >> 3) & 1) << 1)) << 4);
        asm volatile("" : "+v"(a_own));
        // halo row of tile j's pixel at tap (0, 0), kept per tile (7 registers): a tap's fragment address is a handful of VALU ops
        // from it - no division by 28 inside the K loop (quarter-rate multiplies: as much VALU time as the step's MFMAs)
        int hb[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int px = (wp * 6 + j) * 16 + fr;
            const int oy = px / S28_W, ox = px - oy * S28_W;
            hb[j] = (oy + 1) * 32 + ox + 1;
            asm volatile("" : "+v"(hb[j]));
        }
        auto set_tap_one = [&](int j, int dy, int dx, int plane) {
            int l_ = lane;
            asm volatile("" : "+v"(l_));                             // derived here from ONE live register
            // the shared tile: pixels 384 .. 391 = row 13, columns 20 .. 27 (dead lanes: pixel 0), not worth a register
            const int h = (j < 6 ? hb[j < 6 ? j : 0] : ((l_ & 15) < 8 ? 14 * 32 + 21 + (l_ & 15) : 33)) + dy * 32 + dx;
            // key = (hy * 28 + hx) & 7 = (4 * (hy & 1) + hx) & 7
            boff[j] = plane * S28_PLANE + h * 128 + (((l_ >> 4) ^ ((((h >> 3) & 4) + h) & 7)) << 4);
        };
        auto rd_a = [&](int slot, int i) {
            return *reinterpret_cast<const int4v*>(ring + slot * S28_SLOT + (wn * 4096 + ((i + wp) & 3) * 1024) + a_own);
        };
        auto rd_b = [&](int gg, int j) { return *reinterpret_cast<const int4v*>(lds + (gg ? (boff[j] ^ 64) : boff[j])); };
        auto co_of = [&](int i) { return wn * 64 + ((i + wp) & 3) * 16 + fq * 4; };

        S28_STAMP(t0);
        // ---- accumulators start as their pixel's border-class bias (+ the residual: second conv of a block, this tile of x).
        // Here, not in the epilogue: 100 b128 LDS reads per wave are 6 k cycles of LDS time per pass, free while the halo is in flight.
        if (hf == 0) {       // the conv's parameters were issued BEFORE the halo pieces still in flight (kernel start: 16; else waves 4..7: 2)
            if (pass == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (wave >> 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        auto cls_off = [&](int px) {
            const int oy = px / S28_W, ox = px - oy * S28_W, ho = y0 + oy;
            return ((ho == 0 ? 0 : (ho == 27 ? 2 : 1)) * 3 + (ox == 0 ? 0 : (ox == 27 ? 2 : 1))) * S28_C;
        };
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int cj = cls_off((wp * 6 + j) * 16 + fr);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = *reinterpret_cast<const float4v*>(lprm + cj + co_of(i));
        }
        accx = *reinterpret_cast<const float4v*>(lprm + cls_off(fr < 8 ? 384 + fr : 0) + co_of(0));
        // A vector-memory instruction costs the CU ~16 cycles whatever its width, and 8 waves x 100 accumulator tiles of 8 B per
        // lane were 13 k cycles per pass for the stores alone (and as much for the residual loads).  So a lane moves 16 B = 8
        // consecutive couts: v_permlane16_swap trades the odd fq rows of cout fragment i with the even rows of fragment i + 1 -
        // lanes fq = 0, 2 then hold couts 0..7 / 8..15 of fragment i, lanes fq = 1, 3 those of fragment i + 1.
        auto co8 = [&](int ip) { return wn * 64 + ((ip + (fq & 1) + wp) & 3) * 16 + (fq >> 1) * 8; };
        if (second) {
            int4v r[6][2];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
                    r[j][ip] = __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(
                        xrs, (tile0 + (unsigned)((wp * 6 + j) * 16 + fr)) * 256u + (unsigned)co8(2 * ip) * 2u, 0, 16));
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)r[j][ip][0], (unsigned)r[j][ip][2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)r[j][ip][1], (unsigned)r[j][ip][3], false, false);
                    const half4 ha = __builtin_bit_cast(half4, int2v{(int)s0[0], (int)s1[0]});
                    const half4 hb_ = __builtin_bit_cast(half4, int2v{(int)s0[1], (int)s1[1]});
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[j][2 * ip][e] += (float)ha[e]; acc[j][2 * ip + 1][e] += (float)hb_[e]; }
                }
            const int2v rx = __builtin_bit_cast(int2v, __builtin_amdgcn_raw_buffer_load_b64(
                xrs, fr < 8 ? (tile0 + (unsigned)(384 + fr)) * 256u + (unsigned)co_of(0) * 2u : 0x80000000u, 0, 16));
            const half4 hx = __builtin_bit_cast(half4, rx);
#pragma unroll
            for (int e = 0; e < 4; ++e) accx[e] += (float)hx[e];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // halo planes, W(0..2), parameters
        __builtin_amdgcn_s_barrier();
        S28_STAMP(t1);
        // second half of a conv: the weights again; else on to the next conv
        wwrap = (unsigned)(conv + 1) * (S28_STEPS * S28_SLOT);
        wback = hf == 0 ? (unsigned)conv * (S28_STEPS * S28_SLOT) : wwrap;

#pragma unroll
        for (int j = 0; j < 7; ++j) set_tap_one(j, -1, -1, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[i] = rd_a(0, i);
#pragma unroll
        for (int j = 0; j < 3; ++j) bt[j] = rd_b(0, j);
        bx = rd_b(0, 6);
        S28_STAMP(t2);

        // one K step: 25 MFMAs on fragments read during the previous step; the next step's fragment reads between them
        auto step = [&](int4v (&ac)[4], int4v (&an)[4], int k, int dyn, int dxn, int pln, auto pf_tag) {
            constexpr bool PF = decltype(pf_tag)::value;
            const int gg = k & 1, ngg = (k + 1) & 1, nslot = (k + 1) % 3;
            auto dma = [&]() {
                if constexpr (PF) {                                  // the other half's plane 0 into the (dead) plane-0 buffer
                    if (k < 4) {
                        int l_ = lane;
                        asm volatile("" : "+v"(l_));                 // the lane constant derived here, not kept (or spilled) across the loop
                        const unsigned ltp = halo_lane(l_);
                        issue_halo(nxr, 0, 2 * k, nxy0, ltp, nx_ok(2 * k));
                        issue_halo(nxr, 0, 2 * k + 1, nxy0, ltp, nx_ok(2 * k + 1));
                    }
                }
                issue_w(k % 3, lane);
            };
            if constexpr (ROLE == 0) { dma(); S28_PIN(); }
            accx = mm16(ac[0], bx, accx);
            S28_PIN();
#pragma unroll
            for (int j = 0; j < 6; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = mm16(ac[i], bt[j % 3], acc[j][i]);
                // pixel fragments live in a ring of three: tile j's register takes tile j + 3 (this step's, then the next step's)
                if (j == 0) { an[0] = rd_a(nslot, 0); an[1] = rd_a(nslot, 1); }
                if (j == 1) { an[2] = rd_a(nslot, 2); an[3] = rd_a(nslot, 3); }
                if (j < 3) {
                    bt[j] = rd_b(gg, j + 3);
                } else {
                    if (gg == 1) { set_tap_one(j - 3, dyn, dxn, pln); set_tap_one(j, dyn, dxn, pln); }
                    bt[j - 3] = rd_b(ngg, j - 3);
                    if (j == 5) {
                        if (gg == 1) set_tap_one(6, dyn, dxn, pln);
                        bx = rd_b(ngg, 6);
                    }
                }
                S28_PIN();
                if constexpr (ROLE == 1) { if (j == 3) { dma(); S28_PIN(); } }
            }
        };
        // six steps = one kernel row (3 taps x 2 channel groups) of one plane; 6 % 3 == 0: ring slots are compile-time
        auto row = [&](int it, int plane, auto pf_tag) {
            constexpr bool PF = decltype(pf_tag)::value;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                // W(s+1) was the last DMA of step s-2; behind it: step s-1's pieces (PF, steps 0..3: two halo pieces + W(s+2))
                if (PF && k >= 1 && k <= 4) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(1)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                S28_PIN();
                const int tt = k >> 1;
                const bool last = tt == 2 && it == 2;                // behind the plane's last tap: the next plane's first
                const int dyn = last ? -1 : (tt < 2 ? it - 1 : it), dxn = tt < 2 ? tt : -1;
                const int pln = last ? (plane ^ 1) & 1 : plane;
                if ((k & 1) == 0) step(a0, a1, k, dyn, dxn, pln, pf_tag);
                else step(a1, a0, k, dyn, dxn, pln, pf_tag);
            }
        };
#pragma unroll 1
        for (int it = 0; it < 3; ++it) row(it, 0, std::false_type{});
        row(0, 1, std::true_type{});                                  // plane 0's buffer is dead: the other half's plane 0 arrives under this row
#pragma unroll 1
        for (int it = 1; it < 3; ++it) row(it, 1, std::false_type{});

        S28_STAMP(t3);
        // ---- end of the pass.  Plane 1's buffer is dead once every wave is past its last fragment read.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int le = lane;
        asm volatile("" : "+v"(le));
        const unsigned lt = halo_lane(le);
        for (int i = 0; i < 8; ++i) issue_halo(nxr, 1, i, nxy0, lt, nx_ok(i));
        S28_PIN();
        const int fre = le & 15, fqe = le >> 4;
        auto coe = [&](int i) { return wn * 64 + ((i + wp) & 3) * 16 + fqe * 4; };
        float4v sv[4];                                               // read unconditionally: a conditional definition is carried through
#pragma unroll                                                       // the pass loop (16 registers spilled across every K loop)
        for (int i = 0; i < 4; ++i) sv[i] = *reinterpret_cast<const float4v*>(lprm + 9 * S28_C + coe(i));
        auto act = [&](float4v v, int i) {                           // the bias is in the accumulators since the pass start
            if (!second) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sv[i][e];
            }
            return __builtin_bit_cast(int2v, half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]});
        };
        auto co8e = [&](int ip) { return wn * 64 + ((ip + (fqe & 1) + wp) & 3) * 16 + (fqe >> 1) * 8; };
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int2v pa = act(acc[j][2 * ip], 2 * ip), pb = act(acc[j][2 * ip + 1], 2 * ip + 1);
                const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)pa[0], (unsigned)pb[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)pa[1], (unsigned)pb[1], false, false);
                *reinterpret_cast<int4v*>(dst + (size_t)(tile0 + (wp * 6 + j) * 16 + fre) * S28_C + co8e(2 * ip)) =
                    int4v{(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
            }
        if (fre < 8)
            *reinterpret_cast<int2v*>(dst + (size_t)(tile0 + 384 + fre) * S28_C + coe(0)) = act(accx, 0);
        S28_STAMP(t4);
        if (STAMPS) { s_wait += t1 - t0; s_pro += t2 - t1; s_k += t3 - t2; s_epi += t4 - t3; }
        if (hf == 1) {
            // the conv is complete once every wave's stores are: then the next conv's input halo (what this workgroup just wrote)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (conv + 1 < p.nconv) {
                issue_prm(conv + 1, lane);                           // first: the next pass waits for it with two pieces in flight
                if (wave >> 2) {                                     // halo row 15 of both planes: this wave's piece 7
                    issue_halo(nxr, 0, 7, 0, lt, true);
                    issue_halo(nxr, 1, 7, 0, lt, true);
                }
            }
            if (STAMPS) { unsigned long long t5; S28_STAMP(t5); s_drain += t5 - t4; }
        }
    }
    };
    if ((wp >> 1) == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (STAMPS) {
        unsigned long long tZ;
        S28_STAMP(tZ);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rB)::"memory");
        if (p.stamps && (tid & 63) == 0) {
            unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
            o[0] = s_wait; o[1] = s_pro; o[2] = s_k; o[3] = s_epi; o[4] = s_drain; o[5] = tZ - tA; o[6] = rB - rA; o[7] = tA;
        }
    }
#endif
}

// ---------------------------------------------------------------- host side
extern "C" size_t fr_conv_stage28_weight_bytes(int nconv) { return nconv > 0 ? (size_t)nconv * S28_STEPS * S28_SLOT : 0; }

// folded weights [128][9 * 128] f16 (K = tap-major) -> 36 slot images [128 rows][32 channels], slot = plane * 18 + tap * 2 + gg
__global__ void stage28_pack_weights(const half_t* __restrict__ w, half_t* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;             // one thread per 16-B chunk: 36 x 128 rows x 4 chunks
    if (e >= S28_STEPS * 512) return;
    const int q = e / 512, r = e - q * 512, row = r >> 2, cp = r & 3;
    const int chunk = cp ^ (((row >> 3) & 1) << 1);
    const int plane = q / 18, k = q - plane * 18, tap = k >> 1, gg = k & 1;
    const int4v v = *reinterpret_cast<const int4v*>(w + (size_t)row * 1152 + tap * 128 + plane * 64 + gg * 32 + chunk * 8);
    *reinterpret_cast<int4v*>(out + (size_t)q * 4096 + row * 32 + cp * 8) = v;
}

extern "C" int fr_conv_stage28_pack(const void* w, void* out, fr_stream_t stream) {
    FR_REQUIRE(w && out, "fr_conv_stage28_pack: null pointer");
    stage28_pack_weights<<<fr_cdiv(S28_STEPS * 512, 256), 256, 0, fr_stream(stream)>>>((const half_t*)w, (half_t*)out);
    FR_CHECK_LAUNCH("stage28_pack_weights");
    return FR_OK;
}

extern "C" int fr_conv_stage28_f16(void* x, void* mid, const void* wstream, const float* params, int B, int nblocks, fr_stream_t stream) {
    FR_REQUIRE(x && mid && wstream && params && B > 0 && nblocks > 0, "fr_conv_stage28_f16: bad argument");
    FR_REQUIRE(x != mid, "fr_conv_stage28_f16: x and mid must be different buffers");
    FR_REQUIRE((int64_t)B * S28_IMG < (1ll << 31) && (int64_t)nblocks * 2 * S28_STEPS * S28_SLOT < (1ll << 31),
               "fr_conv_stage28_f16: tensor too large (B %d, blocks %d)", B, nblocks);
    Stage28P p;
    p.x = (half_t*)x; p.mid = (half_t*)mid; p.w = (const half_t*)wstream; p.prm = params;
    p.B = B; p.nconv = 2 * nblocks;
    p.xbytes = (unsigned)((int64_t)B * S28_IMG);
    p.wbytes = (unsigned)((int64_t)p.nconv * S28_STEPS * S28_SLOT);
    p.stamps = (unsigned long long*)fr_dbg_ptr("FR_DBG_STAMPS");    // always NULL in the product build
    if constexpr (FR_DEBUG) {                                       // stamped twin: debug build only
        if (p.stamps) {
            static FrDevLatch dl;
            if (!fr_raise_lds(reinterpret_cast<const void*>(conv_stage28_kernel<1>), S28_LDS, dl)) { fr_set_error("fr_conv_stage28_f16: cannot raise dynamic LDS"); return FR_E_LAUNCH; }
            conv_stage28_kernel<1><<<B, 512, S28_LDS, fr_stream(stream)>>>(p);
            FR_CHECK_LAUNCH("conv_stage28_kernel<stamps>");
            return FR_OK;
        }
    }
    static FrDevLatch latch;
    if (!fr_raise_lds(reinterpret_cast<const void*>(conv_stage28_kernel<0>), S28_LDS, latch)) {
        fr_set_error("fr_conv_stage28_f16: cannot raise dynamic LDS to %d bytes", S28_LDS);
        return FR_E_LAUNCH;
    }
    conv_stage28_kernel<0><<<B, 512, S28_LDS, fr_stream(stream)>>>(p);
    FR_CHECK_LAUNCH("conv_stage28_kernel");
    return FR_OK;
}

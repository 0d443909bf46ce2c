// 3x3 / stride 1 / pad 1 convs with 64 INPUT channels at 56x56 and 112x112 (IResNet-100: the first block's conv1 at 112x112,
// the four stride-1 convs of stage 1 and stage 2's entry conv at 56x56: 0.47 TFLOP of a 256-face forward; embed half of
// `FaceAnalysis.get`, /root/reference/infrenceServer.py:528), one launch per layer.
//
// With K = 576 a tile's K loop is short and what the per-tile kernel (conv_halo.hip, two blocks per CU) spends its time on is
// the skeleton around it - halo load, residual load, store, one of each in flight per block: 0.55 PFLOP/s.  Here ONE workgroup
// walks a whole face (x a 64-cout group) region by region with the pass machinery of conv_stage28.hip, so the skeleton of
// region r + 1 runs under the K loop of region r:
//   region 14 rows x 28 columns = 392 output pixels x 64 couts (8 regions per 56x56 image, 32 per 112x112), 8 waves x 3 pixel
//          tiles x all 4 cout tiles + the 25th half tile (every wave computes cout tile `wave & 3` of it with its weight fragment 0 =
//          cout tile (0 + wave) & 3; waves 4..7 discard theirs)
//   LDS    TWO halo buffers of 512 rows x 128 B (16 x 30 input pixels at a pitch of 32, 64 channels; zero border by out-of-range
//          LDS-DMA; 16-B chunk XOR key, key = the pixel's index in a 28-pitch raster & 7): region r + 1's halo is fetched into the
//          other buffer during region r's K loop, one 1 KB piece per wave per step;
//          weight ring 3 slots x [64 couts][64 channels] f16 (8 KB, rows of 128 B with the chunk XOR row & 7): a K step = one
//          TAP = two MFMAs per tile pair (9 steps per region), one LDS-DMA piece per wave per step, the conv's 72 KB of weights
//          streamed again for every region (L2); 9 border-class biases + PReLU slope (f32 [10][64])
//   step   W(s+3) into step s's own slot, then the next region's halo piece; counted vmcnt / lgkmcnt, one barrier; weight
//          fragments double-buffered, pixel fragments re-read in place; role-specialised copies for the two waves of a SIMD
//   end    accumulators start as border-class bias (+ residual: 16-byte L1-bypassing loads via v_permlane16_swap); PReLU -> f16 ->
//          16-byte stores straight from the accumulators; the stores drain under the next region's first steps' waits.
// y must not alias x; residual may alias y.
#include "common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int int2v __attribute__((ext_vector_type(2)));

namespace {

constexpr int K64_RW = 28, K64_RH = 14, K64_C = 64;                  // region width / height, input channels
constexpr int K64_PX = K64_RW * K64_RH;                              // 392
constexpr int K64_HALO = 512 * 128;                                  // one halo buffer: 65 536 B
constexpr int K64_SLOT = 64 * 128;                                   // 8 192
constexpr int K64_RING = 2 * K64_HALO, K64_PRMO = K64_RING + 3 * K64_SLOT;
constexpr int K64_PRM = 10 * 64 * 4;                                 // 2 560
constexpr int K64_LDS = K64_PRMO + K64_PRM;                          // 158 208

struct Walk64P {
    const half_t* x; const half_t* w; half_t* y;
    const float* bias; const float* slope; const half_t* res;
    int B, HW, Cout, bias_mode;                                      // HW: image height = width (56 or 112)
    int nsplit;                                                      // workgroups per (face, cout group): each walks nreg / nsplit regions
    unsigned xbytes, ybytes, wbytes;
};

__device__ __forceinline__ float4v mmk(const int4v& a, const int4v& b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

}  // namespace

#define K64_PIN() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(512, 2) void conv_walk64_kernel(Walk64P p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* ring = lds + K64_RING;
    float* lprm = reinterpret_cast<float*>(lds + K64_PRMO);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncg = p.Cout >> 6;
    const int cg = blockIdx.x % ncg, part = (blockIdx.x / ncg) % p.nsplit, n = blockIdx.x / (ncg * p.nsplit);   // cout groups of a face next to each other: they share its input in L2
    const int HW = p.HW, nbx = HW / K64_RW;
    const int reg0 = part * (nbx * (HW / K64_RH) / p.nsplit), nreg = reg0 + nbx * (HW / K64_RH) / p.nsplit;       // this workgroup's regions: [reg0, nreg)

    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.xbytes, 0x00020000);

    // ---- weight stream of this cout group: 9 slots (taps) of 8 KB, streamed once per region; this wave moves piece `wave`
    const unsigned wbase = (unsigned)cg * (9 * K64_SLOT);
    unsigned wq = 0;                                                 // tap of the next slot to fetch
    auto issue_w = [&](int slot, int ln) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(ring + slot * K64_SLOT + wave * 1024), 16,
                                                 (unsigned)(wave * 1024 + ln * 16), wbase + wq * K64_SLOT, 0, 0);
        wq = wq == 8 ? 0 : wq + 1;
    };
    // One 1 KB piece = 8 pixels x 64 channels of halo row hy (0..15) of a region, quarter (wave & 3) of the row; this wave's
    // pieces are rows of parity (wave >> 2) & 1.  Lane part of the source offset: pixel in the piece, swizzled chunk, and
    // "outside the image" (left / right: only where the region touches the image's edge) as an offset of exactly 2^31.
    auto halo_lane = [&](int ln, bool left_edge, bool right_edge) -> unsigned {
        const int lrow = ln >> 3, ch = ln & 7, q = wave & 3;
        const unsigned out = (unsigned)((q == 0) & (lrow == 0) & left_edge) | (unsigned)((q == 3) & (lrow == 5) & right_edge) |
                             (unsigned)((q == 3) & (lrow >= 6));
        return ((unsigned)((lrow - 1) * 128 + ((ch ^ ((lrow + 4 * ((wave >> 2) & 1)) & 7)) << 4)) & (out - 1u)) | (out << 31);
    };
    auto issue_halo = [&](int buf, int i, int reg, int ln) {         // piece i (0..7) of this wave, region `reg` (>= nreg: zeros)
        int n_ = n;
        asm volatile("" : "+s"(n_));
        const int by = reg / nbx, bx_ = reg - by * nbx;
        const int y0 = by * K64_RH, x0 = bx_ * K64_RW;
        const unsigned lt = halo_lane(ln, x0 == 0, x0 + K64_RW == HW);
        const int hy = 2 * i + (wave >> 2), iy = y0 - 1 + hy;
        const bool ok = reg < nreg && (unsigned)iy < (unsigned)HW;
        const unsigned so = (unsigned)(((n_ * HW + iy) * HW + x0 + (wave & 3) * 8) * 128);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_ptr_t)(lds + buf * K64_HALO + (hy * 4 + (wave & 3)) * 1024), 16,
                                                 ok ? lt + so : 0x80000000u, 0, 0, 0);
    };

    {
        const int ln = tid & 63;
        issue_w(0, ln);
        issue_w(1, ln);
        issue_w(2, ln);
        for (int i = 0; i < 8; ++i) issue_halo(reg0 & 1, i, reg0, ln);
        // parameters of this cout group: 9 bias rows (bias_mode 0: the same row nine times) + slope (none: 1.0)
        for (int e = tid; e < 640; e += 512) {
            const int r = e >> 6, c = e & 63;
            float v;
            if (r < 9) v = p.bias ? p.bias[(p.bias_mode == 1 ? r * p.Cout : 0) + cg * 64 + c] : 0.f;
            else v = p.slope ? p.slope[cg * 64 + c] : 1.f;
            lprm[e] = v;
        }
    }

    float4v acc[3][4], accx;
    int4v a0[2][4], a1[2][4], b[3][2], bx[2];
    int boff[4];

    auto run = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;                  // wave >> 2: which SIMD partner this wave is
#pragma unroll 1
    for (int reg = reg0; reg < nreg; ++reg) {
        const int cur = reg & 1;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));                               // lane constants re-derived per region: nothing hoisted into scratch
        const int fr = lane & 15, fq = lane >> 4;
        const int by = reg / nbx, bx_ = reg - by * nbx;
        const int y0 = by * K64_RH, x0 = bx_ * K64_RW;
        auto gpix = [&](int px) {                                    // region pixel -> global pixel index
            const int oy = px / K64_RW, ox = px - oy * K64_RW;
            return (unsigned)((n * HW + y0 + oy) * HW + x0 + ox);
        };
        auto cls_off = [&](int px) {
            const int oy = px / K64_RW, ox = px - oy * K64_RW, ho = y0 + oy, wo = x0 + ox;
            return ((ho == 0 ? 0 : (ho == HW - 1 ? 2 : 1)) * 3 + (wo == 0 ? 0 : (wo == HW - 1 ? 2 : 1))) * 64;
        };
        // weight fragment f = cout tile (f + wave) & 3, so fragment 0 is the one a wave computes of the shared tile (no register
        // select); lanes fq = 0, 2 / 1, 3 move couts 0..7 / 8..15 of fragments (2 ip, 2 ip + 1) in 16-B pieces
        auto ct = [&](int f) { return (f + wave) & 3; };
        auto co8 = [&](int ip) { return ct(2 * ip + (fq & 1)) * 16 + (fq >> 1) * 8; };
        int a_own = fr * 128 + ((fq ^ (fr & 7)) << 4);               // + i * 2048 (cout tile), ^ 64 (second half of the step's 64 channels)
        asm volatile("" : "+v"(a_own));
        int hb[3];                                                   // halo row of tile t's pixel at tap (0, 0)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int px = (wave * 3 + t) * 16 + fr;
            const int oy = px / K64_RW, ox = px - oy * K64_RW;
            hb[t] = (oy + 1) * 32 + ox + 1;
            asm volatile("" : "+v"(hb[t]));
        }
        auto set_tap_one = [&](int t, int dy, int dx) {
            int l_ = lane;
            asm volatile("" : "+v"(l_));
            // the shared tile: pixels 384 .. 391 = row 13, columns 20 .. 27 (dead lanes: pixel 0)
            const int h = (t < 3 ? hb[t < 3 ? t : 0] : ((l_ & 15) < 8 ? 14 * 32 + 21 + (l_ & 15) : 33)) + dy * 32 + dx;
            boff[t] = cur * K64_HALO + h * 128 + (((l_ >> 4) ^ ((((h >> 3) & 4) + h) & 7)) << 4);     // key = (4 * (hy & 1) + hx) & 7
        };
        auto rd_a = [&](int slot, int kh, int i) { return *reinterpret_cast<const int4v*>(ring + slot * K64_SLOT + ct(i) * 2048 + (kh ? (a_own ^ 64) : a_own)); };
        auto rd_b = [&](int kh, int t) { return *reinterpret_cast<const int4v*>(lds + (kh ? (boff[t] ^ 64) : boff[t])); };

        // ---- accumulators: the pixel's border-class bias (+ residual)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (reg == reg0) __builtin_amdgcn_s_barrier();               // the parameter rows written above
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int cj = cls_off((wave * 3 + t) * 16 + fr);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i] = *reinterpret_cast<const float4v*>(lprm + cj + ct(i) * 16 + fq * 4);
        }
        accx = *reinterpret_cast<const float4v*>(lprm + cls_off(fr < 8 ? 384 + fr : 0) + (wave & 3) * 16 + fq * 4);
        if (p.res) {
            __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.ybytes, 0x00020000);
            int4v r[3][2];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
                    r[t][ip] = __builtin_bit_cast(int4v, __builtin_amdgcn_raw_buffer_load_b128(
                        rrs, (gpix((wave * 3 + t) * 16 + fr) * (unsigned)p.Cout + (unsigned)(cg * 64 + co8(ip))) * 2u, 0, 16));
            const int2v rx = __builtin_bit_cast(int2v, __builtin_amdgcn_raw_buffer_load_b64(
                rrs, fr < 8 ? (gpix(384 + fr) * (unsigned)p.Cout + (unsigned)(cg * 64 + (wave & 3) * 16 + fq * 4)) * 2u : 0x80000000u, 0, 16));
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)r[t][ip][0], (unsigned)r[t][ip][2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)r[t][ip][1], (unsigned)r[t][ip][3], false, false);
                    const half4 ha = __builtin_bit_cast(half4, int2v{(int)s0[0], (int)s1[0]});
                    const half4 hb_ = __builtin_bit_cast(half4, int2v{(int)s0[1], (int)s1[1]});
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[t][2 * ip][e] += (float)ha[e]; acc[t][2 * ip + 1][e] += (float)hb_[e]; }
                }
            const half4 hx = __builtin_bit_cast(half4, rx);
#pragma unroll
            for (int e = 0; e < 4; ++e) accx[e] += (float)hx[e];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this region's halo, W(0..2), the previous region's stores
        __builtin_amdgcn_s_barrier();

#pragma unroll
        for (int t = 0; t < 4; ++t) set_tap_one(t, -1, -1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a0[kh][i] = rd_a(0, kh, i);
#pragma unroll
            for (int t = 0; t < 3; ++t) b[t][kh] = rd_b(kh, t);
            bx[kh] = rd_b(kh, 3);
        }

        // one K step = one tap: 2 x 13 MFMAs on fragments read during the previous step; the next step's fragment reads between them
        auto step = [&](int4v (&ac)[2][4], int4v (&an)[2][4], int s) {
            const int nslot = (s + 1) % 3;
            const int tn = s + 1 < 9 ? s + 1 : 0;                    // next tap (behind the last one: the next region's first - re-read there)
            const int dyn = tn / 3 - 1, dxn = tn % 3 - 1;
            auto dma = [&]() {
                if (s < 8) issue_halo(cur ^ 1, s, reg + 1, lane);    // the next region's halo into the other buffer
                issue_w(s % 3, lane);                                // last: the next step but one waits for it with two pieces behind it
            };
            if constexpr (ROLE == 0) { dma(); K64_PIN(); }
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                accx = mmk(ac[kh][0], bx[kh], accx);
                K64_PIN();
#pragma unroll
                for (int t = 0; t < 3; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][i] = mmk(ac[kh][i], b[t][kh], acc[t][i]);
                    // weights of the next step: two fragments behind each of the first four tile groups
                    if (kh == 0) { an[0][t] = rd_a(nslot, 0, t); if (t == 2) an[0][3] = rd_a(nslot, 0, 3); }
                    else { an[1][t] = rd_a(nslot, 1, t); if (t == 2) an[1][3] = rd_a(nslot, 1, 3); }
                    if (kh == 0) set_tap_one(t, dyn, dxn);           // this tile's fragment of half 1 is in registers already
                    b[t][kh] = rd_b(kh, t);
                    if (t == 2) {
                        if (kh == 0) set_tap_one(3, dyn, dxn);
                        bx[kh] = rd_b(kh, 3);
                    }
                    K64_PIN();
                }
                if constexpr (ROLE == 1) { if (kh == 0) { dma(); K64_PIN(); } }
            }
        };
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            // W(s+1) was the LAST DMA of step s-2; behind it step s-1's two (a halo piece of the next region, its weight piece)
            asm volatile("s_waitcnt vmcnt(2) lgkmcnt(1)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            K64_PIN();
            if ((s & 1) == 0) step(a0, a1, s); else step(a1, a0, s);
        }

        // ---- end of the region: PReLU -> f16 -> 16-byte stores straight from the accumulators
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int le = lane;
        asm volatile("" : "+v"(le));
        const int fre = le & 15, fqe = le >> 4;
        float4v sv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[i] = *reinterpret_cast<const float4v*>(lprm + 9 * 64 + ct(i) * 16 + fqe * 4);
        auto act = [&](float4v v, const float4v& s_) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * s_[e];
            return __builtin_bit_cast(int2v, half4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]});
        };
        half_t* yo = p.y + cg * 64;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const size_t gp = (size_t)gpix((wave * 3 + t) * 16 + fre) * p.Cout;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int2v pa = act(acc[t][2 * ip], sv[2 * ip]), pb = act(acc[t][2 * ip + 1], sv[2 * ip + 1]);
                const auto s0 = __builtin_amdgcn_permlane16_swap((unsigned)pa[0], (unsigned)pb[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap((unsigned)pa[1], (unsigned)pb[1], false, false);
                *reinterpret_cast<int4v*>(yo + gp + ct(2 * ip + (fqe & 1)) * 16 + (fqe >> 1) * 8) = int4v{(int)s0[0], (int)s1[0], (int)s0[1], (int)s1[1]};
            }
        }
        if (wave < 4 && fre < 8)
            *reinterpret_cast<int2v*>(yo + (size_t)gpix(384 + fre) * p.Cout + wave * 16 + fqe * 4) = act(accx, sv[0]);
    }
    };
    if ((wave >> 2) == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
}

// ---------------------------------------------------------------- host side
extern "C" size_t fr_conv_walk64_weight_bytes(int Cout) { return Cout > 0 ? (size_t)(Cout / 64) * 9 * K64_SLOT : 0; }

// folded weights [Cout][9 * 64] f16 (K = tap-major) -> per 64-cout group 9 slot images [64 rows][64 channels], chunk XOR row & 7
__global__ void walk64_pack_weights(const half_t* __restrict__ w, half_t* __restrict__ out, int ncg) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;             // one thread per 16-B chunk: ncg x 9 x 64 rows x 8 chunks
    if (e >= ncg * 9 * 512) return;
    const int q_ = e / 512, r = e - q_ * 512, row = r >> 3, cp = r & 7;
    const int cgi = q_ / 9, tap = q_ - cgi * 9;
    const int chunk = cp ^ (row & 7);
    const int4v v = *reinterpret_cast<const int4v*>(w + (size_t)(cgi * 64 + row) * 576 + tap * 64 + chunk * 8);
    *reinterpret_cast<int4v*>(out + (size_t)q_ * 4096 + row * 64 + cp * 8) = v;
}

extern "C" int fr_conv_walk64_pack(const void* w, void* out, int Cout, fr_stream_t stream) {
    FR_REQUIRE(w && out && Cout > 0 && Cout % 64 == 0, "fr_conv_walk64_pack: bad argument (Cout %d)", Cout);
    const int ncg = Cout / 64;
    walk64_pack_weights<<<fr_cdiv((int64_t)ncg * 9 * 512, 256), 256, 0, fr_stream(stream)>>>((const half_t*)w, (half_t*)out, ncg);
    FR_CHECK_LAUNCH("walk64_pack_weights");
    return FR_OK;
}

extern "C" int fr_conv_walk64_f16(const void* x, const void* wstream, void* y, const float* bias, int bias_mode, const float* slope,
                                  const void* residual, int B, int HW, int Cout, fr_stream_t stream) {
    FR_REQUIRE(x && wstream && y && B > 0, "fr_conv_walk64_f16: bad argument");
    FR_REQUIRE(x != y, "fr_conv_walk64_f16: y must not alias x");
    FR_REQUIRE(HW > 0 && HW % 28 == 0 && Cout > 0 && Cout % 64 == 0, "fr_conv_walk64_f16: image side %d must be a multiple of 28, Cout %d of 64", HW, Cout);
    FR_REQUIRE(bias_mode == 0 || bias_mode == 1, "fr_conv_walk64_f16: bad bias_mode");
    FR_REQUIRE((int64_t)B * HW * HW * 64 * 2 < (1ll << 31) && (int64_t)B * HW * HW * Cout * 2 < (1ll << 32),
               "fr_conv_walk64_f16: tensor too large for 32-bit buffer offsets (split the batch)");
    Walk64P p;
    p.x = (const half_t*)x; p.w = (const half_t*)wstream; p.y = (half_t*)y;
    p.bias = bias; p.slope = slope; p.res = (const half_t*)residual;
    p.B = B; p.HW = HW; p.Cout = Cout; p.bias_mode = bias_mode;
    // fewer (face, cout group) pairs than CUs: cut a face's walk into 2 / 4 / 8 pieces (a piece keeps >= 2 regions: the prefetch)
    const int nregs = (HW / 28) * (HW / 14);
    p.nsplit = 1;
    while (p.nsplit < 8 && (int64_t)B * (Cout / 64) * p.nsplit * 2 <= 256 && nregs % (p.nsplit * 2) == 0 && nregs / (p.nsplit * 2) >= 2) p.nsplit *= 2;
    p.xbytes = (unsigned)((int64_t)B * HW * HW * 64 * 2);
    p.ybytes = (unsigned)((int64_t)B * HW * HW * Cout * 2);
    p.wbytes = (unsigned)((int64_t)(Cout / 64) * 9 * K64_SLOT);
    static FrDevLatch latch;
    if (!fr_raise_lds(reinterpret_cast<const void*>(conv_walk64_kernel), K64_LDS, latch)) {
        fr_set_error("fr_conv_walk64_f16: cannot raise dynamic LDS to %d bytes", K64_LDS);
        return FR_E_LAUNCH;
    }
    conv_walk64_kernel<<<B * (Cout / 64) * p.nsplit, 512, K64_LDS, fr_stream(stream)>>>(p);
    FR_CHECK_LAUNCH("conv_walk64_kernel");
    return FR_OK;
}

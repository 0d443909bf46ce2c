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
//         tile (pixels 192..195) is shared by cout: 2 cout tiles per wave -> 26 accumulator tiles per wave, every wave
//   step  s: wait own W(s+1) pieces -> barrier -> issue W(s+2) (2 LDS-DMA pieces per wave) -> 26 MFMAs on the
//         fragments read during step s-1, with the 13 fragment reads of step s+1 between them
//   conv  end: barrier -> bias (9 border classes) -> PReLU (slope 1 = none) -> [+ residual] -> f16 into the image in
//         place -> barrier; after a block's second conv the image is also copied to HBM (the next block's residual
//         and, at the end, the result): coalesced 16-B stores; residuals are read back with L1-bypassing loads.
//
// Weight stream (host: iresnet.py pack_stage_weights): per conv 72 slots, slot = step q = tap * 8 + g (g = 32-channel
// group), each 16 KB in LDS image order.  Parameters per conv: f32 [10][256] = 9 border-class biases + PReLU slope.
#include "common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

namespace {

constexpr int S14_PX = 196, S14_C = 256, S14_ROWS = 200;            // rows per plane (196 pixels + zero rows)
constexpr int S14_PLANE = S14_ROWS * 128;                           // bytes
constexpr int S14_IMG = 4 * S14_PLANE;                              // 102 400
constexpr int S14_SLOT = 256 * 64;                                  // 16 384
constexpr int S14_LDS = S14_IMG + 3 * S14_SLOT;                     // 151 552
constexpr int S14_STEPS = 72;                                       // per conv

struct StageP {
    const half_t* x;        // [B][196][256] input of the first block
    half_t* y;              // [B][196][256] residual stream / output (written after every block)
    const half_t* w;        // [nconv][72][8192] pre-swizzled weight stream
    const float* prm;       // [nconv][10][256]
    int B, nconv;
    unsigned xbytes, wbytes;
};

__device__ __forceinline__ float4v mfma16(const int4v& a, const int4v& b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

}  // namespace

__global__ __launch_bounds__(512, 2) void conv_stage14_kernel(StageP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* img = lds;
    char* ring = lds + S14_IMG;
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

    // A fragment: row (cout) fr of a 16-cout tile, 16-B chunk fq ^ key(row)
    const int a_lane = fr * 64 + ((fq ^ (((fr >> 3) & 1) << 1)) << 4);
    const int a_own = wn * 4096 + a_lane;                            // + i * 1024: cout tile i of the wave's 64 couts
    const int a_sh = wn * 4096 + wp * 2048 + a_lane;                 // + t * 1024: the wave's 2 cout tiles of the shared pixel tile
    const int px0 = wp * 96 + fr;                                    // pixel of tile 0; tile j: + 16 j; the shared 13th tile: 192 + fr

    float4v acc[6][4], accx[2];
    // fragments: the weights (A) are double-buffered across steps; a pixel fragment (B) is re-read for the NEXT step
    // into its own registers as soon as the current step's MFMAs of that pixel tile are issued
    int4v a0[4], ax0[2], a1[4], ax1[2], b[6], bx;
    int boff[7];        // per tap: byte offset of the lane's tap pixel row + chunk bits, for even g; odd g: ^ 64

    auto set_tap = [&](int dy, int dx) {                             // dy, dx in -1..1
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int px = j < 6 ? px0 + 16 * j : 192 + fr;
            const int oy = px / 14, ox = px - oy * 14;
            const bool ok = px < S14_PX && (unsigned)(oy + dy) < 14u && (unsigned)(ox + dx) < 14u;
            const int pxn = ok ? px + dy * 14 + dx : S14_PX;         // the zero row
            boff[j] = pxn * 128 + ((fq ^ (pxn & 7)) << 4);
        }
    };
    auto rd_a = [&](int slot, int i) { return *reinterpret_cast<const int4v*>(ring + slot * S14_SLOT + a_own + i * 1024); };
    auto rd_ax = [&](int slot, int t) { return *reinterpret_cast<const int4v*>(ring + slot * S14_SLOT + a_sh + t * 1024); };
    auto rd_b = [&](int g, int j) { return *reinterpret_cast<const int4v*>(img + (g >> 1) * S14_PLANE + ((g & 1) ? (boff[j] ^ 64) : boff[j])); };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // image, W(0), W(1)
    __builtin_amdgcn_s_barrier();

#define S14_PIN() __builtin_amdgcn_sched_barrier(0)
    // One K step (local index k of a 24-step group: slot k % 3, channel group g = k & 7) on the fragments (ac, axc, b,
    // bx); meanwhile the NEXT step's fragments are read: weights into (an, axn) from slot (k + 1) % 3, pixel tile j
    // into b[j] right behind the MFMAs that used it.  Before a step with g == 7 reads its successor's pixels, boff
    // moves to the next tap (dyn, dxn).  No branches: the last step of a conv prefetches too - its weight fragments are
    // the next conv's first (that slot has landed), its pixel fragments are dead (the prologue re-reads them).
    auto step = [&](int4v (&ac)[4], int4v (&axc)[2], int4v (&an)[4], int4v (&axn)[2], int k, int dyn, int dxn) {
        const int g = k & 7, ng = (k + 1) & 7, nslot = (k + 1) % 3;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = mfma16(ac[i], b[j], acc[j][i]);
            if (j == 0 && g == 7) set_tap(dyn, dxn);                 // every pixel fragment of THIS step is in registers
            b[j] = rd_b(ng, j);
            if (j < 4) an[j] = rd_a(nslot, j); else axn[j - 4] = rd_ax(nslot, j - 4);
            S14_PIN();
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) accx[t] = mfma16(axc[t], bx, accx[t]);
        bx = rd_b(ng, 6);
        S14_PIN();
    };

#pragma unroll 1
    for (int conv = 0; conv < p.nconv; ++conv) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = float4v{0.f, 0.f, 0.f, 0.f};
        accx[0] = accx[1] = float4v{0.f, 0.f, 0.f, 0.f};
        // prologue: fragments of the conv's first step (its slot, 0, landed before the previous conv's last barrier)
        set_tap(-1, -1);
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[i] = rd_a(0, i);
#pragma unroll
        for (int t = 0; t < 2; ++t) ax0[t] = rd_ax(0, t);
#pragma unroll
        for (int j = 0; j < 6; ++j) b[j] = rd_b(0, j);
        bx = rd_b(0, 6);
#pragma unroll 1
        for (int it = 0; it < 3; ++it) {                             // kernel row dy = it - 1: taps 3 it .. 3 it + 2
#pragma unroll
            for (int k = 0; k < 24; ++k) {                           // 3 taps x 8 channel groups; 24 % 3 == 0: slots are compile-time
                // top of step: own pieces of the next step's slot have landed; after the barrier every wave's have,
                // and slot (k + 2) % 3 is free (its fragments were consumed by MFMAs issued before this barrier)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue_w((k + 2) % 3);
                S14_PIN();
                const int tt = k >> 3;                               // next tap: (it, tt + 1), or (it + 1, 0) after the row's last
                const int dyn = tt < 2 ? it - 1 : it, dxn = tt < 2 ? tt : -1;
                if ((k & 1) == 0) step(a0, ax0, a1, ax1, k, dyn, dxn);
                else step(a1, ax1, a0, ax0, k, dyn, dxn);
            }
        }
        // ---- epilogue of the conv: parameters, residual (second conv of a block), then the image in place
        const float* prm = p.prm + (size_t)conv * 10 * S14_C;
        const bool second = conv & 1;
        half_t* ybase = p.y + (size_t)n * img_elems;
        const half_t* rbase = conv == 1 ? p.x + (size_t)n * img_elems : ybase;     // the block's input: x for the first block
        __builtin_amdgcn_s_barrier();                                // every wave has consumed its last fragments of the old image
        auto finish = [&](float4v v, int px, int co) {
            if (px >= S14_PX) return;
            const int oy = px / 14, ox = px - oy * 14;
            const int cls = (oy == 0 ? 0 : (oy == 13 ? 2 : 1)) * 3 + (ox == 0 ? 0 : (ox == 13 ? 2 : 1));
            v += *reinterpret_cast<const float4v*>(prm + cls * S14_C + co);
            const float4v sl = *reinterpret_cast<const float4v*>(prm + 9 * S14_C + co);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sl[e];
            if (second) {
                // L1-bypassing load: this CU wrote these bytes one block ago; its vector L1 is not refreshed by stores
                const unsigned long long rb = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(rbase + (size_t)px * S14_C + co),
                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const half4 rv = __builtin_bit_cast(half4, rb);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            }
            const half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            const int c64 = co & 63;                                 // channel inside the plane
            *reinterpret_cast<half4*>(img + (co >> 6) * S14_PLANE + px * 128 + ((((c64 >> 3)) ^ (px & 7)) << 4) + (c64 & 4) * 2) = h;
        };
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) finish(acc[j][i], (wp * 6 + j) * 16 + fr, wn * 64 + i * 16 + fq * 4);
#pragma unroll
        for (int t = 0; t < 2; ++t) finish(accx[t], 192 + fr, wn * 64 + wp * 32 + t * 16 + fq * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // the new image is complete
        if (second) {                                                // block output -> HBM (next block's residual; the result)
            for (int e = tid; e < S14_PX * 32; e += 512) {           // 16-B chunks: pixel x 32 chunks
                const int px = e >> 5, c = e & 31;                   // c = plane * 8 + chunk
                const int4v v = *reinterpret_cast<const int4v*>(img + (c >> 3) * S14_PLANE + px * 128 + (((c & 7) ^ (px & 7)) << 4));
                *reinterpret_cast<int4v*>(ybase + (size_t)px * S14_C + c * 8) = v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    static FrDevLatch latch;
    if (!fr_raise_lds(reinterpret_cast<const void*>(conv_stage14_kernel), S14_LDS, latch)) {
        fr_set_error("fr_conv_stage14_f16: cannot raise dynamic LDS to %d bytes", S14_LDS);
        return FR_E_LAUNCH;
    }
    conv_stage14_kernel<<<B, 512, S14_LDS, fr_stream(stream)>>>(p);
    FR_CHECK_LAUNCH("conv_stage14_kernel");
    return FR_OK;
}

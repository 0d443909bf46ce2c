// The small tail layers of MTCNN's R-Net / O-Net - conv 2x2 on a 4x4 map, the dense layers and the O-Net's third conv (R-Net
// conv3 48 -> 64 and dense4 576 -> 128, O-Net conv3 64 -> 64 + pool, conv4 64 -> 128 and dense5 1152 -> 256; detector half of FaceAnalysis.get,
// /root/reference/infrenceServer.py:528) - as ONE split-precision GEMM kernel on the f16 matrix cores.
//
// Why: on the f32 matrix instruction these four launches were 0.36 ms of a 64 x 1080p detector batch (32 768 R-Net and 4 096
// O-Net crops) at a third of that instruction's roof - they are short GEMMs (K = 192 .. 1 152) whose operands fit the caches.
// Every f32 operand is x = hi + lo in f16 and a product hi*hi + lo*hi + hi*lo on v_mfma_f32_16x16x32_f16 (f32 accumulate):
// the layers' outputs differ from the f32 layers' by ~1e-6 of their scale; like the second layers (ro_conv2.hip) they run on
// the batch path only, where the crops near the stage threshold are re-evaluated by the all-f32 layers (the exact pass).
//
// GEMM view: D[cout][row] = sum_k W[cout][k] X[row][k]; a row = one output pixel of one crop slot, and its K values are KH
// contiguous runs of L = KW x Cin floats of the input map (NHWC: the KW pixels of a kernel row lie side by side), L % 32 == 0
// for every layer - so a K step of 32 is 128 contiguous bytes per row and the operand needs no gather.
//   block  = 64 rows x all couts, 4 waves = 4 groups of N / 64 cout tiles x the 4 row tiles (a lane: one row, 4 couts)
//   X      f32 from HBM / L2 (2 float4 per thread and K step, one step ahead in registers) -> hi | lo f16 -> LDS
//          [2 buffers][2 planes][64 rows][64 B], 16-B chunk c of row r at c ^ ((r >> 1) & 3): conflict-free ds_read_b128
//   W      pre-split by the host into MFMA fragment order [K step][cout tile][hi | lo][lane][16 B]: a wave's fragments come
//          straight from global memory (L2-resident: every block reads the same few hundred KB), one step ahead
// Count-aware like the other R-/O-Net layers: a block whose rows belong to empty slots only exits at once.
#include "common.h"

namespace {

struct RgArgs {
    const float* x;            // [slots][Hin][Win][Cin] f32
    const unsigned char* w;    // fragments: [K / 32][N / 16][2][64][16 B] f16
    const float* bias; const float* slope;      // [N]; slope may be NULL (no PReLU)
    float* y;                  // [slots][Ho][Wo][N] f32
    const int32_t* counts; int cap;             // slot s holds a crop iff s % cap < counts[s / cap]
    int nslots;
};

// L: floats of one kernel row's run (KW * Cin); KH runs per row; N couts, NB of them per block (blockIdx.y picks the group: the
// dense layers of the O-Net have 4 096 rows only - 64 row tiles - and want more blocks than that); HIN x WIN input map of CIN
// channels; output HO x WO
// POOL2 (O-Net conv3: 3x3, 64 -> 64, 10x10 -> 8x8): the block's 64 rows are ONE crop's 8 x 8 conv map, tile t = image rows 2t and
// 2t + 1, so the 2x2 / stride-2 max pool is a maximum over lanes li, li ^ 1, li ^ 8, li ^ 9 of one accumulator tile: no LDS pass
template <int L, int KH, int N, int NB, int HIN, int WIN, int CIN, int HO, int WO, bool POOL2 = false>
__global__ __launch_bounds__(256, 2) void ro_gemm_split_kernel(RgArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int MT = 64, K = KH * L, NKS = K / 32, NT = NB / 64, P = HO * WO;
    static_assert(L % 32 == 0 && NB % 64 == 0 && N % NB == 0, "run length and couts");
    static_assert(!POOL2 || (HO == 8 && WO == 8), "the fused pool wants one 8 x 8 map per block");
    __shared__ __attribute__((aligned(16))) char xs[2][2][MT * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int row0 = blockIdx.x * MT;
    const int nrows = a.nslots * P;
    {   // the block's slots [s_lo, s_hi]: any valid?
        const int s_lo = row0 / P, s_hi = min(row0 + MT - 1, nrows - 1) / P;
        bool any = false;
        for (int f = s_lo / a.cap; f * a.cap <= s_hi && !any; ++f) {
            const int first = max(s_lo, f * a.cap);
            any = first - f * a.cap < a.counts[f];
        }
        if (!any) return;
    }
    // ---- this thread's two X pieces per K step: rows r and r + 32, floats [4 c4, 4 c4 + 4) of the step's 32
    const int xr = tid >> 3, c4 = tid & 7;
    const float* xrow[2];
    bool xok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = row0 + xr + 32 * h;
        xok[h] = m < nrows;
        const int mm = xok[h] ? m : 0;
        const int slot = mm / P, p = mm - slot * P, y = p / WO, x = p - y * WO;
        xrow[h] = a.x + ((size_t)slot * HIN * WIN + y * WIN + x) * CIN + c4 * 4;
    }
    auto ldx = [&](int ks, float4v (&v)[2]) {
        const int kh = (ks * 32) / L, kk = ks * 32 - kh * L;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            v[h] = xok[h] ? *reinterpret_cast<const float4v*>(xrow[h] + kh * WIN * CIN + kk) : float4v{0.f, 0.f, 0.f, 0.f};
    };
    auto stx = [&](int buf, const float4v (&v)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = xr + 32 * h;
            half4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const half_t t = (half_t)v[h][e];
                hi[e] = t; lo[e] = (half_t)(v[h][e] - (float)t);
            }
            const int off = r * 64 + (((c4 >> 1) ^ ((r >> 1) & 3)) << 4) + (c4 & 1) * 8;
            *reinterpret_cast<half4*>(xs[buf][0] + off) = hi;
            *reinterpret_cast<half4*>(xs[buf][1] + off) = lo;
        }
    };
    // ---- weight fragments of this wave's NT cout tiles
    const int ct0 = blockIdx.y * (NB / 16) + wave * NT;            // the wave's first cout tile
    const unsigned char* wbase = a.w + ((size_t)ct0 * 2 * 64 + lane) * 16;
    auto ldw = [&](int ks, half8 (&wh)[NT], half8 (&wl)[NT]) {
        const unsigned char* p = wbase + (size_t)ks * (N / 16) * 2 * 1024;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            wh[i] = *reinterpret_cast<const half8*>(p + (size_t)i * 2048);
            wl[i] = *reinterpret_cast<const half8*>(p + (size_t)i * 2048 + 1024);
        }
    };

    float4v acc[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = float4v{0.f, 0.f, 0.f, 0.f};
    float4v xv[2];
    half8 wh[NT], wl[NT], whn[NT], wln[NT];
    ldx(0, xv);
    ldw(0, wh, wl);
    stx(0, xv);
    __syncthreads();
#pragma unroll 1
    for (int ks = 0; ks < NKS; ++ks) {
        const int buf = ks & 1;
        const bool more = ks + 1 < NKS;
        if (more) { ldx(ks + 1, xv); ldw(ks + 1, whn, wln); }      // global loads fly under this step's MFMAs
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = t * 16 + li;
            const int off = r * 64 + ((kq ^ ((r >> 1) & 3)) << 4);
            const half8 bh = *reinterpret_cast<const half8*>(xs[buf][0] + off);
            const half8 bl = *reinterpret_cast<const half8*>(xs[buf][1] + off);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], bh, acc[i][t], 0, 0, 0);
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], bl, acc[i][t], 0, 0, 0);
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], bh, acc[i][t], 0, 0, 0);
            }
        }
        if (more) {
            stx(buf ^ 1, xv);                                   // the other buffer: its readers finished a step ago
#pragma unroll
            for (int i = 0; i < NT; ++i) { wh[i] = whn[i]; wl[i] = wln[i]; }
        }
        __syncthreads();
    }
    // ---- epilogue: lane = row t * 16 + li, couts (wave * NT + i) * 16 + 4 kq .. + 3
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int co = (ct0 + i) * 16 + 4 * kq;
        const float4v bb = *reinterpret_cast<const float4v*>(a.bias + co);
        float4v ss = {1.f, 1.f, 1.f, 1.f};
        if (a.slope) ss = *reinterpret_cast<const float4v*>(a.slope + co);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int m = row0 + t * 16 + li;
            float4v v = acc[i][t] + bb;
            if (a.slope) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * ss[e];
            }
            if constexpr (POOL2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float mx = fmaxf(v[e], __shfl_xor(v[e], 1, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
                    v[e] = mx;
                }
                if ((li & 9) == 0)          // the window's top-left lane: pooled pixel (t, (li & 7) >> 1) of slot blockIdx.x
                    *reinterpret_cast<float4v*>(a.y + ((size_t)blockIdx.x * 16 + t * 4 + ((li & 7) >> 1)) * N + co) = v;
            } else if (m < nrows) {
                *reinterpret_cast<float4v*>(a.y + (size_t)m * N + co) = v;
            }
        }
    }
#endif
}

template <int L, int KH, int N, int NB, int HIN, int WIN, int CIN, int HO, int WO, bool POOL2 = false>
int launch_rg(const RgArgs& a, hipStream_t s) {
    const int64_t nrows = (int64_t)a.nslots * HO * WO;
    ro_gemm_split_kernel<L, KH, N, NB, HIN, WIN, CIN, HO, WO, POOL2><<<dim3((unsigned)((nrows + 63) / 64), N / NB), 256, 0, s>>>(a);
    return FR_OK;
}

// f32 weights [N][K] (K = (kh, kw, ch) ascending: the layer's im2col order) -> fragment order, split
__global__ void ro_gemm_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ out, int N, int K) {
    const int e = blockIdx.x * 256 + threadIdx.x;              // one thread per (K step, cout tile, plane, lane)
    const int total = (K / 32) * (N / 16) * 2 * 64;
    if (e >= total) return;
    const int lane = e & 63, plane = (e >> 6) & 1, rest = e >> 7;
    const int ct = rest % (N / 16), ks = rest / (N / 16);
    const float* src = w + (size_t)(ct * 16 + (lane & 15)) * K + ks * 32 + 8 * (lane >> 4);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = src[j];
        const half_t h = (half_t)v;
        o[j] = plane ? (half_t)(v - (float)h) : h;
    }
    *reinterpret_cast<half8*>(out + (size_t)e * 16) = o;
}

}  // namespace

extern "C" size_t fr_ro_gemm_weight_bytes(int layer) {
    switch (layer) {
        case 12: return (size_t)64 * 192 * 4;
        case 22: return (size_t)64 * 576 * 4;
        case 13: return (size_t)128 * 576 * 4;
        case 23: return (size_t)128 * 256 * 4;
        case 24: return (size_t)256 * 1152 * 4;
        default: return 0;
    }
}

extern "C" int fr_ro_gemm_pack(int layer, const float* w, void* out, fr_stream_t stream) {
    FR_REQUIRE(w && out, "fr_ro_gemm_pack: null pointer");
    int N, K;
    switch (layer) {
        case 12: N = 64; K = 192; break;
        case 22: N = 64; K = 576; break;
        case 13: N = 128; K = 576; break;
        case 23: N = 128; K = 256; break;
        case 24: N = 256; K = 1152; break;
        default: FR_REQUIRE(false, "fr_ro_gemm_pack: layer must be 12, 13, 22, 23 or 24 (got %d)", layer);
    }
    const int total = (K / 32) * (N / 16) * 2 * 64;
    ro_gemm_pack_kernel<<<(total + 255) / 256, 256, 0, fr_stream(stream)>>>(w, (unsigned char*)out, N, K);
    FR_CHECK_LAUNCH("ro_gemm_pack_kernel");
    return FR_OK;
}

extern "C" int fr_ro_gemm_split(int layer, const float* x, const void* w_packed, const float* bias, const float* slope, float* y,
                                int nslots, const int32_t* counts, int cap, fr_stream_t stream) {
    FR_REQUIRE(x && w_packed && bias && y && counts, "fr_ro_gemm_split: null pointer");
    FR_REQUIRE(nslots > 0 && cap > 0 && nslots % cap == 0, "fr_ro_gemm_split: nslots must be frames x cap");
    RgArgs a{x, (const unsigned char*)w_packed, bias, slope, y, counts, cap, nslots};
    hipStream_t s = fr_stream(stream);
    int rc;
    const bool few = (int64_t)nslots <= 16384;                 // few rows: cout groups of 64 per block, more blocks
    switch (layer) {          //                  L   KH   N   NB HIN WIN CIN HO WO
        case 12: rc = launch_rg<96, 2, 64, 64, 4, 4, 48, 3, 3>(a, s); break;        // R-Net conv3: 2x2, 48 -> 64, 4x4 -> 3x3
        case 13: rc = few ? launch_rg<576, 1, 128, 64, 3, 3, 64, 1, 1>(a, s)       // R-Net dense4: the 3x3x64 map -> 128
                          : launch_rg<576, 1, 128, 128, 3, 3, 64, 1, 1>(a, s); break;
        case 22: rc = launch_rg<192, 3, 64, 64, 10, 10, 64, 8, 8, true>(a, s); break;   // O-Net conv3: 3x3, 64 -> 64, 10x10 -> 8x8, + 2x2 / s2 pool -> 4x4
        case 23: rc = launch_rg<128, 2, 128, 128, 4, 4, 64, 3, 3>(a, s); break;     // O-Net conv4: 2x2, 64 -> 128
        case 24: rc = few ? launch_rg<1152, 1, 256, 64, 3, 3, 128, 1, 1>(a, s)     // O-Net dense5: the 3x3x128 map -> 256
                          : launch_rg<1152, 1, 256, 256, 3, 3, 128, 1, 1>(a, s); break;
        default: FR_REQUIRE(false, "fr_ro_gemm_split: layer must be 12, 13, 22, 23 or 24 (got %d)", layer);
    }
    if (rc != FR_OK) return rc;
    FR_CHECK_LAUNCH("ro_gemm_split_kernel");
    return FR_OK;
}

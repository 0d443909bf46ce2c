// MTCNN cascade kernels (detector half of FaceAnalysis.get, /root/reference/infrenceServer.py:528):
// pyramid resize, direct f32 convolutions (P/R/O-Net are 3..128-channel nets: VALU/HBM-bound,
// not MFMA-shaped), max pool, candidate generation, box refinement, crop+resize, stage select.
// Arithmetic order mirrors oracle/detect.py op for op (built with -ffp-contract=off; fused
// multiply-adds are written explicitly where the spec allows them).
#include "common.h"

// ------------------------------------------------------------------ resize helpers
struct Lerp { int i0, i1; float w; };

__device__ __forceinline__ Lerp lerp_coord(int d, float ratio, int n) {
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

// frames u8 [N,H,W,3] BGR -> out f32 [N,hs,ws,3] RGB, (v - 127.5) * 0.0078125
__global__ void pyramid_resize_norm(const uint8_t* __restrict__ frames, int N, int H, int W, int hs, int ws,
                                    float* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)N * hs * ws;
    if (t >= total) return;
    int x = (int)(t % ws);
    int y = (int)((t / ws) % hs);
    int n = (int)(t / ((int64_t)ws * hs));
    const float ry = (float)H / (float)hs, rx = (float)W / (float)ws;
    Lerp ly = lerp_coord(y, ry, H), lx = lerp_coord(x, rx, W);
    const uint8_t* f = frames + (int64_t)n * H * W * 3;
    const uint8_t* r0 = f + (int64_t)ly.i0 * W * 3;
    const uint8_t* r1 = f + (int64_t)ly.i1 * W * 3;
    float* o = out + t * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {            // output channel c (RGB) <- input channel 2-c (BGR)
        int ci = 2 - c;
        float v = bilerp((float)r0[lx.i0 * 3 + ci], (float)r0[lx.i1 * 3 + ci], (float)r1[lx.i0 * 3 + ci],
                         (float)r1[lx.i1 * 3 + ci], lx.w, ly.w);
        o[c] = (v - 127.5f) * 0.0078125f;
    }
}

extern "C" int fr_pyramid_resize_norm(const uint8_t* frames, int nframes, int H, int W, int hs, int ws, float* out,
                                      fr_stream_t stream) {
    FR_REQUIRE(frames && out, "fr_pyramid_resize_norm: null pointer");
    FR_REQUIRE(nframes > 0 && H > 0 && W > 0 && hs > 0 && ws > 0, "fr_pyramid_resize_norm: bad size");
    int64_t total = (int64_t)nframes * hs * ws;
    pyramid_resize_norm<<<fr_cdiv(total, 256), 256, 0, fr_stream(stream)>>>(frames, nframes, H, W, hs, ws, out);
    FR_CHECK_LAUNCH("pyramid_resize_norm");
    return FR_OK;
}

// ------------------------------------------------------------------ direct convolution (valid), f32 NHWC
// One thread = one output pixel (or one 2x2-pooled output pixel) x CT output channels.
// Weights [KH][KW][Cin][CoutP] (CoutP = Cout rounded up to CT, zero padded) are wave-uniform:
// the compiler keeps them in SGPRs (s_load), so the inner loop is v_fmac with a scalar operand.
// Optional fused head (P-Net conv3 -> conv4_1|conv4_2): after PReLU, y2[h] = b2[h] + sum_c act[c]*w2[c][h].
template <int CT, bool POOL2, int NHEAD>
__global__ __launch_bounds__(256) void dconv_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                 const float* __restrict__ bias, const float* __restrict__ slope,
                                                 float* __restrict__ y, int B, int H, int W, int Cin, int Cout,
                                                 int CoutP, int KH, int KW, int Ho, int Wo,
                                                 const float* __restrict__ w2, const float* __restrict__ b2) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * Ho * Wo;
    if (t >= total) return;
    const int co0 = blockIdx.y * CT;
    const int wo = (int)(t % Wo), ho = (int)((t / Wo) % Ho), n = (int)(t / ((int64_t)Wo * Ho));
    constexpr int NP = POOL2 ? 4 : 1;
    const int Hc = H - KH + 1, Wc = W - KW + 1;     // conv output extent (before pooling)
    float best[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) best[c] = -INFINITY;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        const int hy = POOL2 ? ho * 2 + (pp >> 1) : ho;
        const int wx = POOL2 ? wo * 2 + (pp & 1) : wo;
        if (POOL2 && (hy >= Hc || wx >= Wc)) continue;       // ceil-mode window clipped at the border
        float acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = bias[co0 + c];
        for (int kh = 0; kh < KH; ++kh) {
            const float* xr = x + (((int64_t)n * H + hy + kh) * W + wx) * Cin;
            const float* wr = w + (int64_t)kh * KW * Cin * CoutP + co0;
            for (int kc = 0; kc < KW * Cin; ++kc) {           // (kw, ci) are contiguous in NHWC
                const float xv = xr[kc];
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = __builtin_fmaf(xv, wr[(int64_t)kc * CoutP + c], acc[c]);
            }
        }
        if (slope) {
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = acc[c] > 0.f ? acc[c] : acc[c] * slope[co0 + c];
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) best[c] = fmaxf(best[c], acc[c]);
    }
    if (NHEAD > 0) {
        float h[NHEAD > 0 ? NHEAD : 1];
#pragma unroll
        for (int k = 0; k < NHEAD; ++k) h[k] = b2[k];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int k = 0; k < NHEAD; ++k) h[k] = __builtin_fmaf(best[c], w2[c * NHEAD + k], h[k]);
        float* o = y + t * NHEAD;
#pragma unroll
        for (int k = 0; k < NHEAD; ++k) o[k] = h[k];
    } else {
        float* o = y + t * Cout + co0;
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (co0 + c < Cout) o[c] = best[c];
    }
}

extern "C" int fr_dconv_f32(const float* x, const float* w, const float* bias, const float* slope, float* y, int B,
                            int H, int W, int Cin, int Cout, int CoutP, int KH, int KW, int pool2,
                            const float* head_w, const float* head_b, int nhead, fr_stream_t stream) {
    FR_REQUIRE(x && w && bias && y, "fr_dconv_f32: null pointer");
    FR_REQUIRE(B > 0 && H >= KH && W >= KW && Cin > 0 && Cout > 0 && KH > 0 && KW > 0, "fr_dconv_f32: bad geometry");
    const int Hc = H - KH + 1, Wc = W - KW + 1;
    const int Ho = pool2 ? (Hc + 1) / 2 : Hc, Wo = pool2 ? (Wc + 1) / 2 : Wc;
    const int64_t total = (int64_t)B * Ho * Wo;
    hipStream_t s = fr_stream(stream);
    const int gx = fr_cdiv(total, 256);
#define DCONV(CT, P2, NH)                                                                                       \
    do {                                                                                                        \
        FR_REQUIRE(CoutP % CT == 0 && CoutP >= Cout, "fr_dconv_f32: CoutP must be Cout rounded up to %d", CT);  \
        dconv_f32<CT, P2, NH><<<dim3(gx, CoutP / CT), 256, 0, s>>>(x, w, bias, slope, y, B, H, W, Cin, Cout,    \
                                                                    CoutP, KH, KW, Ho, Wo, head_w, head_b);     \
    } while (0)
    if (nhead > 0) {
        FR_REQUIRE(nhead == 6 && CoutP == 32 && Cout == 32 && !pool2 && head_w && head_b,
                   "fr_dconv_f32: fused head needs Cout == 32 and nhead == 6");
        DCONV(32, false, 6);
    } else if (pool2) {
        FR_REQUIRE(CoutP == 16, "fr_dconv_f32: pool2 variant is built for CoutP == 16 (P-Net conv1)");
        DCONV(16, true, 0);
    } else if (CoutP % 32 == 0) {
        DCONV(32, false, 0);
    } else if (CoutP % 16 == 0) {
        DCONV(16, false, 0);
    } else {
        FR_REQUIRE(false, "fr_dconv_f32: CoutP must be a multiple of 16");
    }
#undef DCONV
    FR_CHECK_LAUNCH("dconv_f32");
    return FR_OK;
}

// ------------------------------------------------------------------ max pool (ceil mode), f32 NHWC
__global__ void maxpool_f32(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C, int k,
                            int s, int Ho, int Wo) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)B * Ho * Wo * C;
    if (t >= total) return;
    int c = (int)(t % C);
    int wo = (int)((t / C) % Wo), ho = (int)((t / ((int64_t)C * Wo)) % Ho), n = (int)(t / ((int64_t)C * Wo * Ho));
    float m = -INFINITY;
    for (int i = 0; i < k; ++i) {
        int h = ho * s + i;
        if (h >= H) break;
        for (int j = 0; j < k; ++j) {
            int w = wo * s + j;
            if (w >= W) break;
            m = fmaxf(m, x[(((int64_t)n * H + h) * W + w) * C + c]);
        }
    }
    y[t] = m;
}

static int pool_out(int n, int k, int s) {
    int o = (n - k + s - 1) / s + 1;       // ceil((n-k)/s) + 1
    if ((o - 1) * s >= n) --o;             // last window must start inside the input
    return o;
}

extern "C" int fr_maxpool_f32(const float* x, float* y, int B, int H, int W, int C, int k, int stride,
                              fr_stream_t stream) {
    FR_REQUIRE(x && y && B > 0 && H >= k && W >= k && C > 0 && k > 0 && stride > 0, "fr_maxpool_f32: bad argument");
    int Ho = pool_out(H, k, stride), Wo = pool_out(W, k, stride);
    int64_t total = (int64_t)B * Ho * Wo * C;
    maxpool_f32<<<fr_cdiv(total, 256), 256, 0, fr_stream(stream)>>>(x, y, B, H, W, C, k, stride, Ho, Wo);
    FR_CHECK_LAUNCH("maxpool_f32");
    return FR_OK;
}

// ------------------------------------------------------------------ P-Net candidates
__device__ __forceinline__ float softmax2_face(float a0, float a1) {
    float m = fmaxf(a0, a1);
    float e0 = expf(a0 - m), e1 = expf(a1 - m);
    return e1 / (e0 + e1);
}

// head: f32 [N, hc, wc, 6] = (logit0, logit1, reg0..3).  Pass 1: per-block pass counts.
// dl (optional, fused P-Net): f32 [N, hc, wc] approximate logit1 - logit0; a cell with dl < dl_min is certainly below the
// threshold (pnet_fused.hip re-evaluated every cell at or above dl_min exactly), so its 24-byte head row is not read.
__global__ __launch_bounds__(256) void pnet_count(const float* __restrict__ head, int cells, float thr,
                                                  int32_t* __restrict__ block_counts, float* __restrict__ prob_out,
                                                  const float* __restrict__ dl, float dl_min) {
    const int f = blockIdx.y, b = blockIdx.x;
    const int cell = b * 256 + threadIdx.x;
    bool pass = false;
    if (cell < cells && (!dl || prob_out || dl[(int64_t)f * cells + cell] >= dl_min)) {
        const float* h = head + ((int64_t)f * cells + cell) * 6;
        float p = softmax2_face(h[0], h[1]);
        if (prob_out) prob_out[(int64_t)f * cells + cell] = p;
        pass = p >= thr;
    }
    __shared__ int wsum[4];
    unsigned long long m = __ballot(pass);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[(int64_t)f * gridDim.x + b] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Pass 2: ordered (raster) compaction.  boxes/scores/regs are per-frame lists of `cap` slots.
__global__ __launch_bounds__(256) void pnet_emit(const float* __restrict__ head, int cells, int wc, float scale,
                                                 float thr, int cap, const int32_t* __restrict__ block_counts,
                                                 float* __restrict__ boxes, float* __restrict__ scores,
                                                 float* __restrict__ regs, int32_t* __restrict__ counts,
                                                 const float* __restrict__ dl, float dl_min) {
    const int f = blockIdx.y, b = blockIdx.x, nb = gridDim.x;
    __shared__ int red[256];
    __shared__ int wsum[4];
    // exclusive prefix of the preceding blocks' counts
    int part = 0;
    for (int i = threadIdx.x; i < b; i += 256) part += block_counts[(int64_t)f * nb + i];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const int base = red[0];
    const int cell = b * 256 + threadIdx.x;
    bool pass = false;
    float p = 0.f;
    const float* h = head + ((int64_t)f * cells + (cell < cells ? cell : 0)) * 6;
    if (cell < cells && (!dl || dl[(int64_t)f * cells + cell] >= dl_min)) {
        p = softmax2_face(h[0], h[1]);
        pass = p >= thr;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long m = __ballot(pass);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < wv; ++i) woff += wsum[i];
    const int rank = base + woff + __popcll(m & ((1ull << lane) - 1ull));
    if (pass && rank < cap) {
        const int cy = cell / wc, cx = cell - cy * wc;
        const int64_t o = (int64_t)f * cap + rank;
        float4 bx;
        bx.x = floorf((2.0f * (float)cx + 1.0f) / scale);
        bx.y = floorf((2.0f * (float)cy + 1.0f) / scale);
        bx.z = floorf((2.0f * (float)cx + 12.0f) / scale);
        bx.w = floorf((2.0f * (float)cy + 12.0f) / scale);
        *reinterpret_cast<float4*>(boxes + o * 4) = bx;
        scores[o] = p;
        *reinterpret_cast<float4*>(regs + o * 4) = make_float4(h[2], h[3], h[4], h[5]);
    }
    if (b == nb - 1 && threadIdx.x == 0) {
        int total = base + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        counts[f] = total < cap ? total : cap;
    }
}

// All pyramid levels of a batch in one launch each (fr_pnet_finish_levels): a block finds its level in a table in the kernel
// arguments and runs the per-level kernel's body - the same cells, the same ordered compaction.
#define PCAND_MAXL 16
struct PCandLevels {
    const float* head[PCAND_MAXL]; const float* dl[PCAND_MAXL];
    float* boxes[PCAND_MAXL]; float* scores[PCAND_MAXL]; float* regs[PCAND_MAXL]; int32_t* counts[PCAND_MAXL]; int32_t* block_counts[PCAND_MAXL];
    int cells[PCAND_MAXL], wc[PCAND_MAXL], first[PCAND_MAXL + 1];
    float scale[PCAND_MAXL];
    int nlevels;
};
__global__ __launch_bounds__(256) void pnet_count_levels(PCandLevels t, float thr, float dl_min) {
    int l = 0;
    while (l + 1 < t.nlevels && (int)blockIdx.x >= t.first[l + 1]) ++l;
    const int f = blockIdx.y, b = (int)blockIdx.x - t.first[l], nb = t.first[l + 1] - t.first[l];
    const int cells = t.cells[l];
    const int cell = b * 256 + threadIdx.x;
    bool pass = false;
    if (cell < cells && t.dl[l][(int64_t)f * cells + cell] >= dl_min) {
        const float* h = t.head[l] + ((int64_t)f * cells + cell) * 6;
        pass = softmax2_face(h[0], h[1]) >= thr;
    }
    __shared__ int wsum[4];
    unsigned long long m = __ballot(pass);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) t.block_counts[l][(int64_t)f * nb + b] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void pnet_emit_levels(PCandLevels t, float thr, int cap, float dl_min) {
    int l = 0;
    while (l + 1 < t.nlevels && (int)blockIdx.x >= t.first[l + 1]) ++l;
    const int f = blockIdx.y, b = (int)blockIdx.x - t.first[l], nb = t.first[l + 1] - t.first[l];
    const int cells = t.cells[l], wc = t.wc[l];
    const float scale = t.scale[l];
    const int32_t* block_counts = t.block_counts[l];
    __shared__ int red[256];
    __shared__ int wsum[4];
    int part = 0;
    for (int i = threadIdx.x; i < b; i += 256) part += block_counts[(int64_t)f * nb + i];
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const int base = red[0];
    const int cell = b * 256 + threadIdx.x;
    bool pass = false;
    float p = 0.f;
    const float* h = t.head[l] + ((int64_t)f * cells + (cell < cells ? cell : 0)) * 6;
    if (cell < cells && t.dl[l][(int64_t)f * cells + cell] >= dl_min) {
        p = softmax2_face(h[0], h[1]);
        pass = p >= thr;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long m = __ballot(pass);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < wv; ++i) woff += wsum[i];
    const int rank = base + woff + __popcll(m & ((1ull << lane) - 1ull));
    if (pass && rank < cap) {
        const int cy = cell / wc, cx = cell - cy * wc;
        const int64_t o = (int64_t)f * cap + rank;
        float4 bx;
        bx.x = floorf((2.0f * (float)cx + 1.0f) / scale);
        bx.y = floorf((2.0f * (float)cy + 1.0f) / scale);
        bx.z = floorf((2.0f * (float)cx + 12.0f) / scale);
        bx.w = floorf((2.0f * (float)cy + 12.0f) / scale);
        *reinterpret_cast<float4*>(t.boxes[l] + o * 4) = bx;
        t.scores[l][o] = p;
        *reinterpret_cast<float4*>(t.regs[l] + o * 4) = make_float4(h[2], h[3], h[4], h[5]);
    }
    if (b == nb - 1 && threadIdx.x == 0) {
        int total = base + wsum[0] + wsum[1] + wsum[2] + wsum[3];
        t.counts[l][f] = total < cap ? total : cap;
    }
}

int fr_pnet_candidates_levels_launch(const fr_pnet_level* lv, int nlevels, int nframes, float thr, int cap, float dl_min, hipStream_t s) {
    PCandLevels t;
    t.nlevels = nlevels;
    int total = 0;
    for (int l = 0; l < nlevels; ++l) {
        const int hc = lv[l].H1 - 4, wc = lv[l].W1 - 4;
        t.head[l] = lv[l].head; t.dl[l] = reinterpret_cast<const float*>(lv[l].workspace);
        t.boxes[l] = lv[l].boxes; t.scores[l] = lv[l].scores; t.regs[l] = lv[l].regs; t.counts[l] = lv[l].counts;
        t.block_counts[l] = lv[l].block_counts;
        t.cells[l] = hc * wc; t.wc[l] = wc; t.scale[l] = lv[l].scale;
        t.first[l] = total;
        total += fr_cdiv(hc * wc, 256);
    }
    t.first[nlevels] = total;
    dim3 grid(total, nframes);
    pnet_count_levels<<<grid, 256, 0, s>>>(t, thr, dl_min);
    FR_CHECK_LAUNCH("pnet_count_levels");
    pnet_emit_levels<<<grid, 256, 0, s>>>(t, thr, cap, dl_min);
    FR_CHECK_LAUNCH("pnet_emit_levels");
    return FR_OK;
}

extern "C" int fr_pnet_candidates(const float* head, int nframes, int hc, int wc, float scale, float thr, int cap,
                                  float* boxes, float* scores, float* regs, int32_t* counts, int32_t* block_counts,
                                  float* prob_out, const float* dl, float dl_min, fr_stream_t stream) {
    FR_REQUIRE(head && boxes && scores && regs && counts && block_counts, "fr_pnet_candidates: null pointer");
    FR_REQUIRE(nframes > 0 && hc > 0 && wc > 0 && cap > 0 && scale > 0.f, "fr_pnet_candidates: bad argument");
    const int cells = hc * wc;
    dim3 grid(fr_cdiv(cells, 256), nframes);
    hipStream_t s = fr_stream(stream);
    pnet_count<<<grid, 256, 0, s>>>(head, cells, thr, block_counts, prob_out, dl, dl_min);
    FR_CHECK_LAUNCH("pnet_count");
    pnet_emit<<<grid, 256, 0, s>>>(head, cells, wc, scale, thr, cap, block_counts, boxes, scores, regs, counts, dl, dl_min);
    FR_CHECK_LAUNCH("pnet_emit");
    return FR_OK;
}

// ------------------------------------------------------------------ box refinement (in place)
// mode 0: stage-1 regression (w = x2-x1) then square; mode 1: bbreg (w = x2-x1+1) then square;
// mode 2: bbreg only.  regs are the first 4 floats of each aux row (stride naux).
__global__ void box_refine(float* __restrict__ boxes, const float* __restrict__ aux, int naux,
                           const int32_t* __restrict__ counts, int L, int cap, int mode) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L * cap) return;
    int l = t / cap, i = t - l * cap;
    if (i >= counts[l]) return;
    float4 b = *reinterpret_cast<float4*>(boxes + (int64_t)t * 4);
    const float* r = aux + (int64_t)t * naux;
    float add = mode == 0 ? 0.0f : 1.0f;
    float w = b.z - b.x + add, h = b.w - b.y + add;
    float x1 = b.x + r[0] * w, y1 = b.y + r[1] * h, x2 = b.z + r[2] * w, y2 = b.w + r[3] * h;
    if (mode != 2) {   // rerec
        float hh = y2 - y1, ww = x2 - x1;
        float l2 = fmaxf(ww, hh);
        x1 = x1 + ww * 0.5f - l2 * 0.5f;
        y1 = y1 + hh * 0.5f - l2 * 0.5f;
        x2 = x1 + l2;
        y2 = y1 + l2;
    }
    *reinterpret_cast<float4*>(boxes + (int64_t)t * 4) = make_float4(x1, y1, x2, y2);
}

extern "C" int fr_box_refine(float* boxes, const float* aux, int naux, const int32_t* counts, int L, int cap,
                             int mode, fr_stream_t stream) {
    FR_REQUIRE(boxes && aux && counts && L > 0 && cap > 0 && naux >= 4 && mode >= 0 && mode <= 2,
               "fr_box_refine: bad argument");
    box_refine<<<fr_cdiv((int64_t)L * cap, 256), 256, 0, fr_stream(stream)>>>(boxes, aux, naux, counts, L, cap, mode);
    FR_CHECK_LAUNCH("box_refine");
    return FR_OK;
}

// ------------------------------------------------------------------ crop + resize + normalise
// Zero-padded crop of the (1-based inclusive) box trunc(b), bilinear to size x size, RGB normalised,
// written as 4-channel pixels (RGB0) so that the first R/O-Net conv reads 16-byte pixels.
// One block per candidate slot; invalid slots (>= count, or empty boxes) are zero-filled.
typedef unsigned long long u64_unaligned_t __attribute__((aligned(1)));
__global__ __launch_bounds__(256) void crop_resize_norm(const uint8_t* __restrict__ frames, int H, int W,
                                                        const float* __restrict__ boxes,
                                                        const int32_t* __restrict__ counts, int cap, int size,
                                                        float* __restrict__ out) {
    const int slot = blockIdx.x;
    const int f = slot / cap, i = slot - f * cap;
    float* o = out + (int64_t)slot * size * size * 4;
    const float4 b = *reinterpret_cast<const float4*>(boxes + (int64_t)slot * 4);
    const int x1 = (int)truncf(b.x), y1 = (int)truncf(b.y), x2 = (int)truncf(b.z), y2 = (int)truncf(b.w);
    const int tw = x2 - x1 + 1, th = y2 - y1 + 1;
    const bool valid = i < counts[f] && tw > 0 && th > 0;
    const uint8_t* fr = frames + (int64_t)f * H * W * 3;
    const float ry = (float)th / (float)size, rx = (float)tw / (float)size;
    const int lim = f == (int)(gridDim.x / cap) - 1 ? H * W * 3 - 8 : 0x7fffffff;
    for (int t = threadIdx.x; t < size * size; t += 256) {
        const int oy = t / size, ox = t - oy * size;
        float v[3] = {0.f, 0.f, 0.f};
        if (valid) {
            Lerp ly = lerp_coord(oy, ry, th), lx = lerp_coord(ox, rx, tw);
            const int ys[2] = {y1 - 1 + ly.i0, y1 - 1 + ly.i1};
            const int xs[2] = {x1 - 1 + lx.i0, x1 - 1 + lx.i1};
            float p[2][2][3];
            // both corners of a source row are 6 adjacent bytes (BGR BGR): ONE unaligned 8-byte load per row when the
            // two columns are neighbours inside the frame (all but the crop's clamped last column and crops that
            // stick out of the frame); in the LAST frame the load is pulled back so it never runs past the buffer
            const bool pair = xs[1] == xs[0] + 1 && xs[0] >= 0 && xs[1] < W;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const bool row_in = ys[a] >= 0 && ys[a] < H;
                if (pair && row_in) {
                    const int off = (ys[a] * W + xs[0]) * 3;
                    const int c = min(off, lim);
                    const unsigned long long q = *reinterpret_cast<const u64_unaligned_t*>(fr + c) >> ((off - c) * 8);
                    const unsigned lo = (unsigned)q, hi = (unsigned)(q >> 32);
                    p[a][0][0] = (float)((lo >> 16) & 0xff); p[a][0][1] = (float)((lo >> 8) & 0xff); p[a][0][2] = (float)(lo & 0xff);
                    p[a][1][0] = (float)((hi >> 8) & 0xff); p[a][1][1] = (float)(hi & 0xff); p[a][1][2] = (float)(lo >> 24);
                } else {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const bool in = row_in && xs[c] >= 0 && xs[c] < W;
                        const uint8_t* px = fr + ((int64_t)(in ? ys[a] : 0) * W + (in ? xs[c] : 0)) * 3;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) p[a][c][ch] = in ? (float)px[2 - ch] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                v[ch] = (bilerp(p[0][0][ch], p[0][1][ch], p[1][0][ch], p[1][1][ch], lx.w, ly.w) - 127.5f) * 0.0078125f;
        }
        *reinterpret_cast<float4*>(o + t * 4) = make_float4(v[0], v[1], v[2], 0.f);     // 4th channel = 0 (16-B pixels)
    }
}

extern "C" int fr_crop_resize_norm(const uint8_t* frames, int nframes, int H, int W, const float* boxes,
                                   const int32_t* counts, int cap, int size, float* out, fr_stream_t stream) {
    FR_REQUIRE(frames && boxes && counts && out && nframes > 0 && cap > 0 && size > 0, "fr_crop_resize_norm: bad argument");
    crop_resize_norm<<<nframes * cap, 256, 0, fr_stream(stream)>>>(frames, H, W, boxes, counts, cap, size, out);
    FR_CHECK_LAUNCH("crop_resize_norm");
    return FR_OK;
}

// ------------------------------------------------------------------ R/O-Net stage select
// head: f32 [L*cap, nh] = (logit0, logit1, reg0..3[, lm0..9]).  Keeps slots with softmax face
// prob > thr in slot order (one block per list), emitting trunc(box), score and aux =
// (reg0..3[, landmarks (x1,y1)..(x5,y5) mapped into the frame]).
__global__ __launch_bounds__(256) void stage_select(const float* __restrict__ boxes, const float* __restrict__ head,
                                                    int nh, const int32_t* __restrict__ counts, int cap, float thr,
                                                    float* __restrict__ boxes_out, float* __restrict__ scores_out,
                                                    float* __restrict__ aux_out, int naux,
                                                    int32_t* __restrict__ counts_out, float* __restrict__ prob_out) {
    const int l = blockIdx.x;
    const int n = counts[l];
    __shared__ int wsum[4];
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < cap; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const int64_t slot = (int64_t)l * cap + i;
        bool pass = false;
        float p = 0.f;
        float4 bt = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            const float4 b = *reinterpret_cast<const float4*>(boxes + slot * 4);
            bt = make_float4(truncf(b.x), truncf(b.y), truncf(b.z), truncf(b.w));
            const float* h = head + slot * nh;
            const bool nonempty = (bt.z - bt.x + 1.0f) > 0.f && (bt.w - bt.y + 1.0f) > 0.f;
            p = nonempty ? softmax2_face(h[0], h[1]) : 0.f;
            pass = p > thr;
        }
        if (prob_out && i < cap) prob_out[slot] = p;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        unsigned long long m = __ballot(pass);
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wv; ++k) woff += wsum[k];
        const int rank = base_s + woff + __popcll(m & ((1ull << lane) - 1ull));
        if (pass) {
            const int64_t o = (int64_t)l * cap + rank;
            const float* h = head + slot * nh;
            *reinterpret_cast<float4*>(boxes_out + o * 4) = bt;
            scores_out[o] = p;
            float* a = aux_out + o * naux;
            a[0] = h[2]; a[1] = h[3]; a[2] = h[4]; a[3] = h[5];
            if (naux >= 14) {
                const float w = bt.z - bt.x + 1.0f, hh = bt.w - bt.y + 1.0f;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    a[4 + 2 * k] = w * h[6 + k] + bt.x - 1.0f;
                    a[5 + 2 * k] = hh * h[11 + k] + bt.y - 1.0f;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) counts_out[l] = base_s;
}

extern "C" int fr_stage_select(const float* boxes, const float* head, int nh, const int32_t* counts, int L, int cap,
                               float thr, float* boxes_out, float* scores_out, float* aux_out, int naux,
                               int32_t* counts_out, float* prob_out, fr_stream_t stream) {
    FR_REQUIRE(boxes && head && counts && boxes_out && scores_out && aux_out && counts_out, "fr_stage_select: null pointer");
    FR_REQUIRE(L > 0 && cap > 0 && ((nh == 6 && naux == 4) || (nh == 16 && naux == 14)), "fr_stage_select: bad nh/naux");
    stage_select<<<L, 256, 0, fr_stream(stream)>>>(boxes, head, nh, counts, cap, thr, boxes_out, scores_out, aux_out,
                                                   naux, counts_out, prob_out);
    FR_CHECK_LAUNCH("stage_select");
    return FR_OK;
}

// 5-point alignment: least-squares similarity (closed form; equals Umeyama's SVD solution for
// proper rotations) + bilinear warp to size x size with constant-0 border, uint8 rounding, then
// (x - 127.5) / 127.5, BGR -> RGB, packed NHWC f16 with 8 channels (the stem conv's input).
// Stands in for insightface face_align.norm_crop inside FaceAnalysis.get
// (/root/reference/infrenceServer.py:528); mirrors oracle/align.py (float64 geometry).
#include "common.h"

__constant__ double ARC_DST[5][2] = {{38.2946, 51.6963}, {73.5318, 51.5014}, {56.0252, 71.7366},
                                     {41.5493, 92.3655}, {70.7299, 92.2041}};

typedef unsigned long long u64_unaligned_w __attribute__((aligned(1)));
#define WARP_NB 7
__global__ __launch_bounds__(256) void warp_affine_5pt(const uint8_t* __restrict__ frames, int nframes, int H, int W,
                                                       const float* __restrict__ kps,
                                                       const int32_t* __restrict__ frame_idx,
                                                       const int32_t* __restrict__ count,
                                                       const int32_t* __restrict__ slot_counts, int slot_cap, int size,
                                                       half_t* __restrict__ out, uint8_t* __restrict__ out_u8,
                                                       float* __restrict__ M_out) {
    // WARP_NB blocks per face, one band of rows each: a single face (the single-frame path) was ONE block's 45 us
    const int f = blockIdx.x / WARP_NB, band = blockIdx.x - f * WARP_NB;
    // validity: compact list (f < *count), or fixed per-frame slots (slot f = frame*cap + j valid iff j < counts[frame])
    const bool valid = slot_counts ? ((f % slot_cap) < slot_counts[f / slot_cap]) : (count == nullptr || f < *count);
    __shared__ double inv[6];
    if (threadIdx.x == 0 && valid) {
        const float* k = kps + (int64_t)f * 10;      // valid slots only: their kps are finite
        const double sf = (double)size / 112.0;
        double msx = 0, msy = 0, mdx = 0, mdy = 0;
        for (int i = 0; i < 5; ++i) { msx += k[2 * i]; msy += k[2 * i + 1]; mdx += ARC_DST[i][0] * sf; mdy += ARC_DST[i][1] * sf; }
        msx /= 5; msy /= 5; mdx /= 5; mdy /= 5;
        double num_a = 0, num_b = 0, var = 0;
        for (int i = 0; i < 5; ++i) {
            double sx = k[2 * i] - msx, sy = k[2 * i + 1] - msy;
            double dx = ARC_DST[i][0] * sf - mdx, dy = ARC_DST[i][1] * sf - mdy;
            num_a += sx * dx + sy * dy;
            num_b += sx * dy - sy * dx;
            var += sx * sx + sy * sy;
        }
        const double a = num_a / var, b = num_b / var;
        const double tx = mdx - (a * msx - b * msy), ty = mdy - (b * msx + a * msy);
        if (M_out && band == 0) {
            float* m = M_out + (int64_t)f * 6;
            m[0] = (float)a; m[1] = (float)-b; m[2] = (float)tx; m[3] = (float)b; m[4] = (float)a; m[5] = (float)ty;
        }
        const double det = a * a + b * b;
        inv[0] = a / det; inv[1] = b / det; inv[3] = -b / det; inv[4] = a / det;
        inv[2] = -(inv[0] * tx + inv[1] * ty);
        inv[5] = -(inv[3] * tx + inv[4] * ty);
    }
    __syncthreads();
    const int fidx = valid ? (slot_counts ? f / slot_cap : frame_idx[f]) : 0;
    const uint8_t* fr = frames + (int64_t)fidx * H * W * 3;
    const long long lim = fidx == nframes - 1 ? (long long)H * W * 3 - 8 : (1ll << 62);     // last frame: never read past the buffer
    half_t* o = out + (int64_t)f * size * size * 8;
    const int rows = (size + WARP_NB - 1) / WARP_NB;
    const int t_end = min((band + 1) * rows, size) * size;
    for (int t = band * rows * size + threadIdx.x; t < t_end; t += 256) {
        const int y = t / size, x = t - y * size;
        float rgb[3] = {0.f, 0.f, 0.f};
        unsigned char u[3] = {0, 0, 0};
        if (valid) {
            const double sx = inv[0] * x + inv[1] * y + inv[2];
            const double sy = inv[3] * x + inv[4] * y + inv[5];
            const double fx = floor(sx), fy = floor(sy);
            const double wx = sx - fx, wy = sy - fy;
            const long long x0 = (long long)fx, y0 = (long long)fy;
            // the two columns of a source row are 6 adjacent bytes (BGR BGR): ONE unaligned 8-byte load per row when both
            // lie inside the frame (pulled back at the very end of the last frame's buffer), byte loads otherwise
            double pb[2][2][3];
            const bool pair = x0 >= 0 && x0 + 1 < W;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const long long yy = y0 + a;
                const bool row_in = yy >= 0 && yy < H;
                if (pair && row_in) {
                    const long long off = (yy * W + x0) * 3;
                    const long long c8 = off < lim ? off : lim;
                    const unsigned long long q = *reinterpret_cast<const u64_unaligned_w*>(fr + c8) >> ((off - c8) * 8);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        pb[a][0][c] = (double)(unsigned)((q >> (8 * c)) & 0xff);
                        pb[a][1][c] = (double)(unsigned)((q >> (24 + 8 * c)) & 0xff);
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const long long xx = x0 + b;
                        const bool in = row_in && xx >= 0 && xx < W;
#pragma unroll
                        for (int c = 0; c < 3; ++c) pb[a][b][c] = in ? (double)fr[((int64_t)yy * W + xx) * 3 + c] : 0.0;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double v = (1 - wy) * ((1 - wx) * pb[0][0][c] + wx * pb[0][1][c]) + wy * ((1 - wx) * pb[1][0][c] + wx * pb[1][1][c]);
                double r = floor(v + 0.5);
                r = r < 0 ? 0 : (r > 255 ? 255 : r);
                u[c] = (unsigned char)r;
            }
            // BGR -> RGB, (x - 127.5) / 127.5
            rgb[0] = ((float)u[2] - 127.5f) / 127.5f;
            rgb[1] = ((float)u[1] - 127.5f) / 127.5f;
            rgb[2] = ((float)u[0] - 127.5f) / 127.5f;
        }
        half8 hv = {(half_t)rgb[0], (half_t)rgb[1], (half_t)rgb[2], (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        *reinterpret_cast<half8*>(o + (int64_t)t * 8) = hv;
        if (out_u8) {
            uint8_t* q = out_u8 + ((int64_t)f * size * size + t) * 3;
            q[0] = u[0]; q[1] = u[1]; q[2] = u[2];
        }
    }
}

extern "C" int fr_warp_affine_5pt(const uint8_t* frames, int nframes, int H, int W, const float* kps,
                                  const int32_t* frame_idx, const int32_t* count, int F, int size,
                                  void* out_f16_nhwc8, uint8_t* out_u8_bgr, float* M_out, fr_stream_t stream) {
    if (F <= 0) return FR_OK;
    FR_REQUIRE(frames && kps && frame_idx && out_f16_nhwc8, "fr_warp_affine_5pt: null pointer");
    FR_REQUIRE(nframes > 0 && H > 0 && W > 0 && size > 0, "fr_warp_affine_5pt: bad size");
    warp_affine_5pt<<<F * WARP_NB, 256, 0, fr_stream(stream)>>>(frames, nframes, H, W, kps, frame_idx, count, nullptr, 1, size,
                                                      reinterpret_cast<half_t*>(out_f16_nhwc8), out_u8_bgr, M_out);
    FR_CHECK_LAUNCH("warp_affine_5pt");
    return FR_OK;
}

// Fixed-shape form: faces live in per-frame slots [nframes, cap] with device-side counts (no host sync between
// detection and embedding); slots beyond a frame's count are zero-filled.
extern "C" int fr_warp_affine_5pt_slots(const uint8_t* frames, int nframes, int H, int W, const float* kps,
                                        const int32_t* counts, int cap, int size, void* out_f16_nhwc8,
                                        fr_stream_t stream) {
    FR_REQUIRE(frames && kps && counts && out_f16_nhwc8 && nframes > 0 && cap > 0 && H > 0 && W > 0 && size > 0,
               "fr_warp_affine_5pt_slots: bad argument");
    warp_affine_5pt<<<nframes * cap * WARP_NB, 256, 0, fr_stream(stream)>>>(frames, nframes, H, W, kps, nullptr, nullptr, counts, cap, size,
                                                                  reinterpret_cast<half_t*>(out_f16_nhwc8), nullptr, nullptr);
    FR_CHECK_LAUNCH("warp_affine_5pt");
    return FR_OK;
}

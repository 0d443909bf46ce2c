// Weight preparation for the fp8 body convs (fr_conv_nhwc_f8; BASELINE config C5, reference site
// /root/reference/infrenceServer.py:528): GPTQ rounding of a folded weight matrix to the per-row e4m3 grid.
// Calibration-time only (IResNetHIP.enable_fp8), not on the per-frame path.
#include "common.h"

// nearest OCP e4m3 value of a (ties to even, saturating at 448, subnormal step 2^-9), in double arithmetic
__device__ __forceinline__ double round_e4m3_value(double t) {
    double a = fabs(t);
    if (a > 448.0) a = 448.0;
    int e;
    frexp(a < 0x1p-9 ? 0x1p-9 : a, &e);                // a = m * 2^e, m in [0.5, 1)
    int ex = e - 1; if (ex < -6) ex = -6;
    const double step = ldexp(1.0, ex - 3);
    return copysign(rint(a / step) * step, t);
}

// One workgroup per weight row (output channel): rows are independent in GPTQ.  Column k is rounded, its error
// (scaled by 1 / U[k][k]) is pushed onto the columns still to come along row k of U, the upper Cholesky factor of the
// inverse second-moment matrix of the conv's input patches (Frantar et al. 2022, Algorithm 1 without lazy batching:
// the row lives in LDS, so every update is immediate).
__global__ __launch_bounds__(256) void gptq_round_e4m3(const double* __restrict__ W, const double* __restrict__ U,
                                                       const float* __restrict__ sw, float* __restrict__ Q, int K) {
    extern __shared__ double wrow[];
    const int row = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < K; j += 256) wrow[j] = W[(int64_t)row * K + j];
    const double s = (double)sw[row];
    for (int k = 0; k < K; ++k) {
        __syncthreads();                               // every update of step k-1 has landed
        const double wk = wrow[k];
        const double q = round_e4m3_value(wk / s);
        const double* u = U + (int64_t)k * K;
        const double err = (wk - q * s) / u[k];
        if (tid == 0) Q[(int64_t)row * K + k] = (float)q;
        for (int j = k + 1 + tid; j < K; j += 256) wrow[j] -= err * u[j];
    }
}

extern "C" int fr_gptq_round_e4m3(const double* W, const double* U, const float* sw, float* Q, int rows, int K,
                                  fr_stream_t stream) {
    FR_REQUIRE(W && U && sw && Q && rows > 0 && K > 0 && (size_t)K * 8 <= 64 * 1024, "fr_gptq_round_e4m3: bad argument (rows %d, K %d)", rows, K);
    gptq_round_e4m3<<<rows, 256, (size_t)K * 8, fr_stream(stream)>>>(W, U, sw, Q, K);
    FR_CHECK_LAUNCH("gptq_round_e4m3");
    return FR_OK;
}

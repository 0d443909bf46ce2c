// Sort (descending score, ties by slot = stable) + greedy NMS, one workgroup per list.
// MTCNN post-processing inside FaceAnalysis.get (/root/reference/infrenceServer.py:528); mirrors
// oracle/detect.py nms(): sequential greedy semantics are kept exactly (a box survives iff no
// earlier SURVIVING box overlaps it by more than thr); the suppression sweep of each survivor is
// data-parallel over the workgroup, the survivor scan is a uniform LDS read per step.
//
// A list may be made of `nseg` segments of `seg_cap` slots, each with its own count (used to merge
// the per-pyramid-level lists of one frame without a compaction pass).
#include "common.h"

template <int NMAX, int NMS_T>
__global__ __launch_bounds__(NMS_T) void sort_nms(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                  const float* __restrict__ aux, int naux,
                                                  const int32_t* __restrict__ counts, int nseg, int seg_cap,
                                                  int seg_major, float thr, int mode, int max_keep, float* __restrict__ boxes_out,
                                                  float* __restrict__ scores_out, float* __restrict__ aux_out,
                                                  int32_t* __restrict__ counts_out, int cap_out) {
    __shared__ unsigned long long key[NMAX];
    __shared__ float4 sb[NMAX];
    __shared__ float sarea[NMAX];
    __shared__ unsigned char alive[NMAX];
    __shared__ int keep[NMAX > 1024 ? 1024 : NMAX];
    __shared__ int nvalid_s;
    const int l = blockIdx.x, tid = threadIdx.x;
    const int ntot = nseg * seg_cap;
    int NP = 1;
    while (NP < ntot) NP <<= 1;
    if (tid == 0) nvalid_s = 0;
    __syncthreads();
    const int L = gridDim.x;
    // slot of entry i: segment s = i / seg_cap lives at list index (l*nseg + s) or, seg_major, (s*L + l)
    auto slot_of = [&](int i) -> int64_t {
        int s = i / seg_cap, j = i - s * seg_cap;
        int64_t seg = seg_major ? (int64_t)s * L + l : (int64_t)l * nseg + s;
        return seg * seg_cap + j;
    };
    int myvalid = 0;
    for (int i = tid; i < NP; i += NMS_T) {
        unsigned long long k = 0ull;
        if (i < ntot) {
            int s = i / seg_cap, j = i - s * seg_cap;
            int64_t seg = seg_major ? (int64_t)s * L + l : (int64_t)l * nseg + s;
            if (j < counts[seg]) {
                unsigned sb_ = __float_as_uint(scores[seg * seg_cap + j]);
                k = ((unsigned long long)sb_ << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
                ++myvalid;
            }
        }
        key[i] = k;
    }
    if (myvalid) atomicAdd(&nvalid_s, myvalid);
    __syncthreads();
    // bitonic sort, descending
    for (int k = 2; k <= NP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < NP / 2; t += NMS_T) {
                int i = 2 * t - (t & (j - 1));        // index with bit j clear
                int p = i + j;
                bool desc = (i & k) == 0;
                unsigned long long a = key[i], b = key[p];
                if ((a < b) == desc) { key[i] = b; key[p] = a; }
            }
            __syncthreads();
        }
    }
    const int n = nvalid_s;
    for (int i = tid; i < n; i += NMS_T) {
        unsigned idx = 0xFFFFFFFFu - (unsigned)(key[i] & 0xFFFFFFFFull);
        float4 b = *reinterpret_cast<const float4*>(boxes + slot_of((int)idx) * 4);
        sb[i] = b;
        sarea[i] = (b.z - b.x + 1.0f) * (b.w - b.y + 1.0f);
        alive[i] = 1;
    }
    __syncthreads();
    int nkeep = 0;
    for (int i = 0; i < n && nkeep < max_keep; ++i) {
        if (!alive[i]) continue;                         // uniform: same LDS byte for every thread
        if (tid == 0) keep[nkeep] = i;
        ++nkeep;
        const float4 bi = sb[i];
        const float ai = sarea[i];
        for (int j = i + 1 + tid; j < n; j += NMS_T) {
            if (!alive[j]) continue;
            const float4 bj = sb[j];
            float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
            float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
            float w = fmaxf(0.0f, xx2 - xx1 + 1.0f), h = fmaxf(0.0f, yy2 - yy1 + 1.0f);
            float inter = w * h;
            float o = mode == 1 ? inter / fminf(ai, sarea[j]) : inter / (ai + sarea[j] - inter);
            if (o > thr) alive[j] = 0;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int k = tid; k < nkeep; k += NMS_T) {
        const int i = keep[k];
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(key[i] & 0xFFFFFFFFull);
        const int64_t o = (int64_t)l * cap_out + k;
        *reinterpret_cast<float4*>(boxes_out + o * 4) = sb[i];
        scores_out[o] = __uint_as_float((unsigned)(key[i] >> 32));
        for (int a = 0; a < naux; ++a) aux_out[o * naux + a] = aux[slot_of((int)idx) * naux + a];
    }
    if (tid == 0) counts_out[l] = nkeep;
}

extern "C" int fr_sort_nms(const float* boxes, const float* scores, const float* aux, int naux, const int32_t* counts,
                           int L, int nseg, int seg_cap, int seg_major, float thr, int mode, int max_keep, float* boxes_out,
                           float* scores_out, float* aux_out, int32_t* counts_out, int cap_out, fr_stream_t stream) {
    FR_REQUIRE(boxes && scores && counts && boxes_out && scores_out && counts_out, "fr_sort_nms: null pointer");
    FR_REQUIRE(naux == 0 || (aux && aux_out), "fr_sort_nms: aux pointers missing");
    FR_REQUIRE(L > 0 && nseg > 0 && seg_cap > 0 && max_keep > 0 && max_keep <= cap_out && max_keep <= 1024,
               "fr_sort_nms: bad sizes (max_keep <= min(cap_out, 1024))");
    FR_REQUIRE(mode == 0 || mode == 1, "fr_sort_nms: mode must be 0 (union) or 1 (min)");
    const int ntot = nseg * seg_cap;
    FR_REQUIRE(ntot <= 4096, "fr_sort_nms: list capacity %d exceeds 4096", ntot);
    hipStream_t s = fr_stream(stream);
    if (ntot <= 512)
        sort_nms<512, 512><<<L, 512, 0, s>>>(boxes, scores, aux, naux, counts, nseg, seg_cap, seg_major, thr, mode, max_keep,
                                          boxes_out, scores_out, aux_out, counts_out, cap_out);
    else if (ntot <= 2048)
        sort_nms<2048, 1024><<<L, 1024, 0, s>>>(boxes, scores, aux, naux, counts, nseg, seg_cap, seg_major, thr, mode, max_keep,
                                           boxes_out, scores_out, aux_out, counts_out, cap_out);
    else
        sort_nms<4096, 1024><<<L, 1024, 0, s>>>(boxes, scores, aux, naux, counts, nseg, seg_cap, seg_major, thr, mode, max_keep,
                                           boxes_out, scores_out, aux_out, counts_out, cap_out);
    FR_CHECK_LAUNCH("sort_nms");
    return FR_OK;
}

// Sort (descending score, ties by slot = stable) + greedy NMS, one workgroup per list.
// MTCNN post-processing inside FaceAnalysis.get (/root/reference/infrenceServer.py:528); mirrors
// oracle/detect.py nms(): sequential greedy semantics are kept exactly (a box survives iff no
// earlier SURVIVING box overlaps it by more than thr); candidates are resolved in chunks of 64 (overlap
// bits by ballots, the greedy pass over a chunk by one wave in registers) and a chunk's survivors suppress
// the rest of the list in one data-parallel sweep.
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
    // Keys of the VALID entries only, compacted (the order they land in does not matter: a key carries its entry's
    // index as the tie-break), then a bitonic sort of the next power of two above their number - not of the lists'
    // capacity: a single frame's merged per-level lists (20 480 slots, a few hundred entries) sorted 2 048 + 4 096 keys in
    // 51 + 77 us.  One LDS atomic per wave and pass.
    // (the passes' loads first - segment counts, then scores: two round trips for the whole list instead of two per pass)
    constexpr int NPASS = NMAX / NMS_T;
    int cnt[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const int i = q * NMS_T + tid;
        cnt[q] = 0;
        if (i < ntot) {
            const int s = i / seg_cap;
            cnt[q] = counts[seg_major ? (int64_t)s * L + l : (int64_t)l * nseg + s];
        }
    }
    unsigned long long kk[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const int i = q * NMS_T + tid;
        kk[q] = 0ull;
        if (i < ntot) {
            const int s = i / seg_cap, j = i - s * seg_cap;
            if (j < cnt[q]) {
                const int64_t seg = seg_major ? (int64_t)s * L + l : (int64_t)l * nseg + s;
                kk[q] = ((unsigned long long)__float_as_uint(scores[seg * seg_cap + j]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        if (q * NMS_T >= NP) break;                          // uniform: the list's capacity rounds up to fewer passes
        const unsigned long long k = kk[q];
        // a valid key is never 0: its low word is 0xFFFFFFFF - i with i < 4096
        const unsigned long long bal = __ballot(k != 0ull);
        if (bal) {
            const int ln = tid & 63, first = (int)__builtin_ctzll(bal);
            int base = 0;
            if (ln == first) base = atomicAdd(&nvalid_s, (int)__builtin_popcountll(bal));
            base = __shfl(base, first, 64);
            if (k != 0ull) key[base + (int)__builtin_popcountll(bal & ((1ull << ln) - 1ull))] = k;
        }
    }
    __syncthreads();
    const int nv = nvalid_s;
    int NS = 1;
    while (NS < nv) NS <<= 1;
    for (int i = nv + tid; i < NS; i += NMS_T) key[i] = 0ull;
    __syncthreads();
    // bitonic sort, descending
    for (int k = 2; k <= NS; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < NS / 2; t += NMS_T) {
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
    // Greedy NMS in CHUNKS of up to 64 candidates (exactly the sequential semantics): the next 64 entries that are still
    // alive, in sorted order, are resolved among themselves - their 64 x 64 overlap bits by all waves (one ballot per
    // row), the greedy pass over those bits by one wave in registers - and then ALL the chunk's survivors suppress the
    // entries behind the chunk in ONE data-parallel sweep (a later entry stops at its first hit).  One survivor per
    // pass (the first version) paid a barrier and a walk over every dead entry per survivor: a full 2 048-entry list
    // with 256 survivors took 268 us, ~2 000 cycles per survivor.
    constexpr int NW = NMS_T / 64;
    __shared__ int wave_cnt[NW];
    __shared__ int chunk_idx[64];
    __shared__ unsigned long long cmask[64];
    __shared__ float4 surv_b[64];
    __shared__ float surv_a[64];
    __shared__ int s_pos, s_ns;
    const int lane = tid & 63, wave = tid >> 6;
    auto overlaps = [&](const float4& bi, float ai, const float4& bj, float aj) {
        float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
        float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
        float w = fmaxf(0.0f, xx2 - xx1 + 1.0f), h = fmaxf(0.0f, yy2 - yy1 + 1.0f);
        float inter = w * h;
        float o = mode == 1 ? inter / fminf(ai, aj) : inter / (ai + aj - inter);
        return o > thr;
    };
    int nkeep = 0, pos = 0;
    while (nkeep < max_keep && pos < n) {
        // 1. the next (up to) 64 alive entries of the window [pos, pos + NMS_T), in order
        const int j0 = pos + tid;
        const bool a0 = j0 < n && alive[j0];
        const unsigned long long bal = __ballot(a0);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const int c = wave_cnt[w]; if (w < wave) base += c; total += c; }
        const int rank = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (a0 && rank < 64) chunk_idx[rank] = j0;
        if (total >= 64) { if (a0 && rank == 63) s_pos = j0 + 1; }
        else if (tid == 0) s_pos = min(pos + NMS_T, n);
        const int m = min(total, 64);
        __syncthreads();
        pos = s_pos;
        if (m == 0) continue;                              // nothing alive in this window (uniform)
        // 2. overlap bits inside the chunk: row a = bits of the later entries b > a that a would suppress
        float4 cb = make_float4(0.f, 0.f, 0.f, 0.f); float ca = 1.f;
        if (lane < m) { const int ci = chunk_idx[lane]; cb = sb[ci]; ca = sarea[ci]; }
        for (int a = wave; a < m; a += NW) {
            const int ci = chunk_idx[a];
            const float4 ba = sb[ci]; const float aa = sarea[ci];
            const unsigned long long row = __ballot(lane > a && lane < m && overlaps(ba, aa, cb, ca));
            if (lane == 0) cmask[a] = row;
        }
        __syncthreads();
        // 3. greedy pass over the chunk, one wave, bits in registers
        if (wave == 0) {
            const unsigned long long mine = lane < m ? cmask[lane] : 0ull;
            const unsigned lo = (unsigned)mine, hi = (unsigned)(mine >> 32);
            unsigned long long removed = 0ull, kept = 0ull;
            int cnt = nkeep;
            for (int a = 0; a < m && cnt < max_keep; ++a) {
                if (!((removed >> a) & 1ull)) {
                    kept |= 1ull << a;
                    removed |= ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, a) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)lo, a);
                    ++cnt;
                }
            }
            if (lane == 0) s_ns = cnt - nkeep;
            if (lane < m && ((kept >> lane) & 1ull)) {
                const int k = __popcll(kept & ((1ull << lane) - 1ull));
                const int ci = chunk_idx[lane];
                keep[nkeep + k] = ci;
                surv_b[k] = cb; surv_a[k] = ca;
            }
        }
        __syncthreads();
        const int ns = s_ns;
        nkeep += ns;
        if (nkeep >= max_keep) break;
        // 4. the chunk's survivors against everything behind the chunk (entries before `pos` are never looked at again)
        for (int j = pos + tid; j < n; j += NMS_T) {
            if (!alive[j]) continue;
            const float4 bj = sb[j]; const float aj = sarea[j];
            for (int k = 0; k < ns; ++k)
                if (overlaps(surv_b[k], surv_a[k], bj, aj)) { alive[j] = 0; break; }
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

// Gallery match: re-normalise, linear scan with strict-'>' first-max, decision.
// Replaces the Python loop at /root/reference/infrenceServer.py:530-552.
//
// f32 scan: scores are computed on the f32 MFMA (v_mfma_f32_32x32x2_f32 is an exact
// k-ordered fmaf chain, so scores are plain IEEE f32 dot products).  Orientation:
// A = 32 gallery rows, B = 32 queries, so a lane owns ONE query (column) and 16 gallery
// rows (registers): the running (max, first-argmax) is lane-local, no cross-lane work
// inside the scan.  HBM-bound: N*D*4 bytes per pass per 32-query group.
#include "common.h"

#define GD 512          // embedding dim
#define QPAD 516        // LDS row stride (floats) for the query tile: breaks the 2 KB bank stride
#define QG 32           // queries per group (MFMA N)
#ifndef FR_TOPK
#define FR_TOPK 4
#endif

struct BestPair { float s; int64_t i; };

// query slot q belongs to segment q / seg_len and is real iff its position in the segment < seg_counts[segment]
__device__ __forceinline__ bool slot_valid(const int32_t* seg_counts, int seg_len, int q) {
    const int seg = q / seg_len;
    return q - seg * seg_len < seg_counts[seg];
}
__device__ __forceinline__ bool group_has_valid(const int32_t* seg_counts, int seg_len, int qa, int qb) {
    // [qa, qb) spans at most a few segments; a segment contributes iff its first slot inside the range is real
    for (int q = qa; q < qb;) {
        if (slot_valid(seg_counts, seg_len, q)) return true;
        q = (q / seg_len + 1) * seg_len;
    }
    return false;
}

__device__ __forceinline__ void take_better(float& bs, int64_t& bi, float s, int64_t i) {
    // max score; lowest index on exact ties (== first maximum in row order)
    if (s > bs || (s == bs && i < bi && i >= 0)) { bs = s; bi = i; }
}

// VIEW: row r of the scanned gallery is storage slot view[r] of G (a per-company view of one device-resident
// slab, no copy of the rows); the winner index is the VIEW position, so the order/tie rule is the view's.
template <bool VIEW>
__global__ __launch_bounds__(256) void gallery_scan_f32(const float* __restrict__ Q, const float* __restrict__ G,
                                                        const int64_t* __restrict__ view,
                                                        int F, int64_t N, float* __restrict__ ws_score,
                                                        int64_t* __restrict__ ws_idx,
                                                        const int32_t* __restrict__ seg_counts, int seg_len) {
    __shared__ __attribute__((aligned(16))) float qs[QG * QPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.y * QG;
    // query slots beyond their segment's count are padding (sharded match: SURVEY.md 8(e)); a group made of
    // padding only does no work (the reduce reports (-1, -1) for every padding slot)
    if (seg_counts && !group_has_valid(seg_counts, seg_len, q0, min(q0 + QG, F))) return;
    // stage the query group (zero rows beyond F)
    for (int e = tid; e < QG * (GD / 4); e += 256) {
        int r = e / (GD / 4), c = e % (GD / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + r < F) v = *reinterpret_cast<const float4*>(Q + (int64_t)(q0 + r) * GD + c * 4);
        *reinterpret_cast<float4*>(&qs[r * QPAD + c * 4]) = v;
    }
    __syncthreads();
    const int r = lane & 31, h = lane >> 5;
    float best = -INFINITY;
    int64_t besti = -1;
    const int64_t ntiles = (N + 31) / 32;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
        const int64_t row = t * 32 + r;
        const bool ok = row < N;
        const int64_t slot = VIEW ? (ok ? view[row] : 0) : (ok ? row : 0);
        const float* gp = G + slot * GD + 4 * h;
        const float* qp = &qs[r * QPAD + 4 * h];
        float16v acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int kk = 0; kk < GD / 8; ++kk) {
            float4 a = *reinterpret_cast<const float4*>(gp + kk * 8);
            float4 b = *reinterpret_cast<const float4*>(qp + kk * 8);
            if (!ok) a = make_float4(0.f, 0.f, 0.f, 0.f);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        // acc[reg] = score(gallery row t*32 + (reg&3) + 8*(reg>>2) + 4*h, query q0 + r)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int64_t gi = t * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            float s = acc[reg];
            if (gi < N && s > best) { best = s; besti = gi; }   // rows ascend within a lane
        }
    }
    // merge the two half-waves that hold the same query, then the 4 waves through LDS
    {
        float os = __shfl_xor(best, 32, 64);
        int64_t oi = __shfl_xor(besti, 32, 64);
        if (oi >= 0) take_better(best, besti, os, oi);
    }
    __syncthreads();
    float* ls = qs;                                   // reuse LDS
    int64_t* li = reinterpret_cast<int64_t*>(qs + 256);
    if (h == 0) { ls[wave * 32 + r] = best; li[wave * 32 + r] = besti; }
    __syncthreads();
    if (tid < 32) {
        float bs = ls[tid]; int64_t bi = li[tid];
        for (int w = 1; w < 4; ++w) if (li[w * 32 + tid] >= 0) take_better(bs, bi, ls[w * 32 + tid], li[w * 32 + tid]);
        if (q0 + tid < F) {
            ws_score[(int64_t)blockIdx.x * F + q0 + tid] = bs;
            ws_idx[(int64_t)blockIdx.x * F + q0 + tid] = bi;
        }
    }
}

__global__ void gallery_reduce(const float* __restrict__ ws_score, const int64_t* __restrict__ ws_idx, int nblk,
                               int F, int64_t row_offset, int64_t* __restrict__ out_idx,
                               float* __restrict__ out_score, const int32_t* __restrict__ seg_counts, int seg_len) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    if (seg_counts && !slot_valid(seg_counts, seg_len, f)) { out_idx[f] = -1; out_score[f] = -1.0f; return; }
    float bs = -INFINITY; int64_t bi = -1;
    for (int b = 0; b < nblk; ++b) {
        int64_t i = ws_idx[(int64_t)b * F + f];
        if (i >= 0) take_better(bs, bi, ws_score[(int64_t)b * F + f], i);
    }
    // reference: best_score starts at -1 and only a strictly larger score replaces it
    if (bi < 0 || !(bs > -1.0f)) { out_idx[f] = -1; out_score[f] = -1.0f; }
    else { out_idx[f] = bi + row_offset; out_score[f] = bs; }
}

static int scan_blocks(int64_t N) {
    int64_t tiles = (N + 31) / 32;
    int64_t b = (tiles + 3) / 4;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (int)b;
}

extern "C" size_t fr_gallery_match_workspace(int F, int64_t N) {
    size_t per = (size_t)scan_blocks(N) * (size_t)(F > 0 ? F : 1);
    return per * (sizeof(float) + sizeof(int64_t)) + 512;
}

static int gallery_match_launch(const char* who, const float* Q, const float* G, const int64_t* view, int F, int64_t N,
                                int D, int64_t row_offset, int64_t* out_idx, float* out_score, void* workspace,
                                size_t workspace_bytes, const int32_t* seg_counts, int seg_len, fr_stream_t stream) {
    FR_REQUIRE(!seg_counts || (seg_len > 0 && F % seg_len == 0), "%s: seg_len must divide F", who);
    FR_REQUIRE(D == GD, "%s: D must be %d (got %d)", who, GD, D);
    FR_REQUIRE(F >= 0 && N >= 0, "%s: negative size", who);
    if (F == 0) return FR_OK;
    FR_REQUIRE(Q && out_idx && out_score && (G || N == 0), "%s: null pointer", who);
    FR_REQUIRE(workspace && workspace_bytes >= fr_gallery_match_workspace(F, N),
               "%s: workspace too small (%zu < %zu)", who, workspace_bytes, fr_gallery_match_workspace(F, N));
    const int nblk = scan_blocks(N);
    float* ws_score = reinterpret_cast<float*>(workspace);
    size_t off = ((size_t)nblk * F * sizeof(float) + 255) & ~(size_t)255;
    int64_t* ws_idx = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(workspace) + off);
    hipStream_t s = fr_stream(stream);
    dim3 grid(nblk, (F + QG - 1) / QG);
    if (view) gallery_scan_f32<true><<<grid, 256, 0, s>>>(Q, G, view, F, N, ws_score, ws_idx, seg_counts, seg_len);
    else gallery_scan_f32<false><<<grid, 256, 0, s>>>(Q, G, nullptr, F, N, ws_score, ws_idx, seg_counts, seg_len);
    FR_CHECK_LAUNCH("gallery_scan_f32");
    gallery_reduce<<<fr_cdiv(F, 64), 64, 0, s>>>(ws_score, ws_idx, nblk, F, row_offset, out_idx, out_score, seg_counts, seg_len);
    FR_CHECK_LAUNCH("gallery_reduce");
    return FR_OK;
}

extern "C" int fr_gallery_match_f32(const float* Q, const float* G, int F, int64_t N, int D, int64_t row_offset,
                                    int64_t* out_idx, float* out_score, void* workspace, size_t workspace_bytes,
                                    const int32_t* seg_counts, int seg_len, fr_stream_t stream) {
    return gallery_match_launch("fr_gallery_match_f32", Q, G, nullptr, F, N, D, row_offset, out_idx, out_score,
                                workspace, workspace_bytes, seg_counts, seg_len, stream);
}

extern "C" int fr_gallery_match_view_f32(const float* Q, const float* G, const int64_t* view, int F, int64_t Nview,
                                         int D, int64_t* out_idx, float* out_score, void* workspace,
                                         size_t workspace_bytes, fr_stream_t stream) {
    FR_REQUIRE(view || Nview == 0, "fr_gallery_match_view_f32: null view");
    // Nview == 0: the scan kernel sees no tiles and the reduce writes (-1, -1)
    return gallery_match_launch("fr_gallery_match_view_f32", Q, G, Nview ? view : nullptr, F, Nview, D, 0, out_idx,
                                out_score, workspace, workspace_bytes, nullptr, 0, stream);
}

// ---------------------------------------------------------------- in-place gallery row update
// one wave per row: G[slots[i]] = rows[i] (optionally / ||rows[i]||, the ingest's normalise)
__global__ void gallery_update_rows(float* __restrict__ G, const int64_t* __restrict__ slots,
                                    const float* __restrict__ rows, int n, int normalise) {
    const int i = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const float* p = rows + (int64_t)i * GD;
    float4 v0 = *reinterpret_cast<const float4*>(p + lane * 4);
    float4 v1 = *reinterpret_cast<const float4*>(p + 256 + lane * 4);
    if (normalise) {
        float ss = v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
        ss += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
        ss = wave_sum(ss);
        const float nrm = sqrtf(ss);
        v0.x /= nrm; v0.y /= nrm; v0.z /= nrm; v0.w /= nrm;
        v1.x /= nrm; v1.y /= nrm; v1.z /= nrm; v1.w /= nrm;
    }
    float* o = G + slots[i] * GD;
    *reinterpret_cast<float4*>(o + lane * 4) = v0;
    *reinterpret_cast<float4*>(o + 256 + lane * 4) = v1;
}

extern "C" int fr_gallery_update_rows_f32(float* G, const int64_t* slots, const float* rows, int n, int D,
                                          int normalise, fr_stream_t stream) {
    FR_REQUIRE(D == GD, "fr_gallery_update_rows_f32: D must be %d (got %d)", GD, D);
    FR_REQUIRE(n >= 0, "fr_gallery_update_rows_f32: negative size");
    if (n == 0) return FR_OK;
    FR_REQUIRE(G && slots && rows, "fr_gallery_update_rows_f32: null pointer");
    gallery_update_rows<<<fr_cdiv(n, 4), 256, 0, fr_stream(stream)>>>(G, slots, rows, n, normalise);
    FR_CHECK_LAUNCH("gallery_update_rows");
    return FR_OK;
}

// ---------------------------------------------------------------- l2norm rows
__global__ void l2norm_rows(const float* __restrict__ x, float* __restrict__ out, int rows, int dim) {
    int row = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (int64_t)row * dim;
    float ss = 0.f;
    for (int c = lane * 4; c < dim; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(p + c);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    float nrm = sqrtf(ss);
    for (int c = lane * 4; c < dim; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(p + c);
        v.x /= nrm; v.y /= nrm; v.z /= nrm; v.w /= nrm;
        *reinterpret_cast<float4*>(out + (int64_t)row * dim + c) = v;
    }
}

extern "C" int fr_l2norm_rows_f32(const float* x, float* out, int rows, int dim, fr_stream_t stream) {
    FR_REQUIRE(rows >= 0 && dim > 0 && dim % 4 == 0, "fr_l2norm_rows_f32: dim must be a positive multiple of 4");
    if (rows == 0) return FR_OK;
    FR_REQUIRE(x && out, "fr_l2norm_rows_f32: null pointer");
    l2norm_rows<<<fr_cdiv(rows, 4), 256, 0, fr_stream(stream)>>>(x, out, rows, dim);
    FR_CHECK_LAUNCH("l2norm_rows");
    return FR_OK;
}

// ---------------------------------------------------------------- decision
__global__ void match_decide(const int64_t* idx, const float* score, int F, float thr, float unknown_thr,
                             int32_t* decision) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    bool has = idx[f] >= 0;
    float s = score[f];
    int d;
    if (has && s >= thr) d = 1;
    else if (s < unknown_thr) d = 0;
    else d = 2;
    decision[f] = d;
}

extern "C" int fr_match_decide(const int64_t* idx, const float* score, int F, float thr, float unknown_thr,
                               int32_t* decision, fr_stream_t stream) {
    if (F <= 0) return FR_OK;
    FR_REQUIRE(idx && score && decision, "fr_match_decide: null pointer");
    match_decide<<<fr_cdiv(F, 256), 256, 0, fr_stream(stream)>>>(idx, score, F, thr, unknown_thr, decision);
    FR_CHECK_LAUNCH("match_decide");
    return FR_OK;
}

// ---------------------------------------------------------------- sharded match: exchange payload + reduce
// SURVEY.md 8(e) step 3.  A candidate travels as three int32 words (score bits, row lo, row hi): the RCCL
// all-gather moves raw bits, no float conversion kernels in the exchange.  The reduce applies the scan's own
// rule over the R shards: maximum score, lowest global row on exact ties; (-1, -1.0) when no shard has a row.
__global__ void match_pack_candidates(const int64_t* __restrict__ idx, const float* __restrict__ score, int n,
                                      int32_t* __restrict__ cand) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = idx[i];
    cand[i * 3 + 0] = __float_as_int(score[i]);
    cand[i * 3 + 1] = (int32_t)(r & 0xffffffffll);
    cand[i * 3 + 2] = (int32_t)(r >> 32);
}

__global__ void match_reduce_shards(const int32_t* __restrict__ cand, int R, int n, int q0, int F,
                                    int64_t* __restrict__ out_idx, float* __restrict__ out_score) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    float bs = -INFINITY; int64_t bi = -1;
    for (int r = 0; r < R; ++r) {
        const int32_t* c = cand + ((int64_t)r * n + q0 + f) * 3;
        const int64_t i = (int64_t)(((uint64_t)(uint32_t)c[2] << 32) | (uint32_t)c[1]);
        if (i >= 0) take_better(bs, bi, __int_as_float(c[0]), i);
    }
    if (bi < 0) { out_idx[f] = -1; out_score[f] = -1.0f; }
    else { out_idx[f] = bi; out_score[f] = bs; }
}

extern "C" int fr_match_pack_candidates(const int64_t* idx, const float* score, int n, int32_t* cand,
                                        fr_stream_t stream) {
    if (n <= 0) return FR_OK;
    FR_REQUIRE(idx && score && cand, "fr_match_pack_candidates: null pointer");
    match_pack_candidates<<<fr_cdiv(n, 256), 256, 0, fr_stream(stream)>>>(idx, score, n, cand);
    FR_CHECK_LAUNCH("match_pack_candidates");
    return FR_OK;
}

extern "C" int fr_match_reduce_shards(const int32_t* cand, int R, int n, int q0, int F, int64_t* out_idx,
                                      float* out_score, fr_stream_t stream) {
    FR_REQUIRE(R >= 1 && n >= 0 && q0 >= 0 && F >= 0 && q0 + F <= n, "fr_match_reduce_shards: bad range (R %d n %d q0 %d F %d)", R, n, q0, F);
    if (F == 0) return FR_OK;
    FR_REQUIRE(cand && out_idx && out_score, "fr_match_reduce_shards: null pointer");
    match_reduce_shards<<<fr_cdiv(F, 256), 256, 0, fr_stream(stream)>>>(cand, R, n, q0, F, out_idx, out_score);
    FR_CHECK_LAUNCH("match_reduce_shards");
    return FR_OK;
}

// Exchange glue of the sharded match (distributed.py).  These used to be torch kernels (zeros / slice-assign / fill_ /
// dtype cast) on the embed stream: arithmetic this library does not build shares no stream with the convs.
// send f32 [q_max+1][D]: rows [0,F) = Q, rows [F,q_max) = 0, row q_max = (F, 0, 0, ...) - the count rides along.
__global__ void exchange_pack_queries(const float* __restrict__ Q, int F, int q_max, int D, float* __restrict__ send) {
    const int64_t n4 = (int64_t)(q_max + 1) * D / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i * 4 / D);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < F) v = reinterpret_cast<const float4*>(Q)[i];
        else if (row == q_max && i * 4 == (int64_t)row * D) v.x = (float)F;
        reinterpret_cast<float4*>(send)[i] = v;
    }
}

// counts[r] = (int) gathered[r][q_max][0], clamped to [0, q_max]
__global__ void exchange_counts(const float* __restrict__ gathered, int R, int q_max, int D, int32_t* __restrict__ counts) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float c = gathered[((int64_t)r * (q_max + 1) + q_max) * D];
    int n = (int)c;
    counts[r] = n < 0 ? 0 : (n > q_max ? q_max : n);
}

extern "C" int fr_exchange_pack_queries(const float* Q, int F, int q_max, int D, float* send, fr_stream_t stream) {
    FR_REQUIRE(send && q_max >= 0 && F >= 0 && F <= q_max && D > 0 && D % 4 == 0 && (Q || F == 0),
               "fr_exchange_pack_queries: bad argument (F %d q_max %d D %d)", F, q_max, D);
    const int64_t n4 = (int64_t)(q_max + 1) * D / 4;
    int blocks = fr_cdiv(n4, 256); if (blocks > 2048) blocks = 2048;
    exchange_pack_queries<<<blocks, 256, 0, fr_stream(stream)>>>(Q, F, q_max, D, send);
    FR_CHECK_LAUNCH("exchange_pack_queries");
    return FR_OK;
}

extern "C" int fr_exchange_counts(const float* gathered, int R, int q_max, int D, int32_t* counts, fr_stream_t stream) {
    FR_REQUIRE(gathered && counts && R >= 1 && q_max >= 0 && D > 0, "fr_exchange_counts: bad argument");
    exchange_counts<<<fr_cdiv(R, 64), 64, 0, fr_stream(stream)>>>(gathered, R, q_max, D, counts);
    FR_CHECK_LAUNCH("exchange_counts");
    return FR_OK;
}

__global__ void f32_to_f16_k(const float* __restrict__ x, half_t* __restrict__ out, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            float4 v = *reinterpret_cast<const float4*>(x + i);
            half4 o = {(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
            *reinterpret_cast<half4*>(out + i) = o;
        } else {
            for (int64_t j = i; j < n; ++j) out[j] = (half_t)x[j];
        }
    }
}

extern "C" int fr_f32_to_f16(const float* x, void* out, int64_t n, fr_stream_t stream) {
    if (n <= 0) return FR_OK;
    FR_REQUIRE(x && out, "fr_f32_to_f16: null pointer");
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    f32_to_f16_k<<<blocks, 256, 0, fr_stream(stream)>>>(x, reinterpret_cast<half_t*>(out), n);
    FR_CHECK_LAUNCH("f32_to_f16");
    return FR_OK;
}

// ---------------------------------------------------------------- next-row consumers of the scan
// First row (lowest index) whose dot with the query exceeds a threshold: the enrolment duplicate check
// (`sim > duplicate_threshold`, first hit returns: /root/reference/trainingServer.py:181-194) and the
// unknown-person cluster assignment (`similarity >= 0.65`, first hit breaks:
// /root/reference/peopleCount.py:441-449).  One wave per (query, 64-row slab): coalesced row reads,
// wave reduction of the dot, running minimum of the passing row index.
__global__ __launch_bounds__(256) void gallery_first_above(const float* __restrict__ Q, const float* __restrict__ G,
                                                           int F, int64_t N, float thr, int inclusive,
                                                           unsigned long long* __restrict__ out_min) {
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* q = Q + (int64_t)f * GD;
    float qv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qv[k] = q[lane * 8 + k];
    unsigned long long best = ~0ull;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < N; r += (int64_t)gridDim.x * 4) {
        const float4* g = reinterpret_cast<const float4*>(G + r * GD + lane * 8);
        const float4 g0 = g[0], g1 = g[1];
        float s = qv[0] * g0.x + qv[1] * g0.y + qv[2] * g0.z + qv[3] * g0.w + qv[4] * g1.x + qv[5] * g1.y +
                  qv[6] * g1.z + qv[7] * g1.w;
        s = wave_sum(s);
        const bool pass = inclusive ? (s >= thr) : (s > thr);
        if (pass) {
            unsigned long long key = ((unsigned long long)r << 32) | __float_as_uint(s);
            best = key < best ? key : best;
        }
    }
    if (lane == 0 && best != ~0ull) atomicMin(out_min + f, best);
}

__global__ void first_above_finish(const unsigned long long* __restrict__ mins, int F, int64_t row_offset,
                                   int64_t* __restrict__ out_idx, float* __restrict__ out_score) {
    int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    unsigned long long k = mins[f];
    if (k == ~0ull) { out_idx[f] = -1; out_score[f] = 0.f; }
    else { out_idx[f] = (int64_t)(k >> 32) + row_offset; out_score[f] = __uint_as_float((unsigned)(k & 0xffffffffu)); }
}

extern "C" int fr_gallery_first_above_f32(const float* Q, const float* G, int F, int64_t N, int D, float thr,
                                          int inclusive, int64_t row_offset, int64_t* out_idx, float* out_score,
                                          void* workspace, size_t workspace_bytes, fr_stream_t stream) {
    FR_REQUIRE(D == GD, "fr_gallery_first_above_f32: D must be %d", GD);
    if (F <= 0) return FR_OK;
    FR_REQUIRE(Q && out_idx && out_score && (G || N == 0) && N >= 0 && N < (1ll << 31), "fr_gallery_first_above_f32: bad argument");
    FR_REQUIRE(workspace && workspace_bytes >= (size_t)F * 8, "fr_gallery_first_above_f32: workspace needs %zu bytes", (size_t)F * 8);
    hipStream_t s = fr_stream(stream);
    unsigned long long* mins = reinterpret_cast<unsigned long long*>(workspace);
    if (hipMemsetAsync(mins, 0xff, (size_t)F * 8, s) != hipSuccess) { fr_set_error("fr_gallery_first_above_f32: memset failed"); return FR_E_LAUNCH; }
    if (N > 0) {
        int bx = (int)((N + 3) / 4); if (bx > 1024) bx = 1024;
        gallery_first_above<<<dim3(bx, F), 256, 0, s>>>(Q, G, F, N, thr, inclusive, mins);
        FR_CHECK_LAUNCH("gallery_first_above");
    }
    first_above_finish<<<fr_cdiv(F, 64), 64, 0, s>>>(mins, F, row_offset, out_idx, out_score);
    FR_CHECK_LAUNCH("first_above_finish");
    return FR_OK;
}

// out[f][n] = dot(A[f], B[n]) / (|A[f]| |B[n]|): the pose-consistency matrix of the enrolment path
// (/root/reference/trainingServer.py:202-214); one wave per pair.
__global__ void cosine_matrix(const float* __restrict__ A, const float* __restrict__ B, int F, int N, int D,
                              float* __restrict__ out) {
    const int pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pair >= F * N) return;
    const int f = pair / N, n = pair - f * N;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float a = A[(int64_t)f * D + c], b = B[(int64_t)n * D + c];
        ab += a * b; aa += a * a; bb += b * b;
    }
    ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
    if (lane == 0) out[pair] = ab / (sqrtf(aa) * sqrtf(bb));
}

extern "C" int fr_cosine_matrix_f32(const float* A, const float* B, int F, int N, int D, float* out,
                                    fr_stream_t stream) {
    if (F <= 0 || N <= 0) return FR_OK;
    FR_REQUIRE(A && B && out && D > 0, "fr_cosine_matrix_f32: bad argument");
    cosine_matrix<<<fr_cdiv((int64_t)F * N, 4), 256, 0, fr_stream(stream)>>>(A, B, F, N, D, out);
    FR_CHECK_LAUNCH("cosine_matrix");
    return FR_OK;
}

// mean of K rows (np.mean(face_embeddings, axis=0), /root/reference/trainingServer.py:355; the unknown-person
// running mean /root/reference/peopleCount.py:79): out[c] = (x[0][c] + ... + x[K-1][c]) / K, summed in row order.
__global__ void mean_rows(const float* __restrict__ x, int K, int D, float* __restrict__ out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += x[(int64_t)k * D + c];
    out[c] = s / (float)K;
}

extern "C" int fr_mean_rows_f32(const float* x, int K, int D, float* out, fr_stream_t stream) {
    FR_REQUIRE(x && out && K > 0 && D > 0, "fr_mean_rows_f32: bad argument");
    mean_rows<<<fr_cdiv(D, 256), 256, 0, fr_stream(stream)>>>(x, K, D, out);
    FR_CHECK_LAUNCH("mean_rows");
    return FR_OK;
}

// Gallery scan as a GEMM for MANY queries and LARGE galleries (BASELINE configs C4 / C5), with an exact f32
// re-rank so the answer is still the reference's: best = -1; for id, g in gallery: s = dot(q, g); if s > best ...
// (/root/reference/infrenceServer.py:535-542) - maximum f32 score, lowest row on exact ties.
//
// Coarse pass  (gallery_gemm_scan<FP8>): scores[row][query] = G16/G8 . Q on the f16 matrix cores
//   (v_mfma_f32_16x16x32_f16) or the fp8 ones (v_mfma_scale_f32_16x16x128_f8f6f4, unit scales: twice the f16 rate),
//   ONE pass over the gallery for up to 256 queries per block:
//   * a block = 8 waves; every wave keeps ITS 32 queries STATIONARY IN REGISTERS as MFMA B fragments for the whole
//     K = 512 (f16: 128 VGPRs, fp8: 64), so a block covers 256 queries and the only streamed operand is the gallery;
//   * the gallery streams through LDS in tiles of 64 rows x 512 (f16 64 KB / fp8 32 KB), two buffers, filled by
//     LDS-DMA (buffer_load ... lds, 16 B/lane, whole 128-B lines, XOR-swizzled on the SOURCE side) one tile ahead;
//     one barrier per tile (2048 MFMA cycles per wave), all 8 waves share the tile: gallery bytes cross L2->LDS once
//     per 256 queries, and blocks that scan the same row range for other query tiles share an XCD (its L2);
//   * A = 16 gallery rows, B = 16 queries: a lane owns query (n*16 + lane&15) and rows (m*16 + 4*(lane>>4) + reg): one
//     accumulator quad = 4 CONSECUTIVE gallery rows (a "group", id = row / 4) of one query.  The candidates kept are
//     GROUPS: the running top-K of group maxima per query is LANE-LOCAL (registers), groups arrive in ascending
//     order, strict '>' keeps the earlier group on ties.  Per tile and query the common path is a max tree over the
//     lane's 16 scores and one compare (in-kernel ablation: inserting every row into per-lane row lists cost 30 % of
//     the f16 scan and 60 % of the fp8 scan - 128 slowly warming lists per wave mean a list insert in nearly every
//     tile).  No cross-lane work inside the scan.
// Re-rank (gallery_rerank<K>): one wave per query merges the candidate lists, keeps the K groups with the best coarse
//   maxima and re-scores their 4 rows each EXACTLY in f32 against the f32 rows; final pick = max f32 score, lowest
//   row.  Coarse rounding can only matter if the true winner's group fell out of the coarse top-K groups (K = 4 for
//   f16, 8 for fp8: the fp8 score error is ~2e-3, the gap to the K-th best of 10^6..10^7 random rows is 7 sigma of
//   it; the tests count); near-duplicate neighbouring rows share a group and cannot crowd each other out.
// Algorithmic bytes: N * 512 * b per pass per 256 queries (b = 2 / 1); FLOP: 2 * N * F * 512.
#include "common.h"

#define GD 512
#define SG_ROWS 64            // gallery rows per LDS tile
#define SG_QW 32              // queries per wave (2 MFMA n-tiles)
#define SG_QB 256             // queries per block (8 waves)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef int int8v __attribute__((ext_vector_type(8)));

// query slot q belongs to segment q / seg_len and is real iff its position in the segment < seg_counts[segment]
__device__ __forceinline__ bool sg_slot_valid(const int32_t* seg_counts, int seg_len, int q) {
    const int seg = q / seg_len;
    return q - seg * seg_len < seg_counts[seg];
}

template <int K>
struct TopK { float s[K]; int i[K]; };

template <int K>
__device__ __forceinline__ void topk_insert(TopK<K>& t, float s, int i) {
    // static indices only (a runtime-indexed store would push the lists to scratch memory)
    if (!(s > t.s[K - 1])) return;
    t.s[K - 1] = s; t.i[K - 1] = i;
#pragma unroll
    for (int k = K - 1; k > 0; --k) {
        const bool up = t.s[k] > t.s[k - 1];
        const float hs = up ? t.s[k] : t.s[k - 1], ls = up ? t.s[k - 1] : t.s[k];
        const int hi = up ? t.i[k] : t.i[k - 1], li = up ? t.i[k - 1] : t.i[k];
        t.s[k - 1] = hs; t.s[k] = ls; t.i[k - 1] = hi; t.i[k] = li;
    }
}

struct ScanP {
    const float* Q; const void* G; int F; int64_t N;
    int nqt, nranges; int64_t rows_per_range;          // rows_per_range: multiple of SG_ROWS
    float* ws_score; int* ws_idx;                       // [F][nranges*4][K]
    const int32_t* seg_counts; int seg_len;
    float qscale;                                       // fp8: queries are multiplied by this before conversion
    int abl;                                            // debug build only (FR_SCAN_ABL): 1 no epilogue, 2 one DMA only, 4 no MFMA
};

__device__ __forceinline__ int4v sg_pack_f16(const float* q) {
    const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
    const half8 h = {(half_t)a.x, (half_t)a.y, (half_t)a.z, (half_t)a.w, (half_t)b.x, (half_t)b.y, (half_t)b.z, (half_t)b.w};
    return __builtin_bit_cast(int4v, h);
}

// saturating at +-448: past the range the OCP e4m3 conversion produces NaN, and a NaN code fails every '>' of the
// coarse scan - a non-unit row or query (set_rows(normalise=False), match_device(renormalise=False)) with an element
// beyond 1.75 (x 256) would silently never reach the exact re-rank
__device__ __forceinline__ int sg_pack_fp8x4(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return v;
}

__device__ __forceinline__ int4v sg_pack_fp8(const float* q, float sc) {
    int4v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float4 a = *reinterpret_cast<const float4*>(q + e * 4);
        o[e] = sg_pack_fp8x4(a.x * sc, a.y * sc, a.z * sc, a.w * sc);
    }
    return o;
}

// FP8 = false: G is f16 [N][512] (1 KB rows, 8 chunks of 128 B); true: G is fp8 e4m3 [N][512] (512-B rows, 4 chunks).
// Within a 128-B chunk, lane quarter fq uses bytes [16 fq, +16) ("lo") and [64 + 16 fq, +16) ("hi"): for f16 these
// are the fragments of the chunk's two K = 32 MFMAs; for fp8 both halves feed ONE K = 128 MFMA (the k order inside
// an MFMA is free as long as A and B agree), so the LDS image and its conflict-free ds_read_b128 pattern are shared.
template <bool FP8, int TK>
__global__ __launch_bounds__(512, 2) void gallery_gemm_scan(ScanP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int RB = FP8 ? 512 : 1024;               // gallery row bytes
    constexpr int NKC = RB / 128;                      // 128-B chunks per row
    constexpr int TILE_B = SG_ROWS * RB;               // LDS bytes per tile
    constexpr int NPIECE = TILE_B / 1024 / 8;          // LDS-DMA instructions per wave per tile (8 / 4)
    extern __shared__ __attribute__((aligned(16))) char lds[];      // [2][NKC][64 rows][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // blocks that scan the same row range for different query tiles are adjacent in dispatch order and share an
    // XCD (blocks b and b + 8 do): the range's rows reach that XCD's L2 once.  Speed only.
    const int b = blockIdx.x, xcd = b & 7, g = b >> 3;
    const int qt = g % p.nqt;
    const int rr = (g / p.nqt) * 8 + xcd;
    if (rr >= p.nranges) return;
    const int q0b = qt * SG_QB, q0w = q0b + wave * SG_QW;
    bool any = false;                                    // block-uniform: does this query tile hold a real query?
    if (p.seg_counts) {
        for (int q = q0b; q < min(q0b + SG_QB, p.F);) {
            if (sg_slot_valid(p.seg_counts, p.seg_len, q)) { any = true; break; }
            q = (q / p.seg_len + 1) * p.seg_len;
        }
        if (!any) return;
    }
    const int64_t r0 = (int64_t)rr * p.rows_per_range;
    const int64_t r1 = min(p.N, r0 + p.rows_per_range);
    const int nrows = (int)(r1 - r0);
    const int ntiles = (nrows + SG_ROWS - 1) / SG_ROWS;

    // ---- stationary B operand: this wave's 32 queries, whole K, converted on the way in
    int4v blo[2][NKC], bhi[2][NKC];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int q = q0w + n * 16 + fr;
        const bool ok = q < p.F;
        const float* qp = p.Q + (int64_t)(ok ? q : 0) * GD;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            if (FP8) {
                blo[n][kc] = sg_pack_fp8(qp + kc * 128 + fq * 16, p.qscale);
                bhi[n][kc] = sg_pack_fp8(qp + kc * 128 + 64 + fq * 16, p.qscale);
            } else {
                blo[n][kc] = sg_pack_f16(qp + kc * 64 + fq * 8);
                bhi[n][kc] = sg_pack_f16(qp + kc * 64 + 32 + fq * 8);
            }
            if (!ok) { blo[n][kc] = int4v{0, 0, 0, 0}; bhi[n][kc] = int4v{0, 0, 0, 0}; }
        }
    }

    // ---- gallery stream: wave w fills pieces w*NPIECE + i of a tile; piece = (kc, 8-row group)
    __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(p.G) + r0 * RB), 0, (unsigned)((int64_t)nrows * RB), 0x00020000);
    unsigned voff[NPIECE];
    unsigned ldst[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int pc = wave * NPIECE + i, kc = pc >> 3, rg = pc & 7;
        const int row = rg * 8 + (lane >> 3);
        voff[i] = (unsigned)(row * RB + kc * 128 + (((lane & 7) ^ (row & 7)) << 4));   // rows past the range read 0
        ldst[i] = (unsigned)((kc * SG_ROWS + rg * 8) * 128);
    }
    auto issue_tile = [&](int t) {
        char* dst = lds + (t & 1) * TILE_B;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(grs, (lds_ptr_t)(dst + ldst[i]), 16, voff[i] + (unsigned)t * TILE_B, 0, 0, 0);
    };

    TopK<TK> top[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int k = 0; k < TK; ++k) { top[n].s[k] = -INFINITY; top[n].i[k] = -1; }

    const int key = fr & 7;
    const unsigned a_lo = (unsigned)(fr * 128 + ((fq ^ key) << 4)), a_hi = (unsigned)(fr * 128 + (((4 + fq) ^ key) << 4));

    issue_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of tile t have landed
        __builtin_amdgcn_s_barrier();                               // everyone's have; buffer (t+1)&1 is free again
        if (t + 1 < ntiles && !(FR_DEBUG && (p.abl & 2))) issue_tile(t + 1);
        const char* buf = lds + (t & 1) * TILE_B;
        float4v acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            int4v alo[4], ahi[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const char* rowp = buf + (kc * SG_ROWS + m * 16) * 128;
                alo[m] = *reinterpret_cast<const int4v*>(rowp + a_lo);
                ahi[m] = *reinterpret_cast<const int4v*>(rowp + a_hi);
            }
            if (FR_DEBUG && (p.abl & 4)) {
#pragma unroll
                for (int m = 0; m < 4; ++m) { acc[m][0][0] += __int_as_float(alo[m][0]); acc[m][1][0] += __int_as_float(ahi[m][0]); }
                continue;
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (FP8) {
                        const int8v a = {alo[m][0], alo[m][1], alo[m][2], alo[m][3], ahi[m][0], ahi[m][1], ahi[m][2], ahi[m][3]};
                        const int8v bb = {blo[n][kc][0], blo[n][kc][1], blo[n][kc][2], blo[n][kc][3],
                                          bhi[n][kc][0], bhi[n][kc][1], bhi[n][kc][2], bhi[n][kc][3]};
                        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, bb, acc[m][n], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    } else {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, alo[m]),
                                                                           __builtin_bit_cast(half8, blo[n][kc]), acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ahi[m]),
                                                                           __builtin_bit_cast(half8, bhi[n][kc]), acc[m][n], 0, 0, 0);
                    }
                }
        }
        // acc[m][n] = coarse scores of the 4-row group (t*64 + m*16 + 4*fq)/4 for query q0w + n*16 + fr; groups ascend in m
        if (FR_DEBUG && (p.abl & 1)) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) top[n].s[0] = fmaxf(top[n].s[0], acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3]);
            continue;
        }
        const int rbase = t * SG_ROWS + 4 * fq;
        if (t == ntiles - 1) {                                       // rows past the range end read as zeros: not candidates
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (rbase + m * 16 + reg >= nrows) { acc[m][0][reg] = -INFINITY; acc[m][1][reg] = -INFINITY; }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float gm[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) gm[m] = fmaxf(fmaxf(acc[m][n][0], acc[m][n][1]), fmaxf(acc[m][n][2], acc[m][n][3]));
            const float mx = fmaxf(fmaxf(gm[0], gm[1]), fmaxf(gm[2], gm[3]));
            if (mx > top[n].s[TK - 1]) {                             // rare once the lists have warmed up
                const int g0 = (int)((r0 + rbase) >> 2);             // r0 and rbase are multiples of 4
#pragma unroll
                for (int m = 0; m < 4; ++m) topk_insert<TK>(top[n], gm[m], g0 + m * 4);
            }
        }
    }
    // every (range, lane quarter) writes its candidates; the re-rank kernel merges them
    const int slots = p.nranges * 4, slot = rr * 4 + fq;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int q = q0w + n * 16 + fr;
        if (q < p.F) {
            float* so = p.ws_score + ((int64_t)q * slots + slot) * TK;
            int* io = p.ws_idx + ((int64_t)q * slots + slot) * TK;
#pragma unroll
            for (int k = 0; k < TK; ++k) { so[k] = top[n].s[k]; io[k] = top[n].i[k]; }
        }
    }
#endif
}

// one wave per query: merge the candidate lists (coarse scores), keep the best K, re-score them in f32
template <int K>
__global__ __launch_bounds__(64) void gallery_rerank(const float* __restrict__ Q, const float* __restrict__ G32,
                                                     const float* __restrict__ ws_score, const int* __restrict__ ws_idx,
                                                     int F, int64_t N, int ncand, int64_t row_offset, float coarse_unscale,
                                                     int64_t* __restrict__ out_idx, float* __restrict__ out_score,
                                                     const int32_t* __restrict__ seg_counts, int seg_len) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (seg_counts && !sg_slot_valid(seg_counts, seg_len, q)) {
        if (lane == 0) { out_idx[q] = -1; out_score[q] = -1.0f; }
        return;
    }
    // ONE pass over the candidates: every lane keeps the best K of its strided share (order: coarse score, then lowest
    // group id), then K rounds of wave-wide argmax over the lanes' list heads pop the overall best K.
    TopK<K> loc;
#pragma unroll
    for (int k = 0; k < K; ++k) { loc.s[k] = -INFINITY; loc.i[k] = 0x7fffffff; }
    for (int c = lane; c < ncand; c += 64) {
        const float s = ws_score[(int64_t)q * ncand + c];
        const int i = ws_idx[(int64_t)q * ncand + c];
        if (i < 0 || !(s > loc.s[K - 1] || (s == loc.s[K - 1] && i < loc.i[K - 1]))) continue;
        loc.s[K - 1] = s; loc.i[K - 1] = i;
#pragma unroll
        for (int k = K - 1; k > 0; --k) {
            const bool up = loc.s[k] > loc.s[k - 1] || (loc.s[k] == loc.s[k - 1] && loc.i[k] < loc.i[k - 1]);
            const float hs = up ? loc.s[k] : loc.s[k - 1], ls = up ? loc.s[k - 1] : loc.s[k];
            const int hi = up ? loc.i[k] : loc.i[k - 1], li = up ? loc.i[k - 1] : loc.i[k];
            loc.s[k - 1] = hs; loc.s[k] = ls; loc.i[k - 1] = hi; loc.i[k] = li;
        }
    }
    float bs[K]; int bi[K];
#pragma unroll
    for (int round = 0; round < K; ++round) {
        float ms = loc.s[0]; int mi = loc.i[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float os = __shfl_xor(ms, o, 64); const int oi = __shfl_xor(mi, o, 64);
            if (os > ms || (os == ms && oi < mi)) { ms = os; mi = oi; }
        }
        bs[round] = ms; bi[round] = (mi == 0x7fffffff) ? -1 : mi;
        if (mi != 0x7fffffff && loc.i[0] == mi) {                  // group ids are unique: exactly one lane pops
#pragma unroll
            for (int k = 0; k + 1 < K; ++k) { loc.s[k] = loc.s[k + 1]; loc.i[k] = loc.i[k + 1]; }
            loc.s[K - 1] = -INFINITY; loc.i[K - 1] = 0x7fffffff;
        }
    }
    // bi[k] = group id (4 consecutive rows); re-score the rows of the K groups: 16 lanes per row, 4 rows per group
    float best = -INFINITY; int besti = -1;
    const int sub = lane >> 4, l16 = lane & 15;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (bi[k] < 0) continue;
        const int64_t row = (int64_t)bi[k] * 4 + sub;
        float s = bs[k] * coarse_unscale;                          // G32 == NULL: the group's coarse maximum stands for its rows
        const bool ok = row < N;
        if (G32) {                                                 // exact f32 dot, one row per 16 lanes
            const float* gg = G32 + (ok ? row : 0) * GD;
            const float* qq = Q + (int64_t)q * GD;
            float pp = 0.f;
#pragma unroll
            for (int c = 0; c < GD / 64; ++c) {
                const float4 a = *reinterpret_cast<const float4*>(qq + c * 64 + l16 * 4);
                const float4 bb = *reinterpret_cast<const float4*>(gg + c * 64 + l16 * 4);
                pp += a.x * bb.x + a.y * bb.y + a.z * bb.z + a.w * bb.w;
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) pp += __shfl_xor(pp, o, 64);
            s = pp;
        }
        float cs = ok ? s : -INFINITY;
        int ci = ok ? (int)row : 0x7fffffff;
#pragma unroll
        for (int o = 32; o >= 16; o >>= 1) {                       // best of the group's 4 rows (lowest row on ties)
            const float os = __shfl_xor(cs, o, 64); const int oi = __shfl_xor(ci, o, 64);
            if (os > cs || (os == cs && oi < ci)) { cs = os; ci = oi; }
        }
        if (ci != 0x7fffffff && (cs > best || (cs == best && ci < besti) || besti < 0)) { best = cs; besti = ci; }
    }
    if (lane == 0) {
        if (besti < 0 || !(best > -1.0f)) { out_idx[q] = -1; out_score[q] = -1.0f; }
        else { out_idx[q] = besti + row_offset; out_score[q] = best; }
    }
}

// ---------------------------------------------------------------- host side
struct ScanPlan { int nqt, nranges; int64_t rows_per_range; int grid; };

static ScanPlan scan_plan(int F, int64_t N) {
    ScanPlan pl;
    pl.nqt = (F + SG_QB - 1) / SG_QB;
    if (pl.nqt < 1) pl.nqt = 1;
    // one block per CU (8 waves x up to 256 VGPRs fill it), one round of blocks over the 256 CUs, at least one 64-row tile per range and a
    // multiple of 8 ranges so that every XCD gets the same share
    int64_t tiles = (N + SG_ROWS - 1) / SG_ROWS;
    if (tiles < 1) tiles = 1;
    int64_t want = 256 / pl.nqt;
    if (want < 8) want = 8;
    int64_t nr = tiles < want ? tiles : want;
    int64_t tpr = (tiles + nr - 1) / nr;                 // tiles per range
    const int64_t max_tpr = ((int64_t)1 << 30) / (SG_ROWS * 1024);      // buffer range < 2^31 bytes
    if (tpr > max_tpr) tpr = max_tpr;
    nr = (tiles + tpr - 1) / tpr;
    pl.nranges = (int)nr;
    pl.rows_per_range = tpr * SG_ROWS;
    pl.grid = (int)((nr + 7) / 8) * 8 * pl.nqt;
    return pl;
}

template <int TK>
static size_t scan_ws_bytes(int F, int64_t N) {
    const ScanPlan pl = scan_plan(F, N);
    return (size_t)(F > 0 ? F : 1) * pl.nranges * 4 * TK * 8 + 256;
}

extern "C" size_t fr_gallery_match_f16_workspace(int F, int64_t N) { return scan_ws_bytes<FR_TOPK>(F, N); }
extern "C" size_t fr_gallery_match_f8_workspace(int F, int64_t N) { return scan_ws_bytes<FR_TOPK8>(F, N); }

template <bool FP8, int TK>
static int gemm_scan_launch(const char* who, const float* Q, const void* Gc, const float* G32, int F, int64_t N, int D,
                            int64_t row_offset, int64_t* out_idx, float* out_score, void* workspace,
                            size_t workspace_bytes, const int32_t* seg_counts, int seg_len, float qscale,
                            float coarse_unscale, fr_stream_t stream) {
    FR_REQUIRE(D == GD, "%s: D must be %d (got %d)", who, GD, D);
    FR_REQUIRE(F >= 0 && N >= 0 && N < (1ll << 31), "%s: bad size", who);
    if (F == 0) return FR_OK;
    FR_REQUIRE(Q && out_idx && out_score && (Gc || N == 0), "%s: null pointer", who);
    FR_REQUIRE(!seg_counts || (seg_len > 0 && F % seg_len == 0), "%s: seg_len must divide F", who);
    FR_REQUIRE(workspace && workspace_bytes >= scan_ws_bytes<TK>(F, N), "%s: workspace too small (%zu < %zu)", who,
               workspace_bytes, scan_ws_bytes<TK>(F, N));
    hipStream_t s = fr_stream(stream);
    const ScanPlan pl = scan_plan(F, N);
    const int ncand = pl.nranges * 4 * TK;
    float* ws_score = reinterpret_cast<float*>(workspace);
    int* ws_idx = reinterpret_cast<int*>(ws_score + (size_t)F * ncand);
    if (N > 0) {
        ScanP p;
        p.Q = Q; p.G = Gc; p.F = F; p.N = N; p.nqt = pl.nqt; p.nranges = pl.nranges; p.rows_per_range = pl.rows_per_range;
        p.ws_score = ws_score; p.ws_idx = ws_idx; p.seg_counts = seg_counts; p.seg_len = seg_len; p.qscale = qscale;
        p.abl = fr_dbg_int("FR_SCAN_ABL", 0);
        constexpr int lds = 2 * SG_ROWS * (FP8 ? 512 : 1024);
        static FrDevLatch latch;
        if (!fr_raise_lds(reinterpret_cast<const void*>(gallery_gemm_scan<FP8, TK>), lds, latch)) {
            fr_set_error("%s: cannot raise dynamic LDS", who);
            return FR_E_LAUNCH;
        }
        gallery_gemm_scan<FP8, TK><<<pl.grid, 512, lds, s>>>(p);
        FR_CHECK_LAUNCH("gallery_gemm_scan");
    } else {
        if (hipMemsetAsync(ws_idx, 0xff, (size_t)F * ncand * sizeof(int), s) != hipSuccess) {   // no candidates
            fr_set_error("%s: memset failed", who);
            return FR_E_LAUNCH;
        }
    }
    gallery_rerank<TK><<<F, 64, 0, s>>>(Q, G32, ws_score, ws_idx, F, N, ncand, row_offset, coarse_unscale, out_idx, out_score,
                                        seg_counts, seg_len);
    FR_CHECK_LAUNCH("gallery_rerank");
    return FR_OK;
}

extern "C" int fr_gallery_match_f16(const float* Q, const void* G16, const float* G32, int F, int64_t N, int D,
                                    int64_t row_offset, int64_t* out_idx, float* out_score, void* workspace,
                                    size_t workspace_bytes, const int32_t* seg_counts, int seg_len, fr_stream_t stream) {
    return gemm_scan_launch<false, FR_TOPK>("fr_gallery_match_f16", Q, G16, G32, F, N, D, row_offset, out_idx, out_score,
                                            workspace, workspace_bytes, seg_counts, seg_len, 1.0f, 1.0f, stream);
}

extern "C" int fr_gallery_match_f8(const float* Q, const void* G8, const float* G32, int F, int64_t N, int D,
                                   int64_t row_offset, int64_t* out_idx, float* out_score, void* workspace,
                                   size_t workspace_bytes, const int32_t* seg_counts, int seg_len, fr_stream_t stream) {
    return gemm_scan_launch<true, FR_TOPK8>("fr_gallery_match_f8", Q, G8, G32, F, N, D, row_offset, out_idx, out_score,
                                            workspace, workspace_bytes, seg_counts, seg_len, FR_F8_SCALE,
                                            1.0f / (FR_F8_SCALE * FR_F8_SCALE), stream);
}

// f32 unit rows -> fp8 e4m3 (OCP) rows scaled by FR_F8_SCALE = 256: |element| <= 1 maps into [-256, 256] (e4m3
// max 448), a typical element 1/sqrt(512) to ~11, far above the subnormal range.
__global__ void f32_to_f8_k(const float* __restrict__ x, int* __restrict__ out, int64_t n4, float sc) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
        out[i] = sg_pack_fp8x4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
    }
}

extern "C" int fr_f32_to_f8(const float* x, void* out, int64_t n, fr_stream_t stream) {
    if (n <= 0) return FR_OK;
    FR_REQUIRE(x && out && n % 4 == 0, "fr_f32_to_f8: null pointer or n not a multiple of 4");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    f32_to_f8_k<<<(int)blocks, 256, 0, fr_stream(stream)>>>(x, reinterpret_cast<int*>(out), n / 4, FR_F8_SCALE);
    FR_CHECK_LAUNCH("f32_to_f8");
    return FR_OK;
}

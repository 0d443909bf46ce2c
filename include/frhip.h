/*
 * frhip.h -- C ABI of libfrhip.so: the MI355X (gfx950) detect -> align -> embed -> match hot path.
 *
 * The reference crosses into native code only through ONNX Runtime inside
 * insightface's FaceAnalysis.get (/root/reference/infrenceServer.py:412-416,528) and
 * through NumPy/BLAS for the match loop (/root/reference/infrenceServer.py:530-552).
 * These entry points are what a binding for that path binds instead.  Conventions:
 *   - every function returns 0 on success or a negative FR_E_* code;
 *     fr_last_error_string() gives the thread-local message;
 *   - no allocation inside: every buffer is device memory owned by the caller
 *     (PyTorch-ROCm tensors in the Python host), passed as plain pointers + sizes;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     launches are asynchronous, nothing synchronises;
 *   - no global mutable state and no environment variables: the only process-wide state is an atomic
 *     per-device bit per kernel remembering that its dynamic-LDS limit has been raised there (idempotent);
 *     diagnostic switches and in-kernel cycle stamps exist only in the separate debug build
 *     (`make debug` -> libfrhip_debug.so), never in libfrhip.so.
 * Each declaration cites the reference behaviour it replaces.
 */
#ifndef FRHIP_H
#define FRHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FR_OK 0
#define FR_E_INVALID (-1)   /* bad argument / unsupported shape */
#define FR_E_LAUNCH (-2)    /* HIP launch error */
#define FR_E_NODEVICE (-3)  /* no gfx950 device */

typedef void* fr_stream_t;

/* Version of THIS header's ABI: bumped whenever a signature or an argument struct changes shape (101: fr_conv_args gained
 * x2 / C2, fr_conv_f8_args y8_sub, fr_pnet23_split_f16 all_heads).  fr_version() returns the value the library was built
 * with: a caller compiled against another header must refuse to go on (the Python binding does, _lib.load()). */
#define FR_ABI_VERSION 102
int fr_version(void);
const char* fr_last_error_string(void);
/* number of visible HIP devices (<=0: none); does not create a context on any device */
int fr_device_count(void);

/* ---------------------------------------------------------------- match ----
 * a-6  q = normed_embedding / ||normed_embedding||   (infrenceServer.py:532, peopleCount.py:863)
 * also the gallery-row normalise v/||v||              (infrenceServer.py:271,324) */
int fr_l2norm_rows_f32(const float* x, float* out, int rows, int dim, fr_stream_t stream);

/* a-7  best = -1; for id,g in gallery: s = dot(q,g); if s > best: best,id = s,id
 *      (infrenceServer.py:535-542, peopleCount.py:866-873).  Rows carry their index as id;
 *      strict '>' => the lowest row index wins exact ties.  Q [F,D] f32 (re-normalised),
 *      G [N,D] f32 row-major, D == 512.  out_idx[f] = row + row_offset (or -1 when N == 0),
 *      out_score[f] = best dot (or -1).  workspace: fr_gallery_match_workspace() bytes.
 *      seg_counts (optional, device i32 [F / seg_len]) marks padding: query slot f is real iff
 *      f % seg_len < seg_counts[f / seg_len] (the gathered per-rank face counts of the sharded match, 8(e));
 *      padding slots cost no scan work and report (-1, -1).  NULL: every slot is real. */
size_t fr_gallery_match_workspace(int F, int64_t N);
int fr_gallery_match_f32(const float* Q, const float* G, int F, int64_t N, int D,
                         int64_t row_offset, int64_t* out_idx, float* out_score,
                         void* workspace, size_t workspace_bytes, const int32_t* seg_counts, int seg_len,
                         fr_stream_t stream);
/* Gallery slab + views (replaces the per-frame dict filtering of
 * infrenceServer.py:343-380 and the dict inserts/deletes of :260-341, :234-258).  G is one device-resident
 * slab of row slots [capacity,512]; a view is the int64 slot list of one company in the reference's dict
 * order.  fr_gallery_match_view_f32 scans rows G[view[0..Nview)] without copying them; out_idx is the VIEW
 * position (so "first maximum in iteration order" is the view's order), -1 when Nview == 0.
 * fr_gallery_update_rows_f32 writes rows[i] (optionally v/||v||, :271,324) into slot slots[i] in place. */
int fr_gallery_match_view_f32(const float* Q, const float* G, const int64_t* view, int F, int64_t Nview, int D,
                              int64_t* out_idx, float* out_score, void* workspace, size_t workspace_bytes,
                              fr_stream_t stream);
int fr_gallery_update_rows_f32(float* G, const int64_t* slots, const float* rows, int n, int D, int normalise,
                               fr_stream_t stream);
/* Large galleries / many queries (BASELINE configs C4, C5): ONE pass over a 16-bit or 8-bit copy of the gallery
 * on the f16 / fp8 matrix cores for up to 256 queries per block keeps the top FR_TOPK (f16) / FR_TOPK8 (fp8)
 * rows per query, which are then re-scored EXACTLY in f32 against G32 (may be NULL: scores then come from the
 * coarse scan) and picked by max score / lowest row - the rule of infrenceServer.py:535-542.
 * G16: f16 [N,512] (fr_f32_to_f16 of the unit rows); G8: OCP fp8 e4m3 [N,512] scaled by FR_F8_SCALE
 * (fr_f32_to_f8).  seg_counts / seg_len: see fr_gallery_match_f32. */
#define FR_TOPK 4
#define FR_TOPK8 8
#define FR_F8_SCALE 256.0f
size_t fr_gallery_match_f16_workspace(int F, int64_t N);
int fr_gallery_match_f16(const float* Q, const void* G16, const float* G32, int F, int64_t N, int D,
                         int64_t row_offset, int64_t* out_idx, float* out_score,
                         void* workspace, size_t workspace_bytes, const int32_t* seg_counts, int seg_len,
                         fr_stream_t stream);
size_t fr_gallery_match_f8_workspace(int F, int64_t N);
int fr_gallery_match_f8(const float* Q, const void* G8, const float* G32, int F, int64_t N, int D,
                        int64_t row_offset, int64_t* out_idx, float* out_score,
                        void* workspace, size_t workspace_bytes, const int32_t* seg_counts, int seg_len,
                        fr_stream_t stream);
/* f32 -> fp8 e4m3 row conversion (x * FR_F8_SCALE, round to nearest even, n % 4 == 0) */
int fr_f32_to_f8(const float* x, void* out, int64_t n, fr_stream_t stream);
/* f32 -> f16 row conversion for building the device-resident gallery (infrenceServer.py:271) */
int fr_f32_to_f16(const float* x, void* out, int64_t n, fr_stream_t stream);
/* a-8  known = best_id and best >= thr  (infrenceServer.py:545; peopleCount.py:876-882):
 *      decision[f] = 1 recognised, 0 unknown, 2 dropped (counting path's [unknown_thr, thr) band;
 *      pass unknown_thr = thr for the live path). */
int fr_match_decide(const int64_t* idx, const float* score, int F, float thr, float unknown_thr,
                    int32_t* decision, fr_stream_t stream);


/* e   Sharded match (the reference's only parallelism is one process per camera, each scanning
 *      the whole gallery: infrenceServer.py:606,641-646).  Every rank scans ITS gallery row shard for all gathered
 *      queries (fr_gallery_match_* with row_offset = the shard's first global row); candidates travel as
 *      int32 [n][3] = (score bits, row lo, row hi) so the all-gather moves raw bits; the reduce over the R shards
 *      for queries [q0, q0+F) applies the scan's own rule - maximum score, lowest global row on exact ties,
 *      (-1, -1.0) when no shard has a candidate (infrenceServer.py:535-542).  cand_all: int32 [R][n][3]. */
int fr_match_pack_candidates(const int64_t* idx, const float* score, int n, int32_t* cand, fr_stream_t stream);
int fr_match_reduce_shards(const int32_t* cand_all, int R, int n, int q0, int F, int64_t* out_idx,
                           float* out_score, fr_stream_t stream);
/*      Exchange buffers of the same step: send f32 [q_max+1][D] = the rank's F unit query rows, zero rows up to q_max,
 *      and one trailing row whose first element is F (the face count rides in the one all-gather); after the gather
 *      (f32 [R][q_max+1][D]) counts[r] = that element of rank r's block.  The scan then walks the gathered buffer in
 *      place with seg_len = q_max + 1: the count row is slot q_max >= count, i.e. padding. */
int fr_exchange_pack_queries(const float* Q, int F, int q_max, int D, float* send, fr_stream_t stream);
int fr_exchange_counts(const float* gathered, int R, int q_max, int D, int32_t* counts, fr_stream_t stream);


/* ------------------------------------------------- enrolment / clustering consumers of the scan ----
 * First (lowest) row whose dot with the query is > thr (inclusive != 0: >= thr); idx -1 / score 0 when none.
 * Duplicate check `sim > 0.4`, first hit returns (trainingServer.py:181-194); unknown-person assignment
 * `similarity >= 0.65`, first hit breaks (peopleCount.py:441-449).  Rows must be unit length for the
 * dot to be the reference's cosine.  workspace: F * 8 bytes. */
int fr_gallery_first_above_f32(const float* Q, const float* G, int F, int64_t N, int D, float thr,
                               int inclusive, int64_t row_offset, int64_t* out_idx, float* out_score,
                               void* workspace, size_t workspace_bytes, fr_stream_t stream);
/* out[f][n] = cosine(A[f], B[n]): pose-consistency matrix (trainingServer.py:202-214) */
int fr_cosine_matrix_f32(const float* A, const float* B, int F, int N, int D, float* out,
                         fr_stream_t stream);
/* out = mean of K rows, summed in row order (np.mean(.., axis=0): trainingServer.py:355, peopleCount.py:79) */
int fr_mean_rows_f32(const float* x, int K, int D, float* out, fr_stream_t stream);

/* ---------------------------------------------------------------- embed ----
 * a-4  ArcFace IResNet conv stack (inside FaceAnalysis.get, infrenceServer.py:528).
 * Implicit-GEMM convolution on MFMA, NHWC f16 activations, f32 accumulate:
 *   y[n,ho,wo,co] = epi( sum_{kh,kw,ci} x[n,ho*s+kh-p,wo*s+kw-p,ci] * w[co,kh,kw,ci] )
 *   epi(v) = prelu(v + bias) (+ residual)
 * w: [Cout][KH*KW*Cin] f16 (K contiguous).  Cin % 64 == 0, or Cin == 8 (packed stem: w rows
 * zero-padded to a multiple of 128).  Cout % 64 == 0.
 * bias: f32 [Cout] (bias_mode 0) or [9][Cout] (bias_mode 1: border-class bias
 * [row class: top/mid/bottom][col class: left/mid/right] that carries a folded
 * pre-activation BN shift through the zero padding; 3x3/s1/p1 only), or NULL.
 * slope: f32 [Cout] PReLU slopes or NULL.  residual: f16 [M,Cout] or NULL.
 * out_f32_partial != NULL: split-K mode, raw f32 partial sums [splitk][M][Cout]
 * (no epilogue), used for the FC 25088->512 (H=W=1, Cin=25088).
 * x2 != NULL: a second input [B,H,W,C2] f16 (same H, W as x; C2 % 64 == 0) that enters through a 1x1 tap at
 * (ho * stride, wo * stride) as C2 extra K rows - w is then [Cout][KH*KW*Cin + C2].  An IResNet stage-entry block's
 * 1x1 / stride-2 shortcut conv of the block input, summed with its stride-2 3x3 conv in ONE implicit GEMM (f32
 * accumulation; biases summed by the caller) instead of a launch, an f16 map written and read back. */
typedef struct {
    const void* x; const void* w; void* y;
    const float* bias; const float* slope; const void* residual;
    float* out_f32_partial;
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int bias_mode; int splitk;
    const void* x2; int C2;
} fr_conv_args;
int fr_conv_nhwc_f16(const fr_conv_args* args, fr_stream_t stream);
/* fp8 form of the body convs (BASELINE config C5: "fp8 ArcFace conv path, CDNA4 fp8 MFMA"; same reference site,
 * infrenceServer.py:528): 3x3 / stride 1 / pad 1 layers at 14x14 and 28x28 (78 % of the r100 FLOPs) on
 * v_mfma_scale_f32_16x16x128_f8f6f4.  x8: OCP fp8 e4m3 NHWC [B,H,W,Cin] = x / sx; w8: e4m3 [Cout][9*Cin] =
 * w / sw[cout] (per-output-channel); oscale[cout] = sw[cout] * sx dequantises the f32 accumulator, then the f16
 * kernel's epilogue: + bias (or 9-class border bias) -> PReLU -> + residual (f16) -> one rounding.  Outputs:
 * y16 (f16 NHWC, may be NULL) and/or y8 = fp8((y - y8_sub[cout]) * y8_mul) (the next conv's input, may be NULL;
 * y8_sub: f32 [Cout] per-channel centre the CONSUMER subtracts from its input before rounding - its W.centre term,
 * which depends on which taps fall inside the image, lives in the consumer's 9-class border bias - or NULL = 0).
 * Cin % 128 == 0, Cout % 128 == 0. */
typedef struct {
    const void* x8; const void* w8; void* y16; void* y8;
    const float* oscale; const float* bias; const float* slope; const void* residual;
    int B, H, W, Cin, Cout, bias_mode;
    float y8_mul;
    const float* y8_sub;
} fr_conv_f8_args;
int fr_conv_nhwc_f8(const fr_conv_f8_args* args, fr_stream_t stream);
/* out8 = fp8_e4m3(x16 * mul), saturating at +-448; n % 8 == 0 (input of a stage's first fp8 conv) */
int fr_quantize_f16_f8(const void* x16, void* out8, int64_t n, float mul, fr_stream_t stream);
/* out8 = fp8_e4m3((x16 - sub[i % C]) * mul): the same with a per-channel centre (NHWC, C channels innermost; n % C == 0) */
int fr_quantize_f16_f8_centred(const void* x16, void* out8, int64_t n, int C, const float* sub, float mul,
                               fr_stream_t stream);
/* A run of `nblocks` stride-1 residual blocks at 14x14x256 (IResNet-100 stage 3 after its first block: 29 blocks =
 * 58 convs, 57 % of the network's FLOPs) as ONE launch with the image resident in LDS: one workgroup per image, a
 * conv's output overwrites its input in place, only the weights stream (conv_stage14.hip).  Same math as the
 * per-layer calls: block = conv3x3(+9-class border bias, PReLU) -> conv3x3(+bias) + input, f32 accumulation, one
 * rounding to f16 per conv output.  x: f16 NHWC [B,14,14,256] (input of the first block), y: same shape (the last
 * block's output; nothing else is written - the residual stream stays in LDS), x != y.  wstream: the convs' weights
 * in kernel order (conv1, conv2 of block 0, conv1 of block 1, ...), each re-ordered by fr_conv_stage14_pack from the
 * [256][9*256] f16 layout of fr_conv_nhwc_f16 into 72 x 16 KB slot images (fr_conv_stage14_weight_bytes(1) bytes per
 * conv).  params: f32
 * [2*nblocks][10][256]: rows 0..8 the bias of border class (row class * 3 + column class) - a conv with a plain bias
 * repeats it nine times - row 9 the PReLU slope (1.0 = none). */
size_t fr_conv_stage14_weight_bytes(int nconv);
int fr_conv_stage14_pack(const void* w, void* out, fr_stream_t stream);
int fr_conv_stage14_f16(const void* x, void* y, const void* wstream, const float* params, int B, int nblocks,
                        fr_stream_t stream);
/* The run of stride-1 128 -> 128 residual blocks of IResNet's 28x28 stage (r100: 12 blocks = 24 convs; embed half of
 * FaceAnalysis.get, /root/reference/infrenceServer.py:528) as ONE launch, one workgroup per face walking its own map through
 * all the convs (conv_stage28.hip): half an image per pass, the half's input halo in LDS, weights streaming, outputs
 * straight from the accumulators to HBM, read back by the same workgroup.  Same math as the blocks run through
 * fr_conv_nhwc_f16.  x: f16 [B][28][28][128], OVERWRITTEN with the run's output; mid: scratch of the same size; wstream: the
 * convs' weights in kernel order, each re-ordered by fr_conv_stage28_pack from the [128][9*128] f16 layout into 36 x 8 KB
 * slot images (fr_conv_stage28_weight_bytes(1) bytes per conv); params: f32 [2*nblocks][10][128] = 9 border-class biases
 * (a plain bias nine times) + PReLU slope (1.0 = none). */
size_t fr_conv_stage28_weight_bytes(int nconv);
int fr_conv_stage28_pack(const void* w, void* out, fr_stream_t stream);
int fr_conv_stage28_f16(void* x, void* mid, const void* wstream, const float* params, int B, int nblocks, fr_stream_t stream);
/* 3x3 / stride 1 / pad 1 conv with 64 input channels on square images whose side is a multiple of 28 (IResNet-100: 112x112 and
 * 56x56; embed half of FaceAnalysis.get, /root/reference/infrenceServer.py:528), one launch per layer: a workgroup walks one
 * face (x one 64-cout group) region by region (14 x 28 output pixels), the next region's input halo arriving in a second LDS
 * buffer under the current region's K loop (conv_walk64.hip).  Same arithmetic and argument meaning as fr_conv_nhwc_f16
 * (bias_mode 1 = nine border-class biases [9][Cout]); wstream: the [Cout][9*64] f16 weights re-ordered by
 * fr_conv_walk64_pack (fr_conv_walk64_weight_bytes(Cout) bytes); Cout % 64 == 0; y must not alias x; residual may alias y. */
size_t fr_conv_walk64_weight_bytes(int Cout);
int fr_conv_walk64_pack(const void* w, void* out, int Cout, fr_stream_t stream);
int fr_conv_walk64_f16(const void* x, const void* wstream, void* y, const float* bias, int bias_mode, const float* slope,
                       const void* residual, int B, int HW, int Cout, fr_stream_t stream);
/* The fp8 twin of fr_conv_stage14_f16 (BASELINE config C5): the same run of residual blocks on
 * v_mfma_scale_f32_16x16x128_f8f6f4, the conv inputs as centred e4m3 codes resident in LDS (conv_stage14_f8.hip).  Math and
 * operation order of fr_conv_nhwc_f8 per conv: acc * oscale + 9-class bias -> PReLU -> (+ f16 residual) -> one rounding to
 * f16 -> the next conv's codes fp8((y - mu_next) * inv_sx_next).  x8: e4m3 codes NHWC [B,14,14,256] of the first conv's
 * input, x16: the same tensor in f16 (the first block's residual), y16: f16, receives every block's output in turn (the
 * residual stream goes through HBM: it does not fit beside the codes and the weight ring), x16 != y16.  wstream: the
 * convs' e4m3 weights [256][9*256] re-ordered by fr_conv_stage14_f8_pack into 18 x 32 KB slot images each
 * (fr_conv_stage14_f8_weight_bytes(1) bytes per conv).  params: f32 [2*nblocks][14][256]: row 0 oscale, 1 its reciprocal,
 * 2..10 the nine border-class biases, 11 the PReLU slope (1.0 = none), 12 mu of the NEXT conv's input, 13 whose first
 * element is 1 / sx of the next conv (fr_conv_stage14_f8_param_floats() floats per conv). */
size_t fr_conv_stage14_f8_weight_bytes(int nconv);
size_t fr_conv_stage14_f8_param_floats(void);
int fr_conv_stage14_f8_pack(const void* w8, void* out, fr_stream_t stream);
int fr_conv_stage14_f8(const void* x8, const void* x16, void* y16, const void* wstream, const float* params, int B,
                       int nblocks, fr_stream_t stream);
/* Calibration-time weight rounding for fr_conv_nhwc_f8 (GPTQ, Frantar et al. 2022): W f64 [rows][K] folded weights,
 * U f64 [K][K] = upper Cholesky factor of the inverse of the second-moment matrix of the conv's input patches (K order
 * as W's columns), sw f32 [rows] the per-row scale.  Q f32 [rows][K] receives values ON THE e4m3 GRID (w8 = Q exactly):
 * column k is rounded to the nearest sw * e4m3 value (ties to even, saturating) and its error is pushed onto the
 * columns still to come along row k of U, so that the conv OUTPUT error is what is minimised.  K <= 8192. */
int fr_gptq_round_e4m3(const double* W, const double* U, const float* sw, float* Q, int rows, int K, fr_stream_t stream);
/* Split-K tail of an ordinary conv (small batches): y = epi(sum_z partial[z]) with the epilogue of fr_conv_nhwc_f16
 * (bias or 9-class border bias, PReLU, residual, one rounding to f16).  partial: f32 [splitk][M][Cout]. */
int fr_conv_splitk_epilogue(const float* partial, int splitk, int M, int Cout, int Ho, int Wo,
                            const float* bias, int bias_mode, const float* slope, const void* residual,
                            void* y, fr_stream_t stream);
/* 3x3 / stride 1 / pad 1 conv of a forward of one to eight faces (single frames; same reference site, infrenceServer.py:528),
 * split along K INSIDE a workgroup: sixteen waves share one 16- (past 256 workgroups 32-, then 64-) pixel x 32-cout output tile, a sixteenth of K each, partials
 * summed through LDS in wave order, then the fused epilogue of fr_conv_nhwc_f16 ((border-class) bias -> PReLU -> + residual
 * -> one rounding to f16) - ONE launch where the split-K mode of fr_conv_nhwc_f16 + fr_conv_splitk_epilogue are two
 * (conv_inblock.hip).  Takes fr_conv_args with KH = KW = 3, stride 1, pad 1, Cin % 32 == 0, Cin <= 512, Cout % 32 == 0,
 * x2 / out_f32_partial NULL; splitk is ignored.  Differs from the other modes by f32 summation order only. */
int fr_conv_inblock_f16(const fr_conv_args* args, fr_stream_t stream);
/* A prepared run of convs in ONE call (single frames are launch-bound from an interpreted host: ~100 convs of a few
 * microseconds each).  Step kind 0: fr_conv_nhwc_f16(args).  Kind 1: the small-batch split-K form of the conv that
 * `args` describes as a whole (x, w, y, bias, bias_mode, slope, residual, splitk > 1, out_f32_partial = scratch of
 * splitk * M * Cout floats): the partials launch followed by fr_conv_splitk_epilogue into args.y.  Kind 2:
 * fr_conv_inblock_f16(args).  Same kernels, same order, same bits as the individual calls; stops at the first failing step. */
typedef struct {
    int kind;
    fr_conv_args args;
} fr_conv_step;
int fr_conv_sequence(const fr_conv_step* steps, int nsteps, fr_stream_t stream);
/* FC tail: sum split-K partials + bias -> embedding f32 [B,dim]; then
 * normed_embedding = embedding / ||embedding|| (Face.normed_embedding, infrenceServer.py:532) */
int fr_fc_reduce_l2norm(const float* partial, int splitk, int B, int dim, const float* bias,
                        float* embedding, float* normed, fr_stream_t stream);


/* ---------------------------------------------------------------- align ----
 * a-3  5-point similarity (closed-form least squares == Umeyama for proper rotations) and
 * bilinear warpAffine to size x size, border 0, uint8 rounding, then (x-127.5)/127.5,
 * BGR->RGB, NHWC f16 with C padded to 8 (the stem conv's packed input).  Stands in for
 * insightface face_align.norm_crop inside FaceAnalysis.get (infrenceServer.py:528).
 * frames: u8 [nframes,H,W,3] BGR; kps f32 [F,5,2]; frame_idx i32 [F]; count: device i32 (faces
 * at index >= *count are zero-filled) or NULL; out_u8_bgr u8 [F,size,size,3] and
 * M_out f32 [F,2,3] are optional. */
int fr_warp_affine_5pt(const uint8_t* frames, int nframes, int H, int W, const float* kps,
                       const int32_t* frame_idx, const int32_t* count, int F, int size,
                       void* out_f16_nhwc8, uint8_t* out_u8_bgr, float* M_out, fr_stream_t stream);

/* fixed-shape form: kps f32 [nframes*cap,5,2] in per-frame slots, counts i32 [nframes] on the device; slot
 * (frame, j) is warped iff j < counts[frame], else zero-filled: no host sync between detect and embed. */
int fr_warp_affine_5pt_slots(const uint8_t* frames, int nframes, int H, int W, const float* kps,
                             const int32_t* counts, int cap, int size, void* out_f16_nhwc8,
                             fr_stream_t stream);

/* --------------------------------------------------------------- detect ----
 * a-2  MTCNN cascade (detector half of FaceAnalysis.get, infrenceServer.py:528); the
 * conventions (resize, ordering, capacities) are those of oracle/detect.py. */
/* pyramid level: bilinear resize (half-pixel centres) + BGR->RGB + (x-127.5)*0.0078125 -> f32 NHWC(3) */
int fr_pyramid_resize_norm(const uint8_t* frames, int nframes, int H, int W, int hs, int ws,
                           float* out, fr_stream_t stream);
/* direct valid conv, f32 NHWC, + bias + PReLU (slope may be NULL).  w: [KH][KW][Cin][CoutP] with
 * CoutP = Cout rounded up to 16 (or 32), zero padded; bias/slope: [CoutP].
 * pool2 != 0: fused 2x2/s2 ceil-mode max pool after the PReLU (P-Net conv1).
 * nhead == 6: fused 1x1 head after the PReLU (P-Net conv3 -> conv4_1|conv4_2):
 * y = [.., 6] = head_b + act . head_w[32][6]. */
int fr_dconv_f32(const float* x, const float* w, const float* bias, const float* slope, float* y,
                 int B, int H, int W, int Cin, int Cout, int CoutP, int KH, int KW, int pool2,
                 const float* head_w, const float* head_b, int nhead, fr_stream_t stream);
/* MTCNN layers as LDS-tiled implicit GEMMs on v_mfma_f32_16x16x4_f32 (exact f32), the product path
 * of the detector.  `layer` selects a fixed geometry, with the following max pool FUSED where the net has one:
 * 0/1/2 = P-Net conv1(+PReLU+2x2 pool) / conv2 / conv3(+heads: y = [..,6]);
 * 10..14 = R-Net conv1(+3x3/s2 pool), conv2(+3x3/s2 pool), conv3, dense4, dense5_1|5_2;
 * 20..25 = O-Net conv1(+pool), conv2(+pool), conv3(+2x2 pool), conv4, dense5, dense6_1|6_2|6_3.
 * x f32 NHWC [B,H,W,Cin]; w packed [cout_group][tap][CinP][CP] (mtcnn.py _MConv); bias/slope padded to
 * the group size; slope NULL = no PReLU.  Channel counts are padded to multiples of 4 (P-Net conv1 writes
 * 12 channels, R/O-Net conv1 read 4).  Layer 0 takes `frames` (u8 BGR [B,FH,FW,3]) instead of x: the
 * pyramid level (H x W) is resized on the fly inside the tile load, with the same arithmetic as
 * fr_pyramid_resize_norm.  Blocks walk several tiles, prefetching the next tile into registers.
 * Layer 0 runs as its own kernel on v_mfma_f32_4x4x1 with a broadcast weight operand (pnet_conv1.hip: K = 27
 * exactly, one pixel per lane; needs slope != NULL); layer 3 is the same layer on the 16x16x4 form with the same
 * arguments and bit-identical outputs (same fma chain), kept for the A/B and the equality test.
 * counts / cap (R-/O-Net layers, optional): the batch is B / cap frames x cap crop slots and slot j of frame f
 * holds a candidate iff j < counts[f] (device i32); blocks that cover empty slots only do no work and leave those
 * outputs unwritten.  NULL: every image is computed.
 * y_split (layer 0 only, optional): a second copy of the output as split f16, 64 B per pixel = [hi ch0-7 | hi ch8-15 |
 * lo ch0-7 | lo ch8-15] with x = hi + lo and channels 12-15 zero - the operand format of fr_pnet23_split_f16. */
int fr_dconv_mfma_f32(int layer, const float* x, const float* w, const float* bias, const float* slope,
                      float* y, int B, int H, int W, const float* head_w, const float* head_b,
                      const uint8_t* frames, int FH, int FW, const int32_t* counts, int cap, void* y_split,
                      fr_stream_t stream);
/* R-Net / O-Net first layer fused with the crop that feeds it (net 0: 24x24 crop -> conv 3->28 -> PReLU -> 3x3/s2 ceil
 * pool -> y f32 [nframes*cap, 11, 11, 28]; net 1: 48x48 -> conv 3->32 -> ... -> [nframes*cap, 23, 23, 32]): the
 * arithmetic of fr_crop_resize_norm followed by layer 10 / 20 of fr_dconv_mfma_f32, bit for bit, without the crop
 * tensor in HBM.  boxes f32 [nframes*cap, 4] (trunc'ed inside), counts i32 [nframes]: slot j of frame f is computed
 * iff j < counts[f], other outputs stay unwritten.  w f32 [27][Cout]: k = (kh*3 + kw)*3 + channel (R, G, B);
 * bias / slope f32 [Cout]. */
int fr_crop_conv1_f32(int net, const uint8_t* frames, int nframes, int H, int W, const float* boxes,
                      const int32_t* counts, int cap, const float* w, const float* bias, const float* slope,
                      float* y, fr_stream_t stream);
/* The same layer with its pooled map written as SPLIT f16 (x = hi + lo, hi = f16(x), lo = f16(x - hi)):
 * y_split [nframes*cap][P*P pixels][hi 32 channels | lo 32 channels], 128 B per pixel, channels >= Cout zero (P = 11 / 23) -
 * the operand format of fr_ro_conv2_split.  Same slots computed / left unwritten as fr_crop_conv1_f32.
 * conv_f16 = 0: the conv itself is the f32 form's (the map is hi + lo of fr_crop_conv1_f32's to 2^-21); 1: the conv runs on
 * the f16 matrix cores with split-precision operands too (~1e-6 of the map's scale from the f32 form). */
int fr_crop_conv1_split(int net, const uint8_t* frames, int nframes, int H, int W, const float* boxes,
                        const int32_t* counts, int cap, const float* w, const float* bias, const float* slope,
                        void* y_split, int conv_f16, fr_stream_t stream);
/* fr_crop_conv1_f32 for a compact LIST of slots: row i of y is the f32 map of slot list[i], i < min(*list_count, list_cap)
 * (device-side count; boxes / frames are indexed by list[i], frame = list[i] / cap). */
int fr_crop_conv1_list_f32(int net, const uint8_t* frames, int nframes, int H, int W, const float* boxes, int cap,
                           const int32_t* list, const int32_t* list_count, int list_cap, const float* w,
                           const float* bias, const float* slope, float* y, fr_stream_t stream);
/* R-Net / O-Net SECOND layer on the f16 matrix cores with split-precision operands (three v_mfma_f32_16x16x32_f16 per
 * product term, f32 accumulate; heads differ from the f32 layers by ~1e-6): net 0: [.,11,11,28] -> conv 3x3 -> PReLU ->
 * 3x3/s2 pool -> y f32 [nslots,4,4,48]; net 1: [.,23,23,32] -> ... -> [nslots,10,10,64] - the outputs of layers 11 / 21 of
 * fr_dconv_mfma_f32.  x_split: fr_crop_conv1_split's map; w f32 [Cout][9 taps][32 channels] (channels >= Cin zero);
 * counts / cap as fr_dconv_mfma_f32 (nslots = frames x cap).  The cascade's keep / reject decisions stay those of f32
 * arithmetic through the exact pass below.  zero_word (optional): a device int32 the kernel clears - the list counter of
 * the fr_ro_margin_list call that follows in the stream (saves a memset launch). */
int fr_ro_conv2_split(int net, const void* x_split, const float* w, const float* bias, const float* slope, float* y,
                      int nslots, const int32_t* counts, int cap, int32_t* zero_word, fr_stream_t stream);
/* The small tail layers of R-Net / O-Net as one split-precision GEMM kernel on the f16 matrix cores (x = hi + lo in f16,
 * three MFMAs per product term, f32 accumulate; ~1e-6 of the output's scale from the f32 layers): layer 12 = R-Net conv3
 * (2x2, 48 -> 64, [.,4,4,48] -> [.,3,3,64]), 13 = R-Net dense4 ([.,3,3,64] -> [.,128]), 22 = O-Net conv3 (3x3, 64 -> 64,
 * [.,10,10,64] -> 8x8 -> 2x2/s2 max pool -> [.,4,4,64]), 23 = O-Net conv4 (2x2, 64 -> 128,
 * [.,4,4,64] -> [.,3,3,128]), 24 = O-Net dense5 ([.,3,3,128] -> [.,256]) - the layers of the same ids of fr_dconv_mfma_f32,
 * + bias + PReLU.  w_packed: fr_ro_gemm_pack(layer, w) of the f32 weights [Cout][K], K = (kh, kw, channel) ascending
 * (fr_ro_gemm_weight_bytes(layer) bytes).  counts / cap as fr_dconv_mfma_f32.  Batch path only: see fr_ro_conv2_split. */
size_t fr_ro_gemm_weight_bytes(int layer);
int fr_ro_gemm_pack(int layer, const float* w, void* out, fr_stream_t stream);
int fr_ro_gemm_split(int layer, const float* x, const void* w_packed, const float* bias, const float* slope, float* y,
                     int nslots, const int32_t* counts, int cap, fr_stream_t stream);
/* The exact pass's work list: the valid slots whose logit difference head[s][1] - head[s][0] lies within `margin` of
 * logit_thr = ln(t / (1 - t)) (t: the stage's probability threshold) are appended to list (any order); *list_count = how
 * many there are (may exceed list_cap: entries past it are dropped, consumers clamp).  head f32 [nframes*cap][nhead].
 * *list_count must be 0 on entry (fr_ro_conv2_split's zero_word, or the caller's memset).
 * OVERFLOW: the slots dropped past list_cap keep their split-precision heads - their threshold decision is then NOT the f32
 * one.  Nothing signals it on the device; a caller that must know compares *list_count with list_cap after the cascade
 * (MTCNNHIP._ro_lists; the batch-path tests and bench.py's line do) and raises list_cap or re-runs the stage on the f32 layers.
 * The f32 layers that follow take the list's own counter as their `counts` with cap = list_cap: fr_dconv_mfma_f32 and
 * fr_ro_gemm_split only test slot < counts[f], so a counter ABOVE list_cap is tolerated there (every one of the list_cap rows
 * is computed, none beyond). */
int fr_ro_margin_list(const float* head, int nhead, const int32_t* counts, int nframes, int cap, float logit_thr,
                      float margin, int32_t* list, int32_t* list_count, int list_cap, fr_stream_t stream);
/* dst[list[i]][:] = src[i][:] for i < min(*list_count, list_cap): the exactly re-evaluated head rows go back to their slots. */
int fr_ro_scatter_rows(const float* src, const int32_t* list, const int32_t* list_count, int list_cap, int ncols,
                       float* dst, fr_stream_t stream);
/* P-Net conv2 -> PReLU -> conv3 -> PReLU -> heads fused, on the f16 matrix cores with split-precision operands
 * (x = hi + lo in f16, three MFMAs per product term, f32 accumulate; ~1e-5 logit error), followed by an EXACT f32
 * re-evaluation of every cell whose logit1 - logit0 >= refine_logit_thr (pass ln(t/(1-t)) - 2e-3 for threshold t):
 * every cell that can be kept by fr_pnet_candidates then carries exact f32 logits and regressions, every other cell is
 * below the threshold by more than the approximation error.  x1: P-Net conv1 output (layer 0 of fr_dconv_mfma_f32),
 * f32 [B,H1,W1,12]; w2 [16][10][16] / w3 [32][10][16]: (cout, tap, channel) with tap 9 and unused channels zero;
 * x1s: the same map as split f16 (y_split of layer 0), streamed into LDS by LDS-DMA.
 * b/s: bias and PReLU slope; hw [32][6], hb [6]: conv4_1|conv4_2.  head: f32 [B,H1-4,W1-4,6].
 * all_heads: 0 = only the rows of re-evaluated cells are written (all fr_pnet_candidates reads when it is given the
 * workspace as `dl` and refine_logit_thr as `dl_min`); 1 = the approximate heads of every other cell too (tests, traces).
 * refine_band_hi: <= refine_logit_thr (pass -INFINITY): as above.  > refine_logit_thr: only the cells with refine_logit_thr <=
 * logit1 - logit0 <= refine_band_hi - the band around the face threshold, pass ln(t/(1-t)) + 2e-3 - are re-evaluated
 * exactly (every keep / reject decision is still that of f32 arithmetic); the cells above the band keep their
 * split-precision rows, which the kernel then writes for every cell >= refine_logit_thr (~2e-6 from the f32 heads).
 * refined_count (optional device i32, accumulated): number of re-evaluated cells.
 * workspace: fr_pnet23_workspace_bytes(B, H1, W1) bytes; its first B*(H1-4)*(W1-4) floats are the logit differences
 * (the `dl` argument of fr_pnet_candidates), behind them the per-block lists of the cells the exact pass re-evaluates.
 * all_heads bit 1 (value 2): the exact pass is NOT launched here - fr_pnet_finish_levels runs it for every level at once. */
size_t fr_pnet23_workspace_bytes(int B, int H1, int W1);
int fr_pnet23_split_f16(const float* x1, const void* x1s, int B, int H1, int W1, const float* w2, const float* b2, const float* s2,
                        const float* w3, const float* b3, const float* s3, const float* hw, const float* hb,
                        float* head, int all_heads, float refine_logit_thr, float refine_band_hi, int32_t* refined_count,
                        void* workspace, size_t workspace_bytes, fr_stream_t stream);
/* max pool, ceil mode, f32 NHWC */
int fr_maxpool_f32(const float* x, float* y, int B, int H, int W, int C, int k, int stride,
                   fr_stream_t stream);
/* P-Net level -> per-frame candidate lists: head f32 [nframes,hc,wc,6] = (logit0, logit1, reg0..3);
 * keeps cells with softmax face prob >= thr in raster order (first `cap`), emitting
 * box = floor((2*cell + {1,12}) / scale), score, reg.  boxes [nframes,cap,4], scores [nframes,cap],
 * regs [nframes,cap,4], counts i32 [nframes]; block_counts: i32 scratch [nframes*ceil(hc*wc/256)];
 * prob_out (optional) f32 [nframes,hc,wc].  dl / dl_min (optional, with fr_pnet23_split_f16): f32 [nframes,hc,wc]
 * approximate logit1 - logit0; cells with dl < dl_min are known to be below the threshold and their head rows are
 * not read (pass the workspace of fr_pnet23_split_f16 and its refine_logit_thr). */
int fr_pnet_candidates(const float* head, int nframes, int hc, int wc, float scale, float thr, int cap,
                       float* boxes, float* scores, float* regs, int32_t* counts,
                       int32_t* block_counts, float* prob_out, const float* dl, float dl_min, fr_stream_t stream);
/* per-list sort (descending score, ties by slot) + greedy NMS (mode 0: IoU, 1: IoMin).
 * Input list l = nseg segments of seg_cap slots (segment s at list index l*nseg+s, or s*L+l when
 * seg_major), counts i32 [L*nseg] (nseg*seg_cap <= 4096);
 * aux f32 [.., naux] rides along.  Output: first max_keep survivors in score order,
 * boxes_out [L,cap_out,4], scores_out [L,cap_out], aux_out [L,cap_out,naux], counts_out [L]. */
int fr_sort_nms(const float* boxes, const float* scores, const float* aux, int naux,
                const int32_t* counts, int L, int nseg, int seg_cap, int seg_major, float thr, int mode,
                int max_keep,
                float* boxes_out, float* scores_out, float* aux_out, int32_t* counts_out,
                int cap_out, fr_stream_t stream);
/* box refinement in place, regs = aux[..,0:4]: mode 0 = stage-1 regression (w = x2-x1) + square;
 * mode 1 = bbreg (w = x2-x1+1) + square; mode 2 = bbreg only. */
int fr_box_refine(float* boxes, const float* aux, int naux, const int32_t* counts, int L, int cap,
                  int mode, fr_stream_t stream);
/* zero-padded crop of trunc(box) (1-based inclusive) + bilinear resize to size x size + normalise
 * -> f32 NHWC [nframes*cap,size,size,4] (RGB + a zero channel); slots >= count are zero-filled. */
int fr_crop_resize_norm(const uint8_t* frames, int nframes, int H, int W, const float* boxes,
                        const int32_t* counts, int cap, int size, float* out, fr_stream_t stream);
/* R/O-Net decision: head f32 [L*cap,nh] = (logit0, logit1, reg0..3[, lm0..9]); keeps slots with
 * softmax face prob > thr in slot order, emitting trunc(box), score, aux = (reg0..3[, landmarks
 * (x1,y1)..(x5,y5) in frame coordinates]) (nh,naux) = (6,4) or (16,14). */
int fr_stage_select(const float* boxes, const float* head, int nh, const int32_t* counts, int L, int cap,
                    float thr, float* boxes_out, float* scores_out, float* aux_out, int naux,
                    int32_t* counts_out, float* prob_out, fr_stream_t stream);

/* Band mode with conv1 on the f16 matrix cores (round 4, batches).  fr_pnet_conv1_band mode 0: P-Net conv1 (+ the pyramid
 * resize, PReLU, 2x2 ceil pool: layer 0 of fr_dconv_mfma_f32) with split-precision operands - writes the split map y_split
 * only (y optional: the f32 view of the same ~1e-6-accurate values).  The exact pass then needs an exact f32 map under the
 * windows of the cells on its lists: fr_pnet_band_tiles marks the conv1 tiles (8 x 32 map pixels) those windows touch
 * (tbuf: 1 + ceil(tiles / 32) int32 = [count | bitmap], zeroed here; tiles: int32 [fr_pnet_band_tiles_count(B, H1, W1)]), and
 * fr_pnet_conv1_band mode 1 runs the EXACT f32 kernel over that list (list = tiles, list_count = tbuf, list_cap = the tile
 * count), writing y for those tiles only.  Call order per level: mode 0, fr_pnet23_split_f16 (all_heads | 2, band), band_tiles,
 * mode 1, fr_pnet_finish_levels (one level).  H, W: the level's size; H1, W1: the conv1 map's. */
int fr_pnet_conv1_band(int mode, const uint8_t* frames, int B, int FH, int FW, int H, int W, const float* w, const float* bias,
                       const float* slope, float* y, void* y_split, const int32_t* list, const int32_t* list_count, int list_cap,
                       fr_stream_t stream);
size_t fr_pnet_band_tiles_count(int B, int H1, int W1);
int fr_pnet_band_tiles(const void* workspace, int B, int H1, int W1, int32_t* tbuf, int32_t* tiles, fr_stream_t stream);
/* The exact pass (deferred by all_heads bit 1) and fr_pnet_candidates for ALL pyramid levels of a batch in three launches
 * instead of three per level: the same cells, the same ordered compaction per level and frame.  A level entry repeats what
 * its fr_pnet23_split_f16 / fr_pnet_candidates calls would have been given (workspace = that level's, unchanged since;
 * block_counts: i32 [nframes * ceil((H1-4)*(W1-4)/256)]).  At most 16 levels. */
typedef struct {
    const float* x1; float* head; void* workspace;
    int H1, W1; float scale;
    float* boxes; float* scores; float* regs; int32_t* counts; int32_t* block_counts;
} fr_pnet_level;
int fr_pnet_finish_levels(const fr_pnet_level* levels, int nlevels, int nframes, const float* w2, const float* b2,
                          const float* s2, const float* w3, const float* b3, const float* s3, const float* hw,
                          const float* hb, float thr, int cap, float dl_min, int32_t* refined_count, fr_stream_t stream);

/* A recorded run of detector calls replayed by ONE C call (an eager single-frame get() is bound by the interpreter: ~50
 * ctypes calls per frame; FaceAnalysis.get, infrenceServer.py:528).  `fn` names the entry point, `a` carries its arguments
 * as 8-byte slots in declaration order: pointers and size_t as they are, ints sign-extended, floats as their IEEE bits in
 * the low word.  Two further ids record / wait for an event of the CALLER's (no allocation here): a recorded call may deal
 * the pyramid levels over side streams.  Same kernels, same order per stream, same bits as the individual calls; stops at
 * the first failing call.  `nargs` must be the entry point's arity (18 / 21 / 16 / 18 / 8 / 13 / 14, 2 for the event ids):
 * a short or malformed entry is refused before anything is launched. */
enum { FR_FN_DCONV_MFMA = 1, FR_FN_PNET23 = 2, FR_FN_PNET_CANDIDATES = 3, FR_FN_SORT_NMS = 4, FR_FN_BOX_REFINE = 5,
       FR_FN_CROP_CONV1 = 6, FR_FN_STAGE_SELECT = 7,
       FR_FN_EVENT_RECORD = 8 /* a = (hipEvent_t, hipStream_t) */, FR_FN_STREAM_WAIT = 9 /* a = (hipStream_t, hipEvent_t) */ };
typedef struct {
    int32_t fn; int32_t nargs;
    uint64_t a[22];
} fr_call;
int fr_detect_sequence(const fr_call* calls, int ncalls);

#ifdef __cplusplus
}
#endif
#endif

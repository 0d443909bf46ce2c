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
 *   - no global mutable state.
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
 *      out_score[f] = best dot (or -1).  workspace: fr_gallery_match_workspace() bytes. */
size_t fr_gallery_match_workspace(int F, int64_t N);
int fr_gallery_match_f32(const float* Q, const float* G, int F, int64_t N, int D,
                         int64_t row_offset, int64_t* out_idx, float* out_score,
                         void* workspace, size_t workspace_bytes, fr_stream_t stream);
/* f32 -> f16 row conversion for building the device-resident gallery (infrenceServer.py:271) */
int fr_f32_to_f16(const float* x, void* out, int64_t n, fr_stream_t stream);
/* a-8  known = best_id and best >= thr  (infrenceServer.py:545; peopleCount.py:876-882):
 *      decision[f] = 1 recognised, 0 unknown, 2 dropped (counting path's [unknown_thr, thr) band;
 *      pass unknown_thr = thr for the live path). */
int fr_match_decide(const int64_t* idx, const float* score, int F, float thr, float unknown_thr,
                    int32_t* decision, fr_stream_t stream);


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
 * (no epilogue), used for the FC 25088->512 (H=W=1, Cin=25088). */
typedef struct {
    const void* x; const void* w; void* y;
    const float* bias; const float* slope; const void* residual;
    float* out_f32_partial;
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int bias_mode; int splitk;
} fr_conv_args;
int fr_conv_nhwc_f16(const fr_conv_args* args, fr_stream_t stream);
/* FC tail: sum split-K partials + bias -> embedding f32 [B,dim]; then
 * normed_embedding = embedding / ||embedding|| (Face.normed_embedding, infrenceServer.py:532) */
int fr_fc_reduce_l2norm(const float* partial, int splitk, int B, int dim, const float* bias,
                        float* embedding, float* normed, fr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif

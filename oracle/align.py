"""Oracle: 5-point alignment (Umeyama similarity, float64 SVD) + bilinear warp.

PARITY UNPINNED (see oracle/__init__.py): stands in for insightface's
``face_align.norm_crop`` inside ``FaceAnalysis.get``
(/root/reference/infrenceServer.py:528).  Template = the public 112x112 ArcFace
5-point destination.  The warp is ideal float bilinear
with constant-0 border, rounded to uint8 (half up) like an 8-bit warpAffine,
then (x-127.5)/127.5 and BGR->RGB.

Test infrastructure only.
"""
import numpy as np

ARCFACE_DST = np.array([[38.2946, 51.6963], [73.5318, 51.5014], [56.0252, 71.7366],
                        [41.5493, 92.3655], [70.7299, 92.2041]], np.float64)


def umeyama(src, dst):
    """Least-squares similarity (Umeyama 1991) mapping src -> dst; returns 2x3 float64."""
    src = np.asarray(src, np.float64); dst = np.asarray(dst, np.float64)
    n = src.shape[0]
    mu_s, mu_d = src.mean(0), dst.mean(0)
    sc, dc = src - mu_s, dst - mu_d
    A = dc.T @ sc / n
    d = np.ones(2)
    if np.linalg.det(A) < 0:
        d[1] = -1
    U, S, Vt = np.linalg.svd(A)
    R = U @ np.diag(d) @ Vt
    var_s = (sc ** 2).sum() / n
    scale = (S * d).sum() / var_s
    M = np.zeros((2, 3))
    M[:, :2] = scale * R
    M[:, 2] = mu_d - scale * R @ mu_s
    return M


def warp_affine_u8(img_u8, M, size=112):
    """dst(x,y) = bilinear(img, M^-1 (x,y)); border 0; returns uint8 [size,size,C].
    PRODUCT DECISION: ideal float bilinear sampling (cv2.warpAffine uses 5-bit fixed-point weights; not available
    here to pin against)."""
    H, W = img_u8.shape[:2]
    A = np.vstack([M, [0, 0, 1]])
    Ai = np.linalg.inv(A)
    ys, xs = np.mgrid[0:size, 0:size].astype(np.float64)
    sx = Ai[0, 0] * xs + Ai[0, 1] * ys + Ai[0, 2]
    sy = Ai[1, 0] * xs + Ai[1, 1] * ys + Ai[1, 2]
    x0 = np.floor(sx); y0 = np.floor(sy)
    wx = (sx - x0)[..., None]; wy = (sy - y0)[..., None]
    x0 = x0.astype(np.int64); y0 = y0.astype(np.int64)
    img = img_u8.astype(np.float64)

    def px(y, x):
        ok = (y >= 0) & (y < H) & (x >= 0) & (x < W)
        v = img[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)]
        return v * ok[..., None]

    v = (1 - wy) * ((1 - wx) * px(y0, x0) + wx * px(y0, x0 + 1)) + \
        wy * ((1 - wx) * px(y0 + 1, x0) + wx * px(y0 + 1, x0 + 1))
    return np.clip(np.floor(v + 0.5), 0, 255).astype(np.uint8)


def norm_crop(frame_bgr_u8, kps, size=112):
    M = umeyama(kps, ARCFACE_DST)
    return warp_affine_u8(frame_bgr_u8, M, size), M


def crop_to_net(crop_bgr_u8):
    """uint8 HWC BGR -> float32 [3,H,W] RGB, (x-127.5)/127.5."""
    x = (crop_bgr_u8[:, :, ::-1].astype(np.float32) - np.float32(127.5)) / np.float32(127.5)
    return np.ascontiguousarray(x.transpose(2, 0, 1))

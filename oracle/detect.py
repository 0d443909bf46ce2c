"""Oracle: MTCNN detection pipeline (numpy float32 host logic + torch-CPU nets).

PARITY UNPINNED (see oracle/__init__.py): stands in for the detector half of
``FaceAnalysis.get`` (/root/reference/infrenceServer.py:528), restating the
published MTCNN cascade: image pyramid (minsize 20,
factor 0.709), P-Net FCN -> threshold -> NMS 0.5 per scale / 0.7 across scales
-> regression -> square -> R-Net 24x24 (thr, NMS 0.7) -> O-Net 48x48 (thr,
NMS 0.7 'Min') -> 5 landmarks.  Conventions fixed here (they are the spec the
HIP kernels implement):

* frames are BGR uint8 HWC (reference camera frames); networks see RGB,
  (x - 127.5) * 0.0078125;
* resize = bilinear, half-pixel centres, edge clamp, no antialias, float32;
* candidate order = descending score, ties by slot index (stable);
* fixed capacities (per frame): CAP_SCALE candidates per scale taken in raster
  order before NMS, KEEP_SCALE survivors per scale, CAP_P boxes after stage 1,
  CAP_R after stage 2, CAP_O faces (each the first K in score order).

Test infrastructure only.
"""
import math

import numpy as np
import torch

from . import nets

F32 = np.float32


def pyramid_scales(h, w, minsize=20, factor=0.709):
    m = 12.0 / minsize
    minl = min(h, w) * m
    scales = []
    k = 0
    while minl >= 12:
        scales.append(m * factor ** k)
        minl *= factor
        k += 1
    return scales


# PRODUCT DECISION: ideal float32 bilinear (half-pixel centres, edge clamp, no antialias) for the pyramid levels and
# the R-/O-Net crops.  The third-party path behind the reference (insightface/OpenCV) interpolates with fixed-point
# weights / area averaging; neither is available here to pin against, so this file is the definition.
def resize_bilinear(img, oh, ow):
    """img float32 [H,W,C] -> [oh,ow,C]; half-pixel centres, edge clamp."""
    H, W = img.shape[:2]
    ry, rx = F32(H) / F32(oh), F32(W) / F32(ow)
    fy = (np.arange(oh, dtype=F32) + F32(0.5)) * ry - F32(0.5)
    fx = (np.arange(ow, dtype=F32) + F32(0.5)) * rx - F32(0.5)
    y0 = np.floor(fy); x0 = np.floor(fx)
    wy = (fy - y0).astype(F32)[:, None, None]; wx = (fx - x0).astype(F32)[None, :, None]
    y0 = y0.astype(np.int64); x0 = x0.astype(np.int64)
    y0c = np.clip(y0, 0, H - 1); y1c = np.clip(y0 + 1, 0, H - 1)
    x0c = np.clip(x0, 0, W - 1); x1c = np.clip(x0 + 1, 0, W - 1)
    p00 = img[y0c][:, x0c]; p01 = img[y0c][:, x1c]
    p10 = img[y1c][:, x0c]; p11 = img[y1c][:, x1c]
    one = F32(1)
    top = (one - wx) * p00 + wx * p01
    bot = (one - wx) * p10 + wx * p11
    return ((one - wy) * top + wy * bot).astype(F32)


def _to_net(x_hwc_rgb):
    """float32 HWC RGB [0,255] -> normalised NCHW tensor."""
    x = (x_hwc_rgb - F32(127.5)) * F32(0.0078125)
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None]


def nms(boxes, scores, thr, mode):
    """Greedy NMS over boxes already in priority order (descending score).

    boxes float32 [N,4] (x1,y1,x2,y2, +1 area convention).  Returns kept indices.
    """
    n = boxes.shape[0]
    area = (boxes[:, 2] - boxes[:, 0] + F32(1)) * (boxes[:, 3] - boxes[:, 1] + F32(1))
    alive = np.ones(n, bool)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        j = np.arange(i + 1, n)
        xx1 = np.maximum(boxes[i, 0], boxes[j, 0]); yy1 = np.maximum(boxes[i, 1], boxes[j, 1])
        xx2 = np.minimum(boxes[i, 2], boxes[j, 2]); yy2 = np.minimum(boxes[i, 3], boxes[j, 3])
        w = np.maximum(F32(0), xx2 - xx1 + F32(1)); h = np.maximum(F32(0), yy2 - yy1 + F32(1))
        inter = (w * h).astype(F32)
        if mode == "min":
            o = inter / np.minimum(area[i], area[j])
        else:
            o = inter / (area[i] + area[j] - inter)
        alive[j[o > F32(thr)]] = False
    return np.asarray(keep, np.int64)


def _order(scores):
    """Descending score, ties by original index (stable)."""
    return np.argsort(-scores, kind="stable")


def rerec(b):
    """Square a box about its centre (in place on a copy)."""
    b = b.copy()
    h = b[:, 3] - b[:, 1]; w = b[:, 2] - b[:, 0]
    l = np.maximum(w, h)
    b[:, 0] = b[:, 0] + w * F32(0.5) - l * F32(0.5)
    b[:, 1] = b[:, 1] + h * F32(0.5) - l * F32(0.5)
    b[:, 2] = b[:, 0] + l
    b[:, 3] = b[:, 1] + l
    return b


def bbreg(b, reg):
    w = b[:, 2] - b[:, 0] + F32(1); h = b[:, 3] - b[:, 1] + F32(1)
    out = b.copy()
    out[:, 0] = b[:, 0] + reg[:, 0] * w; out[:, 1] = b[:, 1] + reg[:, 1] * h
    out[:, 2] = b[:, 2] + reg[:, 2] * w; out[:, 3] = b[:, 3] + reg[:, 3] * h
    return out


def crop_resize(frame_rgb_f32, box_i, size):
    """Zero-padded crop of the (1-based inclusive) integer box, bilinear to size x size."""
    H, W = frame_rgb_f32.shape[:2]
    x1, y1, x2, y2 = [int(v) for v in box_i]
    tw, th = x2 - x1 + 1, y2 - y1 + 1
    if tw <= 0 or th <= 0:
        return None
    tmp = np.zeros((th, tw, 3), F32)
    ys, xs = y1 - 1, x1 - 1
    sy0, sy1 = max(ys, 0), min(ys + th, H)
    sx0, sx1 = max(xs, 0), min(xs + tw, W)
    if sy1 > sy0 and sx1 > sx0:
        tmp[sy0 - ys:sy1 - ys, sx0 - xs:sx1 - xs] = frame_rgb_f32[sy0:sy1, sx0:sx1]
    return resize_bilinear(tmp, size, size)


def detect(frame_bgr, pstate, rstate, ostate, minsize=20, factor=0.709,
           thresholds=(0.6, 0.7, 0.7), cap_scale=2048, keep_scale=256, cap_p=512, cap_r=64, cap_o=16,
           trace=None):
    """Returns (bbox float32[F,4], score float32[F], kps float32[F,5,2])."""
    H, W = frame_bgr.shape[:2]
    rgb = frame_bgr[:, :, ::-1].astype(F32)
    t0, t1, t2 = [F32(t) for t in thresholds]
    # ---- stage 1: pyramid + P-Net
    all_boxes, all_scores, all_reg = [], [], []
    for s in pyramid_scales(H, W, minsize, factor):
        hs, ws = int(math.ceil(H * s)), int(math.ceil(W * s))
        im = resize_bilinear(rgb, hs, ws)
        prob, reg = nets.pnet_forward(pstate, _to_net(im))
        prob = prob[0].numpy(); reg = reg[0].numpy()
        if trace is not None:
            trace.setdefault("pnet_prob", []).append(prob)
            trace.setdefault("pnet_reg", []).append(reg)
        ys, xs = np.nonzero(prob >= t0)            # raster order
        # PRODUCT DECISION, not part of the published MTCNN algorithm: a pyramid level hands on at most cap_scale
        # candidates, the FIRST cap_scale cells in raster order (the HIP path has fixed-capacity lists).  On overflow
        # (synthetic weights on a 4K frame: tests/test_gpu_detect.py::test_4k_...) cells further down the level are
        # never candidates.  Trained weights on real frames stay far below the cap; raise cap_scale otherwise.
        ys, xs = ys[:cap_scale], xs[:cap_scale]
        if ys.size == 0:
            continue
        sc = prob[ys, xs].astype(F32)
        s32 = F32(s)
        x1 = np.floor((F32(2) * xs.astype(F32) + F32(1)) / s32)
        y1 = np.floor((F32(2) * ys.astype(F32) + F32(1)) / s32)
        x2 = np.floor((F32(2) * xs.astype(F32) + F32(12)) / s32)
        y2 = np.floor((F32(2) * ys.astype(F32) + F32(12)) / s32)
        b = np.stack([x1, y1, x2, y2], 1).astype(F32)
        r = reg[:, ys, xs].T.astype(F32)
        o = _order(sc); b, sc, r = b[o], sc[o], r[o]
        k = nms(b, sc, 0.5, "union")[:keep_scale]
        all_boxes.append(b[k]); all_scores.append(sc[k]); all_reg.append(r[k])
    empty = (np.zeros((0, 4), F32), np.zeros((0,), F32), np.zeros((0, 5, 2), F32))
    if not all_boxes:
        return empty
    b = np.concatenate(all_boxes); sc = np.concatenate(all_scores); r = np.concatenate(all_reg)
    o = _order(sc); b, sc, r = b[o], sc[o], r[o]
    k = nms(b, sc, 0.7, "union")[:cap_p]
    b, sc, r = b[k], sc[k], r[k]
    rw = b[:, 2] - b[:, 0]; rh = b[:, 3] - b[:, 1]
    b = np.stack([b[:, 0] + r[:, 0] * rw, b[:, 1] + r[:, 1] * rh,
                  b[:, 2] + r[:, 2] * rw, b[:, 3] + r[:, 3] * rh], 1).astype(F32)
    b = rerec(b)
    if trace is not None:
        trace["stage1_boxes"] = b.copy(); trace["stage1_scores"] = sc.copy()
    # ---- stage 2: R-Net
    bi = np.trunc(b).astype(F32)
    crops = [crop_resize(rgb, bb, 24) for bb in bi]
    ok = np.asarray([c is not None for c in crops], bool)
    if not ok.any():
        return empty
    x = torch.cat([_to_net(c) for c in crops if c is not None])
    prob, reg = nets.rnet_forward(rstate, x)
    score = np.zeros(len(crops), F32); regs = np.zeros((len(crops), 4), F32)
    score[ok] = prob.numpy(); regs[ok] = reg.numpy()
    if trace is not None:
        trace["rnet_score"] = score.copy(); trace["rnet_reg"] = regs.copy()
    p = np.nonzero(score > t1)[0]
    if p.size == 0:
        return empty
    b, sc, r = bi[p], score[p], regs[p]
    o = _order(sc); b, sc, r = b[o], sc[o], r[o]
    k = nms(b, sc, 0.7, "union")[:cap_r]
    b = rerec(bbreg(b[k], r[k])); sc = sc[k]
    if trace is not None:
        trace["stage2_boxes"] = b.copy(); trace["stage2_scores"] = sc.copy()
    # ---- stage 3: O-Net
    bi = np.trunc(b).astype(F32)
    crops = [crop_resize(rgb, bb, 48) for bb in bi]
    ok = np.asarray([c is not None for c in crops], bool)
    if not ok.any():
        return empty
    x = torch.cat([_to_net(c) for c in crops if c is not None])
    prob, reg, lm = nets.onet_forward(ostate, x)
    score = np.zeros(len(crops), F32); regs = np.zeros((len(crops), 4), F32)
    lms = np.zeros((len(crops), 10), F32)
    score[ok] = prob.numpy(); regs[ok] = reg.numpy(); lms[ok] = lm.numpy()
    if trace is not None:
        trace["onet_score"] = score.copy(); trace["onet_reg"] = regs.copy(); trace["onet_lm"] = lms.copy()
    p = np.nonzero(score > t2)[0]
    if p.size == 0:
        return empty
    b, sc, r, lm = bi[p], score[p], regs[p], lms[p]
    w = (b[:, 2] - b[:, 0] + F32(1))[:, None]; h = (b[:, 3] - b[:, 1] + F32(1))[:, None]
    px = w * lm[:, 0:5] + b[:, 0:1] - F32(1)
    py = h * lm[:, 5:10] + b[:, 1:2] - F32(1)
    kps = np.stack([px, py], 2).astype(F32)
    b = bbreg(b, r)
    o = _order(sc); b, sc, kps = b[o], sc[o], kps[o]
    k = nms(b, sc, 0.7, "min")[:cap_o]
    return b[k].astype(F32), sc[k].astype(F32), kps[k].astype(F32)

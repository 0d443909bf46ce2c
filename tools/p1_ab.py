"""A/B of P-Net conv1: layer 0 (4x4x1-MFMA kernel, pnet_conv1.hip) against layer 3 (the 16x16x4 form in dconv_mfma.hip).
Bit-equality of the pooled map and of the split-f16 copy on several frame / level sizes, then timing on 64 x 1080p."""
import math
import sys
import time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from facerecognition_infrenceengine_amd import _lib, weights
from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP, pyramid_scales

dev = "cuda:0"
st = weights.synth_mtcnn_states(seed=1234)
det = MTCNNHIP(*st, device=dev)
lib = det.lib
p1 = det.p1


def run(layer, frames, hs, ws, split=True):
    N = frames.shape[0]
    h, w = p1.out_hw(hs, ws)
    y = torch.full((N, h, w, 12), float("nan"), dtype=torch.float32, device=dev)
    xs = torch.full((N, h, w, 64), 0x7f, dtype=torch.uint8, device=dev) if split else None
    rc = lib.fr_dconv_mfma_f32(layer, None, _lib.ptr(p1.w), _lib.ptr(p1.b), _lib.ptr(p1.slope), _lib.ptr(y), N, hs, ws,
                               None, None, _lib.ptr(frames), frames.shape[1], frames.shape[2], None, 0, _lib.ptr(xs),
                               _lib.stream_ptr())
    assert rc == 0, rc
    return y, xs


g = torch.Generator(device=dev).manual_seed(3)
bad = 0
for (N, H, W) in [(1, 480, 640), (3, 120, 160), (2, 1080, 1920), (1, 37, 53), (2, 250, 333)]:
    frames = torch.randint(0, 256, (N, H, W, 3), generator=g, device=dev, dtype=torch.uint8)
    for sc in pyramid_scales(H, W):
        hs, ws = int(math.ceil(H * sc)), int(math.ceil(W * sc))
        if hs < 3 or ws < 3:
            continue
        y0, s0 = run(0, frames, hs, ws)
        y3, s3 = run(3, frames, hs, ws)
        torch.cuda.synchronize()
        ey = not torch.equal(y0.view(torch.int32), y3.view(torch.int32))
        es = not torch.equal(s0, s3)
        if ey or es:
            bad += 1
            d = (y0 - y3).abs()
            print(f"MISMATCH N={N} {H}x{W} level {hs}x{ws}: y {'differs' if ey else 'ok'} max|d|={float(d.nan_to_num(1e9).max()):.3e} "
                  f"n={int((y0.view(torch.int32) != y3.view(torch.int32)).sum())} split {'differs' if es else 'ok'}")
print("levels with mismatches:", bad)

frames = torch.randint(0, 256, (64, 1080, 1920, 3), generator=g, device=dev, dtype=torch.uint8)
scales = pyramid_scales(1080, 1920)
for layer in (3, 0):
    for split in (True,):
        tot = 0.0
        per = []
        for sc in scales:
            hs, ws = int(math.ceil(1080 * sc)), int(math.ceil(1920 * sc))
            run(layer, frames, hs, ws, split)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run(layer, frames, hs, ws, split)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 5
            per.append(round(t * 1e3))
            tot += t
        print(f"layer {layer} split={split}: all levels {tot:.3f} ms  per level us {per}")

"""Timing of P-Net conv1 (layer 0) on 64 x 1080p for a given build of the library (ablation builds: make
EXTRA_pnet_conv1=-DP1_ABL=<bits>).  usage: p1_abl.py <lib.so> [levels]"""
import math
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from facerecognition_infrenceengine_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.use_library(os.path.abspath(sys.argv[1]))
from facerecognition_infrenceengine_amd import weights
from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP, pyramid_scales

nlev = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
det = MTCNNHIP(*weights.synth_mtcnn_states(seed=1234), device=dev)
p1, lib = det.p1, det.lib
g = torch.Generator(device=dev).manual_seed(3)
frames = torch.randint(0, 256, (64, 1080, 1920, 3), generator=g, device=dev, dtype=torch.uint8)
out = []
for sc in pyramid_scales(1080, 1920)[:nlev]:
    hs, ws = int(math.ceil(1080 * sc)), int(math.ceil(1920 * sc))
    h, w = p1.out_hw(hs, ws)
    y = torch.empty((64, h, w, 12), dtype=torch.float32, device=dev)
    xs = torch.empty((64, h, w, 64), dtype=torch.uint8, device=dev)

    def run():
        lib.fr_dconv_mfma_f32(0, None, _lib.ptr(p1.w), _lib.ptr(p1.b), _lib.ptr(p1.slope), _lib.ptr(y), 64, hs, ws,
                              None, None, _lib.ptr(frames), 1080, 1920, None, 0, _lib.ptr(xs), _lib.stream_ptr())
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1) / 5 * 1e3))
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "-", out)

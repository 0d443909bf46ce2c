"""One pyramid level of P-Net conv1 on 64 x 1080p frames, a few launches: target for rocprofv3 --pmc / --kernel-trace.
usage: prof_p1.py [layer (0 | 3)] [level index]"""
import math
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from facerecognition_infrenceengine_amd import _lib, weights
from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP, pyramid_scales

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = "cuda:0"
det = MTCNNHIP(*weights.synth_mtcnn_states(seed=1234), device=dev)
p1, lib = det.p1, det.lib
g = torch.Generator(device=dev).manual_seed(3)
frames = torch.randint(0, 256, (64, 1080, 1920, 3), generator=g, device=dev, dtype=torch.uint8)
sc = pyramid_scales(1080, 1920)[lvl]
hs, ws = int(math.ceil(1080 * sc)), int(math.ceil(1920 * sc))
h, w = p1.out_hw(hs, ws)
y = torch.empty((64, h, w, 12), dtype=torch.float32, device=dev)
xs = torch.empty((64, h, w, 64), dtype=torch.uint8, device=dev)
for _ in range(6):
    lib.fr_dconv_mfma_f32(layer, None, _lib.ptr(p1.w), _lib.ptr(p1.b), _lib.ptr(p1.slope), _lib.ptr(y), 64, hs, ws,
                          None, None, _lib.ptr(frames), 1080, 1920, None, 0, _lib.ptr(xs), _lib.stream_ptr())
torch.cuda.synchronize()
print("done", hs, ws)

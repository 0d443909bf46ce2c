"""Dev tool: P-Net conv1 (fused resize + conv + pool) alone on pyramid level 0 of 64 x 1080p."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from facerecognition_infrenceengine_amd import weights, _lib
from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
det = MTCNNHIP(*weights.synth_mtcnn_states(), device="cuda:0")
frames = bench.synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
det._s = _lib.stream_ptr()
for s in (0.6, 0.6 * 0.709 ** 2):
    hs, ws = math.ceil(1080 * s), math.ceil(1920 * s)
    for _ in range(3):
        det._dconv(None, det.p1, 64, hs, ws, frames=frames)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        det._dconv(None, det.p1, 64, hs, ws, frames=frames)
    e1.record(); torch.cuda.synchronize()
    print(f"P1 level {hs}x{ws}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")

"""Dev tool: detector on 64 frames vs two concurrent half batches on two streams (+ embed on a third)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from facerecognition_infrenceengine_amd import FaceAnalysis

dev = torch.device("cuda:0")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
frames = bench.synth_frames(64, 1080, 1920, 0, dev)
fa, fb = frames[:32].contiguous(), frames[32:].contiguous()
crops = (torch.rand((256, 112, 112, 8), device=dev) * 2 - 1).half()
crops[..., 3:] = 0
sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
import copy
det2 = copy.copy(app.det); det2._sides = {}    # own side streams

def timeit(fn, n=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

def whole():
    app.det.detect_batch(frames)
def halves_serial():
    app.det.detect_batch(fa); app.det.detect_batch(fb)
def halves_par():
    with torch.cuda.stream(sa):
        app.det.detect_batch(fa)
    with torch.cuda.stream(sb):
        det2.detect_batch(fb)
def halves_par_emb():
    halves_par()
    with torch.cuda.stream(sc):
        app.rec.forward(crops)
def whole_emb():
    with torch.cuda.stream(sa):
        app.det.detect_batch(frames)
    with torch.cuda.stream(sc):
        app.rec.forward(crops)
for f in (whole, halves_serial, halves_par, whole_emb, halves_par_emb):
    print(f.__name__, round(timeit(f), 3))

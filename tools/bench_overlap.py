"""Dev tool: does detect(batch i+1) overlap with embed(batch i) when they run on two HIP streams?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from facerecognition_infrenceengine_amd import FaceAnalysis

dev = torch.device("cuda:0")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
frames = bench.synth_frames(64, 1080, 1920, 0, dev)
crops = (torch.rand((256, 112, 112, 8), device=dev) * 2 - 1).half()
crops[..., 3:] = 0
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def det():
    app.det.detect_batch(frames)

def emb():
    app.rec.forward(crops)

def timeit(fn, n=8):
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

def serial():
    det(); emb()

def overlapped():
    with torch.cuda.stream(sa):
        det()
    with torch.cuda.stream(sb):
        emb()

print("detect   ms", round(timeit(det), 3))
print("embed    ms", round(timeit(emb), 3))
print("serial   ms", round(timeit(serial), 3))
print("overlap  ms", round(timeit(overlapped), 3))
print("overlap40 ms", round(timeit(overlapped, 40), 3))
print("serial40 ms", round(timeit(serial, 40), 3))

def chained():
    # as bench does: embed waits for the detector of the same batch
    with torch.cuda.stream(sb):
        app.detect_embed_slots(frames, det_stream=sa)
print("chained40 ms", round(timeit(chained, 40), 3))
import time
def chained_fetch(depth=3):
    from collections import deque
    q = deque()
    def f():
        with torch.cuda.stream(sb):
            r = app.detect_embed_slots(frames, det_stream=sa)
            q.append(r["counts"])
            if len(q) >= depth:
                q.popleft().cpu()
    return f
print("chained_fetch40 ms", round(timeit(chained_fetch(3), 40), 3))


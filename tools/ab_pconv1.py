"""Dev tool: a 64 x 1080p detector batch alone with P-Net conv1 (a) in f32 on the 16x16x4 matrix-core form, (b) on the f16 matrix
cores + exact f32 tiles under the band (MTCNNHIP.split_pconv1): phase times by event marks, interleaved repeats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, warnings
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
frames = bench.synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
for rep in range(3):
    for pc1 in (False, 20000, 40000, 80000):
        app.det.split_pconv1 = bool(pc1)
        app.det.split_pconv1_min_px = pc1 or 0
        for _ in range(3):
            app.det.detect_batch(frames)
        torch.cuda.synchronize()
        acc = {}
        for _ in range(10):
            app.det.phase_marks = []
            app.det.detect_batch(frames)
            torch.cuda.synchronize()
            m = app.det.phase_marks
            for (n0, a), (n1, b) in zip(m[:-1], m[1:]):
                acc[n1] = acc.get(n1, 0.0) + a.elapsed_time(b) / 10
        app.det.phase_marks = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = app.det.detect_batch(frames)
        e1.record(); torch.cuda.synchronize()
        print("split_pconv1", pc1, "detect ms %.3f" % (e0.elapsed_time(e1) / 10), {k: round(v, 3) for k, v in acc.items()},
              "faces", int(out[3].sum()), flush=True)

"""Dev tool: where a 64 x 1080p detector batch spends its time: events at the cascade's phase ends (pyramid + P-Net joined,
stage-1 NMS, R-Net stage, O-Net stage), per-level NMS behind each level (default) or merged into one launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, warnings
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
frames = bench.synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
for merged in (False, True, False, True):
    for sides in (1, 2):
        app.det.merged_level_nms = merged
        for _ in range(3):
            app.det.detect_batch(frames, level_streams=sides)
        torch.cuda.synchronize()
        acc = {}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            app.det.phase_marks = []
            app.det.detect_batch(frames, level_streams=sides)
            torch.cuda.synchronize()
            m = app.det.phase_marks
            for (n0, a), (n1, b) in zip(m[:-1], m[1:]):
                acc[n1] = acc.get(n1, 0.0) + a.elapsed_time(b) / 10
        app.det.phase_marks = None
        e0.record()
        for _ in range(10):
            app.det.detect_batch(frames, level_streams=sides)
        e1.record(); torch.cuda.synchronize()
        print("merged_level_nms", merged, "sides", sides, "detect ms %.3f" % (e0.elapsed_time(e1) / 10), {k: round(v, 3) for k, v in acc.items()}, flush=True)

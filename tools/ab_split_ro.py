"""Dev tool: a 64 x 1080p detector batch alone with the R-/O-Net second layers on the f32 matrix instruction and on the f16
matrix cores with split-precision operands (MTCNNHIP.split_ro), phase times by event marks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, warnings
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
frames = bench.synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
for rep in range(2):
    for split in (False, True):
        app.det.split_ro = split
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
        extra = ""
        if split:
            extra = " exact-pass crops (R, O): %s" % [int(app.det._ro_lists[k][0]) for k in (0, 1)]
        print("split_ro", split, "detect ms %.3f" % (e0.elapsed_time(e1) / 10), {k: round(v, 3) for k, v in acc.items()},
              "faces", int(out[3].sum()), extra, flush=True)

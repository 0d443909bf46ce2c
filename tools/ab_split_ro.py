"""Dev tool: a 64 x 1080p detector batch alone with (a) everything exact (f32 R-/O-Net second layers, every kept P-Net cell
re-evaluated), (b) + split-precision R-/O-Net conv2 (MTCNNHIP.split_ro), (c) + band-only exact P-Net pass (pnet_band): phase
times by event marks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, warnings
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
frames = bench.synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
app.det.refined_cells = torch.zeros(1, dtype=torch.int32, device="cuda")
for rep in range(2):
    for split, band, tail, c1 in ((False, False, False, False), (True, True, False, False), (True, True, True, False), (True, True, True, True)):
        app.det.split_ro, app.det.pnet_band, app.det.split_tail, app.det.split_conv1 = split, band, tail, c1
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
        app.det.refined_cells.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            out = app.det.detect_batch(frames)
        e1.record(); torch.cuda.synchronize()
        extra = " P-Net cells re-evaluated per batch: %d" % (int(app.det.refined_cells[0]) // 10)
        if split:
            extra += "; exact-pass crops (R, O): %s" % [int(app.det._ro_lists[k][0]) for k in (0, 1)]
        print("split_ro", split, "pnet_band", band, "split_tail", tail, "split_conv1", c1, "detect ms %.3f" % (e0.elapsed_time(e1) / 10), {k: round(v, 3) for k, v in acc.items()},
              "faces", int(out[3].sum()), extra, flush=True)

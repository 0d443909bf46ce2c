"""Dev tool: detector-only time per 64x1080p batch, with P-Net conv1 as the 4x4x1-MFMA kernel (layer 0) and as the
16x16x4 form (layer 3), one stream (kernel times add up) and the default two level streams."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for one in ("1", "0"):
        for layer in ("3", "0", "3", "0"):
            subprocess.run([sys.executable, __file__, layer, "", one], check=True)
    sys.exit(0)
import torch, bench, warnings
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
app.det.p1.layer = int(sys.argv[1])
if len(sys.argv) > 2 and sys.argv[2]:
    app.det.fused_crop = sys.argv[2] == "1"
app.det.one_stream = len(sys.argv) > 3 and sys.argv[3] == "1"
frames = bench.synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
for _ in range(3):
    app.det.detect_batch(frames)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    app.det.detect_batch(frames)
e1.record(); torch.cuda.synchronize()
print("one_stream", app.det.one_stream, "p1 layer", sys.argv[1], "fused crop", app.det.fused_crop, "detect ms", round(e0.elapsed_time(e1) / 10, 3))

"""Dev tool: detector-only time per 64x1080p batch (single stream: kernel times add up)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FR_DET_ONE_STREAM", "1")
import torch, bench, warnings
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
frames = bench.synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
for _ in range(3):
    app.det.detect_batch(frames)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    app.det.detect_batch(frames)
e1.record(); torch.cuda.synchronize()
print("detect ms", round(e0.elapsed_time(e1) / 10, 3))

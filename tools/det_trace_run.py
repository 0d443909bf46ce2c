"""Dev tool: a few detector-only 64 x 1080p batches on ONE stream (to be run under rocprofv3 --kernel-trace);
tools/det_trace_sum.py prints the last batch's kernels in launch order."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, warnings
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
app.det.one_stream = True
app.det.refined_cells = torch.zeros(1, dtype=torch.int32, device="cuda")
frames = bench.synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
for _ in range(4):
    app.det.refined_cells.zero_()
    app.det.detect_batch(frames)
    torch.cuda.synchronize()
print("refined cells per batch", int(app.det.refined_cells[0]))

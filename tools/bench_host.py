"""Dev tool: host enqueue time vs GPU time of the two halves of a step (is the Python driver the bottleneck?)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_frames
from facerecognition_infrenceengine_amd import FaceAnalysis
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    app = FaceAnalysis(name="x", cap_o=4).prepare(ctx_id=0)
fr = synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
for _ in range(3): app.detect_embed_device(fr)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); out = app.det.detect_batch(fr); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    crops = torch.zeros((256, 112, 112, 8), dtype=torch.float16, device="cuda")
    torch.cuda.synchronize()
    t3 = time.perf_counter(); app.rec.forward(crops); t4 = time.perf_counter(); torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"detect: host enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms | embed: host enqueue {1e3*(t4-t3):.2f} ms, total {1e3*(t5-t3):.2f} ms")

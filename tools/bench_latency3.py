"""Dev tool: where a single 640x480 frame's latency goes (cap_o = 1): stage times by HIP events, eager launches."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis, _lib
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    app = FaceAnalysis(name="x", cap_o=1).prepare(ctx_id=0)
fr = torch.from_numpy(synth_frame(480, 640, 7)).cuda()[None].contiguous()
def run():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    b, s, k, c = app.det.detect_batch(fr)
    ev[1].record()
    crops = torch.empty((1, 112, 112, 8), dtype=torch.float16, device="cuda")
    app.lib.fr_warp_affine_5pt_slots(_lib.ptr(fr), 1, 480, 640, _lib.ptr(k.contiguous()), _lib.ptr(c), 1, 112, _lib.ptr(crops), _lib.stream_ptr())
    ev[2].record()
    e, n = app.rec.forward(crops)
    ev[3].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
for _ in range(5): run()
ts = np.array([run() for _ in range(20)])
print("eager GPU-side ms: detect %.3f  warp %.3f  embed(B=1) %.3f" % tuple(np.median(ts, 0)))
t0 = time.perf_counter()
for _ in range(20): app.det.detect_batch(fr)
torch.cuda.synchronize(); print("detect wall per call %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
crops = torch.empty((1, 112, 112, 8), dtype=torch.float16, device="cuda")
t0 = time.perf_counter()
for _ in range(20): app.rec.forward(crops)
torch.cuda.synchronize(); print("embed wall per call %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))

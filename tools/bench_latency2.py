"""Dev tool: where a single-frame get() spends its time (graph replay path)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    app = FaceAnalysis(name="x").prepare(ctx_id=0)
app.enable_graphs(True)
fr = synth_frame(480, 640, 0)
for _ in range(3):
    app.get(fr)
g = app._graph_for((1, 480, 640, 3))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    with torch.cuda.stream(g.stream):
        e0.record(); g.graph.replay(); e1.record()
    g.stream.synchronize()
    ts.append(e0.elapsed_time(e1))
print("graph replay GPU time p50 %.3f ms" % np.percentile(ts, 50))
# stage split (eager, events)
d = torch.from_numpy(fr[None]).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for _ in range(3):
    ev[0].record(); out = app.det.detect_batch(d); ev[1].record()
    crops = torch.zeros((16, 112, 112, 8), dtype=torch.float16, device="cuda"); app.rec.forward(crops); ev[2].record()
    torch.cuda.synchronize()
print("eager: detect %.3f ms, embed(16) %.3f ms" % (ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
t = []
for _ in range(20):
    t0 = time.perf_counter(); faces = app.get(fr); t.append((time.perf_counter() - t0) * 1e3)
print("get() wall p50 %.3f ms (%d faces)" % (np.percentile(t, 50), len(faces)))

"""Dev tool: single-frame latency of FaceAnalysis.get (config C1: 640x480) and of a 1080p frame."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis, GalleryMatcher
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    app = FaceAnalysis(name="x").prepare(ctx_id=0)
rng = np.random.default_rng(1)
G = rng.standard_normal((100, 512)).astype(np.float32)
m = GalleryMatcher("cuda:0"); m.set_rows(list(range(100)), G)
for graphs, hw in ((False, (480, 640)), (False, (1080, 1920)), (True, (480, 640)), (True, (1080, 1920))):
    app.enable_graphs(graphs)
    fr = synth_frame(hw[0], hw[1], 0)
    for _ in range(3):
        faces = app.get(fr)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        faces = app.get(fr)
        ids, score, idx = m.match(np.stack([f.normed_embedding for f in faces]))
        ts.append(time.perf_counter() - t0)
    ts = np.asarray(ts) * 1e3
    print(f"graphs={graphs} {hw}: {len(faces)} faces  get+match p50 {np.percentile(ts,50):.2f} ms  p95 {np.percentile(ts,95):.2f} ms  -> {np.percentile(ts,50)/max(len(faces),1):.3f} ms/face")

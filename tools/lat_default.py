import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    app = FaceAnalysis(name="x").prepare(ctx_id=0)
frame = synth_frame(480, 640, 7)
fc = os.environ.get("FR_FUSED_CROP")
for cap, eng in (("16", app), ("16 strict", app.clone_with(cap_o=16, thresholds=(0.6, 0.7, 0.86))), ("4", app.clone_with(cap_o=4)), ("1", app.clone_with(cap_o=1))):
    if fc is not None:
        eng.det.fused_crop = fc == "1"
    for g in (False, True):
        eng.enable_graphs(g)
        ts = []
        for i in range(40):
            t0 = time.perf_counter(); faces = eng.get(frame); ts.append((time.perf_counter() - t0) * 1e3)
        print("cap_o", cap, "graphs", g, "faces", len(faces), "p50 ms", round(float(np.percentile(ts[8:], 50)), 3))
    eng.enable_graphs(False)

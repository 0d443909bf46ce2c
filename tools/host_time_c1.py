"""Dev tool: host time (Python + ctypes + allocator, no sync inside) against GPU time of the single-frame detector and of the
whole get() - is the eager path waiting for the interpreter?"""
import os, sys, time, warnings, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    app = FaceAnalysis(name="x", cap_o=1).prepare(ctx_id=0)
frame = torch.from_numpy(synth_frame(480, 640, 7)[None]).cuda()
for _ in range(10):
    app.det.detect_batch(frame)
torch.cuda.synchronize()
host, tot = [], []
for _ in range(40):
    t0 = time.perf_counter(); app.det.detect_batch(frame); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print("detect_batch: host issue p50 %.3f ms, until the GPU is done p50 %.3f ms" % (np.percentile(host, 50), np.percentile(tot, 50)))
host, tot = [], []
for _ in range(40):
    t0 = time.perf_counter(); r = app.detect_embed_slots(frame); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print("detect_embed_slots: host issue p50 %.3f ms, until the GPU is done p50 %.3f ms" % (np.percentile(host, 50), np.percentile(tot, 50)))
pr = cProfile.Profile(); pr.enable()
for _ in range(40):
    app.det.detect_batch(frame)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)

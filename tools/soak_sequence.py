"""Dev tool: soak of the eager single-frame call list (fr_detect_sequence): many frames of three shapes in random order through an
engine that replays and one that launches call by call - every result must be equal."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
a = FaceAnalysis(name="x", cap_o=4).prepare(ctx_id=0)
b = a.clone_with(cap_o=4)
b.det.use_sequence = False
rng = np.random.default_rng(0)
shapes = [(240, 320), (360, 640), (480, 640)]
n = bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    hw = shapes[rng.integers(len(shapes))]
    f = synth_frame(hw[0], hw[1], int(rng.integers(1 << 20)))
    fa, fb = a.get(f), b.get(f)
    ok = len(fa) == len(fb) and all(np.array_equal(x.bbox, y.bbox) and np.array_equal(x.kps, y.kps) and x.det_score == y.det_score
                                    and np.array_equal(x.embedding, y.embedding) for x, y in zip(fa, fb))
    n += 1; bad += not ok
print("frames", n, "mismatches", bad, "recorded lists", len(a.det._tls.seqs))
sys.exit(1 if bad else 0)

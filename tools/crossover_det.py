"""Dev tool -> profiles/r05_detector_crossover.txt: the two measured gates of the batch detector.
(1) MTCNNHIP.batch_min_pixels: detect_batch ms for N 1080p frames on the batch path (split-precision R-/O-Net, band-only exact
    P-Net pass, level streams) against the all-f32 path, N = 1 .. 16.
(2) MTCNNHIP.split_pconv1_min_px: 64 x 1080p, P-Net conv1 on the f16 matrix cores from which conv1-map size on."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, warnings
from facerecognition_infrenceengine_amd import FaceAnalysis
from facerecognition_infrenceengine_amd.mtcnn import pyramid_scales
import math
warnings.simplefilter("ignore")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
det = app.det
dev = torch.device("cuda:0")
frames = bench.synth_frames(64, 1080, 1920, 0, dev)


def ms(fr, reps=8):
    for _ in range(3):
        det.detect_batch(fr)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); det.detect_batch(fr); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


print("# (1) batch_min_pixels: detect_batch of N x 1080p frames, median of 8, ms (GPU time between HIP events; eager host issue included)")
print("#  N   batch path   all-f32 path   ratio")
det.use_sequence = False            # the recorded single-frame call list is a third thing; compare the two arithmetic paths launch by launch
for n in (1, 2, 3, 4, 6, 8, 12, 16, 32):
    fr = frames[:n].contiguous()
    det.batch_min_pixels = 0
    a = ms(fr)
    assert det._tls.path["batch"] and det._tls.path["split_ro"]
    det.batch_min_pixels = 10 ** 18
    b = ms(fr)
    assert not det._tls.path["batch"]
    print(f"  {n:3d}   {a:8.3f}     {b:8.3f}      {a / b:5.2f}", flush=True)
det.batch_min_pixels = 22_000_000
det.use_sequence = True

print("\n# (2) split_pconv1_min_px: 64 x 1080p, detect_batch ms by the smallest conv1 map (pixels) whose conv1 runs on the f16 matrix cores")
lv = []
for s in pyramid_scales(1080, 1920):
    h, w = det.p1.out_hw(int(math.ceil(1080 * s)), int(math.ceil(1920 * s)))
    lv.append(h * w)
print("# conv1 map pixels per level:", lv)
for thr in (10 ** 9, lv[0], lv[1], lv[2], lv[3], lv[5], 25):
    det.split_pconv1_min_px = thr
    t = ms(frames, reps=10)
    print(f"  min_px {thr:>10d}: levels on the matrix cores {len(det._tls.path['pconv1_mfma_levels'])}   {t:7.3f} ms", flush=True)

"""Dev tool: P-Net conv1 kernels alone on the levels of a 64 x 1080p batch: f32 (16x16x4 form, writes f32 + split map) against
the f16 matrix-core form (split map only)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, warnings
from facerecognition_infrenceengine_amd import FaceAnalysis, _lib
from facerecognition_infrenceengine_amd.mtcnn import pyramid_scales
warnings.simplefilter("ignore")
app = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
det = app.det
N, H, W = 64, 1080, 1920
frames = bench.synth_frames(N, H, W, 0, torch.device("cuda:0"))
s0 = torch.cuda.current_stream().cuda_stream
det._s = s0
tot = [0.0, 0.0]
for s in pyramid_scales(H, W)[:6]:
    hs, ws = int(math.ceil(H * s)), int(math.ceil(W * s))
    h, w = det.p1.out_hw(hs, ws)
    xs = torch.zeros(N, h, w, 64, dtype=torch.uint8, device="cuda")
    def f32():
        det._dconv(None, det.p1, N, hs, ws, frames=frames, y_split=xs)
    def f16():
        det.lib.fr_pnet_conv1_band(0, _lib.ptr(frames), N, H, W, hs, ws, _lib.ptr(det.p1.w), _lib.ptr(det.p1.b), _lib.ptr(det.p1.slope),
                                   None, _lib.ptr(xs), None, None, 0, s0)
    r = []
    for fn in (f32, f16):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) / 20 * 1000)
    tot[0] += r[0]; tot[1] += r[1]
    print(f"level {hs}x{ws}: f32 {r[0]:.1f} us, f16 {r[1]:.1f} us", flush=True)
print("sum", tot)

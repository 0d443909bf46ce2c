"""Dev tool: minimal repro - one simple kernel repeated on stream A while the embedder runs on stream B."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch, warnings
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis, _lib
warnings.simplefilter("ignore")
app = FaceAnalysis(name="buffalo_l").prepare(ctx_id=0)
lib = app.lib
fr = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(240, 320, s) for s in (10, 20)]))).cuda()
N, H, W = 2, 240, 320
cap = 512
g = torch.Generator(device="cuda").manual_seed(0)
x1 = torch.rand((N, cap), device="cuda", generator=g) * 250
y1 = torch.rand((N, cap), device="cuda", generator=g) * 180
sz = torch.rand((N, cap), device="cuda", generator=g) * 60 + 12
boxes = torch.stack([x1, y1, x1 + sz, y1 + sz], -1).contiguous()
counts = torch.full((N,), cap, dtype=torch.int32, device="cuda")
crops_e = (torch.rand((64, 112, 112, 8), device="cuda") * 2 - 1).half()
s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
mode = sys.argv[1] if len(sys.argv) > 1 else "crop"
REP = 200
x14 = (torch.rand((64, 14, 14, 256), device="cuda") - 0.5).half()
agg = sys.argv[2] if len(sys.argv) > 2 else "embed"
Ah = torch.randn((4096, 4096), device="cuda").half()
Af = torch.randn((4096, 4096), device="cuda")
fr1080 = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(1080, 1920, s) for s in (1, 2, 3, 4)]))).cuda()

def crop(out):
    lib.fr_crop_resize_norm(_lib.ptr(fr), N, H, W, _lib.ptr(boxes), _lib.ptr(counts), cap, 24, _lib.ptr(out), _lib.stream_ptr())

want = torch.empty((N * cap, 24, 24, 4), device="cuda")
crop(want)
src = torch.randn((4 << 20,), device="cuda")
want_t = torch.sin(src) * 1.5 + src
torch.cuda.synchronize()
outs = [torch.empty_like(want) for _ in range(REP)]
outs_t = []
for it in range(3):
    with torch.cuda.stream(s_b):
        for _ in range(4):
            if agg == "embed":
                app.rec.forward(crops_e)
            elif agg == "hgemm":
                for _ in range(30):
                    Ch = Ah @ Ah
            elif agg == "sgemm":
                for _ in range(10):
                    Cf = Af @ Af
            elif agg == "halo14":
                for _ in range(40):
                    app.rec._conv(x14, app.rec.blocks[20][0], 64, 14, 14)
            elif agg == "halo14_res":
                for _ in range(40):
                    app.rec._conv(x14, app.rec.blocks[20][1], 64, 14, 14, residual=x14)
            elif agg == "stem":
                for _ in range(10):
                    app.rec._conv(crops_e, app.rec.stem, 64, 112, 112)
            elif agg == "detect":
                app.det.detect_batch(fr1080)
    with torch.cuda.stream(s_a):
        for r in range(REP):
            if mode == "crop":
                crop(outs[r])
            else:
                outs_t.append(torch.sin(src) * 1.5 + src)
    torch.cuda.synchronize()
    bad = 0
    if mode == "crop":
        for r in range(REP):
            d = (outs[r] != want)
            if d.any():
                bad += 1
                w = d.reshape(-1).nonzero().flatten()
                if bad <= 3:
                    print(" rep", r, "n", w.numel(), "crops", (w // 2304).unique().tolist()[:8], "lanes", (((w % 2304) // 4) % 64).unique().tolist(),
                          "ch", (w % 4).unique().tolist())
    else:
        for o in outs_t:
            if not torch.equal(o, want_t):
                bad += 1
        outs_t = []
    print(mode, "iteration", it, "bad launches", bad, "of", REP)

"""Dev tool: hunt nondeterminism in the detector when run back-to-back / beside the embedder."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch, warnings
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="buffalo_l").prepare(ctx_id=0)
batches = [torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(240, 320, s + k) for s in (10, 20)]))).cuda()
           for k in range(3)]
crops = (torch.rand((64, 112, 112, 8), device="cuda") * 2 - 1).half()
mode = sys.argv[1] if len(sys.argv) > 1 else "det_only"
if os.environ.get("HOLD") == "1":
    # keep every tensor the detector allocates alive until the end of the rep: no block reuse inside a call
    held = []
    det = app.det
    f32, i32 = det._f32, det._i32
    det._f32 = lambda *sh: (held.append(f32(*sh)), held[-1])[1]
    det._i32 = lambda *sh: (held.append(i32(*sh)), held[-1])[1]
want = [app.det.detect_batch(b, trace={}) if mode == "trace" else app.det.detect_batch(b) for b in batches]
torch.cuda.synchronize()
s_det, s_emb = torch.cuda.Stream(), torch.cuda.Stream()
A = torch.randn((4096, 4096), device="cuda")
Bs = torch.randn((8 << 20,), device="cuda")
want_e = app.rec.forward(crops)[0].clone(); torch.cuda.synchronize()
embs = []
names = ("boxes", "scores", "kps", "counts")
bad = 0
for rep in range(30):
    got = []
    if mode.startswith("poison"):
        val = float("nan") if mode == "poison_nan" else 3.0e38
        big = [torch.full((64 << 20,), val, device="cuda") for _ in range(4)]      # 1 GiB of poison, back to the pool
        del big
    for b in batches:
        if mode == "det_default" or mode.startswith("poison"):
            got.append(app.det.detect_batch(b))
        else:
            with torch.cuda.stream(s_det):
                got.append(app.det.detect_batch(b))
            if mode == "with_embed":
                with torch.cuda.stream(s_emb):
                    e = app.rec.forward(crops)[0]
                    if want_e is None:
                        pass
                    embs.append(e)
            if mode == "with_matmul":
                with torch.cuda.stream(s_emb):
                    for _ in range(6):
                        C = A @ A
            if mode == "with_small":
                with torch.cuda.stream(s_emb):
                    for _ in range(150):
                        Bs.mul_(1.0001).add_(0.5)
            if mode == "with_alloc":
                with torch.cuda.stream(s_emb):
                    for _ in range(40):
                        t = torch.full((1 << 20,), 7.0, device="cuda")
    torch.cuda.synchronize()
    if os.environ.get("HOLD") == "1":
        held.clear()
    for e in embs:
        if not torch.equal(e, want_e):
            bad += 1; print(rep, "EMBED differs", float((e - want_e).abs().max()))
    embs = []
    for k, (w, g) in enumerate(zip(want, got)):
        cnt = w[3]
        cap = w[0].shape[1]
        valid = (torch.arange(cap, device="cuda")[None, :] < cnt[:, None]).reshape(-1)
        for nm, a, c in zip(names, w, g):
            if nm == "counts":
                if not torch.equal(a, c):
                    bad += 1; print(rep, k, nm, a.tolist(), c.tolist())
                continue
            a2, c2 = a.reshape(valid.numel(), -1)[valid], c.reshape(valid.numel(), -1)[valid]
            if not torch.equal(a2, c2):
                bad += 1
                d = (a2 - c2).abs()
                print(rep, k, nm, "maxdiff", float(d.max()), "rows", d.amax(1).nonzero().flatten().tolist())
print(mode, "mismatches:", bad)

"""Dev tool: first differing detector intermediate when the embedder runs beside it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch, warnings
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="buffalo_l").prepare(ctx_id=0)
fr = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(240, 320, s) for s in (10, 20)]))).cuda()
crops = (torch.rand((64, 112, 112, 8), device="cuda") * 2 - 1).half()
s_det, s_emb = torch.cuda.Stream(), torch.cuda.Stream()
t0 = {}
app.det.detect_batch(fr, trace=t0)
torch.cuda.synchronize()

def flat(t):
    out = {}
    for k, v in t.items():
        if isinstance(v, list):
            for i, x in enumerate(v):
                out[f"{k}[{i}]"] = x
        else:
            out[k] = v
    return out
f0 = flat(t0)
tot = {}
for rep in range(40):
    t1 = {}
    with torch.cuda.stream(s_emb):
        app.rec.forward(crops)
    with torch.cuda.stream(s_det):
        app.det.detect_batch(fr, trace=t1)
    with torch.cuda.stream(s_emb):
        app.rec.forward(crops)
    torch.cuda.synchronize()
    f1 = flat(t1)
    N = fr.shape[0]
    def mask(cnt, cap):
        return (torch.arange(cap, device="cuda")[None, :] < cnt[:, None]).reshape(-1)
    c1, c2 = f0["stage1_counts"], f0["stage2_counts"]
    masks = {"stage1_boxes": mask(c1, 512), "stage1_scores": mask(c1, 512), "rnet_crops": mask(c1, 512),
             "rnet_head": mask(c1, 512), "rnet_prob": mask(c1, 512), "stage2_boxes": mask(c2, 64),
             "stage2_scores": mask(c2, 64), "onet_head": mask(c2, 64), "onet_prob": mask(c2, 64)}
    for k in f0:
        a, b = f0[k], f1[k]
        if k in masks:
            m = masks[k]
            a, b = a.reshape(m.numel(), -1)[m], b.reshape(m.numel(), -1)[m]
        if a.dtype.is_floating_point:
            neq = ((a != b) & ~(a.isnan() & b.isnan())).sum().item()
        else:
            neq = (a != b).sum().item()
        if neq:
            tot.setdefault(k, []).append((rep, neq, a.numel()))
            if k == "rnet_crops" and len(tot[k]) <= 3:
                af, bf = a.reshape(-1), b.reshape(-1)
                w = (af != bf).nonzero().flatten()
                print("rep", rep, "rnet_crops diff idx", w[:6].tolist(), "...", w[-3:].tolist(), "crop", (w // 2304).unique().tolist(),
                      "pix", ((w % 2304) // 4).unique().tolist()[:30])
                print("   want", af[w[:8]].tolist()); print("   got ", bf[w[:8]].tolist())
for k in f0:
    if k in tot:
        print(k, "differs in", len(tot[k]), "reps; e.g.", tot[k][:3])
print("keys", list(f0))

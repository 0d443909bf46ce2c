"""Dev tool: guard bands around every tensor the embedder (or detector) allocates - finds out-of-bounds writes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch, warnings
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="buffalo_l").prepare(ctx_id=0)
real_empty = torch.empty
GUARD = 1 << 18          # elements each side
live = []

def guarded_empty(*shape, dtype=None, device=None, **kw):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    n = int(np.prod(shape)) if len(shape) else 1
    buf = real_empty(n + 2 * GUARD, dtype=dtype, device=device)
    if dtype in (torch.float16, torch.float32):
        buf.fill_(1234.0)
    else:
        buf.fill_(77)
    live.append((buf, n, shape, dtype))
    return buf[GUARD:GUARD + n].view(shape)

def check(tag):
    torch.cuda.synchronize()
    bad = 0
    for buf, n, shape, dtype in live:
        sent = 1234.0 if dtype in (torch.float16, torch.float32) else 77
        lo, hi = buf[:GUARD], buf[GUARD + n:]
        for nm, g in (("before", lo), ("after", hi)):
            w = (g != sent).nonzero().flatten()
            if w.numel():
                bad += 1
                print(tag, "GUARD HIT", nm, tuple(shape), dtype, "count", w.numel(), "first", int(w[0]), "last", int(w[-1]))
    print(tag, "tensors", len(live), "guard hits", bad)
    live.clear()

which = sys.argv[1] if len(sys.argv) > 1 else "embed"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.empty = guarded_empty
try:
    if which == "embed":
        crops = (torch.rand((B, 112, 112, 8), device="cuda") * 2 - 1).half()
        app.rec.forward(crops)
        check(f"embed B={B}")
    else:
        from make_golden import synth_frame
        fr = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(240, 320, s) for s in (10, 20)]))).cuda()
        app.det.detect_batch(fr)
        check("detect 2x240x320")
        fr = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(1080, 1920, s) for s in (1, 2, 3)]))).cuda()
        app.det.detect_batch(fr)
        check("detect 3x1080p")
finally:
    torch.empty = real_empty

"""DESIGN.md 4.7 from the failing end: the round-1 crop_resize_norm compiled WITH packed-f32 ops, in variants that each
remove one ingredient (pkvictim.hip), run beside the product's embed forward on a second stream.  ONE pass per pair."""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, warnings
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis, _lib
warnings.simplefilter("ignore")
app = FaceAnalysis(name="x").prepare(ctx_id=0)
vic = C.CDLL(os.path.join(HERE, "libpkvictim.so"))
r1 = C.CDLL(os.path.join(HERE, "libdetect_ops_r1pk.so"))       # round 1's detect_ops.hip, whole file, packed ops ON
P = C.c_void_p
fr = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(240, 320, s) for s in (10, 20)]))).cuda()
N, H, W, cap = 2, 240, 320, 512
g = torch.Generator(device="cuda").manual_seed(0)
x1 = torch.rand((N, cap), device="cuda", generator=g) * 250
y1 = torch.rand((N, cap), device="cuda", generator=g) * 180
sz = torch.rand((N, cap), device="cuda", generator=g) * 60 + 12
boxes = torch.stack([x1, y1, x1 + sz, y1 + sz], -1).contiguous()
counts = torch.full((N,), cap, dtype=torch.int32, device="cuda")
crops_e = (torch.rand((64, 112, 112, 8), device="cuda") * 2 - 1).half()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
REP = 200
names = {0: "as it was", 1: "no global loads", 2: "four dword stores", 3: "s_nop 7 x2 before the store",
         4: "v_mov copies before the store", 5: "no valid branch", 6: "4th word from a register",
         7: "round 1's file, verbatim (17 v_pk ops)"}


def victim(v, out, stream):
    if v == 7:
        rc = r1.fr_crop_resize_norm(P(fr.data_ptr()), N, H, W, P(boxes.data_ptr()), P(counts.data_ptr()), cap, 24,
                                    P(out.data_ptr()), P(stream.cuda_stream))
        assert rc == 0
        return
    rc = vic.pkv_run(v, P(fr.data_ptr()), N, H, W, P(boxes.data_ptr()), P(counts.data_ptr()), cap, 24, P(out.data_ptr()),
                     P(stream.cuda_stream))
    assert rc == 0


for aggressor in ("none", "embed"):
    for v in (7, 0, 5):
        want = torch.empty((N * cap, 24, 24, 4), device="cuda")
        with torch.cuda.stream(sa):
            victim(v, want, sa)
        torch.cuda.synchronize()
        outs = [torch.empty_like(want) for _ in range(REP)]
        if aggressor == "embed":
            with torch.cuda.stream(sb):
                for _ in range(6):
                    app.rec.forward(crops_e)
        with torch.cuda.stream(sa):
            for o in outs:
                victim(v, o, sa)
        torch.cuda.synchronize()
        bad_launches, words, q = 0, 0, [0, 0, 0, 0]
        for o in outs:
            ne = (o.view(torch.int32) != want.view(torch.int32))
            n = int(ne.sum())
            if n:
                bad_launches += 1
                words += n
                px = ne.any(dim=-1).reshape(N * cap, 576).nonzero()[:, 1]           # pixel index t inside the crop
                lanes = (px % 256) % 64
                for k in range(4):
                    q[k] += int(((lanes // 16) == k).sum())
        print(f"aggressor {aggressor:6s} victim {v} ({names[v]:32s}): launches with mismatches {bad_launches:3d}/{REP}  words {words:7d}  pixels by lane quarter {q}", flush=True)

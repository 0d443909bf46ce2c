"""DESIGN.md 4.7, ISA-level bisect: every code object of tools/hw/pkco (variants of round 1's packed-f32
crop_resize_norm, pkisa_gen.py) beside the product's embed forward on a second stream, compared word by word with the
same variant run alone.  ONE pass per variant."""
import ctypes as C, glob, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, warnings
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import FaceAnalysis
warnings.simplefilter("ignore")
app = FaceAnalysis(name="x").prepare(ctx_id=0)
mod = C.CDLL(os.path.join(HERE, "libpkmod.so"))
mod.pkmod_load.restype = C.c_void_p
mod.pkmod_load.argtypes = [C.c_char_p, C.c_char_p]
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
KERNEL = b"_Z16crop_resize_normPKhiiPKfPKiiiPf"
only = sys.argv[1:]
for path in sorted(glob.glob(os.path.join(HERE, "pkco", "*.co"))):
    name = os.path.basename(path)[:-3]
    if only and not any(o in name for o in only):
        continue
    fn = mod.pkmod_load(path.encode(), KERNEL)
    assert fn, path

    def victim(out, stream):
        rc = mod.pkmod_crop(P(fn), P(fr.data_ptr()), N, H, W, P(boxes.data_ptr()), P(counts.data_ptr()), cap, 24,
                            P(out.data_ptr()), P(stream.cuda_stream))
        assert rc == 0, rc
    want = torch.empty((N * cap, 24, 24, 4), device="cuda")
    with torch.cuda.stream(sa):
        victim(want, sa)
    torch.cuda.synchronize()
    outs = [torch.empty_like(want) for _ in range(REP)]
    with torch.cuda.stream(sb):
        for _ in range(6):
            app.rec.forward(crops_e)
    with torch.cuda.stream(sa):
        for o in outs:
            victim(o, sa)
    torch.cuda.synchronize()
    bad, words, q = 0, 0, [0, 0, 0, 0]
    for o in outs:
        ne = (o.view(torch.int32) != want.view(torch.int32))
        n = int(ne.sum())
        if n:
            bad += 1; words += n
            px = ne.any(dim=-1).reshape(N * cap, 576).nonzero()[:, 1]
            lanes = (px % 256) % 64
            for k in range(4):
                q[k] += int(((lanes // 16) == k).sum())
    ch = [int((torch.stack(outs).view(torch.int32)[..., c] != want.view(torch.int32)[..., c]).sum()) for c in range(4)]
    print(f"{name:24s}: launches with mismatches {bad:3d}/{REP}  words {words:7d}  by lane quarter {q}  by output channel {ch}", flush=True)

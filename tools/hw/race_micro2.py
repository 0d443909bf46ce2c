"""Dev tool: the product's crop kernel (known-sensitive victim) beside micro aggressors."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from make_golden import synth_frame
from facerecognition_infrenceengine_amd import _lib
lib = _lib.load()
mic = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "librace_micro.so"))
P = C.c_void_p
fr = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(240, 320, s) for s in (10, 20)]))).cuda()
N, H, W, cap = 2, 240, 320, 512
g = torch.Generator(device="cuda").manual_seed(0)
x1 = torch.rand((N, cap), device="cuda", generator=g) * 250; y1 = torch.rand((N, cap), device="cuda", generator=g) * 180
sz = torch.rand((N, cap), device="cuda", generator=g) * 60 + 12
boxes = torch.stack([x1, y1, x1 + sz, y1 + sz], -1).contiguous()
counts = torch.full((N,), cap, dtype=torch.int32, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
sink = torch.zeros(8 << 20, device="cuda"); src = torch.randn(16 << 20, device="cuda")
def crop(out, s):
    lib.fr_crop_resize_norm(_lib.ptr(fr), N, H, W, _lib.ptr(boxes), _lib.ptr(counts), cap, 24, _lib.ptr(out), P(s.cuda_stream))
want = torch.empty((N * cap, 24, 24, 4), device="cuda"); crop(want, torch.cuda.current_stream()); torch.cuda.synchronize()
REP = 200
outs = [torch.empty_like(want) for _ in range(REP)]
names = {0: "mfma16x16x32_f16", 1: "mfma32x32x16_f16", 3: "mfma16x16x4_f32", 4: "valu_pk_f16", 5: "ds_read_b128", 6: "buffer_load_b128",
         7: "buffer_load_b128+OOB", 8: "lds_dma16", 9: "lds_dma16+OOB", 10: "global_store16", -1: "none"}
names = {100: "big LDS 32K", 101: "big LDS 60K", 102: "big LDS 80K", 103: "big LDS 80K +mfma", 104: "big LDS 160K +mfma", -1: "none"}
for kind, kname in names.items():
    torch.cuda.synchronize()
    if kind >= 100:
        sz = {100: 32768, 101: 61440, 102: 81920, 103: 81920, 104: 163840}[kind]
        for _ in range(4):
            rc = mic.race_aggressor_big(100000, 1024, sz, 1 if kind >= 103 else 0, P(sink.data_ptr()), P(sb.cuda_stream))
            assert rc == 0, rc
    elif kind >= 6:
        for _ in range(4):
            mic.race_aggressor_mem(kind, 20000, 2048, P(src.data_ptr()), C.c_uint(src.numel() * 4), P(sink.data_ptr()), P(sb.cuda_stream))
    elif kind >= 0:
        for _ in range(4):
            mic.race_aggressor(kind, 200000, 2048, P(sink.data_ptr()), P(sb.cuda_stream))
    for r in range(REP):
        crop(outs[r], sa)
    torch.cuda.synchronize()
    bad = sum(1 for r in range(REP) if not torch.equal(outs[r], want))
    print(f"crop victim vs {kname:22s}: bad launches {bad} / {REP}", flush=True)

"""Dev tool: gallery scan timings (HIP events), f32 exact vs one-pass f16 / fp8 GEMM scan + f32 re-rank."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd.gallery import GalleryMatcher

g = torch.Generator(device="cuda").manual_seed(1)
cases = [(10_000, 256), (125_000, 2048), (1_000_000, 256), (1_000_000, 2048), (1_250_000, 2048)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for N, F in cases:
    G = torch.randn((N, 512), generator=g, device="cuda"); G /= G.norm(dim=1, keepdim=True)
    Q = torch.randn((F, 512), generator=g, device="cuda"); Q /= Q.norm(dim=1, keepdim=True)
    ref = None
    for scan in ("f32", "f16", "f8"):
        if scan == "f32" and N * F > 3e8:
            continue
        m = GalleryMatcher("cuda:0", scan=scan)
        m.set_rows(range(N), G, normalise=False)
        for _ in range(3):
            idx, score = m.match_device(Q, renormalise=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            idx, score = m.match_device(Q, renormalise=False)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        b = {"f32": 4, "f16": 2, "f8": 1}[scan]
        same = "" if ref is None else f" ids==first: {bool(torch.equal(idx, ref))}"
        if ref is None:
            ref = idx
        print(f"N={N} F={F} {scan}: {ms*1e3:8.1f} us  {2*N*F*512/ms/1e9:8.1f} TFLOP/s  gallery bytes/time {N*512*b/ms/1e9:7.2f} TB/s{same}", flush=True)
        del m
    del G, Q

"""Dev tool: ablations of the GEMM scan (debug library): FR_SCAN_ABL bits 1 no epilogue, 2 no DMA after tile 0, 4 no MFMA."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facerecognition_infrenceengine_amd import _lib
_lib.use_library(os.path.join(os.path.dirname(_lib.LIB_PATH), "libfrhip_debug.so"))
import torch
from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
g = torch.Generator(device="cuda").manual_seed(1)
N, F = 1_000_000, 2048
G = torch.randn((N, 512), generator=g, device="cuda"); G /= G.norm(dim=1, keepdim=True)
Q = torch.randn((F, 512), generator=g, device="cuda"); Q /= Q.norm(dim=1, keepdim=True)
for scan in ("f16", "f8"):
    m = GalleryMatcher("cuda:0", scan=scan); m.set_rows(range(N), G, normalise=False)
    for abl in (0, 1, 2, 3, 4, 5, 6, 7):
        os.environ["FR_SCAN_ABL"] = str(abl)
        for _ in range(2): m.match_device(Q, renormalise=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): m.match_device(Q, renormalise=False)
        e1.record(); torch.cuda.synchronize()
        print(f"{scan} abl={abl} (noepi={abl&1} nodma={(abl>>1)&1} nomfma={(abl>>2)&1}): {e0.elapsed_time(e1)/5*1e3:8.1f} us", flush=True)

"""Dev tool: band height / slots-per-block variants of the F16 form of crop_conv1_kernel (csrc/ro_conv1.hip), each built into its
own shared object by `python tools/abl_rc1.py build` (here, no GPU needed) and timed on the GPU box on 64 x 1080p frames'
worth of random boxes by `python tools/abl_rc1.py`."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "hw", "abl")
VARIANTS = {"r4x8_o2x4": (4, 8, 2, 4), "r3x16_o2x4": (3, 16, 2, 4), "r3x32_o2x16": (3, 32, 2, 16), "r3x64_o2x8": (3, 64, 2, 8),
            "r2x16_o2x4": (2, 16, 2, 4), "r5x16_o2x4": (5, 16, 2, 4)}
if len(sys.argv) > 1 and sys.argv[1] == "build":
    src = os.path.join(ROOT, "facerecognition_infrenceengine_amd", "csrc")
    os.makedirs(OUT, exist_ok=True)
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops".split()
    stub = os.path.join(OUT, "stub.cpp")
    open(stub, "w").write('#include <cstdarg>\n#include <cstdio>\nvoid fr_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }\n')
    for name, (pr, rr, po, ro) in VARIANTS.items():
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, f"-DRC1_PB_R={pr}", f"-DRC1_RPB_R={rr}", f"-DRC1_PB_O={po}", f"-DRC1_RPB_O={ro}",
                               "-shared", "-o", os.path.join(OUT, f"rc1_{name}.so"), os.path.join(src, "ro_conv1.hip"), stub])
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from facerecognition_infrenceengine_amd import _lib
import bench
P, I = ctypes.c_void_p, ctypes.c_int
frames = bench.synth_frames(64, 1080, 1920, 0, torch.device("cuda:0"))
g = torch.Generator(device="cuda").manual_seed(3)
for net, cap, p1, c1 in ((0, 512, 11, 28), (1, 64, 23, 32)):
    n = 64 * cap
    x1 = torch.rand(n, generator=g, device="cuda") * 1800
    y1 = torch.rand(n, generator=g, device="cuda") * 1000
    sz = torch.rand(n, generator=g, device="cuda") * 200 + 20
    boxes = torch.stack([x1, y1, x1 + sz, y1 + sz], -1).contiguous()
    counts = torch.full((64,), cap, dtype=torch.int32, device="cuda")
    w = torch.randn((27, c1), generator=g, device="cuda") * 0.2
    b = torch.zeros(c1, device="cuda"); s = torch.full((c1,), 0.25, device="cuda")
    ys = torch.empty((n, p1 * p1, 128), dtype=torch.uint8, device="cuda")
    for name in VARIANTS:
        lib = ctypes.CDLL(os.path.join(OUT, f"rc1_{name}.so"))
        f = lib.fr_crop_conv1_split
        f.argtypes = [I, P, I, I, I, P, P, I, P, P, P, P, I, P]
        args = (net, _lib.ptr(frames), 64, 1080, 1920, _lib.ptr(boxes), _lib.ptr(counts), cap, _lib.ptr(w), _lib.ptr(b), _lib.ptr(s), _lib.ptr(ys), 1, _lib.stream_ptr())
        for _ in range(3):
            assert f(*args) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(*args)
        e1.record(); torch.cuda.synchronize()
        print(f"net {net} {name:12s}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)

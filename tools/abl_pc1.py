"""Dev tool: compile-time ablations of the f16 form of pnet_conv1_kernel (csrc/pnet_conv1.hip, -DP1_ABL=bits: 1 no frame loads,
2 no MFMAs, 8 no fragment reads, 16 no conversion / blend, 32 no split / stores), each built into its own shared object by
`python tools/abl_pc1.py build` (here, no GPU needed) and timed on the GPU box on level 0 of a 64 x 1080p batch by
`python tools/abl_pc1.py`."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "hw", "abl")
VARIANTS = [0, 1, 2, 8, 10, 16, 32, 17, 42, 58, 59]
EXTRA = {"occ4": ["-DP1_OCC=4"], "occ4_ctu1": ["-DP1_OCC=4", "-DP1_CTU=1"], "ctu1": ["-DP1_CTU=1"]}      # other build knobs, timed beside ablation 0
if len(sys.argv) > 1 and sys.argv[1] == "build":
    src = os.path.join(ROOT, "facerecognition_infrenceengine_amd", "csrc")
    os.makedirs(OUT, exist_ok=True)
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops".split()
    stub = os.path.join(OUT, "stub.cpp")       # the one symbol the kernel file needs from abi.cpp
    open(stub, "w").write('#include <cstdarg>\n#include <cstdio>\nvoid fr_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }\n')
    for v in VARIANTS:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, f"-I{ROOT}/include", f"-DP1_ABL={v}", "-shared", "-o", os.path.join(OUT, f"pc1_{v}.so"),
                               os.path.join(src, "pnet_conv1.hip"), stub])
    for k, fl in EXTRA.items():
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, f"-I{ROOT}/include", *fl, "-shared", "-o", os.path.join(OUT, f"pc1_{k}.so"),
                               os.path.join(src, "pnet_conv1.hip"), stub])
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "knobs":
    VARIANTS = [0, *EXTRA]
sys.path.insert(0, ROOT)
import torch, math
from facerecognition_infrenceengine_amd import _lib
P, I = ctypes.c_void_p, ctypes.c_int
N, H, W = 64, 1080, 1920
g = torch.Generator(device="cuda").manual_seed(3)
frames = torch.randint(0, 256, (N, H, W, 3), generator=g, device="cuda", dtype=torch.uint8)
w = torch.randn((9 * 4 * 16,), generator=g, device="cuda") * 0.1
b = torch.zeros(16, device="cuda"); sl = torch.full((16,), 0.25, device="cuda")
for hs, ws in ((648, 1152), (460, 817)):
    h, wd = (hs - 2 + 1) // 2, (ws - 2 + 1) // 2
    xs = torch.zeros(N, h, wd, 64, dtype=torch.uint8, device="cuda")
    for v in VARIANTS:
        lib = ctypes.CDLL(os.path.join(OUT, f"pc1_{v}.so"))
        f = lib.fr_pnet_conv1_band
        f.argtypes = [I, P, I, I, I, I, I, P, P, P, P, P, P, P, I, P]
        args = (0, _lib.ptr(frames), N, H, W, hs, ws, _lib.ptr(w), _lib.ptr(b), _lib.ptr(sl), None, _lib.ptr(xs), None, None, 0, _lib.stream_ptr())
        for _ in range(3):
            assert f(*args) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(*args)
        e1.record(); torch.cuda.synchronize()
        print(f"level {hs}x{ws} P1_ABL={v}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)

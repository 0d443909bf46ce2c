"""Dev tool: compile-time ablations of ro_conv2_split_kernel (csrc/ro_conv2.hip, -DRC2_ABL=bits: 1 no MFMAs, 2 no fragment
reads, 4 no input DMA, 8 no pool epilogue), each built into its own shared object by `python tools/abl_rc2.py build` (here, no
GPU needed) and timed on the GPU box on 64 frames' worth of slots by `python tools/abl_rc2.py`."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "hw", "abl")
VARIANTS = [0, 16, 100, 116, 1000, 1016, 6000, 6016]      # 100+: RC2_CFG=1 (the one-workgroup-per-CU shapes); 1000 s: RC2_STAGGER = s (1000: 0... see below)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    src = os.path.join(ROOT, "facerecognition_infrenceengine_amd", "csrc")
    os.makedirs(OUT, exist_ok=True)
    flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops".split()
    stub = os.path.join(OUT, "stub.cpp")       # the one symbol the kernel file needs from abi.cpp
    open(stub, "w").write('#include <cstdarg>\n#include <cstdio>\nvoid fr_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }\n')
    for v in VARIANTS:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, f"-DRC2_ABL={v % 100}", f"-DRC2_CFG={v // 100 % 10}", *([f"-DRC2_STAGGER={0 if v // 1000 == 1 else v // 1000}"] if v >= 1000 else []), "-shared", "-o", os.path.join(OUT, f"rc2_{v}.so"),
                               os.path.join(src, "ro_conv2.hip"), stub])
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from facerecognition_infrenceengine_amd import _lib
P, I = ctypes.c_void_p, ctypes.c_int
for net, nslots, cap, p1, cout, p2 in ((0, 64 * 512, 512, 11, 48, 4), (1, 64 * 64, 64, 23, 64, 10)):
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.randn((nslots, p1 * p1, 64), generator=g, device="cuda") * 0.5).half()      # any finite f16 pattern
    w = torch.randn((cout, 9, 32), generator=g, device="cuda") * 0.1
    b = torch.zeros(cout, device="cuda"); s = torch.full((cout,), 0.25, device="cuda")
    y = torch.empty((nslots, p2, p2, cout), device="cuda")
    counts = torch.full((nslots // cap,), cap, dtype=torch.int32, device="cuda")
    for v in VARIANTS:
        lib = ctypes.CDLL(os.path.join(OUT, f"rc2_{v}.so"))
        f = lib.fr_ro_conv2_split
        f.argtypes = [I, P, P, P, P, P, I, P, I, P, P]
        args = (net, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(s), _lib.ptr(y), nslots, _lib.ptr(counts), cap, None, _lib.stream_ptr())
        for _ in range(3):
            assert f(*args) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(*args)
        e1.record(); torch.cuda.synchronize()
        print(f"net {net} RC2_ABL={v:2d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
        if v % 100 == 16:         # in-kernel stamps: per-wave cycle sums of the five phases of an item, averaged per item
            st = torch.zeros((4096 * 16 * 8,), dtype=torch.int64, device="cuda")
            args2 = args[:-2] + (_lib.ptr(st), args[-1])
            f(*args2); torch.cuda.synchronize()
            t = st.view(-1, 8).cpu().double()
            t = t[t[:, 5] > 0]
            per = t[:, :5].sum(0) / t[:, 5].sum()
            names = ["wait DMA + barrier", "issue next DMA", "K loop", "barrier + acc -> LDS + barrier", "pool + store"]
            print("   cycles per item and wave: " + ", ".join(f"{n} {c:.0f}" for n, c in zip(names, per.tolist())), f"(sum {per.sum():.0f}; items per wave {t[:, 5].mean():.1f})", flush=True)

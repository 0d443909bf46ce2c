"""Dev tool: A/B of halo-kernel variants in ONE process (debug library, interleaved rounds): FR_HALO_NW4 = 0 (8-wave
lean) vs 3 (4-wave lean).  Also checks that both variants give bit-identical outputs."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facerecognition_infrenceengine_amd import _lib
_lib.use_library(os.path.join(os.path.dirname(_lib.LIB_PATH), "libfrhip_debug.so"))
import torch
lib = _lib.load()
B = 256
for (H, Cin, Cout) in ((14, 256, 256),):
    x = torch.randn((B, H, H, Cin), device="cuda").half()
    w = (torch.randn((Cout, 9 * Cin), device="cuda") * 0.02).half()
    bias = torch.randn(9 * Cout, device="cuda"); slope = torch.rand(Cout, device="cuda")
    res = torch.randn((B, H, H, Cout), device="cuda").half()
    ys = {}
    def run(nw4, y, n):
        os.environ["FR_HALO_NW4"] = str(nw4 if nw4 != 16 else 0)
        os.environ["FR_HALO_W16"] = "1" if nw4 == 16 else "0"
        a = _lib.ConvArgs(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(bias), _lib.ptr(slope), _lib.ptr(res), None,
                          B, H, H, Cin, Cout, 3, 3, 1, 1, H, H, 1, 1)
        for _ in range(n):
            lib.fr_conv_nhwc_f16(ctypes.byref(a), _lib.stream_ptr())
    for nw4 in (0, 16):
        ys[nw4] = torch.empty((B, H, H, Cout), dtype=torch.float16, device="cuda")
        run(nw4, ys[nw4], 3)
    torch.cuda.synchronize()
    print(f"{H}x{H} {Cin}->{Cout}: outputs identical: {bool(torch.equal(ys[0], ys[16]))}")
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    variants = [(0, 0), (16, 0)]
    best = {}
    for rnd in range(3):
        for nw4, stg in variants:
            os.environ["FR_HALO_STAGGER"] = str(stg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(nw4, ys[nw4], 40); e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 40 * 1e3
            best[(nw4, stg)] = min(best.get((nw4, stg), 1e9), us)
    os.environ["FR_HALO_STAGGER"] = "0"
    for (nw4, stg), us in best.items():
        print(f"   nw4={nw4} abl={stg >> 16:2d} (noMFMA={(stg>>16)&1} noReads={(stg>>17)&1} noWdma={(stg>>18)&1} noBarrier={(stg>>19)&1}): best of 3 {us:7.1f} us  {fl/us/1e6:7.1f} TFLOP/s", flush=True)

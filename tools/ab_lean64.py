"""Dev tool: 56x56 64->64 conv (256 faces, bias + PReLU + residual) on the debug library: pipelined schedule (0), lean
schedule (1) and the lean schedule's compile-time ablations (FR_HALO_LEAN64 = 2..6), interleaved rounds in one process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from facerecognition_infrenceengine_amd import _lib
_lib.use_library(os.path.join(os.path.dirname(_lib.LIB_PATH), "libfrhip_debug.so"))
import torch
lib = _lib.load()
B, H, Cin, Cout = 256, 56, 64, 64
x = torch.randn((B, H, H, Cin), device="cuda").half()
w = (torch.randn((Cout, 9 * Cin), device="cuda") * 0.02).half()
bias = torch.randn(9 * Cout, device="cuda"); slope = torch.rand(Cout, device="cuda")
res = torch.randn((B, H, H, Cout), device="cuda").half()
y = torch.empty((B, H, H, Cout), dtype=torch.float16, device="cuda")
NAMES = {0: "pipelined (product)", 1: "lean", 2: "lean, no W DMA", 3: "lean, no MFMA", 4: "lean, no fragment reads", 5: "lean, skeleton", 6: "lean, no barrier"}
def run(v, n):
    os.environ["FR_HALO_LEAN64"] = str(v)
    a = _lib.ConvArgs(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(bias), _lib.ptr(slope), _lib.ptr(res), None,
                      B, H, H, Cin, Cout, 3, 3, 1, 1, H, H, 1, 1)
    for _ in range(n):
        lib.fr_conv_nhwc_f16(ctypes.byref(a), _lib.stream_ptr())
for v in NAMES:
    run(v, 3)
torch.cuda.synchronize()
best = {}
for rnd in range(3):
    for v in NAMES:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(v, 30); e1.record(); torch.cuda.synchronize()
        best[v] = min(best.get(v, 1e9), e0.elapsed_time(e1) / 30 * 1e3)
for v, us in best.items():
    print(f"  {NAMES[v]:28s} {us:7.1f} us", flush=True)

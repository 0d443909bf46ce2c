"""Dev tool: in-kernel s_memtime segment sums of the halo conv (diagnostic build, shares only)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
st = torch.zeros(4096 * 8 * 8, dtype=torch.int64, device="cuda")
os.environ["FR_DBG_STAMPS"] = hex(st.data_ptr())
# stamps exist only in the diagnostic twin of the library (make -C facerecognition_infrenceengine_amd/csrc debug)
from facerecognition_infrenceengine_amd import _lib as _fr_lib
_fr_lib.use_library(os.path.join(os.path.dirname(_fr_lib.LIB_PATH), "libfrhip_debug.so"))
from tools import bench_conv
B = 256
for shape in ((14, 256, 256), (28, 128, 128), (56, 64, 64), (112, 64, 64)):
    st.zero_()
    bench_conv.run(B, shape[0], shape[0], shape[1], shape[2], iters=3, tag="stamped")
    torch.cuda.synchronize()
    nb = {14: B * 2, 28: B * 4, 56: B * 14, 112: B * 56}[shape[0]]   # blocks (stamp slots wrap at 4096)
    nb = min(nb, 4096)
    d = st.reshape(-1, 8)[: nb * 8, :5].double()
    tot = d.sum(1)
    names = ["wait vmcnt", "barrier", "issue W", "reads+MFMA", "tap bookkeeping / halo reload"]
    print(shape, "per-step cycles (s_memtime ticks, 100MHz?) mean per wave:")
    steps = 9 * shape[1] // 64
    for k, nme in enumerate(names):
        print(f"   {nme:16s} {d[:, k].mean().item() / steps:10.1f}  share {d[:, k].sum().item() / tot.sum().item():.3f}")
    print("   total/step", tot.mean().item() / steps)
    ph = st.reshape(-1, 8)[: nb * 8, 5:8].double().mean(0)
    print(f"   phases (cycles/wave): setup {ph[0]:.0f}  K loop {ph[1]:.0f}  epilogue {ph[2]:.0f}")
